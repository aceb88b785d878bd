"""Developer experiment: does the recognise stage overlap with the next step's detector when they run on two streams (two engine
handles = two workspaces)?  Prints det alone, rec alone, both serial on one stream, both on two streams."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ocr-system_amd"))
import numpy as np, torch
from lumina_ocr import arch, synth
from lumina_ocr.engine import Engine
ea, eb = Engine(0), Engine(0)
ea.load_det(arch.make_det_weights()); ea.set_option("det_sub_batch", 64)
eb.load_rec(arch.make_rec_weights())
pages = torch.randint(0, 256, (64, 2000, 1414, 3), dtype=torch.uint8, device="cuda")
rng = np.random.default_rng(1)
crops = torch.from_numpy(np.stack([synth.synth_crop(rng)[0] for _ in range(64)])).cuda().repeat(52, 1, 1, 1)[:3279].contiguous()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def det():
    return ea.det_forward(pages)
def rec():
    return eb.rec_forward(crops)
def T(fn, n=5):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
def serial():
    det(); rec()
def conc():
    with torch.cuda.stream(s1): det()
    with torch.cuda.stream(s2): rec()
print("det alone %.2f ms, rec alone %.2f ms, serial %.2f ms, two streams %.2f ms" % (T(det), T(rec), T(serial), T(conc)))
