import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ocr-system_amd"))
import numpy as np, torch
from lumina_ocr import arch
from lumina_ocr.engine import Engine
eng = Engine()
eng.load_det(arch.make_det_weights(0))
rng = np.random.default_rng(1)
pages = torch.from_numpy(rng.integers(0, 256, (16, 2016, 1440, 3), dtype=np.uint8)).cuda()
outs = {}
for rep in range(2):
    for b in (0, 1):
        eng.set_option("blocked_layout", b)
        for _ in range(2): o = eng.det_forward(pages)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): o = eng.det_forward(pages)
        torch.cuda.synchronize()
        outs[b] = o.clone()
        print("blocked=%d det forward %.3f ms / 16 pages" % (b, (time.perf_counter() - t) / 5 * 1e3), flush=True)
print("equal:", torch.equal(outs[0], outs[1]))
eng.set_option("time_convs", 1)
for b in (0, 1):
    eng.set_option("blocked_layout", b)
    eng.det_forward(pages); torch.cuda.synchronize()
    rows = eng.conv_timing_detail()
    print(b, [(n, round(ms, 3)) for n, k, ms, gf, mb in rows if n.startswith("s0.")])
