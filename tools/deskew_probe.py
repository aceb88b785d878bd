"""Developer probe: time of the de-skew step on 64 A4 pages (2000x1414 after the cap), skewed by a few degrees."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ocr-system_amd"))
import numpy as np
import torch
from PIL import Image
from lumina_ocr import synth
from lumina_ocr.engine import Engine
eng = Engine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
base = [np.asarray(Image.fromarray(synth.synth_page(2000, 1414, 2024 + k, n_lines=60)[0]).rotate(a, resample=Image.BICUBIC, fillcolor=(255, 255, 255)))
        for k, a in enumerate((1.5, -2.0, 0.0, 3.0))]
pages = torch.from_numpy(np.stack([base[i % 4] for i in range(n)])).cuda()
for it in range(4):
    torch.cuda.synchronize(); t = time.perf_counter()
    out, rot = eng.deskew(pages)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) * 1e3
    print("deskew %d pages: %.2f ms  angles %s" % (n, dt, [round(a, 2) for a in Engine.skew_degrees(rot)[:4]]))
