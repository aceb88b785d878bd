"""Developer A/B: det forward of 64 A4 pages for several tail_group sizes (pages per pass of the 1/4-resolution tail)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ocr-system_amd"))
import torch
from lumina_ocr import arch
from lumina_ocr.engine import Engine
eng = Engine(0)
eng.load_det(arch.make_det_weights())
eng.set_option("det_sub_batch", 64)
pages = torch.randint(0, 256, (64, 2000, 1414, 3), dtype=torch.uint8, device="cuda")
prob = eng.det_forward(pages)
for g in (8, 16, 32, 64, 16, 8):
    eng.set_option("tail_group", g)
    prob = eng.det_forward(pages, out=prob)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5):
        prob = eng.det_forward(pages, out=prob)
    torch.cuda.synchronize()
    print("tail_group %2d: %.2f ms per 64 pages" % (g, (time.perf_counter() - t) / 5 * 1e3))
