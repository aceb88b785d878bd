"""Target for rocprofv3 --pmc passes: two det forwards of N A4 pages (2016x1440 padded) in launches of N pages, nothing else.
usage: pmc_target.py [N]   (default 64 = what bench.py launches per step)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ocr-system_amd"))
import torch
from lumina_ocr import arch
from lumina_ocr.engine import Engine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = Engine(0)
eng.load_det(arch.make_det_weights())
eng.set_option("det_sub_batch", N)
pages = torch.randint(0, 256, (N, 2000, 1414, 3), dtype=torch.uint8, device="cuda")
for _ in range(2):
    prob = eng.det_forward(pages)
torch.cuda.synchronize()
