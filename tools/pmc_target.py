"""Target for rocprofv3 --pmc passes: two det forwards of 16 A4 pages (2016x1440 padded), nothing else."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ocr-system_amd"))
import torch
from lumina_ocr import arch
from lumina_ocr.engine import Engine
eng = Engine(0)
eng.load_det(arch.make_det_weights())
pages = torch.randint(0, 256, (16, 2000, 1414, 3), dtype=torch.uint8, device="cuda")
for _ in range(2):
    prob = eng.det_forward(pages)
torch.cuda.synchronize()
