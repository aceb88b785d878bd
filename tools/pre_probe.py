"""Developer probe: the pre-processing kernels only (64 A4 pages), for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ocr-system_amd"))
import torch
import bench
from lumina_ocr import arch
from lumina_ocr.engine import Engine
from lumina_ocr.pipeline import OcrPipeline
eng = Engine(0)
pipe = OcrPipeline(eng, post=arch.TEXT_PATH_POST)
pages = bench.make_pages(torch, 64, 2024, torch.device("cuda", 0))
for _ in range(5):
    proc = pipe.preprocess(pages)
torch.cuda.synchronize()
