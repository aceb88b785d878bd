"""Developer probe: the DB post-process only (64 A4 pages), for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ocr-system_amd"))
import torch
import bench
from lumina_ocr import arch
from lumina_ocr.engine import Engine
from lumina_ocr.pipeline import OcrPipeline
eng = Engine(0); eng.load_det(arch.make_det_weights())
pipe = OcrPipeline(eng, post=arch.TEXT_PATH_POST)
pages = bench.make_pages(torch, 64, 2024, torch.device("cuda", 0))
proc = pipe.preprocess(pages)
prob = eng.det_forward(proc)
torch.cuda.synchronize()
for _ in range(5):
    boxes, scores, counts = eng.det_postprocess(prob, proc.shape[1], proc.shape[2], **pipe.post)
torch.cuda.synchronize()
