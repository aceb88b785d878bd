"""Developer probe: wall time of each pipeline stage (synchronised), 64 A4 pages."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ocr-system_amd"))
import torch
import bench
from lumina_ocr import arch
from lumina_ocr.engine import Engine
from lumina_ocr.pipeline import OcrPipeline
eng = Engine(0); eng.load_det(arch.make_det_weights()); eng.load_rec(arch.make_rec_weights())
pipe = OcrPipeline(eng, post=arch.TEXT_PATH_POST)
pages = bench.make_pages(torch, 64, 2024, torch.device("cuda", 0))
def T(f, *a):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(*a); torch.cuda.synchronize(); return r, (time.perf_counter() - t) * 1e3
for it in range(3):
    proc, t1 = T(pipe.preprocess, pages)
    prob, t2 = T(eng.det_forward, proc)
    (boxes, scores, counts), t3 = T(lambda p: eng.det_postprocess(p, proc.shape[1], proc.shape[2], **pipe.post), prob)
    dets, t4 = T(pipe.recognize, proc, boxes, scores, counts)
    _, t5 = T(pipe.run, pages)
    print("preprocess %.2f  det %.2f  post %.2f  recognize(crop+rec+ctc+host) %.2f | sum %.2f  run() %.2f ms" % (t1, t2, t3, t4, t1 + t2 + t3 + t4, t5))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); pipe.recognize(proc, boxes, scores, counts); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
