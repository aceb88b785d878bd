"""Developer probe: device JPEG decode of 64 A4 pages (the bench's pages, written by the device encoder at quality 95): time per call,
synchronisation passes, bytes.  usage: jpegdec_probe.py [pages] [quality]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ocr-system_amd"))
import torch
import bench
from lumina_ocr.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
q = int(sys.argv[2]) if len(sys.argv) > 2 else 95
eng = Engine(0)
pages = bench.make_pages(torch, n, 2024, torch.device("cuda", 0))
fd, sd = eng.jpeg_encode(pages, q, max_bytes=8 << 20)
sh = sd.cpu().numpy()
files = [fd[i, : int(sh[i])].cpu().numpy().tobytes() for i in range(n)]
out = torch.empty_like(pages)
for it in range(4):
    torch.cuda.synchronize(); t = time.perf_counter()
    out, st = eng.jpeg_decode(files, bench.A4_H, bench.A4_W, out=out)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("decode %d pages: %.2f ms (%.0f pages/s), %d passes, %.2f MB per file, status ok %s" % (n, dt * 1e3, n / dt, eng.jpeg_last_passes, sh.mean() / 1e6, not any(st)))
one, st = eng.jpeg_decode(files[:1], bench.A4_H, bench.A4_W)
torch.cuda.synchronize(); t = time.perf_counter()
one, st = eng.jpeg_decode(files[:1], bench.A4_H, bench.A4_W)
torch.cuda.synchronize()
print("one page: %.2f ms, %d passes" % ((time.perf_counter() - t) * 1e3, eng.jpeg_last_passes))
