"""Developer probe: device JPEG encode of 64 processed A4 pages (for rocprofv3 --kernel-trace --stats) + Pillow on the host."""
import io, os, sys, time
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
proc = pipe.preprocess(pages)
out, sizes = eng.jpeg_encode(proc, 95)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    out, sizes = eng.jpeg_encode(proc, 95)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 5
s = sizes.cpu().numpy()
print("device: %.2f ms per 64 pages (%.0f pages/s), sizes %d..%d bytes" % (dt * 1e3, 64 / dt, s.min(), s.max()))
from PIL import Image
img = proc[0].cpu().numpy()
t = time.perf_counter()
b = io.BytesIO(); Image.fromarray(img).save(b, format="JPEG", quality=95, optimize=True)
print("Pillow on one host core: %.1f ms per page; identical to device: %s" % ((time.perf_counter() - t) * 1e3, b.getvalue() == out[0, : s[0]].cpu().numpy().tobytes()))
