"""A/B of the fused stem (conv1+conv2 in one kernel) against the two-kernel path: det forward time on the bench workload."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ocr-system_amd"))
import numpy as np, torch
from lumina_ocr import arch
from lumina_ocr.engine import Engine

eng = Engine()
eng.load_det(arch.make_det_weights(0))
rng = np.random.default_rng(1)
pages = torch.from_numpy(rng.integers(0, 256, (64, 1654, 1170, 3), dtype=np.uint8)).cuda()
for rep in range(2):
    for f in (1, 0):
        eng.set_option("fuse_stem", f)
        for _ in range(2): eng.det_forward(pages)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): eng.det_forward(pages)
        torch.cuda.synchronize()
        print("fuse_stem=%d det forward %.2f ms / 64 pages" % (f, (time.perf_counter() - t) / 5 * 1e3), flush=True)
