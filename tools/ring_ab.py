"""A/B of the persistent ring conv kernel against the one-tile-per-work-group LDS-DMA kernel: det forward on the bench shape."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ocr-system_amd"))
import numpy as np, torch
from lumina_ocr import arch
from lumina_ocr.engine import Engine

eng = Engine()
eng.load_det(arch.make_det_weights(0))
rng = np.random.default_rng(1)
H, W = (int(a) for a in (sys.argv[1:3] if len(sys.argv) > 2 else (2016, 1440)))
pages = torch.from_numpy(rng.integers(0, 256, (32, H, W, 3), dtype=np.uint8)).cuda()
for rep in range(2):
    for ring, orient in ((0, -1), (1, 0), (1, 1), (1, -1)):
        eng.set_option("conv_ring", ring); eng.set_option("ring_orient", orient)
        for _ in range(2): eng.det_forward(pages)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): eng.det_forward(pages)
        torch.cuda.synchronize()
        print("ring=%d orient=%2d det forward %.2f ms / 32 pages" % (ring, orient, (time.perf_counter() - t) / 5 * 1e3), flush=True)
