"""Developer tool: per-phase wave-cycle breakdown of the ring conv kernel (run with LUMINA_RING_PROF=1), one det forward."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ocr-system_amd"))
import numpy as np, torch
from lumina_ocr import arch
from lumina_ocr.engine import Engine

eng = Engine()
eng.load_det(arch.make_det_weights(0))
rng = np.random.default_rng(1)
pages = torch.from_numpy(rng.integers(0, 256, (16, 2016, 1440, 3), dtype=np.uint8)).cuda()
eng.det_forward(pages)
torch.cuda.synchronize()
