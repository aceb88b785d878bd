"""Developer probe: ONE 3x3 / stride-1 layer through the ring kernel (conv2d hook, variant 2) at a detector shape, for PMC passes.
usage: ring_layer_probe.py [cin cout h w n res]   default: stage-0 layer 64 -> 64 at 504x360, 16 pages, with residual"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ocr-system_amd"))
import numpy as np
import torch
from lumina_ocr import arch
from lumina_ocr.engine import Engine
a = [int(v) for v in sys.argv[1:]] + [64, 64, 504, 360, 16, 1][len(sys.argv) - 1:]
cin, cout, h, w, n, res = a[:6]
eng = Engine(0)
rng = np.random.default_rng(0)
x = torch.from_numpy(arch.bf16_round(rng.standard_normal((n, h, w, cin), dtype=np.float32))).to(torch.bfloat16).cuda()
r = torch.from_numpy(arch.bf16_round(rng.standard_normal((n, h, w, cout), dtype=np.float32))).to(torch.bfloat16).cuda() if res else None
wt = arch.bf16_round(rng.standard_normal((cout, 3, 3, cin), dtype=np.float32) * np.float32(np.sqrt(2.0 / (9 * cin))))
b = rng.standard_normal(cout, dtype=np.float32) * np.float32(0.1)
eng.set_option("conv2d_variant", 2)
for it in range(4):
    torch.cuda.synchronize(); t = time.perf_counter()
    y = eng.conv2d(x, wt, b, 3, 1, 1, r)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
print("conv %d->%d %dx%d n=%d res=%d: last call %.3f ms incl. host packing" % (cin, cout, h, w, n, res, dt * 1e3))
