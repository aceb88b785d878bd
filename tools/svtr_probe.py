"""Developer probe: SVTR forward (variant, storage type) on N synthetic crops next to CRNN (time; for rocprofv3 --kernel-trace --stats).
usage: svtr_probe.py [N] [tiny|base] [bf16|f16]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ocr-system_amd"))
import torch
from lumina_ocr import arch
from lumina_ocr.engine import Engine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3279
variant = sys.argv[2] if len(sys.argv) > 2 else "tiny"
dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
eng = Engine(0); eng.load_svtr(arch.make_svtr_weights(variant=variant, dtype=dtype)); eng.load_rec(arch.make_rec_weights())
crops = torch.randint(0, 256, (N, arch.REC_H, arch.REC_W, 3), dtype=torch.uint8, device="cuda")
for name, f in (("svtr-%s-%s" % (variant, dtype), eng.svtr_forward), ("crnn", eng.rec_forward)):
    f(crops); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3): f(crops)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    print("%s forward N=%d: %.2f ms (%.0f crops/s)" % (name, N, dt * 1e3, N / dt))
