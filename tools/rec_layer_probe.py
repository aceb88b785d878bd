"""Developer probe: per-layer conv timing of one rec forward (event-timed, serialised) on N synthetic crops."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ocr-system_amd"))
import torch
from lumina_ocr import arch
from lumina_ocr.engine import Engine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
eng = Engine(0); eng.load_rec(arch.make_rec_weights())
crops = torch.randint(0, 256, (N, arch.REC_H, arch.REC_W, 3), dtype=torch.uint8, device="cuda")
widths = torch.full((N,), arch.REC_W, dtype=torch.int32, device="cuda")
out = eng.rec_forward(crops, widths)
torch.cuda.synchronize()
eng.set_option("time_convs", 1)
out = eng.rec_forward(crops, widths)
rows = eng.conv_timing_detail()
eng.set_option("time_convs", 0)
tot = sum(r[2] for r in rows)
print("rec forward N=%d: conv kernels %.2f ms" % (N, tot))
for name, kern, ms, gf, mb in rows:
    print("%-18s %-36s %7.3f ms %7.2f GFLOP %7.1f TFLOP/s %8.1f MB %5.2f TB/s" % (name, kern, ms, gf, gf / ms if ms else 0, mb, mb / ms / 1e3 if ms else 0))
