"""Developer probe: per-layer conv timing of one det forward on the GPU box (writes gpurun_out/perf_probe.txt)."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ocr-system_amd"))
import numpy as np, torch
from lumina_ocr import arch
from lumina_ocr.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2000, 1414)
sb = int(sys.argv[4]) if len(sys.argv) > 4 else 4
eng = Engine(0)
eng.load_det(arch.make_det_weights())
eng.set_option("det_sub_batch", sb)
pages = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device="cuda")
prob = eng.det_forward(pages)
torch.cuda.synchronize()
t = time.time()
for _ in range(3):
    prob = eng.det_forward(pages, out=prob)
torch.cuda.synchronize()
dt = (time.time() - t) / 3
hp, wp = prob.shape[1:]
fl = 2 * arch.det_macs_per_page(hp, wp) * B
lines = ["det forward B=%d %dx%d sub_batch=%d: %.2f ms  %.1f pages/s  %.1f TFLOP/s (%.1f%% of 2500)" % (B, hp, wp, sb, dt * 1e3, B / dt, fl / dt / 1e12, fl / dt / 2.5e13)]
eng.set_option("time_convs", 1)
eng.det_forward(pages, out=prob)
rows = eng.conv_timing_detail()
eng.set_option("time_convs", 0)
agg = {}
for name, kern, ms, gf, mb in rows:
    a = agg.setdefault(name, [kern, 0.0, 0.0, 0.0]); a[1] += ms; a[2] += gf; a[3] += mb
tot = sum(a[1] for a in agg.values())
lines.append("conv kernels total %.2f ms (event-timed, serialised), %.1f TFLOP/s" % (tot, sum(a[2] for a in agg.values()) / tot))
for name, (kern, ms, gf, mb) in agg.items():
    lines.append("%-14s %-30s %8.3f ms %8.1f GFLOP %7.1f TFLOP/s %8.1f MB %6.2f TB/s(alg)" % (name, kern, ms, gf, gf / ms, mb, mb / ms / 1e3))
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/perf_probe.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
