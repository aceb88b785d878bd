"""How far do the engine's CTC logits sit from the oracle's?  (choice of the margin eps in tests/test_gpu_rec.py / test_gpu_e2e.py)
Logits are rebuilt in fp64 from the engine's own bf16 LSTM output (tap lstm.l1) and compared with the oracle's fp32 logits."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ocr-system_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lumina_ocr import arch, synth
from lumina_ocr.engine import Engine
from oracle import nets
eng = Engine(0)
w = arch.make_rec_weights(4321)
eng.load_rec(w)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 4321)
crops = np.stack([synth.synth_crop(rng)[0] for _ in range(48)])
eng.set_option("keep_taps", 1)
idx, prob = eng.rec_forward(torch.from_numpy(crops).cuda())
torch.cuda.synchronize()
seq = eng.read_tap("lstm.l1").reshape(-1, 192).astype(np.float64)
lg = seq @ w["ctc.fc.w"].astype(np.float64).T + w["ctc.fc.b"].astype(np.float64)
ridx, rprob, rlog, rseq = nets.rec_forward(w, crops)
rlog = rlog.reshape(-1, rlog.shape[-1]).astype(np.float64)
d = np.abs(lg - rlog)
top2 = np.partition(rlog, -2, axis=1)[:, -2:]
margin = top2[:, 1] - top2[:, 0]
same = idx.cpu().numpy().reshape(-1) == ridx.reshape(-1)
print("logit |gpu - oracle|: mean %.4f  p99 %.4f  max %.4f ; logit std %.3f" % (d.mean(), np.quantile(d, 0.99), d.max(), rlog.std()))
print("per-step max over classes: p50 %.4f p99 %.4f max %.4f" % (np.quantile(d.max(1), .5), np.quantile(d.max(1), .99), d.max(1).max()))
for eps in (0.25, 0.5, 1.0, 1.5, 2.0, 3.0):
    c = margin > eps
    print("eps %.2f: clear steps %.3f, agreement on clear %.5f (%d flips), smallest margin of a flipped step: %.4f" %
          (eps, c.mean(), same[c].mean(), int((~same[c]).sum()), margin[~same].max() if (~same).any() else -1))
lines_clear = (margin.reshape(48, 80) > 1.0).all(1)
print("lines with every step clear at eps 1.0: %d of 48" % lines_clear.sum())
