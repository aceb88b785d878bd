"""GPU parity: the MFMA implicit-GEMM conv kernel vs a plain PyTorch fp32 reference of the same op
(bf16 inputs/weights, fp32 accumulate, one bf16 rounding) — through the C ABI (lumina_ocr_conv2d)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import close_stats
from lumina_ocr import arch

pytestmark = pytest.mark.gpu

ACTS = {0: lambda v: v, 1: torch.relu, 2: lambda v: v * torch.clamp(v + 3, 0, 6) / 6, 4: torch.sigmoid}

CASES = [
    # n, h, w, cin, cout, ks, stride, act, residual
    (2, 40, 70, 32, 32, 3, 1, 1, False),
    (1, 33, 65, 32, 64, 3, 1, 1, False),
    (2, 24, 48, 64, 64, 3, 1, 1, True),
    (1, 19, 37, 128, 128, 3, 1, 1, True),
    (1, 16, 32, 256, 256, 3, 1, 0, False),
    (1, 9, 13, 512, 512, 3, 1, 1, True),
    (1, 34, 66, 64, 128, 3, 2, 1, False),
    (1, 18, 30, 128, 256, 3, 2, 1, False),
    (2, 16, 32, 256, 64, 3, 1, 0, False),
    (1, 20, 44, 64, 64, 1, 1, 0, False),
    (1, 12, 20, 512, 256, 1, 1, 0, False),
    (1, 20, 36, 64, 128, 2, 2, 0, False),
    (3, 8, 160, 48, 128, 1, 1, 2, False),
    (2, 4, 160, 128, 32, 1, 1, 0, True),
    (2, 16, 160, 16, 48, 1, 1, 2, False),
    (1, 64, 96, 256, 64, 3, 1, 1, False),
    (1, 7, 200, 64, 8, 1, 1, 4, False),
    (2, 21, 37, 64, 256, 1, 1, 0, True),     # pixel-stationary pointwise kernel: partial last tile, residual
    (1, 19, 45, 128, 256, 1, 1, 1, False),
    (1, 30, 50, 64, 64, 1, 1, 0, False),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "n%d_%dx%d_c%d-%d_k%ds%d_a%d_r%d" % c)
def test_conv2d_matches_fp32_reference(engine, case):
    n, h, w, cin, cout, ks, stride, act, use_res = case
    rng = np.random.default_rng(hash(case) & 0xFFFF)
    x = arch.bf16_round(rng.standard_normal((n, h, w, cin), dtype=np.float32))
    wt = arch.bf16_round(rng.standard_normal((cout, ks, ks, cin), dtype=np.float32) * np.float32(np.sqrt(2.0 / (ks * ks * cin))))
    bias = rng.standard_normal(cout, dtype=np.float32) * np.float32(0.1)
    ho = (h - 1) // stride + 1 if ks == 3 else h // stride
    wo = (w - 1) // stride + 1 if ks == 3 else w // stride
    res = arch.bf16_round(rng.standard_normal((n, ho, wo, cout), dtype=np.float32)) if use_res else None

    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    ref = F.conv2d(xt, torch.from_numpy(wt).permute(0, 3, 1, 2), torch.from_numpy(bias), stride=stride, padding=1 if ks == 3 else 0)
    if use_res:
        ref = ref + torch.from_numpy(res).permute(0, 3, 1, 2)
    ref = ACTS[act](ref).to(torch.bfloat16).to(torch.float32).permute(0, 2, 3, 1).numpy()

    xd = torch.from_numpy(x).to(torch.bfloat16).cuda()
    rd = torch.from_numpy(res).to(torch.bfloat16).cuda() if use_res else None
    y = engine.conv2d(xd, wt, bias, ks, stride, act, rd).float().cpu().numpy()
    assert y.shape == ref.shape
    st = close_stats(y, ref)
    # fp32 accumulation order differs: a value may land on the other side of a bf16 rounding boundary
    assert st["within1"] > 0.999 and st["within4"] == 1.0, st
