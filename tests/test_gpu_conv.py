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


# 3x3 / stride-1 layers through the kernels that serve them in the detector at BASELINE sizes (engine option conv2d_variant:
# 1 = LDS-DMA fed 16x32-pixel tile, one tile per work-group; 2 = persistent ring kernel), dense random operands: every input
# channel chunk, every output-channel tile, the residual path, partial tiles, several tiles per resident work-group
BIG_CASES = [
    # n, h, w, cin, cout, act, residual
    (2, 24, 48, 64, 64, 1, True),
    (1, 33, 65, 32, 64, 1, False),
    (1, 19, 37, 128, 128, 1, True),
    (2, 16, 32, 256, 64, 0, False),
    (1, 9, 13, 512, 512, 1, True),
    (1, 45, 63, 512, 512, 1, True),
    (3, 90, 126, 64, 64, 1, True),     # 3 x 6 x 4 = 72 tiles of one channel tile
    (1, 64, 96, 256, 64, 1, False),
    (1, 50, 40, 128, 512, 0, True),    # 8 output-channel tiles
]


@pytest.mark.parametrize("variant", [0, 1, 2], ids=["tile8x32", "lds_dma16x32", "ring"])
@pytest.mark.parametrize("case", BIG_CASES, ids=lambda c: "n%d_%dx%d_c%d-%d_a%d_r%d" % c)
def test_conv3x3_variants_match_fp32_reference(engine, case, variant):
    n, h, w, cin, cout, act, use_res = case
    rng = np.random.default_rng((hash(case) & 0xFFFF) + 7)
    x = arch.bf16_round(rng.standard_normal((n, h, w, cin), dtype=np.float32))
    wt = arch.bf16_round(rng.standard_normal((cout, 3, 3, cin), dtype=np.float32) * np.float32(np.sqrt(2.0 / (9 * cin))))
    bias = rng.standard_normal(cout, dtype=np.float32) * np.float32(0.1)
    res = arch.bf16_round(rng.standard_normal((n, h, w, cout), dtype=np.float32)) if use_res else None
    ref = F.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(wt).permute(0, 3, 1, 2), torch.from_numpy(bias), padding=1)
    if use_res:
        ref = ref + torch.from_numpy(res).permute(0, 3, 1, 2)
    ref = ACTS[act](ref).to(torch.bfloat16).to(torch.float32).permute(0, 2, 3, 1).numpy()
    xd = torch.from_numpy(x).to(torch.bfloat16).cuda()
    rd = torch.from_numpy(res).to(torch.bfloat16).cuda() if use_res else None
    engine.conv_timing_detail()
    engine.set_option("time_convs", 1)
    engine.set_option("conv2d_variant", variant)
    try:
        outs = []
        for orient in ((-1,) if variant != 2 else (0, 1)):
            engine.set_option("ring_orient", orient)
            outs.append(engine.conv2d(xd, wt, bias, 3, 1, act, rd).float().cpu().numpy())
        names = [k for _, k, *_ in engine.conv_timing_detail()]
    finally:
        engine.set_option("conv2d_variant", 0)
        engine.set_option("ring_orient", -1)
        engine.set_option("time_convs", 0)
    want = {0: "conv_mfma_kernel<3,1,", 1: "conv_mfma_kernel<3,1,64,16,32,4,3,4>", 2: "conv_ring_kernel<"}[variant]
    assert all(k.startswith(want) for k in names) and (variant != 1 or len(names) == 1), names
    if variant == 2:
        assert len(names) == 2 and names[0].startswith("conv_ring_kernel<0,false,false,") and names[1].startswith("conv_ring_kernel<1,false,false,"), names
        assert np.array_equal(outs[0], outs[1])      # both tile orientations sum in the same order
    st = close_stats(outs[0], ref)
    assert st["within1"] > 0.999 and st["within4"] == 1.0, st
