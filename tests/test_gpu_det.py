"""GPU parity: DBNet forward through the C ABI vs the oracle restatement (oracle/nets.py, mode bf16),
layer by layer (taps) and on the final probability map; then DB post-process vs the C oracle."""
import numpy as np
import pytest
import torch

from conftest import close_stats
from lumina_ocr import arch, synth

pytestmark = pytest.mark.gpu


def _dump(name, obj):
    import json, os
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open(os.path.join("gpurun_out", name), "w") as f:
            json.dump(obj, f, indent=1)
    except OSError:
        pass


# what still exists with every fusion on (engine option keep_taps=2): all of these are compared bit for bit below
TAPS_FUSED = ["stem.pool", "s0.b0", "s0.b1", "s1.b0", "s1.b1", "s2.b0", "s2.b1", "s3.b0", "s3.b1", "head.conv1"]
# the FPN outputs: either the 1/4-resolution concat, or — when head.conv1 reads p5 .. p2 at their own resolution (option fpn_multi, the
# default wherever the ring kernel serves head.conv1) — the four tensors themselves
TAPS_FPN = [["fpn.fuse"], ["fpn.p5", "fpn.p4", "fpn.p3", "fpn.p2"]]


def _forward_all(engine, pages):
    """det forward with keep_taps=2 (no fusion is switched off) -> {"prob": bf16 bits, tap name: values}."""
    engine.set_option("keep_taps", 2)
    try:
        prob = engine.det_forward(pages).clone()
        torch.cuda.synchronize()
        out = {"prob": prob.view(torch.int16).cpu().numpy()}
        for name in TAPS_FUSED:
            out[name] = engine.read_tap(name)
        from lumina_ocr.engine import EngineError
        for group in TAPS_FPN:
            try:
                for name in group:
                    out[name] = engine.read_tap(name)
            except EngineError:
                pass
    finally:
        engine.set_option("keep_taps", 0)
    return out


def _assert_same(a, b, what=""):
    for name in a:
        if name not in b:          # (the two runs materialised the FPN outputs differently: compared through head.conv1 / prob)
            continue
        assert a[name].shape == b[name].shape, (what, name)
        if not np.array_equal(a[name], b[name]):
            d = np.argwhere(a[name] != b[name])
            raise AssertionError("%s: tap %s differs at %d of %d values, first %s" % (what, name, len(d), a[name].size, d[0].tolist()))


def _pages(b, h, w, seed):
    return np.stack([synth.synth_page(h, w, seed + i, n_lines=max(3, h // 40))[0] for i in range(b)])


@pytest.mark.parametrize("shape", [(2, 250, 200), (1, 320, 448), (3, 96, 130)], ids=lambda s: "b%d_%dx%d" % s)
def test_det_forward_taps_and_prob(engine, det_weights, shape):
    from oracle import nets
    b, h, w = shape
    pages = _pages(b, h, w, 7)
    engine.load_det(det_weights)
    engine.set_option("keep_taps", 1)
    engine.set_option("det_sub_batch", 8)
    prob = engine.det_forward(torch.from_numpy(pages).cuda())
    torch.cuda.synchronize()
    engine.set_option("det_sub_batch", 16)
    taps = {}
    ref = nets.det_forward(det_weights, pages, mode="bf16", taps=taps)
    worst = {}
    for name in ["stem.conv1", "stem.conv2", "stem.conv3", "stem.pool", "s0.b0", "s0.b1", "s1.b0", "s1.b1", "s2.b0", "s2.b1",
                 "s3.b0", "s3.b1", "fpn.fuse", "head.conv1", "head.convt2"]:
        got = engine.read_tap(name)
        assert got.shape == taps[name].shape, (name, got.shape, taps[name].shape)
        st = close_stats(got, taps[name])
        worst[name] = st
    engine.set_option("keep_taps", 0)
    p = prob.float().cpu().numpy()
    st = close_stats(p, ref)
    worst["prob"] = st
    _dump("parity_det_b%d_%dx%d.json" % shape, worst)
    # identical arithmetic definition; only the fp32 summation order differs, so almost every value is
    # bit-equal and the rest are off by a bf16 ulp that then propagates through the following layers
    for name, s_ in worst.items():
        if name != "prob":
            assert s_["within4"] > 0.90 and s_["mean_abs"] < 0.01 * max(s_["ref_mean_abs"], 1e-3), (name, s_)
    # the first layers must be (almost) bit-equal: this is what pins the kernel, the rest is bf16 drift
    assert worst["stem.conv1"]["within1"] == 1.0 and worst["stem.conv3"]["within1"] > 0.999 and worst["s0.b0"]["within1"] > 0.995
    assert p.shape == ref.shape
    assert st["within4"] > 0.98 and st["max_abs"] < 0.06, st
    flips = float(((p > arch.DET_THRESH) != (ref > arch.DET_THRESH)).mean())
    assert flips < 1e-2, flips


@pytest.mark.parametrize("ring", [0, 1], ids=["lds_dma_tile", "ring"])
@pytest.mark.parametrize("shape", [(2, 250, 200), (1, 320, 448), (1, 447, 901)], ids=lambda s: "b%d_%dx%d" % s)
def test_det_taps_vs_oracle_with_the_production_kernels(engine, any_det_weights, shape, ring):
    """The kernels that run at BASELINE sizes — persistent ring kernel / LDS-DMA 16x32-tile kernel (conv_big_min=1 selects them
    on these small pages too), fused stem.conv1+conv2, stem.conv3 with the pool in its epilogue, fused DBHead tail — compared
    with the oracle on EVERY tap that still exists (keep_taps=2 leaves the fusions on).  With the dense weights every channel
    of every layer reaches the taps: cross-chunk DMA ring, deferred stores, residual path of every channel tile."""
    from oracle import nets
    b, h, w = shape
    pages = _pages(b, h, w, 9)
    engine.load_det(any_det_weights)
    engine.set_option("conv_big_min", 1)
    engine.set_option("conv_ring", ring)
    engine.set_option("det_sub_batch", 8)
    engine.conv_timing_detail()
    engine.set_option("time_convs", 1)
    try:
        got = _forward_all(engine, torch.from_numpy(pages).cuda())
        names = [k for _, k, *_ in engine.conv_timing_detail()]
    finally:
        engine.set_option("conv_big_min", 1024)
        engine.set_option("conv_ring", 1)
        engine.set_option("time_convs", 0)
        engine.set_option("det_sub_batch", 16)
    want = "conv_ring_kernel" if ring else "conv_mfma_kernel<3,1,64,16,32,4,3,4>"
    assert sum(k.startswith(want) for k in names) >= 10, names
    assert any(k.startswith("conv_ring_kernel<0,false,true") or k.endswith(",4,4>") for k in names), names   # pooled stem.conv3
    taps = {}
    ref = nets.det_forward(any_det_weights, pages, mode="bf16", taps=taps, compose=bool(ring))   # (fpn.p2 is composed with fpn.in2 on the ring kernel only)
    stats = {name: close_stats(got[name], taps[name]) for name in got if name != "prob"}
    assert "fpn.fuse" in stats or "fpn.p2" in stats
    p = arch.bf16_bits_to_f32(got["prob"].view(np.uint16))
    stats["prob"] = close_stats(p, ref)
    _dump("parity_det_prod_ring%d_b%d_%dx%d.json" % ((ring,) + shape), stats)
    for name, s_ in stats.items():
        if name != "prob":
            assert got[name].shape == taps[name].shape
            # (the deepest FPN levels sit behind ~25 bf16-rounded layers: their drift is the largest, ~0.5 % of the mean magnitude)
            assert s_["within4"] > (0.85 if name.startswith("fpn.p") else 0.90) and s_["mean_abs"] < 0.01 * max(s_["ref_mean_abs"], 1e-3), (name, s_)
    assert stats["stem.pool"]["within1"] > 0.999 and stats["s0.b0"]["within1"] > 0.99, stats
    assert stats["prob"]["within4"] > 0.97 and stats["prob"]["max_abs"] < 0.06, stats["prob"]


@pytest.mark.parametrize("orient", [0, 1], ids=["rows", "transposed"])
@pytest.mark.parametrize("shape", [(2, 250, 200), (1, 447, 901), (3, 352, 512)], ids=lambda s: "b%d_%dx%d" % s)
def test_head_reads_fpn_maps_at_their_own_resolution_bit_identically(engine, any_det_weights, shape, orient):
    """Option fpn_multi: head.conv1's halo addressing reads p5 / p4 / p3 / p2 nearest-upsampled from their own tensors (the FPN
    concat is never written) — the same products in the same order as the conv over the materialised concat."""
    b, h, w = shape
    pages = torch.from_numpy(_pages(b, h, w, 51)).cuda()
    engine.load_det(any_det_weights)
    engine.set_option("conv_big_min", 1)
    engine.set_option("ring_orient", orient)
    try:
        engine.set_option("fpn_multi", 0)
        ref = _forward_all(engine, pages)
        engine.set_option("fpn_multi", 1)
        a = _forward_all(engine, pages)
    finally:
        engine.set_option("fpn_multi", 1)
        engine.set_option("ring_orient", -1)
        engine.set_option("conv_big_min", 1024)
    assert "fpn.fuse" in ref and "fpn.p3" in a and "fpn.fuse" not in a
    _assert_same(a, ref, "fpn_multi")
    assert np.array_equal(a["head.conv1"], ref["head.conv1"]) and np.array_equal(a["prob"], ref["prob"])
    # the four maps are the channel slices of the concat, sub-sampled
    for k, (name, s_) in enumerate((("fpn.p5", 8), ("fpn.p4", 4), ("fpn.p3", 2), ("fpn.p2", 1))):
        assert np.array_equal(a[name], ref["fpn.fuse"][:, ::s_, ::s_, 64 * k:64 * k + 64]), name


@pytest.mark.parametrize("shape", [(2, 250, 200), (1, 447, 901)], ids=lambda s: "b%d_%dx%d" % s)
def test_fpn_p2_composed_with_its_lateral(engine, det_weights, dense_det_weights, shape):
    """Default: fpn.p2 is ONE conv over [c2 | up2(out3)] with the lateral fpn.in2 composed into its weights (engine.hip
    compose_fpn_p2 / oracle nets.compose_fpn_p2); option fpn_compose = 0: in2 -> 256-channel lateral -> p2.  Each against its own
    oracle definition; the composed weights themselves bit for bit (the tap of a one-hot input would need a test hook: instead the
    p2 tap of the composed path must be CLOSER to the composed oracle than to the two-step one); on the hand-set text path (exact
    arithmetic) the two definitions give the identical probability map; a lateral with a bias falls back to two kernels."""
    from oracle import nets
    b, h, w = shape
    pages = _pages(b, h, w, 77)
    pd = torch.from_numpy(pages).cuda()
    engine.set_option("conv_big_min", 1)
    try:
        engine.load_det(dense_det_weights)
        got = {}
        for comp in (1, 0):
            engine.set_option("fpn_compose", comp)
            got[comp] = _forward_all(engine, pd)
        ref = {}
        for comp in (1, 0):
            t = {}
            nets.det_forward(dense_det_weights, pages, taps=t, compose=bool(comp))
            ref[comp] = t
        for comp in (1, 0):
            st = close_stats(got[comp]["fpn.p2"], ref[comp]["fpn.p2"])
            assert st["within4"] > 0.85 and st["mean_abs"] < 0.01 * st["ref_mean_abs"], (comp, st)
        assert not np.array_equal(got[1]["fpn.p2"], got[0]["fpn.p2"])
        own = np.abs(got[1]["fpn.p2"].astype(np.float64) - ref[1]["fpn.p2"]).mean()
        other = np.abs(got[1]["fpn.p2"].astype(np.float64) - ref[0]["fpn.p2"]).mean()
        assert own < other, (own, other)
        for name in ("s0.b1", "s1.b1"):                                    # everything upstream of p2 is untouched
            assert np.array_equal(got[1][name], got[0][name])
        # text path: exact arithmetic on channel 0 -> the same probability bits either way
        engine.load_det(det_weights)
        engine.set_option("fpn_compose", 1)
        a = engine.det_forward(pd).clone()
        engine.set_option("fpn_compose", 0)
        bb = engine.det_forward(pd).clone()
        assert torch.equal(a.view(torch.int16), bb.view(torch.int16))
        # a lateral WITH a bias cannot be composed exactly under zero padding: the loader keeps the two-kernel path
        wb = {k: v.copy() for k, v in dense_det_weights.items()}
        wb["fpn.in2.b"][:] = 0.25
        engine.load_det(wb)
        engine.set_option("fpn_compose", 1)
        g = _forward_all(engine, pd)
        t = {}
        nets.det_forward(wb, pages, taps=t)
        assert nets.compose_fpn_p2(wb) is None
        st = close_stats(g["fpn.p2"], t["fpn.p2"])
        assert st["within4"] > 0.85, st
    finally:
        engine.set_option("fpn_compose", 1)
        engine.set_option("conv_big_min", 1024)


@pytest.mark.parametrize("shape", [(2, 250, 200), (1, 447, 901), (3, 96, 130), (1, 1000, 330)], ids=lambda s: "b%d_%dx%d" % s)
def test_fused_block_entry_shortcut_is_bit_identical(engine, any_det_weights, shape):
    """Option fuse_short (default): in stages 1-3 the block entry's 3x3 / stride-2 conv0 kernel also computes the block's 2x2 /
    stride-2 vd shortcut from the halo it staged — the shortcut's taps are taps (1,1) .. (2,2) of the 3x3 window — with its own
    accumulators, in the separate kernel's (chunk, tap) order: every tap downstream must be bit-identical, and the launch list
    must show the fused instantiation instead of the 2x2 / s2 kernel."""
    b, h, w = shape
    pages = torch.from_numpy(_pages(b, h, w, 83)).cuda()
    engine.load_det(any_det_weights)
    try:
        engine.set_option("fuse_short", 0)
        ref = _forward_all(engine, pages)
        engine.set_option("fuse_short", 1)
        engine.conv_timing_detail()
        engine.set_option("time_convs", 1)
        a = _forward_all(engine, pages)
        rows = engine.conv_timing_detail()
    finally:
        engine.set_option("fuse_short", 1)
        engine.set_option("time_convs", 0)
    names = [n for n, *_ in rows]
    kerns = [k for _, k, *_ in rows]
    assert sum(k == "conv_mfma_kernel<3,2,64,16,32,4,5,2>" for k in kerns) == 3 and not any(k.startswith("conv_mfma_kernel<2,2,") for k in kerns), kerns
    assert "s1.b0.conv0+short" in names and "s1.b0.short" not in names
    _assert_same(a, ref, "fused shortcut")
    assert np.array_equal(a["prob"], ref["prob"])


def test_grouped_quarter_resolution_tail_is_invisible(engine, any_det_weights):
    """Option tail_group: the lateral in2 -> p2 -> head.conv1 -> DBHead tail section runs in groups of pages inside one forward
    (producer -> consumer locality in the Infinity Cache; the 256-channel lateral exists for one group only): launch order only."""
    pages = torch.from_numpy(_pages(5, 250, 330, 61)).cuda()
    engine.load_det(any_det_weights)
    engine.set_option("conv_big_min", 1)          # head.conv1 on the ring kernel: the multi-source path the grouping belongs to
    try:
        engine.set_option("tail_group", 0)
        ref = engine.det_forward(pages).clone()
        engine.set_option("tail_group", 2)        # groups of 2 + 2 + 1 pages
        a = engine.det_forward(pages).clone()
        engine.set_option("fuse_head", 0)         # (the two-launch DBHead tail inside the group loop)
        b = engine.det_forward(pages).clone()
    finally:
        engine.set_option("fuse_head", 1)
        engine.set_option("tail_group", 16)
        engine.set_option("conv_big_min", 1024)
    torch.cuda.synchronize()
    assert torch.equal(a, ref)
    d = (a.float() - b.float()).abs()
    assert float(d.max()) <= 2.0 ** -8            # fused vs two-launch tail: see test_fused_head_equals_unfused


def test_det_sub_batching_is_invisible(engine, any_det_weights):
    pages = torch.from_numpy(_pages(5, 128, 160, 3)).cuda()
    engine.load_det(any_det_weights)
    engine.set_option("det_sub_batch", 8)
    a = engine.det_forward(pages).clone()
    engine.set_option("det_sub_batch", 2)
    b = engine.det_forward(pages).clone()
    engine.set_option("det_sub_batch", 16)
    torch.cuda.synchronize()
    assert torch.equal(a.view(torch.int16), b.view(torch.int16))


def test_normalize_kernel_bit_exact(engine):
    from oracle import nets
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (2, 37, 53, 3), dtype=np.uint8)
    mean, std = np.asarray(arch.DET_MEAN, np.float32), np.asarray(arch.DET_STD, np.float32)
    scale = (np.float32(1.0) / (np.float32(255.0) * std)).astype(np.float32)
    shift = (-mean / std).astype(np.float32)
    out = engine.normalize(torch.from_numpy(img).cuda(), 64, 64, scale.tolist(), shift.tolist(), nchw=True)
    ref = nets.det_normalize(img, 64, 64).numpy()
    assert np.array_equal(out.float().cpu().numpy(), ref)


@pytest.mark.parametrize("seed", [0, 1])
def test_db_postprocess_bit_exact_on_engine_prob(engine, det_weights, seed):
    """Integer path: feed the SAME bf16 probability map to the HIP post-process and to the C oracle."""
    from oracle import dbpost
    b, h, w = 2, 300, 420
    pages = _pages(b, h, w, 11 + seed)
    engine.load_det(det_weights)
    prob = engine.det_forward(torch.from_numpy(pages).cuda())
    boxes, scores, counts = engine.det_postprocess(prob, h, w)
    torch.cuda.synchronize()
    pb = prob.view(torch.int16).cpu().numpy().view(np.uint16)
    for i in range(b):
        rb, rs, ncomp = dbpost.db_postprocess(pb[i], h, w)
        n = int(counts[i])
        assert n == len(rb), (n, len(rb), ncomp)
        assert np.array_equal(boxes[i, :n].cpu().numpy(), rb)
        assert np.array_equal(scores[i, :n].cpu().numpy(), rs)


def test_db_postprocess_structured_maps(engine):
    """Hand-made probability maps: rotated bars, touching blobs, 1-pixel specks, blobs on the border, > cap."""
    from oracle import dbpost
    rng = np.random.default_rng(3)
    hp, wp, vh, vw = 256, 320, 250, 310
    maps = []
    yy, xx = np.mgrid[0:hp, 0:wp]
    for k in range(4):
        p = rng.random((hp, wp), dtype=np.float32) * 0.25
        for _ in range(14):
            cx, cy = rng.uniform(0, wp), rng.uniform(0, hp)
            ang = rng.uniform(-1.5, 1.5)
            hl, hw = rng.uniform(8, 90), rng.uniform(2, 14)
            u = (xx - cx) * np.cos(ang) + (yy - cy) * np.sin(ang)
            v = -(xx - cx) * np.sin(ang) + (yy - cy) * np.cos(ang)
            p[(np.abs(u) < hl) & (np.abs(v) < hw)] = rng.uniform(0.5, 1.0)
        if k == 3:  # speckle: thousands of tiny components, more than max_candidates
            p[rng.random((hp, wp)) > 0.97] = 0.9
        maps.append(p)
    prob = np.stack(maps)
    bits = arch.f32_to_bf16_bits(prob)
    pd = torch.from_numpy(bits.view(np.int16)).cuda().view(torch.bfloat16)
    boxes, scores, counts = engine.det_postprocess(pd, vh, vw)
    torch.cuda.synchronize()
    total = 0
    for i in range(len(maps)):
        rb, rs, ncomp = dbpost.db_postprocess(bits[i], vh, vw)
        n = int(counts[i])
        assert n == len(rb), (i, n, len(rb), ncomp)
        assert np.array_equal(boxes[i, :n].cpu().numpy(), rb), i
        assert np.array_equal(scores[i, :n].cpu().numpy(), rs), i
        total += n
    assert total > 5


def _post_vs_oracle(engine, prob, vh, vw):
    from oracle import dbpost
    bits = arch.f32_to_bf16_bits(prob)
    pd = torch.from_numpy(bits.view(np.int16)).cuda().view(torch.bfloat16)
    boxes, scores, counts = engine.det_postprocess(pd, vh, vw)
    torch.cuda.synchronize()
    total = 0
    for i in range(prob.shape[0]):
        rb, rs, ncomp = dbpost.db_postprocess(bits[i], vh, vw)
        n = int(counts[i])
        assert n == len(rb), (i, n, len(rb), ncomp)
        assert np.array_equal(boxes[i, :n].cpu().numpy(), rb), i
        assert np.array_equal(scores[i, :n].cpu().numpy(), rs), i
        total += n
    return total


def test_db_postprocess_run_list_edge_cases(engine):
    """The components are found over row RUNS (dbpost.hip): the shapes that stress a run list — a run per other pixel (the worst case
    the run buffer is sized for), runs that span and end exactly on 64-pixel segment borders, a width that is not a multiple of 64,
    staircases that only touch diagonally, a comb whose teeth merge in the LAST row (late unions), empty and full maps."""
    hp, wp, vh, vw = 96, 200, 90, 197
    yy, xx = np.mgrid[0:hp, 0:wp]
    maps = []
    maps.append(np.where((xx + yy) % 2 == 0, 0.9, 0.0))                                  # checkerboard: one 8-connected component, a run per 2 pixels
    maps.append(np.where(xx % 2 == 0, 0.9, 0.0) * (yy % 3 != 2))                           # vertical 1-pixel bars, broken every third row: thousands of small components
    seg = np.zeros((hp, wp)); seg[10:14, 0:64] = 0.9; seg[20:24, 63:129] = 0.9; seg[30:34, 64:128] = 0.9; seg[40:44, 1:192] = 0.9; seg[50:54, 127:197] = 0.9
    maps.append(seg)                                                                         # runs ending / starting / spanning segment borders
    stair = np.zeros((hp, wp))
    for k in range(40):
        stair[5 + k, 10 + 3 * k:13 + 3 * k] = 0.9                                            # touches the row above only through a corner
    maps.append(stair)
    comb = np.zeros((hp, wp)); comb[5:80, 4:180:6] = 0.9; comb[79, 4:180] = 0.9               # teeth joined by the bottom row only
    maps.append(comb)
    maps.append(np.zeros((hp, wp)))
    maps.append(np.full((hp, wp), 0.9))
    rng = np.random.default_rng(9)
    maps.append(np.where(rng.random((hp, wp)) > 0.5, 0.9, 0.0))                              # dense noise: long union chains
    total = _post_vs_oracle(engine, np.stack(maps).astype(np.float32), vh, vw)
    assert total >= 5   # (most of these maps are one big component; the bar map's pieces are below min_size)


def test_db_postprocess_wide_map(engine):
    """More than 64 segments per row (the run-fill kernel walks the segments in groups of 64 lanes)."""
    hp, wp = 32, 4160
    rng = np.random.default_rng(4)
    p = np.zeros((2, hp, wp), np.float32)
    for k in range(60):
        x0, y0 = int(rng.integers(0, wp - 300)), int(rng.integers(0, hp - 8))
        p[k % 2, y0:y0 + int(rng.integers(3, 8)), x0:x0 + int(rng.integers(20, 300))] = 0.8
    p[0, 12:16, 4000:4160] = 0.8; p[1, 3:9, 4090:4100] = 0.9
    assert _post_vs_oracle(engine, p, hp, wp) > 10


def test_fused_head_equals_unfused(engine, any_det_weights):
    """head.convt3 fused into head.convt2's epilogue must give the same bf16 map as the two-launch path (dense weights: every
    one of the 64 channels of head.convt2 feeds the map)."""
    pages = torch.from_numpy(_pages(2, 160, 224, 21)).cuda()
    engine.load_det(any_det_weights)
    engine.set_option("fuse_head", 1)
    a = engine.det_forward(pages).clone()
    engine.set_option("fuse_head", 0)
    b = engine.det_forward(pages).clone()
    engine.set_option("fuse_head", 1)
    torch.cuda.synchronize()
    d = (a.float() - b.float()).abs()
    # the unfused path rounds the 64-channel half-resolution tensor to bf16 and sums the 64-term dot in another order: a few values
    # land one bf16 ulp of a probability (<= 2^-8) apart
    assert float(d.max()) <= 2.0 ** -8 and float((d > 0).float().mean()) < 2e-2, (float(d.max()), float((d > 0).float().mean()))


@pytest.mark.parametrize("shape", [(2, 250, 200), (1, 447, 901), (3, 96, 130)], ids=lambda s: "b%d_%dx%d" % s)
def test_fused_stem_pool_is_bit_identical(engine, any_det_weights, shape):
    """stem.conv3 with the 3x3/s2 max pool in its epilogue (overlapping 16x32 conv tiles, pooled 7x15 per tile, image borders,
    partial tiles) must reproduce the conv + maxpool kernel pair exactly: same conv arithmetic, max picks existing values.
    conv_big_min=1 makes the un-fused stem.conv3 run the same 16-channel-chunk kernel family as the fused one (the 8x32-tile
    kernel sums its 32 input channels as one chunk: a different fp32 order, 1 bf16 ulp apart on ~2e-5 of the values — found
    by this test when it started to compare the taps instead of the channel-0 probability map)."""
    b, h, w = shape
    pages = torch.from_numpy(_pages(b, h, w, 33)).cuda()
    engine.load_det(any_det_weights)
    engine.set_option("conv_big_min", 1)
    try:
        engine.set_option("fuse_pool", 1)
        a = _forward_all(engine, pages)
        engine.set_option("fuse_pool", 0)
        ref = _forward_all(engine, pages)
    finally:
        engine.set_option("fuse_pool", 1)
        engine.set_option("conv_big_min", 1024)
    _assert_same(a, ref, "fuse_pool")


@pytest.mark.parametrize("shape", [(2, 250, 200), (1, 447, 901), (3, 96, 130), (1, 33, 35)], ids=lambda s: "b%d_%dx%d" % s)
def test_fused_stem_convs_are_bit_identical(engine, any_det_weights, shape):
    """stem.conv1 + stem.conv2 in one kernel (the 32-channel half-resolution tensor only exists as a 10x34-pixel LDS tile per
    work-group) must reproduce the two-kernel path exactly: same bf16 rounding of the intermediate, same chunk->tap summation
    order, conv2's zero padding at the map borders, pages whose size is not a multiple of the tile."""
    b, h, w = shape
    pages = torch.from_numpy(_pages(b, h, w, 35)).cuda()
    engine.load_det(any_det_weights)
    engine.set_option("fuse_stem", 1)
    a = _forward_all(engine, pages)
    engine.set_option("fuse_stem", 0)
    ref = _forward_all(engine, pages)
    engine.set_option("fuse_stem", 1)
    _assert_same(a, ref, "fuse_stem")


@pytest.mark.parametrize("orient", [-1, 0, 1], ids=["auto", "rows", "transposed"])
@pytest.mark.parametrize("shape", [(2, 250, 200), (1, 447, 901), (5, 352, 512), (1, 1000, 330)], ids=lambda s: "b%d_%dx%d" % s)
def test_ring_conv_kernel_is_bit_identical(engine, any_det_weights, shape, orient):
    """The persistent LDS-DMA ring kernel (conv_ring.hip: tiles strided over resident work-groups, register epilogue, stores
    deferred behind the next tile's DMA, either tile orientation) sums the same products in the same order as the
    one-tile-per-work-group kernel it replaces: the probability maps must be equal bit for bit (residual and plain layers,
    channel-offset output into the FPN concat buffer, partial tiles, one or many tiles per work-group)."""
    b, h, w = shape
    pages = torch.from_numpy(_pages(b, h, w, 41)).cuda()
    engine.load_det(any_det_weights)
    engine.set_option("conv_big_min", 1)      # the 16x32-tile kernels on every layer that has them, however small the page
    engine.set_option("fpn_compose", 0)       # (kernels are compared here, not definitions: fpn.p2 in its two-step form on both sides)
    try:
        engine.set_option("conv_ring", 0)
        ref = _forward_all(engine, pages)
        engine.set_option("conv_ring", 1)
        engine.set_option("ring_orient", orient)
        engine.conv_timing_detail()
        engine.set_option("time_convs", 1)
        a = _forward_all(engine, pages)
        names = [k for _, k, *_ in engine.conv_timing_detail()]
    finally:
        engine.set_option("ring_orient", -1)
        engine.set_option("conv_big_min", 1024)
        engine.set_option("time_convs", 0)
        engine.set_option("fpn_compose", 1)
    assert sum(k.startswith("conv_ring_kernel") for k in names) >= 10, names
    _assert_same(a, ref, "ring vs one-tile")


def test_ring_conv_kernel_on_random_page_shapes(engine, any_det_weights):
    """Ten seeded random page shapes (1-3 pages, 64-700 px sides: maps from 2x2 to 175x175 pixels, tiles per work-group from a
    fraction to many, every partial-tile remainder) through the detector with the ring kernel on every layer that has it,
    against the one-tile-per-work-group kernels."""
    rng = np.random.default_rng(2025)
    engine.load_det(any_det_weights)
    engine.set_option("conv_big_min", 1)
    engine.set_option("fpn_compose", 0)       # (two-step fpn.p2 on both sides, as above)
    try:
        for _ in range(10):
            b, h, w = int(rng.integers(1, 4)), int(rng.integers(64, 700)), int(rng.integers(64, 700))
            pages = torch.from_numpy(_pages(b, h, w, int(rng.integers(1 << 30)))).cuda()
            engine.set_option("conv_ring", 0)
            ref = _forward_all(engine, pages)
            engine.set_option("conv_ring", 1)
            a = _forward_all(engine, pages)
            _assert_same(a, ref, "ring, page batch %dx%dx%d" % (b, h, w))
    finally:
        engine.set_option("conv_ring", 1)
        engine.set_option("conv_big_min", 1024)
        engine.set_option("fpn_compose", 1)


@pytest.mark.parametrize("shape", [(2, 250, 200), (1, 447, 901)], ids=lambda s: "b%d_%dx%d" % s)
def test_channel_blocked_stage0_tensors_are_bit_identical(engine, any_det_weights, shape):
    """Engine option blocked_layout (an experiment): the stage-0 tensors that only the ring kernel reads and
    writes are stored [n][C/16][H][W][16] instead of NHWC — input halo, residual and output addressing change, the
    arithmetic does not."""
    b, h, w = shape
    pages = torch.from_numpy(_pages(b, h, w, 43)).cuda()
    engine.load_det(any_det_weights)
    engine.set_option("conv_big_min", 1)
    try:
        engine.set_option("blocked_layout", 0)
        ref = engine.det_forward(pages).clone()
        engine.set_option("blocked_layout", 1)
        a = engine.det_forward(pages).clone()
    finally:
        engine.set_option("blocked_layout", 0)
        engine.set_option("conv_big_min", 1024)
    torch.cuda.synchronize()
    assert torch.equal(a, ref)


def test_page_result_is_independent_of_batch_size(engine, any_det_weights):
    """The kernel variant of every layer is chosen from the layer geometry and the configured sub-batch, never from the number
    of pages in the call (different tilings sum the same products in a different order): one page alone == the same page in a batch."""
    pages = torch.from_numpy(_pages(5, 352, 512, 77)).cuda()
    engine.load_det(any_det_weights)
    batch = engine.det_forward(pages).clone()
    single = engine.det_forward(pages[2:3]).clone()
    torch.cuda.synchronize()
    assert torch.equal(batch[2:3], single)
