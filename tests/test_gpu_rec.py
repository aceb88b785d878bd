"""GPU parity: CRNN forward + CTC (fused FC/argmax/softmax) + greedy collapse + crop, through the C ABI."""
import numpy as np
import pytest
import torch

from conftest import close_stats
from lumina_ocr import arch, synth

pytestmark = pytest.mark.gpu

MARGIN_EPS = 1.0   # logit units; see test_rec_forward_taps
AGREE_MIN = 0.95   # arg-max agreement of the plain seeded set (near-ties at most steps); the code-path set below is held to EQUALITY


def _crops(n, seed):
    rng = np.random.default_rng(seed)
    return np.stack([synth.synth_crop(rng)[0] for _ in range(n)])


def test_rec_forward_taps(engine, rec_weights):
    from oracle import nets
    crops = _crops(6, 4321)
    widths = np.array([320, 320, 200, 77, 320, 131], np.int32)
    crops_m = crops.copy()
    engine.load_rec(rec_weights)
    engine.set_option("keep_taps", 1)
    idx, prob = engine.rec_forward(torch.from_numpy(crops).cuda(), torch.from_numpy(widths).cuda())
    torch.cuda.synchronize()
    # oracle: columns >= width are zero in normalised space
    taps = {}
    x = nets.rec_normalize(crops_m)
    for i, wv in enumerate(widths):
        x[i, :, :, wv:] = 0
    with torch.no_grad():
        feat = nets.rec_backbone(rec_weights, x, "bf16", taps)
        ridx, rprob, logits, seq = nets.rec_head(rec_weights, feat)
    for name in ["rec.conv1"] + ["rec.b%d" % i for i in range(11)] + ["rec.conv2"]:
        got = engine.read_tap(name)
        ref = taps[name]
        got = got[..., : ref.shape[-1]]  # drop channel padding
        st = close_stats(got, ref)
        assert st["within4"] > 0.90 and st["mean_abs"] < 0.01 * max(st["ref_mean_abs"], 1e-3), (name, st)
    got = engine.read_tap("lstm.l1").reshape(6, 80, 192)
    st = close_stats(got, seq)
    # 80 recurrent steps amplify single-ulp differences of the (saturating) gate inputs: bound the drift, not the ulps
    assert st["mean_abs"] < 0.02 and st["within4"] > 0.6, st
    engine.set_option("keep_taps", 0)
    agree = float((idx.cpu().numpy() == ridx).mean())
    assert agree > AGREE_MIN, agree
    # The string-parity statement (north_star: "exact or within stated edit-distance"): the seeded network's logits (std ~2.0)
    # have near-ties at most steps, and bf16 drift through 2 x 80 recurrent steps moves a logit by <= ~0.9 (p99; measured with
    # tools/margin_probe.py: largest margin of a flipped arg-max 0.52).  EVERY time step whose oracle top-1 / top-2 logit margin
    # exceeds MARGIN_EPS must carry the same class id; the looser agreement bound above only covers the near-ties.
    top2 = np.partition(logits, -2, axis=2)[:, :, -2:]
    clear = (top2[:, :, 1] - top2[:, :, 0]) > MARGIN_EPS
    assert clear.sum() >= 10, int(clear.sum())           # (a few % of the steps of a seeded network)
    assert np.array_equal(idx.cpu().numpy()[clear], ridx[clear]), "arg-max differs on a step with a clear margin"
    same = idx.cpu().numpy() == ridx
    gp, rp = prob.cpu().numpy()[same], rprob[same]
    rel = float(np.abs(gp - rp).mean() / max(rp.mean(), 1e-9))
    try:
        import json, os
        os.makedirs("gpurun_out", exist_ok=True)
        json.dump(dict(argmax_agreement=agree, prob_mean_rel_err=rel, margin_eps=MARGIN_EPS, clear_step_share=float(clear.mean()),
                       clear_step_agreement=1.0), open("gpurun_out/parity_rec.json", "w"))
    except OSError:
        pass
    assert rel < 0.1, rel


def test_code_path_recogniser_strings_are_equal(engine, code_rec_weights):
    """North_star: "recognised strings exact".  With the code-path weight set every time step has a top-1 / top-2 margin of >= 14
    logits in the oracle (asserted), and the path that decides it is exact arithmetic: class ids, strings and the path channels
    themselves must be IDENTICAL to the oracle's — 48 crops incl. ragged widths, every one of the 80 steps."""
    from oracle import nets
    crops = _crops(48, 1357)
    widths = np.full(48, 320, np.int32)
    widths[::5] = [200, 77, 131, 33, 250, 320, 64, 301, 18, 160]
    engine.load_rec(code_rec_weights)
    engine.set_option("keep_taps", 1)
    idx, prob = engine.rec_forward(torch.from_numpy(crops).cuda(), torch.from_numpy(widths).cuda())
    text, length, score = engine.ctc_decode(idx, prob)
    torch.cuda.synchronize()
    x = nets.rec_normalize(crops)
    for i, wv in enumerate(widths):
        x[i, :, :, wv:] = 0
    taps = {}
    with torch.no_grad():
        feat = nets.rec_backbone(code_rec_weights, x, "bf16", taps)
        ridx, rprob, logits, seq = nets.rec_head(code_rec_weights, feat)
    top2 = np.partition(logits, -2, axis=2)[:, :, -2:]
    margin = top2[:, :, 1] - top2[:, :, 0]
    assert margin.min() > 8.0, float(margin.min())                        # every step is clear (MARGIN_EPS = 1.0 with a lot of room)
    # the path channels are bit-identical in every tensor they cross (exact arithmetic: one non-zero product per contraction)
    for name, nch in [("rec.conv1", 1), ("rec.b0", 1), ("rec.b1", 5), ("rec.b3", 5), ("rec.b7", 5), ("rec.b8", 5), ("rec.b10", 5), ("rec.conv2", 5)]:
        assert np.array_equal(engine.read_tap(name)[..., :nch], taps[name][..., :nch]), name
    got_seq = engine.read_tap("lstm.l1").reshape(48, 80, 192)
    engine.set_option("keep_taps", 0)
    assert np.array_equal(got_seq[..., :5], seq[..., :5])                 # the five code bits after both LSTM layers: +-0.76171875
    assert set(np.unique(np.abs(seq[..., :5]))) == {np.float32(0.76171875)}
    assert np.array_equal(idx.cpu().numpy(), ridx)                        # all 48 x 80 class ids
    cs = arch.ctc_charset()
    ref = nets.ctc_greedy(ridx, rprob, cs)
    got = arch.TextDecoder(cs).decode(text.cpu().numpy(), length.cpu().numpy())
    assert got == [r[0] for r in ref]
    assert len(set(got)) >= 40 and sum(" " in g for g in got) >= 10        # real variety: distinct strings, with spaces
    assert np.abs(score.cpu().numpy() - np.array([r[1] for r in ref], np.float32)).max() < 1e-4


def test_fused_expand_depthwise_is_bit_identical(engine, rec_weights):
    """The fused expand+depthwise kernel (expanded tensor in LDS) and the unfused conv + depthwise pair compute the same
    arithmetic in the same order: every block output and the final argmax / max-prob must match bit for bit."""
    crops = _crops(9, 99)
    widths = np.array([320, 320, 200, 77, 320, 131, 33, 320, 250], np.int32)
    engine.load_rec(rec_weights)
    engine.set_option("keep_taps", 1)
    outs = []
    for fuse in (1, 0):
        engine.set_option("fuse_mb", fuse)
        idx, prob = engine.rec_forward(torch.from_numpy(crops).cuda(), torch.from_numpy(widths).cuda())
        torch.cuda.synchronize()
        outs.append(([engine.read_tap("rec.b%d" % i).copy() for i in range(11)], idx.cpu().numpy(), prob.cpu().numpy()))
    engine.set_option("fuse_mb", 1)
    engine.set_option("keep_taps", 0)
    for i, (a, b) in enumerate(zip(outs[0][0], outs[1][0])):
        assert np.array_equal(a, b), "rec.b%d" % i
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])


def test_ctc_fc_argmax_exact_on_same_sequence(engine, rec_weights):
    """The fused FC+argmax+softmax kernel vs numpy on the engine's own bf16 LSTM output (same inputs)."""
    crops = _crops(5, 99)
    engine.load_rec(rec_weights)
    engine.set_option("keep_taps", 1)
    idx, prob = engine.rec_forward(torch.from_numpy(crops).cuda())
    torch.cuda.synchronize()
    seq = engine.read_tap("lstm.l1").reshape(-1, 192).astype(np.float64)
    engine.set_option("keep_taps", 0)
    logits = seq @ rec_weights["ctc.fc.w"].astype(np.float64).T + rec_weights["ctc.fc.b"].astype(np.float64)
    ref_idx = logits.argmax(1)
    srt = np.sort(logits, axis=1)
    clear = (srt[:, -1] - srt[:, -2]) > 1e-4  # fp32 accumulation can only flip near-ties
    got = idx.cpu().numpy().reshape(-1)
    assert clear.mean() > 0.95
    assert np.array_equal(got[clear], ref_idx[clear])
    ref_p = 1.0 / np.exp(logits - logits.max(1, keepdims=True)).sum(1)
    assert np.allclose(prob.cpu().numpy().reshape(-1)[clear], ref_p[clear], rtol=2e-4)


def test_ctc_decode_bit_exact(engine):
    from oracle import nets
    rng = np.random.default_rng(0)
    n, t = 37, 80
    idx = rng.integers(0, 6, (n, t)).astype(np.int32)  # many blanks and repeats
    idx[0] = 0
    idx[1] = 3
    idx[2, ::2] = 0
    prob = rng.random((n, t), dtype=np.float32)
    text, length, score = engine.ctc_decode(torch.from_numpy(idx).cuda(), torch.from_numpy(prob).cuda())
    torch.cuda.synchronize()
    cs = [chr(65 + i) for i in range(6)]
    ref = nets.ctc_greedy(idx, prob, cs)
    for i in range(n):
        ln = int(length[i])
        got = "".join(cs[k] for k in text[i, :ln].cpu().tolist())
        assert got == ref[i][0]
        assert np.float32(score[i].item()) == np.float32(ref[i][1]), (i, score[i].item(), ref[i][1])
        assert (text[i, ln:] == -1).all()


def test_rec_crop_bit_exact(engine):
    from oracle import dbpost
    rng = np.random.default_rng(1)
    pages = rng.integers(0, 256, (2, 180, 260, 3), dtype=np.uint8)
    quads = np.array([[10, 20, 200, 24, 199, 50, 9, 46], [30, 10, 60, 12, 50, 170, 20, 168], [0, 0, 260, 0, 260, 180, 0, 180],
                      [100, 100, 140, 100, 140, 108, 100, 108], [250, 170, 270, 172, 268, 190, 248, 188], [5, 5, 5, 5, 5, 5, 5, 5]], np.int32)
    pidx = np.array([0, 1, 1, 0, 1, 0], np.int32)
    crops, widths = engine.rec_crop(torch.from_numpy(pages).cuda(), torch.from_numpy(quads).cuda(), torch.from_numpy(pidx).cuda())
    torch.cuda.synchronize()
    for i in range(len(quads)):
        ref, wc = dbpost.rec_crop(pages[pidx[i]], quads[i])
        assert int(widths[i]) == wc, (i, int(widths[i]), wc)
        assert np.array_equal(crops[i].cpu().numpy(), ref), i
