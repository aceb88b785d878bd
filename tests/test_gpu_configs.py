"""GPU parity at BASELINE.json's full configuration sizes (configs[1]..[3]) through size-independent properties
(batch invariance, idempotence) plus a direct oracle comparison on a bounded sample of the same inputs."""
import numpy as np
import pytest
import torch

from conftest import close_stats
from lumina_ocr import arch, synth
from lumina_ocr.pipeline import OcrPipeline

pytestmark = pytest.mark.gpu


def test_config2_det_batch32_1024(engine, det_weights):
    """DBNet-R18 detection only, batch=32 1024x1024 synthetic pages."""
    from oracle import nets
    base = np.stack([synth.synth_page(1024, 1024, 1234 + i, n_lines=24)[0] for i in range(4)])
    pages = torch.from_numpy(np.concatenate([base] * 8)).cuda()            # 32 pages, 4 distinct
    engine.load_det(det_weights)
    engine.set_option("det_sub_batch", 16)                                 # the bench's sub-batch (the kernel choice depends on it)
    engine.set_option("keep_taps", 2)                                      # every fusion stays on; taps of the last sub-batch (pages 16..31)
    engine.set_option("time_convs", 1)
    engine.conv_timing_detail()
    try:
        prob = engine.det_forward(pages)
        torch.cuda.synchronize()
        names = [k for _, k, *_ in engine.conv_timing_detail()]
        got = {name: engine.read_tap(name)[0] for name in ("s0.b1", "s3.b1", "fpn.p3", "fpn.p2", "head.conv1")}   # page 16 == base[0]
    finally:
        engine.set_option("keep_taps", 0)
        engine.set_option("time_convs", 0)
    assert sum(k.startswith("conv_ring_kernel") for k in names) >= 2 * 10, names   # the ring kernel is what runs at this size
    assert any(k.startswith("conv_ring_kernel") and k.endswith(",8>") for k in names), names   # ... its 8-wave tile on the 64-channel stage
    assert prob.shape == (32, 1024, 1024)
    bits = prob.view(torch.int16)
    for r in range(1, 8):                                                  # a page's result cannot depend on its batch slot
        assert torch.equal(bits[:4], bits[4 * r:4 * r + 4])
    taps = {}
    ref = nets.det_forward(det_weights, base[:1], mode="bf16", taps=taps)  # one full-size page against the oracle, taps included
    for name, g in got.items():
        s_ = close_stats(g, taps[name][0])
        assert s_["within4"] > (0.85 if name.startswith("fpn.p") else 0.90) and s_["mean_abs"] < 0.01 * max(s_["ref_mean_abs"], 1e-3), (name, s_)
    st = close_stats(prob[0].float().cpu().numpy(), ref[0])
    assert st["within4"] > 0.98 and st["max_abs"] < 0.06, st
    boxes, scores, counts = engine.det_postprocess(prob, 1024, 1024, **arch.TEXT_PATH_POST)
    c = counts.cpu().numpy()
    assert (c[:4] > 10).all() and all((c[:4] == c[4 * r:4 * r + 4]).all() for r in range(1, 8))
    assert torch.equal(boxes[:4], boxes[4:8])


def test_config3_rec_batch512(engine, rec_weights):
    """CRNN-MobileNetV3 recognition + CTC greedy, batch=512 32x320 line crops."""
    from oracle import nets
    rng = np.random.default_rng(4321)
    base = np.stack([synth.synth_crop(rng)[0] for _ in range(64)])
    crops = torch.from_numpy(np.concatenate([base] * 8)).cuda()            # 512 crops, 64 distinct
    engine.load_rec(rec_weights)
    engine.set_option("rec_sub_batch", 200)                                # ragged sub-batches: 200 + 200 + 112
    idx, prob = engine.rec_forward(crops)
    text, length, score = engine.ctc_decode(idx, prob)
    engine.set_option("rec_sub_batch", 2048)
    idx2, prob2 = engine.rec_forward(crops)
    torch.cuda.synchronize()
    assert torch.equal(idx, idx2) and torch.equal(prob, prob2)            # sub-batching is invisible
    for r in range(1, 8):
        assert torch.equal(idx[:64], idx[64 * r:64 * r + 64])
    ridx, rprob, _, _ = nets.rec_forward(rec_weights, base[:8])
    agree = float((idx[:8].cpu().numpy() == ridx).mean())
    assert agree > 0.95, agree
    ref = nets.ctc_greedy(idx[:8].cpu().numpy(), prob[:8].cpu().numpy(), arch.ctc_charset())   # decode is exact on the same ids
    cs = arch.ctc_charset()
    for i in range(8):
        got = "".join(cs[k] for k in text[i, : int(length[i])].cpu().tolist())
        assert got == ref[i][0] and np.float32(score[i].item()) == np.float32(ref[i][1])


def test_config3_rec_batch512_strings_equal_the_oracle(engine, code_rec_weights):
    """The same configuration with the code-path weight set (trained-like margins): per-crop STRING EQUALITY with the oracle
    on all 64 distinct crops of the batch of 512, in ragged sub-batches."""
    from oracle import nets
    rng = np.random.default_rng(8642)
    base = np.stack([synth.synth_crop(rng)[0] for _ in range(64)])
    crops = torch.from_numpy(np.concatenate([base] * 8)).cuda()
    engine.load_rec(code_rec_weights)
    engine.set_option("rec_sub_batch", 200)
    idx, prob = engine.rec_forward(crops)
    text, length, score = engine.ctc_decode(idx, prob)
    torch.cuda.synchronize()
    engine.set_option("rec_sub_batch", 2048)
    ridx, rprob, logits, _ = nets.rec_forward(code_rec_weights, base)
    top2 = np.partition(logits, -2, axis=2)[:, :, -2:]
    assert (top2[:, :, 1] - top2[:, :, 0]).min() > 8.0
    cs = arch.ctc_charset()
    ref = [r[0] for r in nets.ctc_greedy(ridx, rprob, cs)]
    got = arch.TextDecoder(cs).decode(text.cpu().numpy(), length.cpu().numpy())
    assert got == ref * 8
    assert np.array_equal(idx.cpu().numpy(), np.concatenate([ridx] * 8))


def test_config4_end_to_end_a4_pages_and_detector_recall(engine, det_weights, rec_weights):
    """End-to-end det+rec on A4@200DPI pages: batch invariance + the hand-set text path really finds the rendered lines."""
    page, gt = synth.synth_page(2339, 1654, 2024)
    pages = torch.from_numpy(np.stack([page] * 3)).cuda()
    engine.load_det(det_weights)
    engine.load_rec(rec_weights)
    pipe = OcrPipeline(engine, post=arch.TEXT_PATH_POST)
    dets, processed = pipe.run(pages)
    assert processed.shape == (3, 2000, 1414, 3)                           # the reference's 2000-px cap with int() truncation
    assert np.array_equal(dets[0].quads, dets[1].quads) and dets[0].texts == dets[2].texts
    sc = 1414 / 1654
    hit = 0
    for g in gt:                                                           # ground-truth line boxes of the synthetic page
        gx0, gy0, gx1, gy1 = [v * sc for v in g["box"]]
        for q in dets[0].quads:
            x0, y0, x1, y1 = q[0], q[1], q[4], q[5]
            ix = max(0.0, min(gx1, x1) - max(gx0, x0)); iy = max(0.0, min(gy1, y1) - max(gy0, y0))
            inter = ix * iy
            union = (gx1 - gx0) * (gy1 - gy0) + (x1 - x0) * (y1 - y0) - inter
            if union > 0 and inter / union > 0.5:
                hit += 1
                break
    assert hit >= 0.75 * len(gt), (hit, len(gt), len(dets[0].quads))


def test_config5_svtr_base_fp16_batch512_with_a_dictionary_file(engine, tmp_path):
    """BASELINE configs[4]: SVTR-BASE, fp16 MFMA, Hindi dictionary, batch 512 — batch and sub-batch invariance at size, the oracle
    (fp16 storage) on a bounded sample incl. taps of the last block of every stage, CTC decode exact on the same ids, strings through
    a dictionary FILE (PP-OCR key-file format) loaded the way the provider loads LUMINA_OCR_REC_DICT.  No Hindi dictionary, image or
    weights ship offline: the file is written from the build's own list (arch.devanagari_charset), weights are seeded."""
    from oracle import nets
    f = tmp_path / "devanagari_dict.txt"
    arch.save_charset(f, arch.devanagari_charset())
    cs = arch.load_charset(f)
    assert cs == arch.devanagari_charset()
    weights = arch.make_svtr_weights(num_classes=len(cs), variant="base", dtype="f16")
    rng = np.random.default_rng(55)
    base = np.stack([synth.synth_crop(rng)[0] for _ in range(64)])
    crops = torch.from_numpy(np.concatenate([base] * 8)).cuda()
    engine.load_svtr(weights)
    assert engine.svtr_dtype == "f16" and engine.svtr_num_classes == len(cs)
    engine.set_option("rec_sub_batch", 600)                                # SVTR-Base sub-batches of 150: 150 + 150 + 150 + 62
    idx, prob = engine.svtr_forward(crops)
    engine.set_option("rec_sub_batch", 4096)
    engine.set_option("keep_taps", 1)
    idx2, prob2 = engine.svtr_forward(crops)
    torch.cuda.synchronize()
    engine.set_option("keep_taps", 0)
    assert torch.equal(idx, idx2) and torch.equal(prob, prob2)
    for r in range(1, 8):
        assert torch.equal(idx[:64], idx[64 * r:64 * r + 64])
    assert int(idx.max()) < len(cs)
    taps = {}
    ridx, rprob, _, _ = nets.svtr_forward(weights, base[:4], None, taps)
    for name in ("svtr.b2", "svtr.b8", "svtr.b17", "svtr.seq"):           # last block of each stage + the sequence (first 4 crops)
        got = engine.read_tap(name, "f16")
        got = got.reshape((got.shape[0], -1, got.shape[-1]))[:4]
        st = close_stats(got, taps[name], "f16")                          # graded in fp16 ulps (2^-10 relative)
        assert st["within4"] > 0.7 and st["mean_abs"] < 0.005 * max(st["ref_mean_abs"], 1e-3), (name, st)
    assert float((idx[:4].cpu().numpy() == ridx).mean()) > 0.97
    text, length, score = engine.ctc_decode(idx, prob)
    ref = nets.ctc_greedy(idx[:6].cpu().numpy(), prob[:6].cpu().numpy(), cs)
    dec = arch.TextDecoder(cs)
    got = dec.decode(text[:6].cpu().numpy(), length[:6].cpu().numpy())
    for i in range(6):
        assert got[i] == ref[i][0] and np.float32(score[i].item()) == np.float32(ref[i][1])
