"""CPU: the recogniser's hand-set code path (arch._install_code_path) gives the ORACLE trained-like arg-max margins — the premise
of the string-equality tests on the GPU (tests/test_gpu_rec.py, test_gpu_e2e.py, test_gpu_configs.py)."""
import numpy as np

from lumina_ocr import arch, synth


def test_code_path_margins_and_variety():
    from oracle import nets
    w = arch.make_rec_weights(4321, code_path=True)
    plain = arch.make_rec_weights(4321)
    changed = [k for k in w if not np.array_equal(w[k], plain[k])]
    assert all(k.split(".")[0] in ("rec", "lstm", "ctc") for k in changed)
    # only rows of the path were touched: the rest of every tensor is the seeded-random one
    assert np.array_equal(w["ctc.fc.w"][:, 5:], plain["ctc.fc.w"][:, 5:]) and np.array_equal(w["rec.b4.expand.w"], plain["rec.b4.expand.w"])
    rng = np.random.default_rng(7)
    crops = np.stack([synth.synth_crop(rng)[0] for _ in range(12)])
    idx, prob, logits, seq = nets.rec_forward(w, crops)
    top2 = np.partition(logits, -2, axis=2)[:, :, -2:]
    margin = top2[..., 1] - top2[..., 0]
    assert (margin > 1.0).mean() >= 0.9 and margin.min() > 8.0, (float((margin > 1.0).mean()), float(margin.min()))
    assert set(np.unique(np.abs(seq[..., :5]))) == {np.float32(0.76171875)}       # saturated code bits after both LSTM layers
    cs = arch.ctc_charset()
    texts = [t for t, _ in nets.ctc_greedy(idx, prob, cs)]
    assert len(set(texts)) == 12 and all(set(t) <= set(arch.CODE_PATH_SYMBOLS) for t in texts)
    assert len(set("".join(texts))) >= 20                                         # most of the 31 code symbols occur
    # the plain seeded set for contrast: near-ties at most steps
    _, _, lp, _ = nets.rec_forward(plain, crops[:3])
    t2 = np.partition(lp, -2, axis=2)[:, :, -2:]
    assert ((t2[..., 1] - t2[..., 0]) > 1.0).mean() < 0.3
