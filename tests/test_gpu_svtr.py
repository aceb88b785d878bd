"""GPU parity: the SVTR-Tiny recogniser (BASELINE configs[4] family) through the C ABI vs the oracle restatement
(oracle/nets.py svtr_forward, mode bf16).  Parity unpinned: no reference implementation or weights exist offline, the
restatement defines the arithmetic; tolerances below bound the drift of fp32 summation-order differences through 12 blocks."""
import numpy as np
import pytest
import torch

from conftest import close_stats
from lumina_ocr import arch, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def svtr_weights():
    return arch.make_svtr_weights()


def _crops(n, seed):
    rng = np.random.default_rng(seed)
    return np.stack([synth.synth_crop(rng)[0] for _ in range(n)])


# share of a tap's values within 4 ulps of the storage type: (embedding + first two blocks, deeper taps)
FP_LO4 = {"bf16": (0.97, 0.7), "f16": (0.97, 0.7)}


@pytest.mark.parametrize("variant,dtype", [("tiny", "bf16"), ("tiny", "f16"), ("base", "bf16"), ("base", "f16")])
def test_svtr_forward_taps(engine, variant, dtype):
    """Every tap of the model (patch embedding + positional embedding, all mixing blocks, both merging stages, the sequence) vs the
    oracle in the SAME storage type (bf16 or fp16: BASELINE configs[4] asks for fp16 MFMA); Tiny and Base dimensions."""
    from oracle import nets
    weights = arch.make_svtr_weights(variant=variant, dtype=dtype, num_classes=500)
    crops = _crops(5, 777)
    widths = np.array([320, 211, 320, 77, 150], np.int32)
    engine.load_svtr(weights)
    assert engine.svtr_dtype == dtype and engine.svtr_num_classes == 500
    engine.set_option("keep_taps", 1)
    idx, prob = engine.svtr_forward(torch.from_numpy(crops).cuda(), torch.from_numpy(widths).cuda())
    torch.cuda.synchronize()
    taps = {}
    ridx, rprob, logits, seq = nets.svtr_forward(weights, crops, None, taps, widths=widths)
    nblocks = sum(arch.SVTR_VARIANTS[variant]["depths"])
    stats = {}
    for name in ["svtr.embed"] + ["svtr.b%d" % i for i in range(nblocks)] + ["svtr.sub0", "svtr.sub1", "svtr.seq"]:
        got = engine.read_tap(name, dtype).reshape(taps[name].shape)
        st = stats[name] = close_stats(got, taps[name], dtype)
        # LayerNorm keeps magnitudes at O(1); one-ulp differences of fp32 summation order accumulate over the blocks.  The ulp is
        # the STORAGE type's: 2^-7 relative for bf16, 2^-10 for fp16 (an fp16 tensor graded in bf16 ulps would pass 8x too easily).
        early = name in ("svtr.embed", "svtr.b0", "svtr.b1")
        lo4 = FP_LO4[dtype][0 if early else 1]
        rel = (0.01 if early else 0.02) * (1.0 if dtype == "bf16" else 0.25)
        assert st["within4"] > lo4 and st["mean_abs"] < rel * max(st["ref_mean_abs"], 1e-3), (name, st)
    engine.set_option("keep_taps", 0)
    try:
        import json, os
        os.makedirs("gpurun_out", exist_ok=True)
        json.dump(stats, open("gpurun_out/parity_svtr_%s_%s.json" % (variant, dtype), "w"), indent=1)
    except OSError:
        pass
    agree = float((idx.cpu().numpy() == ridx).mean())
    assert agree > (0.9 if dtype == "bf16" else 0.97), agree
    top2 = np.partition(logits, -2, axis=2)[:, :, -2:]
    clear = (top2[:, :, 1] - top2[:, :, 0]) > 1.0        # every step with a clear top-1 / top-2 logit margin: same class id
    assert np.array_equal(idx.cpu().numpy()[clear], ridx[clear])
    same = idx.cpu().numpy() == ridx
    rel = float(np.abs(prob.cpu().numpy()[same] - rprob[same]).mean() / max(rprob[same].mean(), 1e-9))
    assert rel < 0.05, rel


def test_svtr_storage_type_override_and_bad_config(engine):
    """Option svtr_f16 overrides the blob's type; an inconsistent svtr.config is an error, not a crash."""
    w = arch.make_svtr_weights(variant="tiny", dtype="bf16", num_classes=200)
    engine.load_svtr(w, f16=True)
    assert engine.svtr_dtype == "f16"
    engine.load_svtr(w)
    assert engine.svtr_dtype == "bf16"
    bad = dict(w)
    bad["svtr.config"] = np.array([64, 128, 256, 3, 6, 3, 3, 4, 8, 6, 192, 0], np.float32)     # 3 heads of 32 != 64 channels
    from lumina_ocr.engine import EngineError
    with pytest.raises(EngineError):
        engine.load_svtr(bad)
    engine.load_svtr(w)


def test_svtr_batch_invariance_and_pipeline(engine, svtr_weights, det_weights):
    """A crop's result cannot depend on its batch slot or on the sub-batch split; the pipeline switches recognisers."""
    from lumina_ocr.pipeline import OcrPipeline
    base = _crops(8, 31)
    crops = torch.from_numpy(np.concatenate([base] * 5)).cuda()
    engine.load_svtr(svtr_weights)
    idx, prob = engine.svtr_forward(crops)
    engine.set_option("rec_sub_batch", 12)            # -> sub-batches of 6 crops
    idx2, prob2 = engine.svtr_forward(crops)
    engine.set_option("rec_sub_batch", 4096)
    assert torch.equal(idx, idx2) and torch.equal(prob, prob2)
    for r in range(1, 5):
        assert torch.equal(idx[:8], idx[8 * r:8 * r + 8])
    engine.load_det(det_weights)
    pages = torch.from_numpy(np.stack([synth.synth_page(560, 800, 90 + i, n_lines=10)[0] for i in range(2)])).cuda()
    pipe = OcrPipeline(engine, max_dimension=800, post=arch.TEXT_PATH_POST, recognizer="svtr")
    dets, _ = pipe.run(pages)
    assert sum(len(d.texts) for d in dets) > 10 and all(isinstance(t, str) for d in dets for t in d.texts)
