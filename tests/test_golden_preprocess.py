"""CPU: the oracle's restatement of the reference pre-processing vs vectors produced by the reference module itself
(tools/make_golden.py imported /root/reference/backend/utils/image_preprocessing.py)."""
import hashlib
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import preprocess as P
from lumina_ocr import synth
from lumina_ocr.utils.image_preprocessing import ImagePreprocessor, get_optimal_size

G = Path(__file__).parent / "golden"


def test_size_table_matches_reference():
    for w, h, ow, oh, err in json.loads((G / "resize_sizes.json").read_text()):
        assert P.target_size(w, h) == (ow, oh)
        assert get_optimal_size(w, h) == (ow, oh)                      # product host mirror
        assert ImagePreprocessor().get_optimal_size(w, h) == (ow, oh)
        if err:  # degenerate target: the reference raises ValueError inside PIL
            assert ow == 0 or oh == 0


@pytest.mark.parametrize("i", range(7))
def test_pixel_vectors(i):
    v = np.load(G / "preprocess_vectors.npz")
    x = v[f"in{i}"]
    assert np.array_equal(P.resize_if_needed(x, 120), v[f"resize{i}"])
    assert np.array_equal(P.enhance_contrast(x, 1.2), v[f"contrast{i}"])
    assert np.array_equal(P.enhance_sharpness(x, 1.1), v[f"sharp{i}"])
    assert np.array_equal(P.optimize_for_ocr(x, 120), v[f"optimize{i}"])


@pytest.mark.parametrize("i", range(7))
def test_optional_steps_grayscale_and_denoise(i):
    """optimize_for_ocr's optional steps (image_preprocessing.py:160-169, :225-231): oracle restatement vs the reference's own output,
    alone and inside the whole chain (resize -> [grayscale] -> [denoise] -> contrast -> sharpness)."""
    v = np.load(G / "preprocess_vectors.npz")
    x = v[f"in{i}"]
    assert np.array_equal(P.grayscale(x), v[f"gray{i}"])
    assert np.array_equal(P.denoise(x), v[f"denoise{i}"])
    r = P.resize_if_needed(x, 120)
    dn = P.enhance_sharpness(P.enhance_contrast(P.denoise(r), 1.2), 1.1)
    assert np.array_equal(dn, v[f"optimize_dn{i}"])
    dg = P.enhance_sharpness(P.enhance_contrast(P.denoise(P.grayscale(r)), 1.2), 1.1)
    assert np.array_equal(dg, v[f"optimize_dn_gray{i}"]) and dg.ndim == 2


def test_a4_200dpi_page_hashes():
    """BASELINE's page shape: 1654x2339 -> 1414x2000 (int truncation), pinned by SHA-256 of the reference's output."""
    a4 = json.loads((G / "a4_page.json").read_text())
    page = synth.synth_page(2339, 1654, 2024)[0]
    assert list(page.shape) == a4["in_shape"] and hashlib.sha256(page.tobytes()).hexdigest() == a4["in_sha256"]
    res = P.resize_if_needed(page)
    assert list(res.shape) == a4["out_shape"] == [2000, 1414, 3]
    y, x = a4["crop_origin"]
    assert np.array_equal(res[y:y + 64, x:x + 64], np.array(a4["resize_crop"], np.uint8))
    assert hashlib.sha256(res.tobytes()).hexdigest() == a4["resize_sha256"]
    opt = P.enhance_sharpness(P.enhance_contrast(res, 1.2), 1.1)
    assert hashlib.sha256(opt.tobytes()).hexdigest() == a4["optimize_sha256"]


def test_binarize_restatement_matches_the_reference_vectors():
    """`binarize` (image_preprocessing.py:175-185) — also what the reference's `adaptive_binarize` (:462-494) returns in this container,
    where OpenCV is absent (:473-475): both captured by tools/make_golden.py."""
    from oracle import preprocess as op
    g = np.load(Path(__file__).parent / "golden" / "preprocess_vectors.npz")
    for i in range(7):
        assert np.array_equal(op.binarize(g["in%d" % i]), g["binarize%d" % i])
        assert np.array_equal(g["adaptive_nocv%d" % i], g["binarize%d" % i])


def test_adaptive_binarize_restatement_against_an_independent_gaussian():
    """cv2.adaptiveThreshold(GAUSSIAN_C, 11, 2) restated (parity unpinned: no OpenCV offline).  Cross-check of the restatement's
    indexing / borders / threshold rule with scipy's separable Gaussian correlation (another summation order: the rounded mean may
    differ by one grey level on a few pixels)."""
    from scipy import ndimage
    from oracle import preprocess as op
    g = np.load(Path(__file__).parent / "golden" / "preprocess_vectors.npz")
    for i in (0, 2, 6):
        img = g["in%d" % i]
        L = op.gray_L(img).astype(np.float32)
        m = ndimage.correlate1d(ndimage.correlate1d(L, op.GAUSS11, axis=1, mode="nearest"), op.GAUSS11, axis=0, mode="nearest")
        ref = np.where(L - np.clip(np.rint(m), 0, 255) > -2, 255, 0).astype(np.uint8)
        got = op.adaptive_binarize(img)
        assert got.shape == ref.shape and (got != ref).mean() < 0.005, float((got != ref).mean())
    assert abs(float(op.GAUSS11.sum()) - 1.0) < 1e-6 and np.array_equal(op.GAUSS11, op.GAUSS11[::-1])
