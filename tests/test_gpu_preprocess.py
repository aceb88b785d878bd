"""GPU parity (bit-exact, byte work): LANCZOS resize / contrast / sharpness kernels vs the reference's own outputs
(tests/golden/, produced by /root/reference/backend/utils/image_preprocessing.py) and vs the oracle restatement."""
import hashlib
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from lumina_ocr import synth
from lumina_ocr.utils.image_preprocessing import ImagePreprocessor

pytestmark = pytest.mark.gpu
G = Path(__file__).parent / "golden"


def _dev(x):
    x = x if x.ndim == 3 else x[..., None]
    return torch.from_numpy(np.ascontiguousarray(x))[None].cuda()


@pytest.mark.parametrize("i", range(7))
def test_golden_vectors_bit_exact(engine, i):
    v = np.load(G / "preprocess_vectors.npz")
    x = v[f"in{i}"]
    pre = ImagePreprocessor(max_dimension=120, engine=engine)
    r = pre.resize_if_needed(_dev(x))[0].cpu().numpy()
    ref = v[f"resize{i}"]
    assert np.array_equal(r if x.ndim == 3 else r[..., 0], ref)
    if x.ndim == 3:
        assert np.array_equal(engine.enhance(_dev(x), 1.2, 1.0)[0].cpu().numpy(), v[f"contrast{i}"])
        assert np.array_equal(engine.enhance(_dev(x), 1.0, 1.1)[0].cpu().numpy(), v[f"sharp{i}"])
        assert np.array_equal(pre.optimize_for_ocr(_dev(x))[0].cpu().numpy(), v[f"optimize{i}"])


def test_a4_page_matches_reference_hashes(engine):
    a4 = json.loads((G / "a4_page.json").read_text())
    page = synth.synth_page(2339, 1654, 2024)[0]
    assert hashlib.sha256(page.tobytes()).hexdigest() == a4["in_sha256"]
    pre = ImagePreprocessor(engine=engine)
    res = pre.resize_if_needed(_dev(page))
    assert list(res.shape[1:]) == [2000, 1414, 3]
    assert hashlib.sha256(res[0].cpu().numpy().tobytes()).hexdigest() == a4["resize_sha256"]
    opt = engine.enhance(res, 1.2, 1.1)
    assert hashlib.sha256(opt[0].cpu().numpy().tobytes()).hexdigest() == a4["optimize_sha256"]


def test_batch_and_edge_sizes_vs_oracle(engine):
    from oracle import preprocess as P
    rng = np.random.default_rng(9)
    x = rng.integers(0, 256, (3, 90, 260, 3), dtype=np.uint8)
    got = engine.resize_lanczos(torch.from_numpy(x).cuda(), 69, 200).cpu().numpy()      # batch of 3, downscale
    for k in range(3):
        assert np.array_equal(got[k], P.resize_lanczos(x[k], 200, 69))
    up = engine.resize_lanczos(torch.from_numpy(x[:1]).cuda(), 180, 300).cpu().numpy()[0]  # upscale (filterscale clamps to 1)
    assert np.array_equal(up, P.resize_lanczos(x[0], 300, 180))
    tiny = rng.integers(0, 256, (1, 2, 5, 3), dtype=np.uint8)                             # < 3 px: sharpen copies borders
    assert np.array_equal(engine.enhance(torch.from_numpy(tiny).cuda(), 1.2, 1.1).cpu().numpy()[0],
                          P.enhance_sharpness(P.enhance_contrast(tiny[0], 1.2), 1.1))
    flat = np.full((1, 40, 40, 3), 255, np.uint8)                                          # saturated page: clipping branches
    assert np.array_equal(engine.enhance(torch.from_numpy(flat).cuda(), 1.2, 1.1).cpu().numpy()[0],
                          P.enhance_sharpness(P.enhance_contrast(flat[0], 1.2), 1.1))


@pytest.mark.parametrize("shape", [(3, 1, 5), (2, 2, 6), (3, 3, 7), (3, 9, 8), (2, 17, 345), (2, 11, 683), (3, 21, 342), (1, 5, 1369)])
def test_enhance_alignment_cases_vs_oracle(engine, shape):
    """Rows of W*3 bytes start at every 4-byte misalignment; batches make image starts misaligned too; strips > 1024 bytes."""
    from oracle import preprocess as P
    n, h, w = shape
    rng = np.random.default_rng(h * 1000 + w)
    x = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    got = engine.enhance(torch.from_numpy(x).cuda(), 1.2, 1.1).cpu().numpy()
    for k in range(n):
        assert np.array_equal(got[k], P.enhance_sharpness(P.enhance_contrast(x[k], 1.2), 1.1)), (shape, k)
    # a view that starts at an odd byte address (sub-tensor of a larger allocation)
    big = torch.from_numpy(np.concatenate([np.zeros(1, np.uint8), x.reshape(-1)])).cuda()
    view = big[1:].view(n, h, w, 3)
    got2 = engine.enhance(view, 1.2, 1.1).cpu().numpy()
    assert np.array_equal(got2, got)


@pytest.mark.parametrize("case", [(2, 37, 61, 29, 50), (1, 64, 343, 55, 300), (3, 50, 90, 41, 77), (1, 200, 130, 20, 13), (2, 33, 47, 70, 99),
                                  (1, 19, 1400, 17, 1250)])
def test_resize_alignment_cases_vs_oracle(engine, case):
    """Both passes at every row misalignment, batches, the LDS-tiled and the fallback (strong down-scale) vertical kernels."""
    from oracle import preprocess as P
    n, h, w, oh, ow = case
    rng = np.random.default_rng(h * 977 + w)
    x = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    got = engine.resize_lanczos(torch.from_numpy(x).cuda(), oh, ow).cpu().numpy()
    for k in range(n):
        assert np.array_equal(got[k], P.resize_lanczos(x[k], ow, oh)), (case, k)
    big = torch.from_numpy(np.concatenate([np.zeros(3, np.uint8), x.reshape(-1)])).cuda()
    got2 = engine.resize_lanczos(big[3:].view(n, h, w, 3), oh, ow).cpu().numpy()
    assert np.array_equal(got2, got)


def test_exif_transpose_equals_pillow_for_every_orientation(engine):
    """auto_orient (image_preprocessing.py:171-173 = ImageOps.exif_transpose) on the device: all eight EXIF orientations, odd sizes, a batch."""
    import io
    from PIL import Image, ImageOps
    rng = np.random.default_rng(8)
    pages = rng.integers(0, 256, (3, 37, 53, 3), dtype=np.uint8)
    d = torch.from_numpy(pages).cuda()
    for o in range(1, 9):
        got = engine.exif_transpose(d, o).cpu().numpy()
        for k in range(3):
            ex = Image.Exif()
            ex[0x0112] = o
            buf = io.BytesIO()
            Image.fromarray(pages[k]).save(buf, format="PNG", exif=ex)
            buf.seek(0)
            want = np.asarray(ImageOps.exif_transpose(Image.open(buf)))
            assert got[k].shape == want.shape and np.array_equal(got[k], want), (o, k)


def test_grayscale_and_denoise_are_byte_exact_with_the_reference(engine):
    """convert_to_grayscale / denoise (image_preprocessing.py:160-169) and optimize_for_ocr(apply_denoise=True, grayscale=True) (:191-242)
    on the device vs the reference module's own outputs (tests/golden/preprocess_vectors.npz: gray*, denoise*, optimize_dn*, optimize_dn_gray*)."""
    from lumina_ocr.utils.image_preprocessing import ImagePreprocessor
    g = np.load(Path(__file__).parent / "golden" / "preprocess_vectors.npz")
    pre = ImagePreprocessor(max_dimension=120, engine=engine)
    for i in range(7):
        x = g["in%d" % i]
        rgb = x if x.ndim == 3 else np.repeat(x[..., None], 3, axis=2)          # (an L input: the provider's page format is three-channel)
        d = torch.from_numpy(np.ascontiguousarray(rgb))[None].cuda()
        gray = engine.grayscale(d)[0].cpu().numpy()
        assert np.array_equal(gray[..., 0], g["gray%d" % i]) and np.array_equal(gray[..., 0], gray[..., 1]) and np.array_equal(gray[..., 0], gray[..., 2])
        dn = engine.denoise(d)[0].cpu().numpy()
        want = g["denoise%d" % i]
        assert np.array_equal(dn if want.ndim == 3 else dn[..., 0], want), i
        if x.ndim == 3:
            assert np.array_equal(pre.optimize_for_ocr(d, apply_denoise=True)[0].cpu().numpy(), g["optimize_dn%d" % i]), i
        full = pre.optimize_for_ocr(d, apply_denoise=True, grayscale=True)[0].cpu().numpy()
        assert np.array_equal(full[..., 0], g["optimize_dn_gray%d" % i]) and np.array_equal(full[..., 0], full[..., 2]), i
    # a batch of odd-sized pages: every page on its own
    rng = np.random.default_rng(3)
    pages = rng.integers(0, 256, (3, 37, 131, 3), dtype=np.uint8)
    from oracle import preprocess as op
    out = engine.denoise(torch.from_numpy(pages).cuda()).cpu().numpy()
    for k in range(3):
        assert np.array_equal(out[k], op.denoise(pages[k]))


def test_binarize_simple_is_the_reference_and_adaptive_matches_its_restatement(engine):
    """image_preprocessing.py:175-185 / :462-494.  simple (L > 128) == the reference's own output in this container (its
    adaptive_binarize falls back to it without OpenCV): pinned by tests/golden (binarize*, adaptive_nocv*).  adaptive == the
    restatement of cv2.adaptiveThreshold (oracle/preprocess.py; parity unpinned), bit for bit incl. borders and ragged tiles."""
    from oracle import preprocess as op
    g = np.load(Path(__file__).parent / "golden" / "preprocess_vectors.npz")
    for i in range(7):
        img = g["in%d" % i]
        rgb = img if img.ndim == 3 else np.stack([img] * 3, -1)
        d = torch.from_numpy(np.ascontiguousarray(rgb[None])).cuda()
        simple = engine.binarize(d, adaptive=False)[0].cpu().numpy()
        adaptive = engine.binarize(d, adaptive=True)[0].cpu().numpy()
        assert np.array_equal(simple[..., 0], simple[..., 1]) and np.array_equal(simple[..., 0], simple[..., 2])
        if img.ndim == 3:      # (an L image converted to RGB and back gives the same L: checked for the RGB cases through the reference)
            assert np.array_equal(simple[..., 0], g["binarize%d" % i]) and np.array_equal(g["adaptive_nocv%d" % i], g["binarize%d" % i])
        assert np.array_equal(simple[..., 0], op.binarize(rgb))
        assert np.array_equal(adaptive[..., 0], op.adaptive_binarize(rgb)), i
        assert np.array_equal(adaptive[..., 0], adaptive[..., 2])
    page = np.stack([synth_page_for_binarize(k) for k in range(2)])
    out = engine.binarize(torch.from_numpy(page).cuda(), adaptive=True).cpu().numpy()
    for k in range(2):
        assert np.array_equal(out[k, ..., 0], op.adaptive_binarize(page[k]))
        assert 0.5 < (out[k] == 255).mean() < 0.99         # text on paper: mostly white, ink black


def synth_page_for_binarize(k):
    from lumina_ocr import synth
    return synth.synth_page(333, 517, 40 + k, n_lines=8)[0]
