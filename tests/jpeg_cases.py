"""Seeded JPEG files for the decoder tests (CPU: oracle vs Pillow; GPU: device vs oracle vs Pillow).  The files are produced with
Pillow at test time — the encoder the reference's inputs come from — from seeded images; tests/golden/jpegdec_digests.json
(tools/make_golden.py) pins the SHA-256 of every file and of Pillow's decode of it, so a changed Pillow cannot move both sides at once."""
import io

import numpy as np
from PIL import Image


def _image(kind, w, h, seed):
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if kind == "smooth":
        yy, xx = np.mgrid[0:h, 0:w]
        a = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) * 255 // max(w + h - 2, 1))], -1)
        return (a + rng.integers(-6, 7, a.shape)).clip(0, 255).astype(np.uint8)
    if kind == "text":
        from lumina_ocr import synth
        return synth.synth_page(h, w, seed, n_lines=max(3, h // 40))[0]
    raise ValueError(kind)


# (name, kind, w, h, seed, save kwargs, mode)
CASES = [
    ("n444_q95", "noise", 64, 48, 1, dict(quality=95, subsampling=0), "RGB"),
    ("n420_q95", "noise", 64, 48, 2, dict(quality=95, subsampling=2), "RGB"),
    ("n422_q75", "noise", 64, 48, 3, dict(quality=75, subsampling=1), "RGB"),
    ("odd420", "noise", 37, 53, 4, dict(quality=90, subsampling=2), "RGB"),
    ("odd422", "noise", 37, 53, 5, dict(quality=90, subsampling=1), "RGB"),
    ("odd444", "smooth", 101, 67, 6, dict(quality=85, subsampling=0), "RGB"),
    ("tiny1", "noise", 1, 1, 7, dict(quality=95), "RGB"),
    ("tiny3x5", "noise", 3, 5, 8, dict(quality=95), "RGB"),
    ("narrow4", "smooth", 4, 40, 9, dict(quality=95, subsampling=2), "RGB"),          # down-sampled width 2: replication, no fancy up-sampling
    ("narrow5", "smooth", 5, 40, 10, dict(quality=95, subsampling=2), "RGB"),         # down-sampled width 3: fancy
    ("w16h16", "smooth", 16, 16, 11, dict(quality=60, subsampling=2), "RGB"),
    ("w17h17", "smooth", 17, 17, 12, dict(quality=60, subsampling=2), "RGB"),
    ("text420_q95", "text", 320, 240, 13, dict(quality=95, optimize=True), "RGB"),    # what compress_for_azure writes
    ("text420_q30", "text", 320, 240, 14, dict(quality=30, optimize=True), "RGB"),
    ("text444_q85", "text", 333, 211, 15, dict(quality=85, subsampling=0, optimize=True), "RGB"),
    ("grey_text", "text", 200, 150, 16, dict(quality=90), "L"),
    ("grey_noise_odd", "noise", 45, 31, 17, dict(quality=75, optimize=True), "L"),
    ("rst1_420", "noise", 80, 64, 18, dict(quality=90, subsampling=2, restart_marker_blocks=1), "RGB"),
    ("rst3_444", "smooth", 90, 50, 19, dict(quality=90, subsampling=0, restart_marker_blocks=3), "RGB"),
    ("rstrow_420", "text", 256, 192, 20, dict(quality=80, restart_marker_rows=1), "RGB"),
    ("q100_noise", "noise", 48, 48, 21, dict(quality=100, subsampling=2), "RGB"),     # large coefficients: 10-bit AC categories
    ("q5_noise", "noise", 64, 64, 22, dict(quality=5), "RGB"),                        # 16-bit quantisation steps stay 8-bit (<= 255): still baseline
    ("page_a5", "text", 620, 877, 23, dict(quality=95, optimize=True), "RGB"),
    ("wide", "text", 1200, 96, 24, dict(quality=92), "RGB"),
]

UNSUPPORTED = [
    ("progressive", "smooth", 64, 64, 30, dict(quality=90, progressive=True), "RGB"),
    ("cmyk", "noise", 32, 32, 31, dict(quality=90), "CMYK"),
]


def make_file(case) -> bytes:
    name, kind, w, h, seed, kw, mode = case
    arr = _image(kind, w, h, seed)
    im = Image.fromarray(arr)
    if mode == "L":
        im = im.convert("L")
    elif mode == "CMYK":
        im = im.convert("CMYK")
    buf = io.BytesIO()
    im.save(buf, format="JPEG", **kw)
    return buf.getvalue()


def pil_decode(data: bytes) -> np.ndarray:
    """What the reference's load_image_bytes hands on (image_preprocessing.py:70-75: modes RGB and L pass), as three channels."""
    im = Image.open(io.BytesIO(data))
    im.load()
    a = np.asarray(im if im.mode in ("RGB", "L") else im.convert("RGB"))
    return a if a.ndim == 3 else np.repeat(a[..., None], 3, axis=2)
