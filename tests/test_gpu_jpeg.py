"""GPU parity (bit-exact, integer work): the device JPEG encoder vs the pinned C oracle and vs Pillow — the encoder the
reference calls in compress_for_azure (/root/reference/backend/utils/image_preprocessing.py:526-538)."""
import io

import numpy as np
import pytest
import torch

from lumina_ocr import synth

pytestmark = pytest.mark.gpu


def _images(n, h, w, seed, kind):
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    return np.stack([synth.synth_page(h, w, seed + i, n_lines=max(2, h // 40))[0] for i in range(n)])


@pytest.mark.parametrize("shape", [(1, 1, 1), (2, 8, 8), (3, 37, 53), (1, 64, 48), (2, 17, 31), (1, 100, 75), (2, 250, 333)])
@pytest.mark.parametrize("quality", [95, 50])
def test_coefficients_match_oracle(engine, shape, quality):
    from oracle import jpeg as oj
    n, h, w = shape
    imgs = _images(n, h, w, 11, "noise")
    got = engine.jpeg_coefficients(torch.from_numpy(imgs).cuda(), quality).cpu().numpy()
    zz = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])
    for i in range(n):
        ref = oj.coefficients(imgs[i], quality)[:, :, zz]      # oracle: natural order -> zig-zag
        assert np.array_equal(got[i], ref), (shape, quality, i)


@pytest.mark.parametrize("case", [(1, 1, 1, "noise"), (2, 8, 8, "noise"), (3, 37, 53, "noise"), (2, 100, 75, "noise"), (2, 250, 333, "page"),
                                  (2, 640, 448, "page"), (1, 1000, 707, "page")])
@pytest.mark.parametrize("quality", [95, 85, 30])
def test_files_match_oracle_and_pillow(engine, case, quality):
    from PIL import Image
    from oracle import jpeg as oj
    n, h, w, kind = case
    imgs = _images(n, h, w, 23, kind)
    out, sizes = engine.jpeg_encode(torch.from_numpy(imgs).cuda(), quality)
    out, sizes = out.cpu().numpy(), sizes.cpu().numpy()
    for i in range(n):
        assert sizes[i] > 0
        got = out[i, : sizes[i]].tobytes()
        assert got == oj.encode(imgs[i], quality), (case, quality, i)
        b = io.BytesIO()
        Image.fromarray(imgs[i]).save(b, format="JPEG", quality=quality, optimize=True)
        assert got == b.getvalue(), (case, quality, i)


def test_too_small_output_reports_negative_size(engine):
    imgs = _images(1, 200, 300, 5, "noise")
    out, sizes = engine.jpeg_encode(torch.from_numpy(imgs).cuda(), 95, max_bytes=4096)
    assert int(sizes[0]) < 0
