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


@pytest.mark.parametrize("quality", [95, 30])
def test_standard_table_mode_matches_pillow(engine, quality):
    """optimize=0: the Annex K.3 tables — the reference's size probe `image.save(buffer, 'JPEG', quality=min_quality)` (:548)."""
    from PIL import Image
    from oracle import jpeg as oj
    imgs = _images(2, 123, 211, 31, "page")
    out, sizes = engine.jpeg_encode(torch.from_numpy(imgs).cuda(), quality, optimize=False)
    out, sizes = out.cpu().numpy(), sizes.cpu().numpy()
    for i in range(2):
        b = io.BytesIO()
        Image.fromarray(imgs[i]).save(b, format="JPEG", quality=quality)
        got = out[i, : sizes[i]].tobytes()
        assert got == b.getvalue() and got == oj.encode(imgs[i], quality, optimize=False)


def test_compress_for_azure_device_mirrors_the_reference_loop(engine):
    """Quality loop 95 -> 30 and the resize fallback (image_preprocessing.py:495-557) on the device vs the same loop run with PIL."""
    from PIL import Image
    from lumina_ocr.utils.image_preprocessing import ImagePreprocessor
    pre = ImagePreprocessor(engine=engine)
    rng = np.random.default_rng(8)
    page = synth.synth_page(400, 560, 3, n_lines=10)[0]
    noisy = np.clip(page.astype(np.int16) + rng.normal(0, 12, page.shape), 0, 255).astype(np.uint8)
    batch = np.stack([page, noisy])
    dev = torch.from_numpy(batch).cuda()
    full = [len(pre.compress_for_azure(Image.fromarray(b))) for b in batch]
    for target in (2.0, max(full) * 0.7 / 2 ** 20, min(full) * 0.45 / 2 ** 20, 0.004):   # fits / lower quality / mixed / resize fallback
        got = pre.compress_for_azure_device(dev, target_size_mb=target)
        for i in range(2):
            ref = pre.compress_for_azure(Image.fromarray(batch[i]), target_size_mb=target)
            assert got[i] == ref, (target, i, len(got[i]), len(ref))
