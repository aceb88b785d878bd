"""CPU: the JPEG oracle restatement (oracle/csrc/jpeg_oracle.c) is pinned, byte for byte, to the encoder the reference
calls — Pillow's image.save(buffer, format='JPEG', quality=q, optimize=True)
(/root/reference/backend/utils/image_preprocessing.py:526-538) — on committed digests (tests/golden/jpeg_digests.json,
written by tools/make_golden.py) and live against the Pillow of this environment."""
import hashlib
import io
import json
from pathlib import Path

import numpy as np
import pytest

from lumina_ocr import synth
from oracle import jpeg

G = Path(__file__).parent / "golden"


def _image(kind, h, w, seed):
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if kind == "ramp":
        return np.ascontiguousarray((np.linspace(0, 255, w)[None, :, None] * np.ones((h, 1, 3))).astype(np.uint8))
    return synth.synth_page(h, w, seed, n_lines=max(2, h // 40))[0]


def test_oracle_matches_committed_pillow_digests():
    cases = json.loads((G / "jpeg_digests.json").read_text())["cases"]
    assert len(cases) >= 20
    for c in cases:
        out = jpeg.encode(_image(c["kind"], c["h"], c["w"], c["seed"]), c["quality"])
        assert len(out) == c["size"] and hashlib.sha256(out).hexdigest() == c["sha256"], c


@pytest.mark.parametrize("shape", [(1, 1), (8, 8), (16, 16), (17, 31), (37, 53), (64, 48), (100, 75), (15, 200), (250, 333)])
def test_oracle_matches_pillow_live(shape):
    from PIL import Image
    h, w = shape
    for kind in ("noise", "ramp", "page"):
        if kind == "page" and min(h, w) < 32:
            continue
        img = _image(kind, h, w, 3)
        for q in (95, 85, 50, 30):
            b = io.BytesIO()
            Image.fromarray(img).save(b, format="JPEG", quality=q, optimize=True)
            assert jpeg.encode(img, q) == b.getvalue(), (shape, kind, q)


def test_quality_loop_mirrors_reference():
    page = synth.synth_page(600, 420, 5, n_lines=14)[0]
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(page).save(b, format="JPEG", quality=95, optimize=True)
    assert jpeg.compress_for_azure(page) == b.getvalue()          # fits 2 MB at the first quality, like the reference's loop
    small = jpeg.compress_for_azure(page, target_size_mb=len(b.getvalue()) * 0.6 / (1024 * 1024))
    assert len(small) <= len(b.getvalue()) * 0.6
