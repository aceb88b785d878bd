"""CPU: the oracle's JPEG decoder (oracle/csrc/jpegdec_oracle.c) vs Pillow — the decoder behind the reference's load_image /
load_image_bytes (/root/reference/backend/utils/image_preprocessing.py:57-75) — byte for byte, on files Pillow itself wrote from
seeded images: 4:4:4 / 4:2:2 / 4:2:0 / grey, odd and tiny sizes, optimised and default Huffman tables, restart intervals,
qualities 5 .. 100; and the committed digests of those files and of their decoded pixels (tools/make_golden.py)."""
import hashlib
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import jpeg as oj
from jpeg_cases import CASES, UNSUPPORTED, make_file, pil_decode

G = Path(__file__).parent / "golden" / "jpegdec_digests.json"


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_oracle_decode_equals_pillow(case):
    data = make_file(case)
    rc, info = oj.info(data)
    assert rc == 0 and (info["width"], info["height"]) == (case[2], case[3])
    want = pil_decode(data)
    got = oj.decode(data)
    assert got.shape == want.shape
    if not np.array_equal(got, want):
        d = np.argwhere(got != want)
        raise AssertionError("%s: %d of %d values differ, first at %s: got %s want %s" % (case[0], len(d), got.size, d[0].tolist(), got[tuple(d[0])], want[tuple(d[0])]))
    pinned = json.loads(G.read_text())[case[0]]
    assert hashlib.sha256(data).hexdigest() == pinned["file_sha256"], "Pillow writes a different file than the one the digests were made from"
    assert hashlib.sha256(want.tobytes()).hexdigest() == pinned["rgb_sha256"]


def test_restart_interval_files_really_carry_markers():
    for case in CASES:
        if "restart_marker_blocks" in case[5] or "restart_marker_rows" in case[5]:
            data = make_file(case)
            assert oj.info(data)[1]["restart"] > 0 and any(bytes([0xFF, 0xD0 + k]) in data for k in range(8)), case[0]


@pytest.mark.parametrize("case", UNSUPPORTED, ids=[c[0] for c in UNSUPPORTED])
def test_unsupported_files_are_reported_not_misdecoded(case):
    rc, _ = oj.info(make_file(case))
    assert rc == -2
    with pytest.raises(ValueError):
        oj.decode(make_file(case))


def test_corrupt_and_truncated_files():
    data = make_file(CASES[1])
    assert oj.info(b"not a jpeg")[0] == -1 and oj.info(data[:100])[0] == -1
    with pytest.raises(ValueError):
        oj.decode(data[: len(data) // 2][:-2] + b"\xff\xd9" if False else data[:200])


def test_coefficients_round_trip_through_the_encoder_oracle():
    """decode_coefficients(encode(x)) == the encoder oracle's own quantised coefficients (4:2:0, MCU scan order)."""
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)
    data = oj.encode(img, 90)
    want = oj.coefficients(img, 90).reshape(-1, 64)
    assert np.array_equal(oj.decode_coefficients(data), want)
