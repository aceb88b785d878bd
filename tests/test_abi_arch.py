"""CPU: the C-ABI library loads and exports every symbol include/lumina_ocr.h declares (no compute calls);
architecture tables / weight container / oracle C restatement known answers."""
import ctypes
import re
from pathlib import Path

import numpy as np

from lumina_ocr import arch, engine
from lumina_ocr.dist import shard_range

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    hdr = (ROOT / "include" / "lumina_ocr.h").read_text()
    declared = sorted(set(re.findall(r"\b(lumina_ocr_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 20
    lib = engine.load_library()
    assert lib._missing == []
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(engine.EXPORTED_SYMBOLS) == declared
    assert b"gfx950" in lib.lumina_ocr_version()
    assert lib.lumina_ocr_last_error(None) == b"null handle"


def test_blob_roundtrip_and_mac_count():
    w = arch.make_det_weights(1)
    back = arch.read_blob(arch.write_blob(w))
    assert set(back) == set(w) and all(np.array_equal(back[k], w[k]) for k in w)
    assert arch.det_macs_per_page(1024, 1024) == 68828528640        # 68.83 GMAC
    assert arch.det_macs_per_page(2016, 1440) == 190555545600       # 190.6 GMAC per padded A4 page
    x = np.array([1.0, 1.00390625, 1.01171875, -3.14159], np.float32)
    assert np.array_equal(arch.bf16_bits_to_f32(arch.f32_to_bf16_bits(x)), arch.bf16_round(x))
    assert arch.bf16_round(x)[1] == 1.0 and arch.bf16_round(x)[2] == 1.015625   # ties to even
    cs = arch.ctc_charset()
    assert len(cs) == 6625 and cs[0] == "\x00" and cs[-1] == " " and len(set(cs)) == 6625


def test_shard_range_partitions():
    for n in (0, 1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_oracle_dbpost_known_answers():
    from oracle import dbpost
    p = np.zeros((64, 96), np.float32)
    p[10:20, 5:60] = 0.9                     # axis-aligned 55 x 10 bar: wd=54, wn=9 -> E = 1.5*54*9/(2*63) = 5.786
    p[40:44, 70:74] = 0.95                   # 4x4 blob: short side 3 -> kept by min_size, unclip E = 1.5*9/12=1.125 -> short side 5.25 >= 5
    p[30, 30] = 0.99                         # single pixel: skipped
    p[50:60, 10:40] = 0.45                   # above thresh, below box_thresh 0.6: dropped
    boxes, scores, ncomp = dbpost.db_postprocess(arch.f32_to_bf16_bits(p), 64, 96)
    assert ncomp == 4
    assert boxes.tolist() == [[0, 4, 65, 4, 65, 25, 0, 25], [69, 39, 74, 39, 74, 44, 69, 44]]
    assert np.allclose(scores, [0.8984375, 0.94921875])
    # empty / all-foreground / valid-region clipping
    assert len(dbpost.db_postprocess(arch.f32_to_bf16_bits(np.zeros((32, 32), np.float32)), 32, 32)[0]) == 0
    b, s, n = dbpost.db_postprocess(arch.f32_to_bf16_bits(np.ones((32, 64), np.float32)), 20, 50)
    assert n == 1 and b.tolist() == [[0, 0, 50, 0, 50, 20, 0, 20]]
    # rotation invariance of the short-side / score logic: a 45-degree bar is found with a rotated quad
    yy, xx = np.mgrid[0:96, 0:96]
    bar = (np.abs((xx - 48) - (yy - 48)) <= 4) & (np.abs((xx - 48) + (yy - 48)) <= 50)
    b, s, n = dbpost.db_postprocess(arch.f32_to_bf16_bits(np.where(bar, 0.9, 0.0).astype(np.float32)), 96, 96)
    assert n == 1 and len(b) == 1
    q = b[0].reshape(4, 2)
    assert abs((q[1] - q[0])[0]) == abs((q[1] - q[0])[1])          # edges at 45 degrees


def test_oracle_crop_identity_and_rotation():
    from oracle import dbpost
    rng = np.random.default_rng(0)
    page = rng.integers(0, 256, (64, 400, 3), dtype=np.uint8)
    crop, wc = dbpost.rec_crop(page, [0, 0, 320, 0, 320, 32, 0, 32])      # 320x32 region -> 1:1 sampling at pixel centres + 0.5
    assert wc == 320
    crop2, wc2 = dbpost.rec_crop(page, [10, 0, 42, 0, 42, 64, 10, 64])    # tall box (h/w = 2 >= 1.5): rotated 90 degrees
    assert wc2 == 64 and crop2[:, 64:].max() == 0
    _, wc3 = dbpost.rec_crop(page, [5, 5, 5, 5, 5, 5, 5, 5])
    assert wc3 == 0
