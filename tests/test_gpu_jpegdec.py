"""GPU: the device JPEG decoder (csrc/jpegdec.hip, through lumina_ocr_jpeg_decode) vs Pillow and the oracle, byte for byte:
the 24 seeded files of tests/jpeg_cases.py (4:4:4 / 4:2:2 / 4:2:0 / grey, odd and tiny sizes, default and optimised Huffman tables,
restart intervals, qualities 5 .. 100), batches of same-size files, the engine's OWN encoder's output, an A4 page, unsupported and
corrupt files reported per page.  Reference: Image.open in /root/reference/backend/utils/image_preprocessing.py:57-75."""
import io

import numpy as np
import pytest
import torch
from PIL import Image

from jpeg_cases import CASES, UNSUPPORTED, make_file, pil_decode
from lumina_ocr import synth
from lumina_ocr.engine import Engine

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_device_decode_equals_pillow(engine, case):
    from oracle import jpeg as oj
    data = make_file(case)
    rc, info = Engine.jpeg_probe(data)
    assert rc == 0 and (info["width"], info["height"]) == (case[2], case[3])
    out, status = engine.jpeg_decode([data], case[3], case[2])
    torch.cuda.synchronize()
    assert status == [0]
    got, want = out[0].cpu().numpy(), pil_decode(data)
    if not np.array_equal(got, want):
        d = np.argwhere(got != want)
        raise AssertionError("%s: %d of %d values differ, first at %s: got %s want %s" % (case[0], len(d), got.size, d[0].tolist(), got[tuple(d[0])], want[tuple(d[0])]))
    assert np.array_equal(got, oj.decode(data))


def test_batch_of_pages_with_different_tables_and_sampling(engine):
    """One call, eight same-size files that share nothing else: sampling, quality, Huffman tables, restart intervals, grey."""
    rng = np.random.default_rng(11)
    w, h = 333, 211
    files = []
    for k, kw in enumerate([dict(quality=95, optimize=True), dict(quality=40, subsampling=0), dict(quality=75, subsampling=1), dict(quality=88, restart_marker_blocks=5),
                            dict(quality=92, subsampling=2, optimize=True), dict(quality=20), dict(quality=99, subsampling=0), dict(quality=80)]):
        arr = synth.synth_page(h, w, 50 + k, n_lines=5)[0] if k % 2 else rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        im = Image.fromarray(arr)
        if k == 7:
            im = im.convert("L")
        buf = io.BytesIO()
        im.save(buf, format="JPEG", **kw)
        files.append(buf.getvalue())
    out, status = engine.jpeg_decode(files, h, w)
    torch.cuda.synchronize()
    assert status == [0] * 8
    for k, f in enumerate(files):
        assert np.array_equal(out[k].cpu().numpy(), pil_decode(f)), k


def test_a4_page_written_by_the_engines_own_encoder(engine):
    """A4 @ 200 DPI through the device encoder (byte-identical to Pillow's file, tests/test_gpu_jpeg.py) and back through the device
    decoder: equal to Pillow's decode of that file; ~1 MB of entropy-coded data in 256-byte chunks (thousands of chunk decoders)."""
    page = synth.synth_page(2339, 1654, 2024)[0]
    d = torch.from_numpy(page)[None].cuda()
    files, sizes = engine.jpeg_encode(d, 95, max_bytes=8 << 20)
    n = int(sizes[0])
    assert n > 0
    data = files[0, :n].cpu().numpy().tobytes()
    out, status = engine.jpeg_decode([data, data], 2339, 1654)
    torch.cuda.synchronize()
    assert status == [0, 0]
    want = pil_decode(data)
    assert np.array_equal(out[0].cpu().numpy(), want) and torch.equal(out[0], out[1])
    assert np.abs(want.astype(int) - page.astype(int)).mean() < 3.0          # (and it is the page, up to JPEG loss)


def test_unsupported_corrupt_and_wrong_size_files_are_reported_per_page(engine):
    good = make_file(CASES[0])                     # 64 x 48
    prog, cmyk = make_file(UNSUPPORTED[0]), make_file(UNSUPPORTED[1])
    assert Engine.jpeg_probe(prog)[0] == -2 and Engine.jpeg_probe(cmyk)[0] == -2 and Engine.jpeg_probe(b"nope")[0] == -1
    other = make_file(CASES[3])                    # 37 x 53
    broken = good[: len(good) // 2] + bytes(len(good) - len(good) // 2 - 2) + b"\xff\xd9"       # the second half of the scan zeroed
    out = torch.zeros((6, 48, 64, 3), dtype=torch.uint8, device="cuda")
    out, status = engine.jpeg_decode([good, prog, other, b"nope", broken, good], 48, 64, out=out)
    torch.cuda.synchronize()
    assert status[0] == 0 and status[5] == 0 and status[1] == -2 and status[2] == -4 and status[3] == -1 and status[4] == -1, status
    want = pil_decode(good)
    assert np.array_equal(out[0].cpu().numpy(), want) and np.array_equal(out[5].cpu().numpy(), want)
    assert int(out[1].sum()) == 0 and int(out[2].sum()) == 0 and int(out[3].sum()) == 0     # pages that were not decoded are not written


def test_async_decode_equals_the_synchronous_one_and_reports_too_few_passes(engine):
    """lumina_ocr_jpeg_decode_async: no host synchronisation inside the call; the pinned status is valid after the stream has run.
    Same bytes as the synchronous form; with too few passes a busy file reports -5 instead of wrong pixels going unnoticed."""
    rng = np.random.default_rng(4)
    files = []
    for k in range(3):
        buf = io.BytesIO()
        Image.fromarray(rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)).save(buf, format="JPEG", quality=95)
        files.append(buf.getvalue())
    files.append(make_file(UNSUPPORTED[0]))
    ref, st_ref = engine.jpeg_decode(files, 480, 640)
    out, st = engine.jpeg_decode_async(files, 480, 640, passes=16)
    out2, st2 = engine.jpeg_decode_async(files, 480, 640, passes=16)          # back to back: the staging buffers alternate
    torch.cuda.synchronize()
    assert st_ref == [0, 0, 0, -2] and st.tolist() == [0, 0, 0, -2] and st2.tolist() == [0, 0, 0, -2]
    assert torch.equal(out[:3], ref[:3]) and torch.equal(out2[:3], ref[:3])
    few, st_few = engine.jpeg_decode_async(files, 480, 640, passes=2)
    torch.cuda.synchronize()
    assert st_few.tolist()[3] == -2 and -5 in st_few.tolist()[:3]             # ~100 KB of noise per file: two passes cannot synchronise ~100 chunks
