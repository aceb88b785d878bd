"""GPU parity: de-skew (the reference's image_preprocessing.py:372-460: Canny + Hough segments + median angle + cubic warp) through
the C ABI vs oracle/csrc/deskew_oracle.c.  Integer / fixed-point / correctly rounded fp64 throughout: everything is BIT-EXACT —
edge map, the SET of segments, (sin, cos, flag), the warped bytes.  "Parity unpinned" w.r.t. OpenCV (absent offline): the oracle's
header states what is restated from OpenCV and the one documented deviation (deterministic instead of random-order Hough walk)."""
import numpy as np
import pytest
import torch
from PIL import Image

from lumina_ocr import synth
from lumina_ocr.engine import Engine

pytestmark = pytest.mark.gpu


def _skewed(kind, h, w, seed, angle):
    page = synth.synth_form_page(seed)[0] if kind == "form" else synth.synth_page(h, w, seed, n_lines=max(4, h // 45))[0]
    if kind == "form":
        page = np.asarray(Image.fromarray(page).resize((w, h), Image.BILINEAR))
    if angle:
        page = np.asarray(Image.fromarray(page).rotate(angle, resample=Image.BICUBIC, fillcolor=(255, 255, 255)))
    return np.ascontiguousarray(page)


def _segset(segs, nsegs):
    out = []
    for p in range(segs.shape[0]):
        for k in range(int(nsegs[p])):
            out.append(tuple(int(v) for v in segs[p, k]))
    return sorted(out)


CASES = [("text", 700, 1000, 1, 3.0), ("text", 1000, 1414, 2, -2.0), ("form", 545, 1000, 0, 1.2), ("text", 600, 800, 3, 0.0),
         ("text", 640, 480, 4, 0.3), ("form", 545, 1000, 0, -7.5), ("text", 333, 517, 5, 12.0), ("text", 900, 700, 6, 50.0)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s_%dx%d_s%d_rot%g" % c)
def test_deskew_every_stage_is_bit_exact(engine, case):
    from oracle import deskew as od
    kind, h, w, seed, ang = case
    page = _skewed(kind, h, w, seed, ang)
    out, rot, info, edges, segs, nsegs = engine.deskew(torch.from_numpy(page[None]).cuda(), debug=True)
    torch.cuda.synchronize()
    ref_edges = od.canny(page)
    assert np.array_equal(edges[0].cpu().numpy(), ref_edges), "Canny edge map"
    ref_segs, ref_peaks = od.segments(ref_edges)
    got = _segset(segs[0].cpu().numpy(), nsegs[0].cpu().numpy())
    assert got == sorted(tuple(int(v) for v in s) for s in ref_segs), "segment set"
    assert info[0].tolist() == [len(ref_segs), ref_peaks]
    ref_rot = od.angle(ref_segs)
    assert np.array_equal(rot[0].cpu().numpy(), ref_rot), (rot[0].tolist(), ref_rot.tolist())      # sin, cos, flag: same doubles
    ref_out, ref_angle, ref_info = od.deskew(page)
    assert np.array_equal(out[0].cpu().numpy(), ref_out), "warped page"
    assert Engine.skew_degrees(rot) == [ref_angle]
    if abs(ang) >= 1.0 and abs(ang) < 40:     # the estimate is the skew that was put in (to the Hough resolution), and it is undone
        assert abs(ref_angle + ang) < 0.6 and int(ref_rot[2]) == 3, (ref_angle, ang)
        again = engine.deskew(out, estimate_only=True)[1]
        assert abs(Engine.skew_degrees(again)[0]) < 0.5


@pytest.mark.parametrize("hw", [(100, 130), (257, 321), (64, 64), (16, 200), (513, 65)])
def test_hysteresis_on_noise_crosses_every_tile_border(engine, hw):
    """The hysteresis is a union-find inside 16x64 tiles (in LDS, in the Canny kernel) plus joins across tile borders: smoothed noise
    gives weak / strong edge chains that wander across all of them, on sizes that are and are not multiples of the tile."""
    from oracle import deskew as od
    h, w = hw
    rng = np.random.default_rng(h * 1000 + w)
    base = rng.random((h // 4 + 2, w // 4 + 2))
    img = np.asarray(Image.fromarray((base * 255).astype(np.uint8)).resize((w, h), Image.BICUBIC), np.float32)
    img = np.clip(img + rng.normal(0, 6, (h, w)), 0, 255).astype(np.uint8)
    page = np.ascontiguousarray(np.repeat(img[..., None], 3, axis=2))
    pages = np.stack([page, np.ascontiguousarray(page[::-1, ::-1])])
    edges = engine.deskew(torch.from_numpy(pages).cuda(), debug=True)[3]
    torch.cuda.synchronize()
    for i in range(2):
        ref = od.canny(pages[i])
        assert np.array_equal(edges[i].cpu().numpy(), ref), (hw, i, int((ref > 0).sum()))
    assert int((od.canny(pages[0]) > 0).sum()) > h * w // 50


@pytest.mark.parametrize("hw", [(1, 1), (5, 7), (300, 7), (3, 1000), (7, 64), (2, 130)])
def test_strips_narrower_than_a_tile(engine, hw):
    """Degenerate pages (a side shorter than the 16x64 tile, down to one pixel): every stage still agrees with the oracle."""
    from oracle import deskew as od
    h, w = hw
    rng = np.random.default_rng(h * 31 + w)
    page = np.ascontiguousarray((rng.integers(0, 2, (h, w, 1)) * 200 + rng.integers(0, 40, (h, w, 3))).astype(np.uint8))
    out, rot, info, edges, segs, nsegs = engine.deskew(torch.from_numpy(page[None].copy()).cuda(), debug=True)
    torch.cuda.synchronize()
    ref_edges = od.canny(page)
    assert np.array_equal(edges[0].cpu().numpy(), ref_edges)
    ref_segs, ref_peaks = od.segments(ref_edges)
    assert _segset(segs[0].cpu().numpy(), nsegs[0].cpu().numpy()) == sorted(tuple(int(v) for v in s) for s in ref_segs)
    assert info[0].tolist() == [len(ref_segs), ref_peaks]
    assert np.array_equal(rot[0].cpu().numpy(), od.angle(ref_segs))
    assert np.array_equal(out[0].cpu().numpy(), od.deskew(page)[0])


def test_deskew_flags_skip_small_and_large_angles_and_blank_pages(engine):
    """< 0.5 degrees: unchanged, angle reported; > 45: unchanged, angle 0.0 (:441-447); no line at all: unchanged (:409-411)."""
    from oracle import deskew as od
    blank = np.full((300, 400, 3), 255, np.uint8)
    steep = _skewed("text", 900, 700, 6, 50.0)          # text lines at 50 degrees: folded to ~-40 by the reference's normalisation
    flat = _skewed("text", 600, 800, 3, 0.0)
    for page, want in ((blank, 0), (flat, 1)):
        out, rot = engine.deskew(torch.from_numpy(page[None]).cuda())
        torch.cuda.synchronize()
        assert int(rot[0, 2]) == want and np.array_equal(out[0].cpu().numpy(), page)
    out, rot = engine.deskew(torch.from_numpy(steep[None]).cuda())
    _, ref_angle, ref_info = od.deskew(steep)
    assert int(rot[0, 2]) == ref_info["flag"] and Engine.skew_degrees(rot) == [ref_angle]


def test_deskew_batch_and_page_groups_are_invisible(engine):
    """Pages of a batch are independent; the page-group size that bounds the workspace does not change a byte."""
    pages = np.stack([_skewed("text", 500, 700, 10 + i, a) for i, a in enumerate((2.0, 0.0, -4.0, 1.0, 7.0))])
    pd = torch.from_numpy(pages).cuda()
    a, ra = engine.deskew(pd)
    engine.set_option("post_group", 2)
    try:
        b, rb = engine.deskew(pd)
    finally:
        engine.set_option("post_group", 64)
    single, rs = engine.deskew(pd[2:3].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(ra, rb) and torch.equal(a[2:3], single) and torch.equal(ra[2:3], rs)


@pytest.mark.parametrize("angle", [0.7, -3.3, 20.0, -44.0, 89.0])
def test_warp_alone_matches_oracle_for_any_angle(engine, angle):
    from oracle import deskew as od
    rng = np.random.default_rng(int(abs(angle) * 10))
    page = rng.integers(0, 256, (211, 317, 3), dtype=np.uint8)       # noise: every tap of every phase matters
    s, c = np.sin(np.radians(angle)), np.cos(np.radians(angle))
    rot = torch.tensor([[s, c, 3.0]], dtype=torch.float64).cuda()
    out = engine.deskew_warp(torch.from_numpy(page[None]).cuda(), rot)
    torch.cuda.synchronize()
    assert np.array_equal(out[0].cpu().numpy(), od.warp(page, float(s), float(c)))


def test_full_size_a4_page_round_trip(engine):
    """BASELINE size (A4@200DPI after the 2000-px cap: 2000x1414): skew in -> estimated -> undone (size-independent property),
    and the de-skewed page is what the oracle produces."""
    from oracle import deskew as od
    page = _skewed("text", 2000, 1414, 2024, 1.7)
    out, rot = engine.deskew(torch.from_numpy(page[None]).cuda())
    torch.cuda.synchronize()
    ang = Engine.skew_degrees(rot)[0]
    assert abs(ang + 1.7) < 0.5 and int(rot[0, 2]) == 3
    again = engine.deskew(out, estimate_only=True)[1]
    assert abs(Engine.skew_degrees(again)[0]) < 0.5
    ref_out, ref_angle, _ = od.deskew(page)
    assert ref_angle == ang and np.array_equal(out[0].cpu().numpy(), ref_out)
