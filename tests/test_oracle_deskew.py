"""CPU: the de-skew restatement (oracle/csrc/deskew_oracle.c; reference: /root/reference/backend/utils/image_preprocessing.py:372-460).
OpenCV is absent offline and the reference holds no fixtures for this step ("parity unpinned"), so the oracle is checked through the
properties the reference's algorithm guarantees: a page skewed by a known angle is estimated to the Hough resolution (theta step 1
degree) with the reference's sign convention and comes back upright; angles below 0.5 degrees leave the page untouched but are
reported; folding maps near-vertical segments onto near-horizontal ones; the cubic warp is the identity for a zero angle."""
import numpy as np
import pytest
from PIL import Image

from lumina_ocr import synth
from oracle import deskew as od


def _rot(page, angle):
    return np.ascontiguousarray(np.asarray(Image.fromarray(page).rotate(angle, resample=Image.BICUBIC, fillcolor=(255, 255, 255))))


@pytest.mark.parametrize("angle", [3.0, -2.0, 7.0])
def test_known_skew_is_estimated_and_undone(angle):
    page = synth.synth_page(600, 800, 11, n_lines=14)[0]
    skewed = _rot(page, angle)                       # PIL rotates counter-clockwise: text lines rise to the right -> negative atan2
    out, est, info = od.deskew(skewed)
    assert info["flag"] == 3 and abs(est + angle) < 0.6, (est, info)
    _, again, info2 = od.deskew(out)
    assert abs(again) < 0.5 and info2["flag"] in (1, 3) or abs(again) < 1.01      # residual below the Hough resolution


def test_small_angles_blank_pages_and_identity_warp():
    page = synth.synth_page(500, 700, 5, n_lines=10)[0]
    out, est, info = od.deskew(page)
    assert info["flag"] == 1 and abs(est) < 0.5 and np.array_equal(out, page)     # (:441-443) unchanged, angle reported
    blank = np.full((200, 300, 3), 255, np.uint8)
    out, est, info = od.deskew(blank)
    assert info["flag"] == 0 and est == 0.0 and info["segments"] == 0 and np.array_equal(out, blank)   # (:409-411)
    rng = np.random.default_rng(0)
    noise = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
    assert np.array_equal(od.warp(noise, 0.0, 1.0), noise)                        # phase (0, 0): weight 2^15 on the centre tap


def test_angle_folding_and_median():
    seg = lambda x1, y1, x2, y2: [x1, y1, x2, y2]
    # a vertical segment walked bottom -> top (angle -90) folds to 0; one at -88 folds to +2; horizontal ones stay
    rot = od.angle(np.array([seg(100, 500, 100, 100), seg(0, 0, 400, 0), seg(0, 10, 400, 24)], np.int32))
    assert int(rot[2]) in (1, 3)
    segs = np.array([seg(0, 0, 1000, 35)] * 3 + [seg(0, 0, 1000, 0)] * 2, np.int32)      # median of (2, 2, 2, 0, 0) degrees ~ 2
    rot = od.angle(segs)
    assert int(rot[2]) == 3 and abs(np.degrees(np.arctan2(rot[0], rot[1])) - np.degrees(np.arctan2(35, 1000))) < 1e-9
    steep = np.array([seg(0, 0, 100, 300)] * 3, np.int32)                                # 71.6 degrees -> folded to -18.4
    rot = od.angle(steep)
    assert abs(np.degrees(np.arctan2(rot[0], rot[1])) + 18.43494882292201) < 1e-9 and int(rot[2]) == 3
    even = np.array([seg(0, 0, 1000, 0), seg(0, 0, 1000, 35)], np.int32)                 # even count: mean of the two middle angles
    rot = od.angle(even)
    assert abs(np.degrees(np.arctan2(rot[0], rot[1])) - 0.5 * np.degrees(np.arctan2(35, 1000))) < 1e-9


def _canny_numpy(rgb):
    """Independent (vectorised numpy / scipy) statement of steps 1-2 of the oracle's header: cross-checks the C code's indexing."""
    from scipy import ndimage
    r, g, b = (rgb[..., k].astype(np.int64) for k in range(3))
    gray = ((4899 * r + 9617 * g + 1868 * b + 8192) >> 14).astype(np.int64)
    p = np.pad(gray, 1, mode="edge")
    H, W = gray.shape
    sl = lambda dy, dx: p[1 + dy:1 + dy + H, 1 + dx:1 + dx + W]
    dx = (sl(-1, 1) + 2 * sl(0, 1) + sl(1, 1)) - (sl(-1, -1) + 2 * sl(0, -1) + sl(1, -1))
    dy = (sl(1, -1) + 2 * sl(1, 0) + sl(1, 1)) - (sl(-1, -1) + 2 * sl(-1, 0) + sl(-1, 1))
    mag = np.abs(dx) + np.abs(dy)
    mp = np.pad(mag, 1)                                   # the magnitude is 0 outside the image
    m = lambda oy, ox: mp[1 + oy:1 + oy + H, 1 + ox:1 + ox + W]
    ax, ay = np.abs(dx), np.abs(dy) << 15
    tg22 = ax * 13573
    horiz = ay < tg22
    vert = ~horiz & (ay > tg22 + (ax << 16))
    diag = ~horiz & ~vert
    s_pos = (dx ^ dy) >= 0                                # s = +1: neighbours (y-1, x-1) and (y+1, x+1)
    keep = np.zeros_like(mag, bool)
    keep |= horiz & (mag > m(0, -1)) & (mag >= m(0, 1))
    keep |= vert & (mag > m(-1, 0)) & (mag >= m(1, 0))
    keep |= diag & s_pos & (mag > m(-1, -1)) & (mag > m(1, 1))
    keep |= diag & ~s_pos & (mag > m(-1, 1)) & (mag > m(1, -1))
    keep &= mag > 50
    strong = keep & (mag > 150)
    lab, n = ndimage.label(keep, structure=np.ones((3, 3), int))
    good = np.zeros(n + 1, bool)
    good[np.unique(lab[strong])] = True
    good[0] = False
    return (good[lab] * 255).astype(np.uint8)


@pytest.mark.parametrize("seed,angle", [(3, 0.0), (8, 4.0)])
def test_c_canny_and_hough_votes_match_an_independent_numpy_statement(seed, angle):
    page = synth.synth_page(300, 420, seed, n_lines=8)[0]
    if angle:
        page = _rot(page, angle)
    edges = od.canny(page)
    assert np.array_equal(edges, _canny_numpy(page))
    assert 0.01 < (edges > 0).mean() < 0.4
    segs, npk, acc = od.segments(edges, want_accum=True)
    ys, xs = np.nonzero(edges)
    half = (acc.shape[1] - 1) // 2
    for n in (0, 1, 45, 87, 90, 135, 179):               # votes of a few angles, float32 arithmetic as HoughLinesP's tables
        c, s_ = np.float32(np.cos(n * np.pi / 180)), np.float32(np.sin(n * np.pi / 180))
        rho = np.rint(xs.astype(np.float32) * c + ys.astype(np.float32) * s_).astype(np.int64) + half
        assert np.array_equal(np.bincount(rho, minlength=acc.shape[1]), acc[n]), n
    assert acc.sum() == 180 * len(xs)


def test_warp_agrees_with_an_independent_bicubic_rotation_on_a_smooth_image():
    """Direction, centre and interpolation of the fixed-point warp against scipy's cubic-spline rotation of a smooth image
    (different cubic kernels: agreement to a couple of grey levels in the interior; a wrong sign or centre would be off by tens)."""
    from scipy import ndimage
    yy, xx = np.mgrid[0:160, 0:220].astype(np.float64)
    img = (127 + 60 * np.sin(xx / 17.0) * np.cos(yy / 23.0) + 40 * np.sin((xx + yy) / 31.0))
    rgb = np.stack([img, img * 0.9, 255 - img], -1).round().clip(0, 255).astype(np.uint8)
    ang = 5.0
    out = od.warp(rgb, float(np.sin(np.radians(ang))), float(np.cos(np.radians(ang))))
    # cv2.getRotationMatrix2D: positive angle = counter-clockwise about (W // 2, H // 2); dst(x, y) = src(M^-1 (x, y))
    c, s_ = np.cos(np.radians(ang)), np.sin(np.radians(ang))
    cx, cy = 220 // 2, 160 // 2
    xs_ = c * (xx - cx) - s_ * (yy - cy) + cx            # inverse of [[c, s], [-s, c]] applied about the centre
    ys_ = s_ * (xx - cx) + c * (yy - cy) + cy
    for ch in range(3):
        ref = ndimage.map_coordinates(rgb[..., ch].astype(np.float64), [ys_, xs_], order=3, mode="nearest")
        d = np.abs(out[20:-20, 20:-20, ch].astype(np.float64) - ref[20:-20, 20:-20])
        assert d.max() < 4.0 and d.mean() < 0.7, (ch, d.max(), d.mean())
