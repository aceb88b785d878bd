"""Seeded synthetic pages and line crops (no reference images exist offline —
backend/test_image.png is listed in /root/reference/.MISSING_LARGE_BLOBS).  Used by bench.py,
tests and tools; pure PIL/numpy, no network, no files."""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
from PIL import Image, ImageDraw, ImageFont

ALPHABET = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789 .,:-/()#&@%+=?!'\"$;"  # 80 chars
A4_200DPI = (2339, 1654)  # (H, W)

_fonts = {}


def _font(size: int):
    if size not in _fonts:
        _fonts[size] = ImageFont.load_default(size=size)
    return _fonts[size]


def random_text(rng: np.random.Generator, lo: int = 4, hi: int = 24) -> str:
    n = int(rng.integers(lo, hi + 1))
    s = "".join(ALPHABET[int(i)] for i in rng.integers(0, len(ALPHABET), n)).strip()
    return s or "A"


def synth_page(h: int, w: int, seed: int, n_lines: int = 60, noise: float = 3.0) -> Tuple[np.ndarray, List[dict]]:
    """White page with rendered text lines (10-14 pt at 200 DPI ~ 28-39 px) + Gaussian noise.
    -> (uint8 [h,w,3], [{'text', 'box': (x0,y0,x1,y1)}])"""
    rng = np.random.default_rng(seed)
    img = Image.new("RGB", (w, h), (255, 255, 255))
    d = ImageDraw.Draw(img)
    gt = []
    scale = min(1.0, h / 2339.0 * 1.6 + 0.2)
    margin = max(4, int(0.06 * w))
    y = max(4, int(0.04 * h))
    pitch = max(14, (h - 2 * y) // max(n_lines, 1))
    for _ in range(n_lines):
        size = max(10, int(rng.integers(28, 40) * scale))
        if y + size + 4 >= h:
            break
        x = margin + int(rng.integers(0, max(1, w // 10)))
        words = []
        while True:
            words.append(random_text(rng, 3, 12))
            txt = " ".join(words)
            if d.textlength(txt, font=_font(size)) > (w - margin - x) * float(rng.uniform(0.35, 0.95)) or len(words) > 12:
                break
        while len(txt) > 1 and x + d.textlength(txt, font=_font(size)) > w - margin:
            txt = txt[:-1]
        shade = int(rng.integers(0, 41))
        d.text((x, y), txt, fill=(shade, shade, shade), font=_font(size))
        bb = d.textbbox((x, y), txt, font=_font(size))
        gt.append(dict(text=txt, box=bb))
        y += max(pitch, size + 6)
    arr = np.asarray(img, np.float32)
    if noise > 0:
        arr = arr + rng.normal(0.0, noise, arr.shape).astype(np.float32)
    return np.clip(np.rint(arr), 0, 255).astype(np.uint8), gt


def synth_crop(rng: np.random.Generator) -> Tuple[np.ndarray, str]:
    """One 32x320 line crop with a rendered random string (len 4-24), right-padded with 0."""
    txt = random_text(rng)
    img = Image.new("RGB", (320, 32), (255, 255, 255))
    d = ImageDraw.Draw(img)
    size = 22
    while size > 8 and d.textlength(txt, font=_font(size)) > 312:
        size -= 1
    d.text((4, max(0, (32 - size) // 2 - 2)), txt, fill=(20, 20, 20), font=_font(size))
    arr = np.asarray(img, np.uint8).copy()
    tw = int(min(320, d.textlength(txt, font=_font(size)) + 8))
    arr[:, tw:] = 0
    return arr, txt


def synth_form_page(seed: int = 0, w: int = 2000, h: int = 1090) -> Tuple[np.ndarray, List[dict]]:
    """A 2000x1090 form-like page mimicking the layout of the reference's captured sample
    (/root/reference/azure_debug_output.json:172-173: page 2000.0 x 1090.0)."""
    rng = np.random.default_rng(seed)
    img = Image.new("RGB", (w, h), (255, 255, 255))
    d = ImageDraw.Draw(img)
    gt = []

    def put(x, y, txt, size):
        d.text((x, y), txt, fill=(15, 15, 15), font=_font(size))
        gt.append(dict(text=txt, box=d.textbbox((x, y), txt, font=_font(size))))

    put(385, 150, "SPRINGFIELD UNIVERSITY - UNDERGRADUATE ADMISSION", 40)
    rows = [("Applicant Name:", "Jordan A. Whitfield"), ("Date of Birth:", "14/03/2004"), ("Program:", "B.Sc. Computer Science"),
            ("Student ID:", "SU-2024-%05d" % int(rng.integers(0, 99999))), ("Email:", "j.whitfield@example.edu"),
            ("Phone:", "+1 555 0134"), ("Address:", "221 Maple Avenue, Springfield"), ("Guardian:", "Morgan Whitfield")]
    y = 280
    for k, v in rows:
        put(140, y, k, 32)
        put(620, y, v, 32)
        y += 88
    return np.asarray(img, np.uint8).copy(), gt
