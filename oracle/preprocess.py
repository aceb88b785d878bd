"""Oracle (test infrastructure): numpy restatement of the reference's image pre-processing.

Follows /root/reference/backend/utils/image_preprocessing.py:
  * resize_if_needed            :81-110  (max dimension 2000, int() truncation, Image.Resampling.LANCZOS)
  * enhance_contrast(1.2)       :132-144 / :234-236   (ImageEnhance.Contrast -> Image.blend with the mean-gray image)
  * enhance_sharpness(1.1)      :146-158 / :238-240   (ImageEnhance.Sharpness -> Image.blend with ImageFilter.SMOOTH)
  * optimize_for_ocr            :191-242 (EXIF -> resize -> contrast 1.2 -> sharpness 1.1)
The arithmetic inside PIL (a third-party dependency, Pillow 12.2.0 here; not vendored in the reference) is
restated from its published algorithm: 8-bit resampling with 22-bit fixed-point coefficients, two passes
(horizontal then vertical); blend in float32 with truncation; 3x3 SMOOTH kernel (1,1,1,1,5,1,1,1,1)/13 in
float32 with +0.5 offset and untouched borders.  PINNED: tests/test_golden_preprocess.py checks every function
here against vectors produced by the reference module itself (tools/make_golden.py).
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2
MAX_DIM = 2000  # settings.OCR_MAX_IMAGE_DIMENSION default (/root/reference/backend/config.py:69)


def target_size(width: int, height: int, max_dim: int = MAX_DIM) -> Tuple[int, int]:
    """image_preprocessing.py:93-105 (also get_optimal_size :112-126)."""
    if max(width, height) <= max_dim:
        return width, height
    if width > height:
        return max_dim, int(height * (max_dim / width))
    return int(width * (max_dim / height)), max_dim


def _sinc(x: float) -> float:
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x: float) -> float:
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


def lanczos_coeffs(in_size: int, out_size: int):
    """Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc for box (0, in_size).
    -> (ksize, bounds int32 [out,2] (xmin, count), kk int32 [out, ksize])"""
    scale = filterscale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 3.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ws = [_lanczos((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in ws:
            ww += w
        for x, w in enumerate(ws):
            v = w / ww if ww != 0.0 else w
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _resample_axis(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    in_size = img.shape[axis]
    ksize, bounds, kk = lanczos_coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        xmin, cnt = bounds[xx]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for k in range(cnt):
            acc += src[xmin + k] * int(kk[xx, k])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_lanczos(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """[H,W,C] u8 -> [out_h,out_w,C] u8: horizontal pass, then vertical pass (Pillow ImagingResample)."""
    h, w = img.shape[:2]
    x = img
    if out_w != w:
        x = _resample_axis(x, out_w, 1)
    if out_h != h:
        x = _resample_axis(x, out_h, 0)
    return x


def resize_if_needed(img: np.ndarray, max_dim: int = MAX_DIM) -> np.ndarray:
    h, w = img.shape[:2]
    nw, nh = target_size(w, h, max_dim)
    if (nw, nh) == (w, h):
        return img
    return resize_lanczos(img, nw, nh)


def gray_mean(img: np.ndarray) -> int:
    """int(ImageStat.Stat(image.convert('L')).mean[0] + 0.5); L = (R*19595 + G*38470 + B*7471 + 0x8000) >> 16."""
    if img.ndim == 2:
        lum = img.astype(np.int64)
    else:
        r, g, b = (img[..., i].astype(np.int64) for i in range(3))
        lum = (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16
    return int(float(lum.sum()) / lum.size + 0.5)


def _blend(deg: np.ndarray, img: np.ndarray, alpha: float) -> np.ndarray:
    """Pillow Blend.c: temp = (float)(in1 + alpha*(in2 - in1)) in float32; extrapolation clips, result truncated."""
    a = np.float32(alpha)
    d = deg.astype(np.int32)
    diff = (img.astype(np.int32) - d).astype(np.float32)
    t = (a * diff).astype(np.float32)
    t = (d.astype(np.float32) + t).astype(np.float32)
    if 0.0 <= alpha <= 1.0:
        return t.astype(np.uint8)  # truncation
    out = np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t)).astype(np.float32)
    return out.astype(np.uint8)


def enhance_contrast(img: np.ndarray, factor: float = 1.2) -> np.ndarray:
    m = gray_mean(img)
    return _blend(np.full_like(img, m), img, factor)


def smooth3x3(img: np.ndarray) -> np.ndarray:
    """ImageFilter.SMOOTH: kernel (1,1,1,1,5,1,1,1,1)/13 (float32), offset 0 + 0.5, borders copied."""
    k = (np.array([1, 1, 1, 1, 5, 1, 1, 1, 1], np.float32) / np.float32(13.0)).astype(np.float32)
    f = img.astype(np.float32)
    out = img.copy()
    h, w = img.shape[:2]
    if h < 3 or w < 3:
        return out
    ss = np.full(f[1:-1, 1:-1].shape, np.float32(0.5), np.float32)
    # Pillow order: row y+1 with kernel[0:3], row y with kernel[3:6], row y-1 with kernel[6:9]
    for rows, kr in ((slice(2, None), 0), (slice(1, -1), 3), (slice(0, -2), 6)):
        a = (f[rows, 0:-2] * k[kr]).astype(np.float32)
        a = (a + (f[rows, 1:-1] * k[kr + 1]).astype(np.float32)).astype(np.float32)
        a = (a + (f[rows, 2:] * k[kr + 2]).astype(np.float32)).astype(np.float32)
        ss = (ss + a).astype(np.float32)
    res = np.where(ss <= 0.0, 0, np.where(ss >= 255.0, 255, ss)).astype(np.float32).astype(np.uint8)
    out[1:-1, 1:-1] = res
    return out


def enhance_sharpness(img: np.ndarray, factor: float = 1.1) -> np.ndarray:
    return _blend(smooth3x3(img), img, factor)


def optimize_for_ocr(img: np.ndarray, max_dim: int = MAX_DIM, contrast: float = 1.2, sharpness: float = 1.1) -> np.ndarray:
    """image_preprocessing.py:191-242 with the defaults the providers use (no denoise, no grayscale)."""
    x = resize_if_needed(img, max_dim)
    x = enhance_contrast(x, contrast)
    return enhance_sharpness(x, sharpness)


# --------------------------------------------------------------------------------------
# binarisation (reference: backend/utils/image_preprocessing.py:175-185 `binarize`, :462-494 `adaptive_binarize`; off by default,
# settings.PREPROCESSING_APPLY_BINARIZE -> preprocess_for_azure(apply_binarize=...), :613-622)
# --------------------------------------------------------------------------------------
def gray_L(img: np.ndarray) -> np.ndarray:
    """PIL image.convert('L'): (R*19595 + G*38470 + B*7471 + 0x8000) >> 16."""
    if img.ndim == 2:
        return img.astype(np.uint8)
    r, g, b = (img[..., i].astype(np.int64) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def binarize(img: np.ndarray, threshold: int = 128) -> np.ndarray:
    """`binarize` (:175-185) — what the reference's adaptive_binarize falls back to without OpenCV (:473-475), i.e. what it computes
    in this container: L > threshold -> 255 else 0.  PINNED by tests/golden/preprocess_vectors.npz (binarize*)."""
    return np.where(gray_L(img) > threshold, 255, 0).astype(np.uint8)


# (float)(exp(-(i - 5)^2 / (2 sigma^2)) / sum), sigma = 0.3 * ((11 - 1) * 0.5 - 1) + 0.8 = 2.0: cv2.getGaussianKernel(11, -1, CV_32F)
GAUSS11 = np.array([float.fromhex(v) for v in (
    "0x1.20c256p-7", "0x1.bcb86ap-6", "0x1.0ab50ap-4", "0x1.f2464cp-4", "0x1.6a7e1ep-3", "0x1.9ac20ap-3",
    "0x1.6a7e1ep-3", "0x1.f2464cp-4", "0x1.0ab50ap-4", "0x1.bcb86ap-6", "0x1.20c256p-7")], np.float32)


def adaptive_binarize(img: np.ndarray) -> np.ndarray:
    """cv2.adaptiveThreshold(gray, 255, ADAPTIVE_THRESH_GAUSSIAN_C, THRESH_BINARY, blockSize=11, C=2) (:485-491) restated — OpenCV is
    absent offline: "parity unpinned".  mean = 11x11 Gaussian of the L image (separable, float32, taps in ascending order, one mul
    and one add per tap, replicated border, rounded to uint8 once); dst = 255 where L - mean > -2, else 0."""
    L = gray_L(img)
    h, w = L.shape
    f = np.pad(L.astype(np.float32), ((0, 0), (5, 5)), mode="edge")
    row = np.zeros((h, w), np.float32)
    for k in range(11):
        row = (row + (f[:, k:k + w] * GAUSS11[k]).astype(np.float32)).astype(np.float32)
    f2 = np.pad(row, ((5, 5), (0, 0)), mode="edge")
    acc = np.zeros((h, w), np.float32)
    for k in range(11):
        acc = (acc + (f2[k:k + h, :] * GAUSS11[k]).astype(np.float32)).astype(np.float32)
    mean = np.clip(np.rint(acc), 0, 255).astype(np.int32)
    return np.where(L.astype(np.int32) - mean > -2, 255, 0).astype(np.uint8)


# --------------------------------------------------------------------------------------
# optimize_for_ocr's optional steps (reference: backend/utils/image_preprocessing.py:160-169, :225-231; off by default).
# PINNED by tests/golden/preprocess_vectors.npz (gray*, denoise*, optimize_dn*, optimize_dn_gray*: produced by the reference module).
# --------------------------------------------------------------------------------------
def grayscale(img: np.ndarray) -> np.ndarray:
    """`convert_to_grayscale` (:167-169) = PIL convert('L') of an RGB image: (19595 R + 38470 G + 7471 B + 0x8000) >> 16.  [H,W,3] -> [H,W]."""
    if img.ndim == 2:
        return img.copy()
    r = img.astype(np.uint32)
    return ((r[..., 0] * 19595 + r[..., 1] * 38470 + r[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)


def denoise(img: np.ndarray) -> np.ndarray:
    """`denoise` (:160-165) = PIL ImageFilter.MedianFilter(3): per channel the median of the 3x3 neighbourhood of the edge-replicated image."""
    x = img if img.ndim == 3 else img[..., None]
    p = np.pad(x, ((1, 1), (1, 1), (0, 0)), mode="edge")
    st = np.stack([p[i:i + x.shape[0], j:j + x.shape[1]] for i in range(3) for j in range(3)])
    out = np.sort(st, axis=0)[4]
    return out if img.ndim == 3 else out[..., 0]
