"""ctypes wrapper around the plain-C JPEG oracle (oracle/csrc/jpeg_oracle.c). Test infrastructure."""
from __future__ import annotations

import ctypes

import numpy as np

from .dbpost import lib as _lib


def encode(rgb: np.ndarray, quality: int = 95, optimize: bool = True) -> bytes:
    """uint8 [H,W,3] -> the JFIF byte stream of PIL's save(format='JPEG', quality=quality, optimize=optimize)."""
    a = np.ascontiguousarray(rgb, np.uint8)
    h, w, _ = a.shape
    L = _lib()
    L.oracle_jpeg_encode_ex.restype = ctypes.c_size_t
    cap = w * h * 3 + 65536
    out = np.empty(cap, np.uint8)
    n = L.oracle_jpeg_encode_ex(a.ctypes.data_as(ctypes.c_void_p), w, h, int(quality), int(bool(optimize)), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(cap))
    assert n <= cap
    return out[:n].tobytes()


def coefficients(rgb: np.ndarray, quality: int = 95) -> np.ndarray:
    """Quantised DCT coefficients in MCU scan order: int16 [mcus, 6, 64] (4 Y, Cb, Cr; natural order inside a block)."""
    a = np.ascontiguousarray(rgb, np.uint8)
    h, w, _ = a.shape
    mcus = ((w + 15) // 16) * ((h + 15) // 16)
    out = np.empty((mcus, 6, 64), np.int16)
    _lib().oracle_jpeg_coefficients(a.ctypes.data_as(ctypes.c_void_p), w, h, int(quality), out.ctypes.data_as(ctypes.c_void_p))
    return out


def compress_for_azure(rgb: np.ndarray, target_size_mb: float = 2.0, initial_quality: int = 95, min_quality: int = 30) -> bytes:
    """The reference's quality loop (image_preprocessing.py:495-538) on top of encode(); the resize fallback is not restated."""
    target = int(target_size_mb * 1024 * 1024)
    q = initial_quality
    while q >= min_quality:
        b = encode(rgb, q)
        if len(b) <= target:
            return b
        q -= 10
    raise ValueError("image does not fit the target size without the resize fallback")
