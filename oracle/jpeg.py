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


# ---- decoder (oracle/csrc/jpegdec_oracle.c) ------------------------------------------------------------------------------
def info(data: bytes):
    """-> (rc, dict(width, height, ncomp, h, v, restart)); rc 0 = the subset the decoders handle, -2 = valid but unsupported, -1 = corrupt."""
    L = _lib()
    a = np.frombuffer(data, np.uint8)
    out = (ctypes.c_int * 6)()
    rc = L.oracle_jpeg_info(a.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(len(data)), out)
    return rc, dict(width=out[0], height=out[1], ncomp=out[2], h=out[3], v=out[4], restart=out[5])


def decode(data: bytes) -> np.ndarray:
    """JPEG file bytes -> uint8 [H,W,3] as Pillow decodes it (a grey-scale file: the grey value on all three channels)."""
    rc, i = info(data)
    if rc:
        raise ValueError("jpeg oracle: %s" % ("unsupported file" if rc == -2 else "corrupt file"))
    L = _lib()
    a = np.frombuffer(data, np.uint8)
    out = np.empty((i["height"], i["width"], 3), np.uint8)
    nc = ctypes.c_int(0)
    rc = L.oracle_jpeg_decode_rgb(a.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(len(data)), out.ctypes.data_as(ctypes.c_void_p), i["width"], i["height"], ctypes.byref(nc))
    if rc:
        raise ValueError("jpeg oracle: decode failed (%d)" % rc)
    return out


def decode_coefficients(data: bytes) -> np.ndarray:
    """Quantised coefficients of every block in scan order (natural order inside a block, absolute DC): int16 [blocks, 64]."""
    rc, i = info(data)
    if rc:
        raise ValueError("jpeg oracle: unsupported / corrupt file")
    mcux = -(-i["width"] // (8 * i["h"])); mcuy = -(-i["height"] // (8 * i["v"]))
    bpm = i["h"] * i["v"] + (2 if i["ncomp"] == 3 else 0)
    n = mcux * mcuy * bpm
    out = np.empty((n, 64), np.int16)
    L = _lib()
    L.oracle_jpeg_decode_coefficients.restype = ctypes.c_long
    a = np.frombuffer(data, np.uint8)
    got = L.oracle_jpeg_decode_coefficients(a.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(len(data)), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(n))
    if got != n:
        raise ValueError("jpeg oracle: coefficient decode failed (%d)" % got)
    return out
