"""ctypes wrapper around the plain-C oracle (oracle/csrc/dbpost_oracle.c). Test infrastructure."""
from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None


def build() -> Path:
    so = _HERE / "_build" / "liboracle.so"
    newest = max(f.stat().st_mtime for f in (_HERE / "csrc").glob("*.c"))
    if not so.exists() or so.stat().st_mtime < newest:
        subprocess.check_call(["make", "-s", "-C", str(_HERE)])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(str(build()))
        _LIB.oracle_db_postprocess.restype = ctypes.c_int
        _LIB.oracle_rec_crop.restype = ctypes.c_int
    return _LIB


def db_postprocess(prob_bf16_bits: np.ndarray, valid_h: int, valid_w: int, thresh=0.3, box_thresh=0.6,
                   unclip_ratio=1.5, min_size=3, max_candidates=1000):
    """prob [H,W] uint16 (bf16 bits) -> (boxes int32 [n,8], scores f32 [n], n_components)."""
    p = np.ascontiguousarray(prob_bf16_bits, np.uint16)
    h, w = p.shape
    boxes = np.zeros((max_candidates, 8), np.int32)
    scores = np.zeros(max_candidates, np.float32)
    ncomp = ctypes.c_int32(0)
    n = lib().oracle_db_postprocess(
        p.ctypes.data_as(ctypes.c_void_p), h, w, valid_h, valid_w, ctypes.c_float(thresh),
        ctypes.c_float(box_thresh), ctypes.c_float(unclip_ratio), min_size, max_candidates,
        boxes.ctypes.data_as(ctypes.c_void_p), scores.ctypes.data_as(ctypes.c_void_p), max_candidates,
        ctypes.byref(ncomp))
    return boxes[:n].copy(), scores[:n].copy(), int(ncomp.value)


def rec_crop(page_u8: np.ndarray, box8) -> tuple:
    """page [H,W,3] u8, box 8 ints -> (crop [32,320,3] u8, valid width)."""
    pg = np.ascontiguousarray(page_u8, np.uint8)
    h, w, _ = pg.shape
    b = np.ascontiguousarray(box8, np.int32)
    out = np.zeros((32, 320, 3), np.uint8)
    wc = lib().oracle_rec_crop(pg.ctypes.data_as(ctypes.c_void_p), h, w, b.ctypes.data_as(ctypes.c_void_p),
                               out.ctypes.data_as(ctypes.c_void_p))
    return out, wc
