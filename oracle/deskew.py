"""ctypes wrapper around oracle/csrc/deskew_oracle.c (test infrastructure): the reference's de-skew step
(/root/reference/backend/utils/image_preprocessing.py:372-460), "parity unpinned" — see the C file's header."""
from __future__ import annotations

import ctypes

import numpy as np

from .dbpost import lib

MAX_PEAKS, SEG_PER_PEAK = 512, 8


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def canny(rgb: np.ndarray) -> np.ndarray:
    """[H,W,3] u8 -> edge map [H,W] u8 (0 / 255)."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w, _ = rgb.shape
    out = np.zeros((h, w), np.uint8)
    lib().oracle_deskew_canny(_p(rgb), h, w, _p(out))
    return out


def segments(edges: np.ndarray, want_accum: bool = False):
    """edge map -> (segments int32 [n,4] (x1,y1,x2,y2), peaks visited[, accumulator int32 [180, 2(W+H)+1]])."""
    e = np.ascontiguousarray(edges, np.uint8)
    h, w = e.shape
    segs = np.zeros((MAX_PEAKS * SEG_PER_PEAK, 4), np.int32)
    npk = ctypes.c_int32(0)
    acc = np.zeros((180, 2 * (w + h) + 1), np.int32) if want_accum else None
    f = lib().oracle_deskew_segments
    f.restype = ctypes.c_int
    n = f(_p(e), h, w, _p(segs), len(segs), ctypes.byref(npk), _p(acc) if want_accum else None)
    return (segs[:n].copy(), int(npk.value)) + ((acc,) if want_accum else ())


def angle(segs: np.ndarray) -> np.ndarray:
    """segments -> [sin, cos, flag] (flag: 0 none, 1 below 0.5 deg, 2 above 45 deg, 3 rotate)."""
    s = np.ascontiguousarray(segs, np.int32).reshape(-1, 4)
    rot = np.zeros(3, np.float64)
    lib().oracle_deskew_angle(_p(s), len(s), _p(rot))
    return rot


def warp(rgb: np.ndarray, sin: float, cos: float) -> np.ndarray:
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w, _ = rgb.shape
    out = np.zeros_like(rgb)
    lib().oracle_deskew_warp(_p(rgb), h, w, ctypes.c_double(sin), ctypes.c_double(cos), _p(out))
    return out


def deskew(rgb: np.ndarray):
    """The whole step: -> (image, angle in degrees as the reference reports it, info dict)."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w, _ = rgb.shape
    out = np.zeros_like(rgb)
    rot = np.zeros(3, np.float64)
    ns, npk = ctypes.c_int32(0), ctypes.c_int32(0)
    f = lib().oracle_deskew
    f.restype = ctypes.c_int
    flag = f(_p(rgb), h, w, _p(out), _p(rot), ctypes.byref(ns), ctypes.byref(npk))
    return out, angle_degrees(rot), dict(flag=flag, sin=float(rot[0]), cos=float(rot[1]), segments=int(ns.value), peaks=int(npk.value))


def angle_degrees(rot) -> float:
    """What the reference returns next to the image (:446-447, :441-443, :460): the angle when it rotated or was below 0.5 deg, 0.0 otherwise."""
    return float(np.degrees(np.arctan2(rot[0], rot[1]))) if int(rot[2]) in (1, 3) else 0.0
