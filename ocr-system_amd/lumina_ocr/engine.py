"""ctypes binding of liblumina_ocr.so (include/lumina_ocr.h) — the only way the host reaches the GPU.

PyTorch-ROCm is used for device memory and streams only (tensor.data_ptr(), current stream);
no torch op runs on the hot path.  There is NO CPU fallback: a missing library or device raises
EngineUnavailable, which the provider turns into an error *result* (the reference's convention,
/root/reference/backend/services/ocr_service.py:464-475).
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path
from typing import Dict, Optional, Tuple

import numpy as np

from . import arch

_LIB_PATH = Path(__file__).resolve().parent.parent / "lib" / "liblumina_ocr.so"

REC_H, REC_W, REC_T = 32, 320, 80
MAX_BOXES = 1000


class EngineUnavailable(RuntimeError):
    pass


class EngineError(RuntimeError):
    pass


_lib = None


def load_library() -> ctypes.CDLL:
    """Load the HIP engine; fails loudly when it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("LUMINA_OCR_LIB", str(_LIB_PATH)))
    if not path.exists():
        raise EngineUnavailable(f"{path} not found: build it with `make -C ocr-system_amd` (hipcc, gfx950)")
    lib = ctypes.CDLL(str(path))
    c = ctypes
    vp, i32, f32, sz = c.c_void_p, c.c_int, c.c_float, c.c_size_t
    sig = {
        "lumina_ocr_create": (i32, [i32, c.POINTER(vp)]),
        "lumina_ocr_destroy": (None, [vp]),
        "lumina_ocr_last_error": (c.c_char_p, [vp]),
        "lumina_ocr_version": (c.c_char_p, []),
        "lumina_ocr_set_option": (i32, [vp, c.c_char_p, i32]),
        "lumina_ocr_load_det_weights": (i32, [vp, vp, sz]),
        "lumina_ocr_load_rec_weights": (i32, [vp, vp, sz]),
        "lumina_ocr_num_classes": (i32, [vp]),
        "lumina_ocr_normalize": (i32, [vp, vp, i32, i32, i32, i32, i32, c.POINTER(f32), c.POINTER(f32), i32, vp, vp]),
        "lumina_ocr_det_forward": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, vp]),
        "lumina_ocr_det_postprocess": (i32, [vp, vp, i32, i32, i32, i32, i32, f32, f32, f32, i32, i32, vp, vp, vp, vp]),
        "lumina_ocr_rec_crop": (i32, [vp, vp, i32, i32, i32, vp, vp, i32, vp, vp, vp]),
        "lumina_ocr_rec_forward": (i32, [vp, vp, vp, i32, vp, vp, vp]),
        "lumina_ocr_ctc_decode": (i32, [vp, vp, vp, i32, vp, vp, vp, vp]),
        "lumina_ocr_conv2d": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, i32, i32, i32, i32, vp, vp, vp]),
        "lumina_ocr_read_tap": (i32, [vp, c.c_char_p, vp, sz, c.POINTER(i32)]),
        "lumina_ocr_conv_timing": (i32, [vp, c.POINTER(c.c_double), c.POINTER(c.c_double), c.POINTER(i32)]),
        "lumina_ocr_conv_timing_detail": (i32, [vp, c.c_char_p, sz]),
        "lumina_ocr_resize_lanczos": (i32, [vp, vp, i32, i32, i32, i32, vp, i32, i32, vp]),
        "lumina_ocr_enhance": (i32, [vp, vp, i32, i32, i32, f32, f32, vp, vp, vp]),
        "lumina_ocr_jpeg_encode": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, sz, vp, vp]),
        "lumina_ocr_jpeg_probe": (i32, [vp, sz, vp]),
        "lumina_ocr_jpeg_last_passes": (i32, [vp]),
        "lumina_ocr_jpeg_decode_async": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, i32, vp]),
        "lumina_ocr_jpeg_decode": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp]),
        "lumina_ocr_jpeg_coefficients": (i32, [vp, vp, i32, i32, i32, i32, vp, vp]),
        "lumina_ocr_load_svtr_weights": (i32, [vp, vp, sz]),
        "lumina_ocr_svtr_forward": (i32, [vp, vp, vp, i32, vp, vp, vp]),
        "lumina_ocr_svtr_num_classes": (i32, [vp]),
        "lumina_ocr_svtr_dtype": (i32, [vp]),
        "lumina_ocr_binarize": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, vp]),
        "lumina_ocr_grayscale": (i32, [vp, vp, i32, i32, i32, vp, vp]),
        "lumina_ocr_exif_transpose": (i32, [vp, vp, i32, i32, i32, i32, vp, vp]),
        "lumina_ocr_denoise": (i32, [vp, vp, i32, i32, i32, vp, vp]),
        "lumina_ocr_deskew": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]),
        "lumina_ocr_deskew_warp": (i32, [vp, vp, i32, i32, i32, vp, vp, vp]),
    }
    missing = []
    for name, (res, args) in sig.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    lib._missing = missing  # exported-symbol check lives in tests/test_abi.py
    _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "lumina_ocr_create", "lumina_ocr_destroy", "lumina_ocr_last_error", "lumina_ocr_version", "lumina_ocr_set_option",
    "lumina_ocr_load_det_weights", "lumina_ocr_load_rec_weights", "lumina_ocr_num_classes", "lumina_ocr_normalize",
    "lumina_ocr_det_forward", "lumina_ocr_det_postprocess", "lumina_ocr_rec_crop", "lumina_ocr_rec_forward",
    "lumina_ocr_ctc_decode", "lumina_ocr_conv2d", "lumina_ocr_read_tap", "lumina_ocr_conv_timing", "lumina_ocr_conv_timing_detail",
    "lumina_ocr_resize_lanczos", "lumina_ocr_enhance", "lumina_ocr_jpeg_encode", "lumina_ocr_jpeg_coefficients", "lumina_ocr_jpeg_probe", "lumina_ocr_jpeg_decode", "lumina_ocr_jpeg_decode_async", "lumina_ocr_jpeg_last_passes",
    "lumina_ocr_load_svtr_weights", "lumina_ocr_svtr_forward", "lumina_ocr_svtr_num_classes", "lumina_ocr_svtr_dtype", "lumina_ocr_binarize", "lumina_ocr_exif_transpose", "lumina_ocr_grayscale", "lumina_ocr_denoise", "lumina_ocr_deskew", "lumina_ocr_deskew_warp",
]


def _torch():
    import torch
    return torch


def _ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


class Engine:
    """One engine handle == one GPU.  Not re-entrant (the reference serialises pages the same way)."""

    def __init__(self, device: int = 0):
        torch = _torch()
        if not torch.cuda.is_available():
            raise EngineUnavailable("no ROCm device visible to torch (torch.cuda.is_available() is False)")
        self.lib = load_library()
        self.device = device
        torch.cuda.set_device(device)
        h = ctypes.c_void_p()
        rc = self.lib.lumina_ocr_create(device, ctypes.byref(h))
        self._h = h
        if rc != 0:
            msg = self.lib.lumina_ocr_last_error(h).decode() if h else "create failed"
            raise EngineUnavailable(msg)
        self.num_classes = self.svtr_num_classes = 0
        self.det_loaded = self.rec_loaded = self.svtr_loaded = False

    # -- plumbing -------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self.lib.lumina_ocr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int):
        if rc != 0:
            raise EngineError(self.lib.lumina_ocr_last_error(self._h).decode())

    def _stream(self) -> int:
        return _torch().cuda.current_stream().cuda_stream

    def jpeg_encode(self, pages, quality: int = 95, max_bytes: int = 2 * 1024 * 1024, optimize: bool = True):
        """uint8 [n,H,W,3] device -> (files uint8 [n, stride] device, sizes int32 [n] device); sizes[i] < 0: file i needs more than
        max_bytes (the reference's cue to lower the quality).  Asynchronous; byte-identical to PIL save(JPEG, quality, optimize=True)."""
        torch = _torch()
        n, h, w, c = pages.shape
        assert c == 3 and pages.dtype == torch.uint8
        stride = (int(max_bytes) + 1023) // 1024 * 1024
        out = torch.empty((n, stride), dtype=torch.uint8, device=pages.device)
        sizes = torch.empty((n,), dtype=torch.int32, device=pages.device)
        self._chk(self.lib.lumina_ocr_jpeg_encode(self._h, _ptr(pages), n, h, w, int(quality), int(bool(optimize)), _ptr(out), stride, _ptr(sizes),
                                                  self._stream()))
        return out, sizes

    @staticmethod
    def jpeg_probe(data: bytes):
        """Host only. -> (rc, dict(width, height, ncomp, h, v, restart)); rc 0: the device decodes this file, -2: valid JPEG outside the
        subset (progressive, CMYK ...: decode with Pillow as the reference does), -1: corrupt / not a JPEG."""
        lib = load_library()
        info = (ctypes.c_int * 6)()
        buf = ctypes.c_char_p(data)
        rc = lib.lumina_ocr_jpeg_probe(buf, len(data), info)
        return rc, dict(width=info[0], height=info[1], ncomp=info[2], h=info[3], v=info[4], restart=info[5])

    def jpeg_decode(self, files, height: int, width: int, out=None):
        """JPEG file images (a sequence of bytes objects, all height x width) -> (uint8 [n,H,W,3] device, status list): the pixel work of
        the reference's Image.open for .jpg inputs (image_preprocessing.py:57-75), byte-identical to Pillow's decode.  status[i] != 0:
        page i was not decoded (-1 corrupt, -2 outside the device subset, -4 another size): the caller falls back to Pillow for it."""
        torch = _torch()
        n = len(files)
        if out is None:
            out = torch.empty((n, height, width, 3), dtype=torch.uint8, device=torch.device("cuda", self.device))
        ptrs = (ctypes.c_char_p * n)(*files)
        sizes = (ctypes.c_size_t * n)(*[len(f) for f in files])
        status = (ctypes.c_int * n)()
        self._chk(self.lib.lumina_ocr_jpeg_decode(self._h, ptrs, sizes, n, int(height), int(width), _ptr(out), status, self._stream()))
        return out, list(status)

    def jpeg_decode_async(self, files, height: int, width: int, out=None, passes: int = 12):
        """jpeg_decode without host synchronisation: -> (pages uint8 [n,H,W,3] device, status int32 [n] PINNED host tensor).  The status is
        valid once the current stream has run (e.g. after the event of the results that depend on the pages): 0 ok, -1 / -2 / -4 as
        jpeg_decode, -5 = `passes` synchronisation passes were not enough for that file (decode the batch again with jpeg_decode)."""
        torch = _torch()
        n = len(files)
        if out is None:
            out = torch.empty((n, height, width, 3), dtype=torch.uint8, device=torch.device("cuda", self.device))
        status = torch.zeros((n,), dtype=torch.int32).pin_memory()
        ptrs = (ctypes.c_char_p * n)(*files)
        sizes = (ctypes.c_size_t * n)(*[len(f) for f in files])
        self._chk(self.lib.lumina_ocr_jpeg_decode_async(self._h, ptrs, sizes, n, int(height), int(width), _ptr(out), status.data_ptr(), int(passes), self._stream()))
        return out, status

    @property
    def jpeg_last_passes(self) -> int:
        return int(self.lib.lumina_ocr_jpeg_last_passes(self._h))

    def jpeg_coefficients(self, pages, quality: int = 95):
        torch = _torch()
        n, h, w, _ = pages.shape
        mcus = ((w + 15) // 16) * ((h + 15) // 16)
        coefs = torch.empty((n, mcus, 6, 64), dtype=torch.int16, device=pages.device)
        self._chk(self.lib.lumina_ocr_jpeg_coefficients(self._h, _ptr(pages), n, h, w, int(quality), _ptr(coefs), self._stream()))
        return coefs

    def set_option(self, key: str, value: int):
        self._chk(self.lib.lumina_ocr_set_option(self._h, key.encode(), int(value)))

    def version(self) -> str:
        return self.lib.lumina_ocr_version().decode()

    # -- weights --------------------------------------------------------------------------
    def load_det(self, weights):
        blob = weights if isinstance(weights, (bytes, bytearray)) else arch.write_blob(weights)
        buf = ctypes.create_string_buffer(bytes(blob), len(blob))
        self._chk(self.lib.lumina_ocr_load_det_weights(self._h, ctypes.cast(buf, ctypes.c_void_p), len(blob)))
        self.det_loaded = True

    def load_rec(self, weights):
        blob = weights if isinstance(weights, (bytes, bytearray)) else arch.write_blob(weights)
        buf = ctypes.create_string_buffer(bytes(blob), len(blob))
        self._chk(self.lib.lumina_ocr_load_rec_weights(self._h, ctypes.cast(buf, ctypes.c_void_p), len(blob)))
        self.num_classes = self.lib.lumina_ocr_num_classes(self._h)
        self.rec_loaded = True

    def load_svtr(self, weights, f16=None):
        """SVTR recogniser weights (arch.make_svtr_weights or a LOCW blob with the svtr.* tensors): Tiny or Base, bf16 or fp16 as the
        blob's svtr.config says; f16 = True / False overrides the storage / MFMA type."""
        self.set_option("svtr_f16", -1 if f16 is None else int(bool(f16)))
        blob = weights if isinstance(weights, (bytes, bytearray)) else arch.write_blob(weights)
        buf = ctypes.create_string_buffer(bytes(blob), len(blob))
        self._chk(self.lib.lumina_ocr_load_svtr_weights(self._h, ctypes.cast(buf, ctypes.c_void_p), len(blob)))
        self.svtr_loaded = True
        self.svtr_num_classes = self.lib.lumina_ocr_svtr_num_classes(self._h)
        self.svtr_dtype = "f16" if self.lib.lumina_ocr_svtr_dtype(self._h) else "bf16"

    # -- hot path -------------------------------------------------------------------------
    def normalize(self, img, hp: int, wp: int, scale, shift, nchw: bool = False):
        torch = _torch()
        n, h, w, _ = img.shape
        out = torch.empty((n, 3, hp, wp) if nchw else (n, hp, wp, 3), dtype=torch.bfloat16, device=img.device)
        sc = (ctypes.c_float * 3)(*scale)
        sh = (ctypes.c_float * 3)(*shift)
        self._chk(self.lib.lumina_ocr_normalize(self._h, _ptr(img), n, h, w, hp, wp, sc, sh, int(nchw), _ptr(out), self._stream()))
        return out

    def det_forward(self, pages, hp: Optional[int] = None, wp: Optional[int] = None, out=None):
        """pages uint8 [B,H,W,3] (device) -> probability map bf16 [B,Hp,Wp]."""
        torch = _torch()
        assert pages.dtype == torch.uint8 and pages.is_cuda and pages.is_contiguous() and pages.shape[-1] == 3
        b, h, w, _ = pages.shape
        hp = hp or (h + 31) // 32 * 32
        wp = wp or (w + 31) // 32 * 32
        if out is None:
            out = torch.empty((b, hp, wp), dtype=torch.bfloat16, device=pages.device)
        self._chk(self.lib.lumina_ocr_det_forward(self._h, _ptr(pages), b, h, w, hp, wp, _ptr(out), self._stream()))
        return out

    def det_postprocess(self, prob, valid_h: int, valid_w: int, thresh=arch.DET_THRESH, box_thresh=arch.DET_BOX_THRESH,
                        unclip_ratio=arch.DET_UNCLIP_RATIO, min_size=arch.DET_MIN_SIZE, max_boxes=MAX_BOXES):
        torch = _torch()
        b, hp, wp = prob.shape
        boxes = torch.zeros((b, max_boxes, 8), dtype=torch.int32, device=prob.device)
        scores = torch.zeros((b, max_boxes), dtype=torch.float32, device=prob.device)
        counts = torch.zeros((b,), dtype=torch.int32, device=prob.device)
        self._chk(self.lib.lumina_ocr_det_postprocess(self._h, _ptr(prob), b, hp, wp, valid_h, valid_w, thresh, box_thresh,
                                                      unclip_ratio, min_size, max_boxes, _ptr(boxes), _ptr(scores), _ptr(counts),
                                                      self._stream()))
        return boxes, scores, counts

    def rec_crop(self, pages, quads, page_idx):
        torch = _torch()
        b, h, w, _ = pages.shape
        n = quads.shape[0]
        crops = torch.empty((n, REC_H, REC_W, 3), dtype=torch.uint8, device=pages.device)
        widths = torch.empty((n,), dtype=torch.int32, device=pages.device)
        if n:
            self._chk(self.lib.lumina_ocr_rec_crop(self._h, _ptr(pages), b, h, w, _ptr(quads), _ptr(page_idx), n, _ptr(crops),
                                                   _ptr(widths), self._stream()))
        return crops, widths

    def rec_forward(self, crops, widths=None):
        torch = _torch()
        n = crops.shape[0]
        idx = torch.empty((n, REC_T), dtype=torch.int32, device=crops.device)
        prob = torch.empty((n, REC_T), dtype=torch.float32, device=crops.device)
        if n:
            self._chk(self.lib.lumina_ocr_rec_forward(self._h, _ptr(crops), _ptr(widths), n, _ptr(idx), _ptr(prob), self._stream()))
        return idx, prob

    def svtr_forward(self, crops, widths=None):
        """Same contract as rec_forward, SVTR-Tiny backbone."""
        torch = _torch()
        n = crops.shape[0]
        idx = torch.empty((n, REC_T), dtype=torch.int32, device=crops.device)
        prob = torch.empty((n, REC_T), dtype=torch.float32, device=crops.device)
        if n:
            self._chk(self.lib.lumina_ocr_svtr_forward(self._h, _ptr(crops), _ptr(widths), n, _ptr(idx), _ptr(prob), self._stream()))
        return idx, prob

    def ctc_decode(self, idx, prob):
        torch = _torch()
        n = idx.shape[0]
        text = torch.empty((n, REC_T), dtype=torch.int32, device=idx.device)
        length = torch.empty((n,), dtype=torch.int32, device=idx.device)
        score = torch.empty((n,), dtype=torch.float32, device=idx.device)
        if n:
            self._chk(self.lib.lumina_ocr_ctc_decode(self._h, _ptr(idx), _ptr(prob), n, _ptr(text), _ptr(length), _ptr(score),
                                                     self._stream()))
        return text, length, score

    # -- kernel-level ---------------------------------------------------------------------
    def conv2d(self, x, w_ohwi_f32: np.ndarray, bias: np.ndarray, ks: int, stride: int, act: int = 0, res=None):
        """x bf16 [N,H,W,Cin] device; weights OHWI float32 (bf16-exact) host -> y bf16 [N,Ho,Wo,Cout]."""
        torch = _torch()
        n, h, w, cin = x.shape
        cout = w_ohwi_f32.shape[0]
        ho = (h - 1) // stride + 1 if ks == 3 else h // stride
        wo = (w - 1) // stride + 1 if ks == 3 else w // stride
        y = torch.empty((n, ho, wo, cout), dtype=torch.bfloat16, device=x.device)
        wb = np.ascontiguousarray(arch.f32_to_bf16_bits(w_ohwi_f32))
        bb = np.ascontiguousarray(bias, np.float32)
        self._chk(self.lib.lumina_ocr_conv2d(self._h, _ptr(x), n, h, w, cin, wb.ctypes.data, bb.ctypes.data, cout, ks, stride, act,
                                             _ptr(res), _ptr(y), self._stream()))
        return y

    def read_tap(self, name: str, dtype: str = "bf16") -> np.ndarray:
        """Intermediate tensor of the last forward as float32 (dtype: the storage type of that tensor — "f16" for an SVTR fp16 model)."""
        dims = (ctypes.c_int * 4)()
        self._chk(self.lib.lumina_ocr_read_tap(self._h, name.encode(), None, 0, dims))
        n = int(np.prod(list(dims)))
        buf = np.empty(n, np.uint16)
        self._chk(self.lib.lumina_ocr_read_tap(self._h, name.encode(), buf.ctypes.data, n, dims))
        vals = buf.view(np.float16).astype(np.float32) if dtype == "f16" else arch.bf16_bits_to_f32(buf)
        return vals.reshape(tuple(dims))

    def conv_timing(self) -> Tuple[float, float, int]:
        ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        self._chk(self.lib.lumina_ocr_conv_timing(self._h, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n)))
        return ms.value, fl.value, n.value

    def conv_timing_detail(self):
        buf = ctypes.create_string_buffer(1 << 20)
        self._chk(self.lib.lumina_ocr_conv_timing_detail(self._h, buf, len(buf)))
        rows = []
        for line in buf.value.decode().splitlines():
            name, kern, ms, gf, mb = line.split()
            rows.append((name, kern, float(ms), float(gf), float(mb)))
        return rows

    # -- pre-processing on device (image_preprocessing.py:81-110, :132-158) -----------------
    def resize_lanczos(self, img, out_h: int, out_w: int):
        torch = _torch()
        n, h, w, c = img.shape
        out = torch.empty((n, out_h, out_w, c), dtype=torch.uint8, device=img.device)
        self._chk(self.lib.lumina_ocr_resize_lanczos(self._h, _ptr(img), n, h, w, c, _ptr(out), out_h, out_w, self._stream()))
        return out

    # -- de-skew (image_preprocessing.py:372-460) ---------------------------------------------
    def deskew(self, pages, debug: bool = False, estimate_only: bool = False):
        """uint8 [n,H,W,3] device -> (de-skewed pages (or None), rot float64 [n,3] device = sin, cos, flag).  Asynchronous: the
        rotation of each page is estimated and applied on the device.  debug=True also returns (info int32 [n,2] = segments,
        peaks; Canny edge maps uint8 [n,H,W]; segments int32 [n,512,8,4]; segments per peak slot int32 [n,512])."""
        torch = _torch()
        n, h, w, c = pages.shape
        assert c == 3 and pages.dtype == torch.uint8 and pages.is_contiguous()
        out = None if estimate_only else torch.empty_like(pages)
        rot = torch.empty((n, 3), dtype=torch.float64, device=pages.device)
        info = torch.empty((n, 2), dtype=torch.int32, device=pages.device)
        edges = segs = nsegs = None
        if debug:
            edges = torch.empty((n, h, w), dtype=torch.uint8, device=pages.device)
            segs = torch.zeros((n, 512, 8, 4), dtype=torch.int32, device=pages.device)
            nsegs = torch.zeros((n, 512), dtype=torch.int32, device=pages.device)
        self._chk(self.lib.lumina_ocr_deskew(self._h, _ptr(pages), n, h, w, _ptr(out), _ptr(rot), _ptr(info), _ptr(edges), _ptr(segs), _ptr(nsegs),
                                             self._stream()))
        return (out, rot, info, edges, segs, nsegs) if debug else (out, rot)

    def deskew_warp(self, pages, rot):
        """The cubic warp alone for given (sin, cos, flag) triples (float64 [n,3] device); flag != 3 copies the page."""
        torch = _torch()
        n, h, w, _ = pages.shape
        out = torch.empty_like(pages)
        self._chk(self.lib.lumina_ocr_deskew_warp(self._h, _ptr(pages), n, h, w, _ptr(rot), _ptr(out), self._stream()))
        return out

    @staticmethod
    def skew_degrees(rot) -> list:
        """rot (device or host [n,3]) -> the angle the reference's deskew() returns next to the image (:441-447, :460):
        the detected angle when the page was rotated or left alone below 0.5 degrees, 0.0 otherwise.  Synchronises."""
        r = rot.cpu().numpy() if hasattr(rot, "cpu") else np.asarray(rot)
        return [float(np.degrees(np.arctan2(s, c))) if int(f) in (1, 3) else 0.0 for s, c, f in r]

    def binarize(self, img, adaptive: bool = True, threshold: int = 128):
        """image_preprocessing.py:462-494 (adaptive=True: Gaussian 11x11 adaptive threshold, C = 2) / :175-185 (adaptive=False: L > threshold,
        the reference's behaviour without OpenCV).  uint8 [n,H,W,3] device -> 0 / 255 on all three channels."""
        torch = _torch()
        n, h, w, c = img.shape
        assert c == 3 and img.dtype == torch.uint8
        out = torch.empty_like(img)
        self._chk(self.lib.lumina_ocr_binarize(self._h, _ptr(img), n, h, w, int(bool(adaptive)), int(threshold), _ptr(out), self._stream()))
        return out

    def exif_transpose(self, img, orientation: int):
        """auto_orient / ImageOps.exif_transpose (image_preprocessing.py:171-173) for EXIF orientation 1..8.  uint8 [n,H,W,3] device -> [n,H,W,3]
        (1..4) or [n,W,H,3] (5..8)."""
        torch = _torch()
        n, h, w, c = img.shape
        assert c == 3 and img.dtype == torch.uint8
        if orientation in (0, 1):
            return img
        out = torch.empty((n, w, h, 3) if orientation >= 5 else (n, h, w, 3), dtype=torch.uint8, device=img.device)
        self._chk(self.lib.lumina_ocr_exif_transpose(self._h, _ptr(img), n, h, w, int(orientation), _ptr(out), self._stream()))
        return out

    def grayscale(self, img):
        """convert_to_grayscale (image_preprocessing.py:167-169): PIL convert('L'), on all three channels.  uint8 [n,H,W,3] device."""
        torch = _torch()
        n, h, w, c = img.shape
        assert c == 3 and img.dtype == torch.uint8
        out = torch.empty_like(img)
        self._chk(self.lib.lumina_ocr_grayscale(self._h, _ptr(img), n, h, w, _ptr(out), self._stream()))
        return out

    def denoise(self, img):
        """denoise (image_preprocessing.py:160-165): PIL MedianFilter(3) per channel.  uint8 [n,H,W,3] device."""
        torch = _torch()
        n, h, w, c = img.shape
        assert c == 3 and img.dtype == torch.uint8
        out = torch.empty_like(img)
        self._chk(self.lib.lumina_ocr_denoise(self._h, _ptr(img), n, h, w, _ptr(out), self._stream()))
        return out

    def enhance(self, img, contrast: float = 1.2, sharpness: float = 1.1):
        torch = _torch()
        n, h, w, c = img.shape
        out = torch.empty_like(img)
        tmp = torch.empty_like(img)
        self._chk(self.lib.lumina_ocr_enhance(self._h, _ptr(img), n, h, w, contrast, sharpness, _ptr(tmp), _ptr(out), self._stream()))
        return out
