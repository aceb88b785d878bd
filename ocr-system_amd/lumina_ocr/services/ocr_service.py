"""OCR provider backed by the MI355X det+rec engine — a drop-in for the reference's provider module.

Replace /root/reference/backend/services/ocr_service.py with this module: callers do
`from services.ocr_service import OCRService, DocumentOCRResult` and `OCRService()`
(/root/reference/backend/services/extraction_service.py:46, :207).  Mirrored surface (reference file:line):
  OCROutput :48-79, DocumentOCRResult :82-104 (same fields, same to_dict keys; processed_image_bytes excluded)
  OCRService: singleton :126-135, process_image_sync :477-502, _process_single_image_sync :398-475,
    process_pdf_sync :508-602, process_pdf_as_images_sync :604-660, process_image/process_pdf (async) :666-693,
    process_document :695-731, get_status :759-771, preload_model :773-775, is_model_loaded :777-780, cleanup :782-795
  module level: ocr_service :802, ocr_node :805-829, preload_ocr_model :832-837, get_ocr_status :840-842
Errors are data, never exceptions (:464-475, :653-660).  Box coordinates, page_width_inches/page_height_inches
and processed_image_bytes (JPEG) all refer to the processed image's pixel grid.
There is no CPU fallback: without the HIP library / a GPU every call returns success=False with the reason.
"""
from __future__ import annotations

import asyncio
import io
import logging
import os
import threading
import time
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Dict, List, Optional, Union

import numpy as np
from PIL import Image

from .. import arch
from ..utils import layout
from ..utils.image_preprocessing import ImagePreprocessor, get_optimal_size

logger = logging.getLogger(__name__)

SUPPORTED_IMAGE_TYPES = ("png", "jpg", "jpeg", "webp", "bmp", "tiff")


@dataclass
class OCROutput:
    markdown: str = ""
    html: str = ""
    json_output: Dict[str, Any] = field(default_factory=dict)
    processing_time_ms: int = 0
    success: bool = True
    error: Optional[str] = None
    page_number: int = 1
    image_width: int = 0
    image_height: int = 0
    layout_boxes: List[Dict[str, Any]] = field(default_factory=list)
    processed_image_bytes: Optional[bytes] = None
    page_width_inches: float = 0.0
    page_height_inches: float = 0.0

    def to_dict(self) -> Dict[str, Any]:
        keys = ("markdown", "html", "json_output", "processing_time_ms", "success", "error", "page_number", "image_width",
                "image_height", "layout_boxes", "page_width_inches", "page_height_inches")
        return {k: getattr(self, k) for k in keys}


@dataclass
class DocumentOCRResult:
    pages: List[OCROutput] = field(default_factory=list)
    total_pages: int = 0
    total_processing_time_ms: int = 0
    success: bool = True
    error: Optional[str] = None
    combined_markdown: str = ""
    combined_html: str = ""
    combined_layout_boxes: List[Dict[str, Any]] = field(default_factory=list)

    def to_dict(self) -> Dict[str, Any]:
        d = {k: getattr(self, k) for k in ("total_pages", "total_processing_time_ms", "success", "error", "combined_markdown",
                                           "combined_html", "combined_layout_boxes")}
        return {"pages": [p.to_dict() for p in self.pages], **d}


def _ms_since(t0: float) -> int:
    return int((time.time() - t0) * 1000)


class OCRService:
    """Process-wide singleton; pages are serialised by a semaphore exactly like the reference (:157, :404)."""

    _instance = None
    _lock = threading.Lock()

    def __new__(cls):
        if cls._instance is None:
            with cls._lock:
                if cls._instance is None:
                    cls._instance = super().__new__(cls)
                    cls._instance._initialized = False
        return cls._instance

    def __init__(self):
        if getattr(self, "_initialized", False):
            return
        self._engine = None
        self._pipeline = None
        self._engine_lock = threading.Lock()
        self._semaphore = threading.Semaphore(1)
        self.max_dimension = int(os.environ.get("OCR_MAX_IMAGE_DIMENSION", 2000))
        # settings.OCR_APPLY_DESKEW (/root/reference/backend/config.py:85, default True; used at ocr_service.py:150, :412-417)
        self.apply_deskew = os.environ.get("OCR_APPLY_DESKEW", "true").lower() not in ("0", "false", "no")
        # settings.PREPROCESSING_APPLY_BINARIZE (ocr_service.py:151, default False): "true" / "adaptive" = cv2.adaptiveThreshold semantics,
        # "simple" = the L > 128 threshold the reference falls back to without OpenCV (image_preprocessing.py:473-475)
        b = os.environ.get("PREPROCESSING_APPLY_BINARIZE", "false").lower()
        self.device_jpeg = os.environ.get("LUMINA_OCR_DEVICE_JPEG", "1").lower() not in ("0", "false", "no")   # baseline JPEG inputs are decoded on the device
        self.apply_binarize = "adaptive" if b in ("1", "true", "yes", "adaptive") else ("simple" if b == "simple" else None)
        self._device = int(os.environ.get("LUMINA_OCR_DEVICE", os.environ.get("LOCAL_RANK", 0)))
        self._det_weights = os.environ.get("LUMINA_OCR_DET_WEIGHTS", "")
        self._rec_weights = os.environ.get("LUMINA_OCR_REC_WEIGHTS", "")
        self._recognizer = os.environ.get("LUMINA_OCR_RECOGNIZER", "crnn")        # "crnn" | "svtr" (BASELINE configs[4] family)
        self._svtr_weights = os.environ.get("LUMINA_OCR_SVTR_WEIGHTS", "")
        self._rec_dict = os.environ.get("LUMINA_OCR_REC_DICT", "")               # dictionary file: one symbol per line (PP-OCR key-file format)
        self._allow_synthetic = os.environ.get("LUMINA_OCR_ALLOW_SYNTHETIC", "") == "1"
        self._weights_kind = "unloaded"
        self._pre = ImagePreprocessor(self.max_dimension)
        self._initialized = True

    # ---- engine management (reference: _ensure_client_initialized :166-207) ----
    def _ensure_engine(self) -> None:
        """Builds the engine once.  Like the reference without Azure credentials (:175-195), a provider without its weights is
        an ERROR, returned as data by the callers: seeded synthetic networks are only used when LUMINA_OCR_ALLOW_SYNTHETIC=1
        says so (tests, demos, the benchmark), never silently."""
        if self._pipeline is not None:
            return
        with self._engine_lock:
            if self._pipeline is not None:
                return
            import torch
            from ..engine import Engine
            from ..pipeline import OcrPipeline
            svtr = self._recognizer == "svtr"
            have_files = bool(self._det_weights) and bool(self._svtr_weights if svtr else self._rec_weights)
            if not have_files and not self._allow_synthetic:
                missing = [k for k, v in (("LUMINA_OCR_DET_WEIGHTS", self._det_weights),
                                          ("LUMINA_OCR_SVTR_WEIGHTS" if svtr else "LUMINA_OCR_REC_WEIGHTS", self._svtr_weights if svtr else self._rec_weights)) if not v]
                raise RuntimeError("OCR weights not configured: set %s (LOCW blobs) and LUMINA_OCR_REC_DICT, or LUMINA_OCR_ALLOW_SYNTHETIC=1 "
                                   "for seeded synthetic networks" % " and ".join(missing))
            if have_files and not self._rec_dict:
                raise RuntimeError("LUMINA_OCR_REC_DICT (the dictionary file the recogniser was trained with) is required with weight files")
            eng = Engine(self._device)  # raises EngineUnavailable without the HIP library / a GPU
            try:
                with torch.cuda.device(self._device):
                    if have_files:
                        eng.load_det(Path(self._det_weights).read_bytes())
                        if svtr:
                            eng.load_svtr(Path(self._svtr_weights).read_bytes())
                        else:
                            eng.load_rec(Path(self._rec_weights).read_bytes())
                        charset = arch.load_charset(self._rec_dict)
                        kind, post = "files", arch.DEFAULT_POST
                    else:  # no trained weights ship offline: deterministic seeded networks, on request only
                        logger.warning("OCR provider is running SEEDED SYNTHETIC networks (LUMINA_OCR_ALLOW_SYNTHETIC=1): recognised text is not meaningful")
                        eng.load_det(arch.make_det_weights())
                        charset = arch.load_charset(self._rec_dict) if self._rec_dict else arch.ctc_charset()
                        if svtr:
                            eng.load_svtr(arch.make_svtr_weights(num_classes=len(charset)))
                        else:
                            eng.load_rec(arch.make_rec_weights(num_classes=len(charset), code_path=True))
                        kind, post = "seeded-synthetic", arch.TEXT_PATH_POST
                    n_cls = eng.svtr_num_classes if svtr else eng.num_classes
                    if len(charset) != n_cls:
                        raise RuntimeError("dictionary has %d classes (blank + symbols + space) but the %s head has %d"
                                           % (len(charset), "SVTR" if svtr else "CRNN", n_cls))
                    pipeline = OcrPipeline(eng, charset=charset, max_dimension=self.max_dimension, post=post, recognizer=self._recognizer)
            except Exception:
                eng.close()
                raise
            self._weights_kind = kind
            self._engine = eng
            self._pre._engine = eng
            pipeline.binarize = self.apply_binarize
            self._pipeline = pipeline

    def _device_ctx(self):
        """Binds the calling thread (possibly an asyncio.to_thread worker, which starts on device 0) to the engine's GPU."""
        import torch
        return torch.cuda.device(self._device)

    # ---- single image (:398-475) ----
    def _prepare(self, image: Image.Image) -> Image.Image:
        """EXIF orientation, RGB, size check (optimize_for_ocr's first steps, image_preprocessing.py:206-215) -> PIL RGB image."""
        image = self._pre.auto_orient(image)
        if image.mode != "RGB":
            image = image.convert("RGB")
        w, h = image.size
        nw, nh = get_optimal_size(w, h, self.max_dimension)
        if nw <= 0 or nh <= 0:
            raise ValueError("height and width must be > 0")
        return image

    def _stage_pages(self, images: List[Image.Image]):
        """Same-size PIL RGB pages -> one host uint8 tensor [n,H,W,3] (pinned when there is a GPU).  The pixels go from Pillow's raw
        encoder straight into the staging buffer: np.asarray(image) + np.stack + a pageable upload were 5 ms of the 7.5 ms host
        time per A4 page (tobytes() joins 64 KB chunks, stack copies them again)."""
        import torch
        w, h = images[0].size
        n = len(images)
        key = (n, h, w)
        if getattr(self, "_stage_key", None) != key:
            self._stage = torch.empty((n, h, w, 3), dtype=torch.uint8)
            if torch.cuda.is_available():
                self._stage = self._stage.pin_memory()
            self._stage_key = key
        flat = self._stage.view(n, -1).numpy()
        row_bytes = w * 3
        chunk = max(row_bytes, (4 << 20) // row_bytes * row_bytes)   # whole rows, ~4 MB per encoder call
        for i, im in enumerate(images):
            if not self._raw_copy(im, flat[i], h * row_bytes, chunk):
                self._stage[i] = torch.from_numpy(np.asarray(im, np.uint8))   # any Pillow whose raw encoder is not driven this way
        return self._stage

    @staticmethod
    def _raw_copy(im: Image.Image, dst: np.ndarray, nbytes: int, chunk: int) -> bool:
        """What Image.tobytes() does, minus the join: the raw encoder's chunks are written where they are wanted."""
        try:
            im.load()
            enc = Image._getencoder("RGB", "raw", "RGB")
            try:
                enc.setimage(im.im, (0, 0) + im.size)
            except TypeError:   # Pillow < 10: setimage(im)
                enc.setimage(im.im)
            off = 0
            while True:
                _, status, data = enc.encode(chunk)
                dst[off:off + len(data)] = np.frombuffer(data, np.uint8)
                off += len(data)
                if status:
                    break
            return status > 0 and off == nbytes
        except Exception:
            return False

    def _upload(self, staged):
        """Staging buffer -> the engine's device (worker threads start on device 0 whatever LUMINA_OCR_DEVICE says)."""
        import torch
        dev = torch.device("cuda", self._device)
        out = staged.to(dev, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()   # the staging buffer is reused by the next call
        return out

    def _finish_page(self, det, jpeg: bytes, processed_hw, page_number: int, original_size, t0: float) -> OCROutput:
        merged, ordered = layout.reading_order(det.triples())
        md = layout.page_markdown(merged)
        paragraphs = layout.build_paragraph_boxes(merged, page_number)
        boxes = layout.build_layout_boxes(ordered, page_number) + paragraphs     # words, lines, ..., paragraphs: the order of ocr_service.py:285-367
        ph, pw = processed_hw
        return OCROutput(markdown=md, html=layout.html_from_markdown(md),
                         json_output={"page_count": 1, "words_count": sum(1 for b in boxes if b["type"] == "word"),
                                      "lines_count": len(ordered), "tables_count": 0, "paragraphs_count": len(paragraphs)},
                         processing_time_ms=_ms_since(t0), success=True, page_number=page_number, image_width=original_size[0],
                         image_height=original_size[1], layout_boxes=boxes, processed_image_bytes=jpeg,
                         page_width_inches=float(pw), page_height_inches=float(ph))

    def _process_single_image_sync(self, image: Image.Image, page_number: int = 1, decoded=None) -> OCROutput:
        """decoded: the page already on the device (uint8 [1,H,W,3], from the device JPEG decoder) — `image` is then only consulted
        for its size."""
        with self._semaphore:
            t0 = time.time()
            original_size = image.size
            try:
                import torch
                self._ensure_engine()
                with self._device_ctx():
                    if decoded is None:
                        decoded = self._upload(self._stage_pages([self._prepare(image)]))
                    else:
                        w, h = original_size
                        nw, nh = get_optimal_size(w, h, self.max_dimension)
                        if nw <= 0 or nh <= 0:
                            raise ValueError("height and width must be > 0")
                    dets, processed = self._pipeline.run(decoded, deskew=self.apply_deskew)
                    jpeg = self._pre.compress_for_azure_device(processed)[0]   # processed_image_bytes: encoded on the device
                return self._finish_page(dets[0], jpeg, tuple(processed.shape[1:3]), page_number, original_size, t0)
            except Exception as e:  # errors are data (:464-475)
                logger.error("OCR failed: %s", e)
                return OCROutput(success=False, error=str(e), processing_time_ms=_ms_since(t0), page_number=page_number,
                                 image_width=original_size[0], image_height=original_size[1])

    def _decode_jpeg_on_device(self, data: bytes, image: Image.Image):
        """The reference decodes every input with Image.open (image_preprocessing.py:57-75).  For a baseline JPEG
        the pixels are produced on the device instead (lumina_ocr_jpeg_decode: byte-identical to Pillow's decode; grey files arrive with
        their value on all three channels, which is what convert('RGB') gives): nothing is decoded on the host, the file's
        entropy-coded bytes are what crosses PCIe (~10x less than the pixels).  -> device tensor [1,H,W,3], or None: Pillow decodes."""
        if not self.device_jpeg or image.format != "JPEG" or data[:2] != b"\xff\xd8":
            return None
        try:
            orientation = image.getexif().get(0x0112, 1)
            if orientation not in range(0, 9):
                return None
            from ..engine import Engine
            rc, info = Engine.jpeg_probe(data)
            if rc != 0 or (info["width"], info["height"]) != image.size:
                return None
            self._ensure_engine()
            with self._device_ctx():
                out, status = self._engine.jpeg_decode([data], info["height"], info["width"])
                if status != [0]:
                    return None
                return self._engine.exif_transpose(out, orientation)     # auto_orient (image_preprocessing.py:213), on the device as well
        except Exception as e:       # any doubt: the reference's own path
            logger.warning("device JPEG decode not used: %s", e)
            return None

    def process_image_sync(self, image_source: Union[str, Path, Image.Image, bytes], page_number: int = 1) -> OCROutput:
        data = None
        if isinstance(image_source, bytes):
            data = image_source
            image = self._pre.load_image_bytes(image_source)
        elif isinstance(image_source, (str, Path)):
            image = self._pre.load_image(image_source)
            if getattr(image, "format", None) == "JPEG" and self.device_jpeg:
                try:
                    data = Path(image_source).read_bytes()
                except OSError:
                    data = None
        elif isinstance(image_source, Image.Image):
            image = image_source
        else:
            raise ValueError(f"Unsupported image type: {type(image_source)}")
        decoded = self._decode_jpeg_on_device(data, image) if data is not None else None      # (Image.open is lazy: no pixel was decoded yet)
        return self._process_single_image_sync(image, page_number, decoded=decoded)

    # ---- page batches: the data-parallel unit (reference loops pages serially, :620-627) ----
    def process_pages_sync(self, images: List[Image.Image], first_page_number: int = 1) -> List[OCROutput]:
        """Same-size pages go through the engine as one batch; results are identical to the per-page path."""
        out: List[Optional[OCROutput]] = [None] * len(images)
        with self._semaphore:
            groups: Dict[Any, List[int]] = {}
            prepared: List[Optional[Image.Image]] = [None] * len(images)
            for i, im in enumerate(images):
                try:
                    prepared[i] = self._prepare(im)   # (EXIF orientation may swap width and height: group by the prepared size)
                    groups.setdefault(prepared[i].size, []).append(i)
                except Exception as e:
                    out[i] = OCROutput(success=False, error=str(e), page_number=first_page_number + i, image_width=im.size[0], image_height=im.size[1])
            for size, idxs in groups.items():
                t0 = time.time()
                try:
                    import torch
                    self._ensure_engine()
                    batch = [prepared[i] for i in idxs]
                    with self._device_ctx():
                        dets, processed = self._pipeline.run(self._upload(self._stage_pages(batch)), deskew=self.apply_deskew)
                        jpegs = self._pre.compress_for_azure_device(processed)
                    for j, i in enumerate(idxs):
                        out[i] = self._finish_page(dets[j], jpegs[j], tuple(processed.shape[1:3]), first_page_number + i, images[i].size, t0)
                except Exception as e:
                    for i in idxs:
                        out[i] = OCROutput(success=False, error=str(e), processing_time_ms=_ms_since(t0),
                                           page_number=first_page_number + i, image_width=images[i].size[0], image_height=images[i].size[1])
        return out  # type: ignore[return-value]

    def _document_from_pages(self, pages: List[OCROutput], t0: float) -> DocumentOCRResult:
        ok = all(p.success for p in pages)
        boxes: List[Dict[str, Any]] = []
        for p in pages:
            boxes.extend(p.layout_boxes)
        return DocumentOCRResult(pages=pages, total_pages=len(pages), total_processing_time_ms=_ms_since(t0), success=ok,
                                 error=None if ok else "Some pages failed", combined_markdown=layout.combine_markdown(pages),
                                 combined_html=layout.combine_html(pages), combined_layout_boxes=boxes)

    # ---- PDF (:508-660): every PDF goes through the rasterise-and-batch path ----
    def process_pdf_as_images_sync(self, pdf_path: Union[str, Path]) -> DocumentOCRResult:
        t0 = time.time()
        try:
            images = self._pre.pdf_to_images(pdf_path)
            if not images:
                return DocumentOCRResult(success=False, error="No pages found in PDF")
            return self._document_from_pages(self.process_pages_sync(images), t0)
        except Exception as e:
            return DocumentOCRResult(success=False, error=str(e), total_processing_time_ms=_ms_since(t0))

    def process_pdf_sync(self, pdf_path: Union[str, Path]) -> DocumentOCRResult:
        if not Path(pdf_path).exists():
            return DocumentOCRResult(success=False, error=f"File not found: {Path(pdf_path)}")
        return self.process_pdf_as_images_sync(pdf_path)

    # ---- async wrappers (:666-731) ----
    async def process_image(self, image_source, page_number: int = 1, timeout: float = 120.0) -> OCROutput:
        try:
            return await asyncio.wait_for(asyncio.to_thread(self.process_image_sync, image_source, page_number), timeout=timeout)
        except asyncio.TimeoutError:
            return OCROutput(success=False, error=f"Timed out after {timeout}s")

    async def process_pdf(self, pdf_path, timeout: float = 600.0) -> DocumentOCRResult:
        try:
            return await asyncio.wait_for(asyncio.to_thread(self.process_pdf_sync, pdf_path), timeout=timeout)
        except asyncio.TimeoutError:
            return DocumentOCRResult(success=False, error=f"Timed out after {timeout}s")

    async def process_document(self, file_path: Union[str, Path], file_type: str) -> DocumentOCRResult:
        file_type = file_type.lower().strip(".")
        path = Path(file_path)
        if not path.exists():
            return DocumentOCRResult(success=False, error=f"File not found: {path}")
        if file_type == "pdf":
            return await self.process_pdf(path)
        if file_type in SUPPORTED_IMAGE_TYPES:
            r = await self.process_image(path)
            return DocumentOCRResult(pages=[r], total_pages=1, total_processing_time_ms=r.processing_time_ms, success=r.success,
                                     error=r.error, combined_markdown=r.markdown, combined_html=r.html, combined_layout_boxes=r.layout_boxes)
        return DocumentOCRResult(success=False, error=f"Unsupported file type: {file_type}")

    # ---- status (:759-795) ----
    def get_status(self) -> Dict[str, Any]:
        st = {"client_initialized": self._pipeline is not None, "model_id": "dbnet-r18vd+crnn-mv3", "max_dimension": self.max_dimension,
              "device": self._device, "weights": self._weights_kind, "recognizer": self._recognizer, "apply_deskew": self.apply_deskew, "apply_binarize": bool(self.apply_binarize), "engine": "Lumina MI355X det+rec (HIP, gfx950)"}
        if self._engine is not None:
            st["engine_version"] = self._engine.version()
            st["num_classes"] = self._engine.num_classes
        return st

    def preload_model(self) -> None:
        self._ensure_engine()

    @property
    def is_model_loaded(self) -> bool:
        return self._pipeline is not None

    def cleanup(self) -> None:
        with self._engine_lock:
            if self._engine is not None:
                self._engine.close()
            self._engine = self._pipeline = None


ocr_service = OCRService()


async def ocr_node(state: Dict[str, Any]) -> Dict[str, Any]:
    """LangGraph node (:805-829)."""
    path = state.get("document_path")
    if not path:
        return {**state, "ocr_result": None, "ocr_markdown": "", "ocr_success": False, "ocr_error": "No document_path in state", "ocr_time_ms": 0}
    r = await ocr_service.process_document(path, state.get("file_type", ""))
    return {**state, "ocr_result": r.to_dict(), "ocr_markdown": r.combined_markdown, "ocr_success": r.success, "ocr_error": r.error,
            "ocr_time_ms": r.total_processing_time_ms}


def preload_ocr_model() -> None:
    try:
        ocr_service.preload_model()
    except Exception as e:
        logger.warning("OCR preload failed: %s", e)


async def get_ocr_status() -> Dict[str, Any]:
    return ocr_service.get_status()
