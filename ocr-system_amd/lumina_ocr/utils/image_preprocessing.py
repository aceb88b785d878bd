"""Host-side mirror of /root/reference/backend/utils/image_preprocessing.py for the MI355X engine.

Same public names and argument meaning; decode / EXIF / JPEG stay PIL host plumbing (they are in the reference
too), while the pixel work — LANCZOS resize (:81-110), contrast (:132-144), sharpness (:146-158) — runs on the GPU
through liblumina_ocr.so, byte-exact with PIL (tests/golden/preprocess_vectors.npz).  `deskew` / adaptive
binarisation need OpenCV in the reference and degrade to no-ops without it (:383-385, :473-475); they are
no-ops here as well (SURVEY.md §8f rank 2 lists deskew as a later row).
"""
from __future__ import annotations

import io
from pathlib import Path
from typing import List, Optional, Tuple, Union

import numpy as np
from PIL import Image, ImageOps

DEFAULT_MAX_DIMENSION = 2000  # settings.OCR_MAX_IMAGE_DIMENSION (/root/reference/backend/config.py:69)


def get_optimal_size(width: int, height: int, max_dimension: int = DEFAULT_MAX_DIMENSION) -> Tuple[int, int]:
    """:112-126 — long side capped at max_dimension, short side int()-truncated."""
    if max(width, height) <= max_dimension:
        return width, height
    if width > height:
        return max_dimension, int(height * (max_dimension / width))
    return int(width * (max_dimension / height)), max_dimension


class ImagePreprocessor:
    def __init__(self, max_dimension: Optional[int] = None, target_dpi: int = 300, engine=None):
        self.max_dimension = max_dimension or DEFAULT_MAX_DIMENSION
        self.target_dpi = target_dpi
        self._engine = engine

    # ---- loading (:57-75) ----
    def load_image(self, image_path: Union[str, Path]) -> Image.Image:
        path = Path(image_path)
        if not path.exists():
            raise FileNotFoundError(f"Image not found: {path}")
        return self._normalise_mode(Image.open(path))

    def load_image_bytes(self, image_bytes: bytes) -> Image.Image:
        return self._normalise_mode(Image.open(io.BytesIO(image_bytes)))

    @staticmethod
    def _normalise_mode(image: Image.Image) -> Image.Image:
        return image if image.mode in ("RGB", "L") else image.convert("RGB")

    def auto_orient(self, image: Image.Image) -> Image.Image:
        """ImageOps.exif_transpose (image_preprocessing.py:206-211 of the reference).  Pillow copies the image even when there is
        nothing to transpose (0.75 ms per A4 page): without an Orientation tag other than 1 the image itself is the answer."""
        try:
            if image.getexif().get(0x0112, 1) in (0, 1):
                return image
        except Exception:  # an unreadable EXIF block: let Pillow decide
            pass
        return ImageOps.exif_transpose(image)

    def get_optimal_size(self, width: int, height: int, max_dimension: Optional[int] = None) -> Tuple[int, int]:
        return get_optimal_size(width, height, max_dimension or self.max_dimension)

    # ---- device work ----
    def _eng(self):
        if self._engine is None:
            from ..engine import Engine
            self._engine = Engine(0)
        return self._engine

    def to_device(self, image: Union[Image.Image, np.ndarray]):
        import torch
        arr = np.asarray(image)
        if arr.ndim == 2:
            arr = arr[..., None]
        return torch.from_numpy(np.ascontiguousarray(arr))[None].cuda()

    def resize_if_needed(self, image, max_dimension: Optional[int] = None):
        """PIL image / HWC array -> resized uint8 device tensor [1,H',W',C] (unchanged size: plain upload)."""
        x = image if hasattr(image, "is_cuda") else self.to_device(image)
        _, h, w, _ = x.shape
        nw, nh = get_optimal_size(w, h, max_dimension or self.max_dimension)
        if nw <= 0 or nh <= 0:
            raise ValueError("height and width must be > 0")  # the reference fails the same way inside PIL
        return x if (nw, nh) == (w, h) else self._eng().resize_lanczos(x, nh, nw)

    def optimize_for_ocr(self, image, apply_contrast: bool = True, apply_sharpness: bool = True, apply_denoise: bool = False, grayscale: bool = False):
        """:191-242, same arguments and order: EXIF -> resize -> [grayscale] -> [denoise: 3x3 median] -> contrast 1.2 -> sharpness 1.1, every
        pixel step on the device.  The page stays three-channel: with grayscale=True all channels carry the reference's L value (its
        ImageEnhance steps give the same bytes on an L image and on the R = G = B replica)."""
        if isinstance(image, bytes):
            image = self.load_image_bytes(image)
        elif isinstance(image, (str, Path)):
            image = self.load_image(image)
        if isinstance(image, Image.Image):
            image = self.auto_orient(image)
            if image.mode != "RGB":
                image = image.convert("RGB")
        x = self.resize_if_needed(image)
        if grayscale:
            x = self._eng().grayscale(x)
        if apply_denoise:
            x = self._eng().denoise(x)
        if apply_contrast or apply_sharpness:
            x = self._eng().enhance(x, 1.2 if apply_contrast else 1.0, 1.1 if apply_sharpness else 1.0)
        return x

    # ---- JPEG hand-off (:496-557): processed_image_bytes must be JPEG (file_manager.py:283-287) ----
    def compress_for_azure(self, image: Image.Image, target_size_mb: float = 2.0, initial_quality: int = 95, min_quality: int = 30) -> bytes:
        target = int(target_size_mb * 1024 * 1024)
        if image.mode != "RGB":
            image = image.convert("RGB")
        q = initial_quality
        while q >= min_quality:
            buf = io.BytesIO()
            image.save(buf, format="JPEG", quality=q, optimize=True)
            if buf.tell() <= target:
                return buf.getvalue()
            q -= 10
        buf = io.BytesIO()
        image.save(buf, format="JPEG", quality=min_quality)
        scale = (target / buf.tell()) ** 0.5
        small = image.resize((int(image.width * scale), int(image.height * scale)), Image.Resampling.LANCZOS)
        buf = io.BytesIO()
        small.save(buf, format="JPEG", quality=min_quality, optimize=True)
        return buf.getvalue()

    def compress_for_azure_device(self, pages, target_size_mb: float = 2.0, initial_quality: int = 95, min_quality: int = 30) -> List[bytes]:
        """compress_for_azure (:495-557) for a batch of processed pages that are already on the device (uint8 [n,H,W,3]):
        the same quality loop — 95, 85, ... down to min_quality until a page's file fits target_size_mb — with every encode done by
        the engine (lumina_ocr_jpeg_encode, byte-identical to the PIL call inside the reference's loop); pages that still do not
        fit are LANCZOS-resized on the device by sqrt(target / size) and encoded at min_quality, as the reference does."""
        eng = self._eng()
        target = int(target_size_mb * 1024 * 1024)
        n = pages.shape[0]
        result: List[Optional[bytes]] = [None] * n
        todo = list(range(n))
        q = initial_quality
        while todo and q >= min_quality:
            sub = pages if len(todo) == n else pages[todo]
            files, sizes = eng.jpeg_encode(sub, q, max_bytes=target)
            sizes_h = sizes.cpu().numpy()
            fit = [k for k in range(len(todo)) if 0 < sizes_h[k] <= target]
            if fit:
                files_h = files[fit].cpu().numpy()
                for j, k in enumerate(fit):
                    result[todo[k]] = files_h[j, : sizes_h[k]].tobytes()
            todo = [todo[k] for k in range(len(todo)) if not (0 < sizes_h[k] <= target)]
            q -= 10
        for i in todo:  # quality reduction was not enough: also resize (:540-557)
            page = pages[i:i + 1]
            h, w = int(page.shape[1]), int(page.shape[2])
            _, probe = eng.jpeg_encode(page, min_quality, max_bytes=target, optimize=False)   # size probe WITHOUT optimised tables (:548)
            current = abs(int(probe[0]))
            scale = (target / current) ** 0.5
            small = eng.resize_lanczos(page, int(h * scale), int(w * scale))
            files, sizes = eng.jpeg_encode(small, min_quality, max_bytes=2 * target)
            sz = int(sizes[0])
            if sz <= 0:
                raise ValueError("resized page still exceeds twice the JPEG size target")
            result[i] = files[0, :sz].cpu().numpy().tobytes()
        return result  # type: ignore[return-value]

    def pdf_to_images(self, pdf_path: Union[str, Path], dpi: Optional[int] = None) -> List[Image.Image]:
        """:248-295 — needs pdf2image + poppler exactly like the reference; absent here -> ImportError as data upstream."""
        try:
            from pdf2image import convert_from_path
        except ImportError:
            raise ImportError("pdf2image not installed. Install with: pip install pdf2image (and poppler)")
        path = Path(pdf_path)
        if not path.exists():
            raise FileNotFoundError(f"PDF not found: {path}")
        return list(convert_from_path(str(path), dpi=dpi or self.target_dpi, fmt="png"))


image_preprocessor = ImagePreprocessor()
