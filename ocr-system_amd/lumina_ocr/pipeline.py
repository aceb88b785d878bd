"""Batched device pipeline: pages -> (resize/enhance) -> DBNet -> DB post-process -> crops -> CRNN -> CTC.

This is the arithmetic that fills the reference's engine slot
(/root/reference/backend/services/ocr_service.py:420 `_analyze_with_azure`, :428 `_extract_layout_boxes`);
every stage is a C-ABI call into liblumina_ocr.so.  torch is used for buffers, the stream and two tiny
index-gather ops between stages; one host sync (box counts) separates det from rec.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import arch
from .engine import MAX_BOXES, Engine
from .utils.image_preprocessing import get_optimal_size


@dataclass
class PageDetections:
    quads: np.ndarray          # int32 [n, 8]  TL,TR,BR,BL in processed-image pixels
    texts: List[str]
    scores: np.ndarray         # float32 [n]  CTC mean max-prob
    det_scores: np.ndarray     # float32 [n]  DB box score
    width: int = 0             # processed image size
    height: int = 0
    text_ids: Optional[np.ndarray] = None   # int32 [n, 80] class ids (-1 padded): what travels in the multi-GPU gather
    lens: Optional[np.ndarray] = None       # int32 [n]

    def triples(self) -> List[Tuple[Sequence[int], str, float]]:
        return [(self.quads[i].tolist(), self.texts[i], float(self.scores[i])) for i in range(len(self.texts))]


@dataclass
class _Pending:
    """A batch whose device work is enqueued: pinned host copies of the outputs + the event that completes them."""
    b: int
    w: int
    h: int
    counts_h: np.ndarray
    n: int
    processed: object
    host: Optional[list] = None
    event: Optional[object] = None
    gathered: Optional[object] = None   # multi-GPU: handle of dist.PageGather.submit (the batch's results of ALL ranks)


class OcrPipeline:
    def __init__(self, engine: Engine, charset: Optional[List[str]] = None, max_dimension: int = 2000, post: Optional[dict] = None,
                 recognizer: str = "crnn", gather=None):
        """recognizer: "crnn" (CRNN-MobileNetV3 + BiLSTM, engine.load_rec) or "svtr" (SVTR, engine.load_svtr).
        gather: a dist.PageGather — multi-GPU runs: every batch's results are all-gathered from the device tensors and
        finish() returns a GatheredPages over the pages of ALL ranks instead of this rank's PageDetections."""
        assert recognizer in ("crnn", "svtr")
        self.recognizer = recognizer
        self.gather = gather
        self.binarize = None     # None | "adaptive" (cv2.adaptiveThreshold semantics) | "simple" (L > 128: the reference without OpenCV)
        self.eng = engine
        self.charset = charset or arch.ctc_charset(engine.num_classes or 6625)
        self._decoder = arch.TextDecoder(self.charset)   # class ids -> strings (vectorised for single-code-point dictionaries)
        self.max_dimension = max_dimension
        self.post = dict(arch.DEFAULT_POST if post is None else post)

    # ---- stages -------------------------------------------------------------------------
    def preprocess(self, pages, enhance: bool = True, deskew: bool = False):
        """uint8 [B,H,W,3] device -> processed uint8 [B,H',W',3]: the reference's order (image_preprocessing.py:559-628 /
        :191-242): resize -> [deskew] -> contrast 1.2 -> sharpness 1.1 (JPEG hand-off is the provider's)."""
        b, h, w, _ = pages.shape
        nw, nh = get_optimal_size(w, h, self.max_dimension)
        x = pages if (nw, nh) == (w, h) else self.eng.resize_lanczos(pages, nh, nw)
        if deskew:
            x, self.last_skew_angles = self.eng.deskew(x)
        if self.binarize:        # (:613-622) binarisation replaces contrast + sharpness
            return self.eng.binarize(x, adaptive=self.binarize == "adaptive")
        return self.eng.enhance(x, 1.2, 1.1) if enhance else x

    def detect(self, processed):
        b, h, w, _ = processed.shape
        prob = self.eng.det_forward(processed)
        return self.eng.det_postprocess(prob, h, w, **self.post)

    def recognize(self, processed, boxes, scores, counts) -> List[PageDetections]:
        return self.finish(self.submit_recognize(processed, boxes, scores, counts))[0]

    # ---- split submission: lets the host-side decode of batch k overlap the device work of batch k+1 ----------------
    def submit_detect(self, pages, enhance: bool = True, deskew: bool = False):
        """Enqueue resize/[deskew]/enhance + DBNet + DB post-process; no host synchronisation. -> handle for submit_recognize."""
        processed = self.preprocess(pages, enhance, deskew)
        return (processed,) + tuple(self.detect(processed))

    def submit_recognize(self, processed, boxes, scores, counts) -> "_Pending":
        """One host sync (box counts), then crop + CRNN + CTC and the device->pinned-host copies are enqueued. -> pending."""
        import torch
        b, h, w, _ = processed.shape
        counts_h = counts.cpu().numpy()  # the one host sync of the pipeline
        n = int(counts_h.sum())
        pend = _Pending(b=b, w=w, h=h, counts_h=counts_h, n=n, processed=processed)
        if self.gather is not None:
            self.gather.begin(counts_h)          # capacity all-reduce runs beside the recogniser
        if n == 0:
            if self.gather is not None:          # every rank takes part in the collective, with or without lines
                e = lambda *shape, dt=torch.int32: torch.empty(shape, dtype=dt, device=boxes.device)
                pend.gathered = self.gather.submit(counts_h, e(0, 8), e(0, dt=torch.float32), e(0, 80), e(0), e(0, dt=torch.float32))
            return pend
        # the valid (page, slot) pairs are known on the host (counts): one small index upload + three gathers, instead of boolean-mask
        # indexing (each of those runs a nonzero kernel and synchronises to learn its output size)
        cap = boxes.shape[1]
        page_h = np.repeat(np.arange(b, dtype=np.int64), counts_h)
        slot_h = np.arange(n, dtype=np.int64) - np.repeat(np.cumsum(counts_h) - counts_h, counts_h)
        flat = torch.from_numpy(page_h * cap + slot_h).to(boxes.device, non_blocking=True)
        quads = boxes.view(-1, 8).index_select(0, flat)
        det_sc = scores.view(-1).index_select(0, flat)
        page_idx = torch.from_numpy(page_h.astype(np.int32)).to(boxes.device, non_blocking=True)
        crops, widths = self.eng.rec_crop(processed, quads, page_idx)
        idx, prob = (self.eng.svtr_forward if self.recognizer == "svtr" else self.eng.rec_forward)(crops, widths)
        text, length, score = self.eng.ctc_decode(idx, prob)
        if self.gather is not None:
            pend.gathered = self.gather.submit(counts_h, quads, det_sc, text, length, score)
            return pend
        pend.host = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True).copy_(t, non_blocking=True) for t in (text, length, score, quads, det_sc)]
        pend.event = torch.cuda.Event()
        pend.event.record(torch.cuda.current_stream(processed.device))
        return pend

    def finish(self, pend: "_Pending") -> Tuple[List[PageDetections], "object"]:
        """Wait for the pending batch's copies and build the per-page results (string decode on the host)."""
        b, w, h = pend.b, pend.w, pend.h
        if pend.gathered is not None:
            return self.gather.finish(pend.gathered), pend.processed
        if pend.n == 0:
            return [PageDetections(np.zeros((0, 8), np.int32), [], np.zeros(0, np.float32), np.zeros(0, np.float32), w, h) for _ in range(b)], pend.processed
        pend.event.synchronize()
        text_h, len_h, score_h, quads_h, det_h = (t.numpy() for t in pend.host)
        all_texts = self._decoder.decode(text_h, len_h)
        out, off = [], 0
        for p in range(b):
            c = int(pend.counts_h[p])
            texts = all_texts[off:off + c]
            out.append(PageDetections(quads_h[off:off + c], texts, score_h[off:off + c], det_h[off:off + c], w, h,
                                      text_h[off:off + c], len_h[off:off + c]))
            off += c
        return out, pend.processed

    def run_many(self, batches, enhance: bool = True, deskew: bool = False):
        """Generator over batches: yields (detections, processed) per batch, in order, with batch k's host decode running
        while the device works on batch k+1's detection half."""
        pending = None
        for pages in batches:
            h = self.submit_detect(pages, enhance, deskew)
            if pending is not None:
                yield self.finish(pending)
            pending = self.submit_recognize(*h)
        if pending is not None:
            yield self.finish(pending)

    def run(self, pages, enhance: bool = True, deskew: bool = False) -> Tuple[List[PageDetections], "object"]:
        """-> (per-page detections, processed pages on device)."""
        return self.finish(self.submit_recognize(*self.submit_detect(pages, enhance, deskew)))
