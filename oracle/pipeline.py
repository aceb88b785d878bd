"""Oracle (test infrastructure): the whole hot path on the CPU, stage by stage, mirroring the order of
/root/reference/backend/services/ocr_service.py:398-475 (preprocess -> engine -> boxes -> OCROutput fields).
det+rec numerics: "parity unpinned" (see oracle/__init__.py); pre/post-processing: pinned by tests/golden/."""
from __future__ import annotations

from typing import List

import numpy as np

from lumina_ocr import arch

from . import dbpost, nets, preprocess, reading_order


def run_pages(det_w, rec_w, pages_u8: np.ndarray, charset: List[str], enhance: bool = True, max_dim: int = 2000, mode: str = "bf16",
              post: dict = None):
    """pages [B,H,W,3] u8 -> (list per page of dict(quads int32 [n,8], texts, scores, det_scores), processed [B,H',W',3])."""
    processed = []
    for pg in pages_u8:
        x = preprocess.resize_if_needed(pg, max_dim)
        if enhance:
            x = preprocess.enhance_sharpness(preprocess.enhance_contrast(x, 1.2), 1.1)
        processed.append(x)
    processed = np.stack(processed)
    b, h, w, _ = processed.shape
    prob = nets.det_forward(det_w, processed, mode=mode)
    bits = arch.f32_to_bf16_bits(prob)
    out = []
    for i in range(b):
        quads, dsc, _ = dbpost.db_postprocess(bits[i], h, w, **(post or {}))
        if len(quads) == 0:
            out.append(dict(quads=quads, texts=[], scores=np.zeros(0, np.float32), det_scores=dsc, idx=np.zeros((0, 80), np.int64),
                            margin=np.zeros((0, 80), np.float32)))
            continue
        crops, widths = zip(*[dbpost.rec_crop(processed[i], q) for q in quads])
        crops = np.stack(crops)
        x = nets.rec_normalize(crops, mode)
        for j, wv in enumerate(widths):
            x[j, :, :, wv:] = 0
        import torch
        with torch.no_grad():
            feat = nets.rec_backbone(rec_w, x, mode)
            idx, prob_t, logits, _ = nets.rec_head(rec_w, feat, mode)
        dec = nets.ctc_greedy(idx, prob_t, charset)
        top2 = np.partition(logits, -2, axis=2)[:, :, -2:]
        out.append(dict(quads=quads, texts=[d[0] for d in dec], scores=np.array([d[1] for d in dec], np.float32), det_scores=dsc,
                        idx=idx, margin=(top2[:, :, 1] - top2[:, :, 0]).astype(np.float32)))   # per-step top-1 minus top-2 logit
    return out, processed


def layout_items(page):
    """(box 4x2, text, score) triples for oracle.reading_order from one page dict."""
    return [([[float(q[0]), float(q[1])], [float(q[2]), float(q[3])], [float(q[4]), float(q[5])], [float(q[6]), float(q[7])]], t, float(s))
            for q, t, s in zip(page["quads"], page["texts"], page["scores"])]


def quad_iou(a, b) -> float:
    """IoU of two convex quads (8 numbers each) by polygon clipping (Sutherland-Hodgman); test helper."""
    def area(p):
        return 0.5 * abs(sum(p[i][0] * p[(i + 1) % len(p)][1] - p[(i + 1) % len(p)][0] * p[i][1] for i in range(len(p))))

    def clip(subject, c1, c2):
        def inside(p):
            return (c2[0] - c1[0]) * (p[1] - c1[1]) - (c2[1] - c1[1]) * (p[0] - c1[0]) >= 0

        def inter(p, q):
            x1, y1, x2, y2 = *p, *q
            x3, y3, x4, y4 = *c1, *c2
            den = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4)
            if den == 0:
                return q
            t = ((x1 - x3) * (y3 - y4) - (y1 - y3) * (x3 - x4)) / den
            return (x1 + t * (x2 - x1), y1 + t * (y2 - y1))
        res = []
        for i in range(len(subject)):
            p, q = subject[i - 1], subject[i]
            if inside(q):
                if not inside(p):
                    res.append(inter(p, q))
                res.append(q)
            elif inside(p):
                res.append(inter(p, q))
        return res
    pa = [(float(a[2 * i]), float(a[2 * i + 1])) for i in range(4)]
    pb = [(float(b[2 * i]), float(b[2 * i + 1])) for i in range(4)]

    def orient(p):
        s = sum(p[i][0] * p[(i + 1) % 4][1] - p[(i + 1) % 4][0] * p[i][1] for i in range(4))
        return p if s >= 0 else p[::-1]
    pa, pb = orient(pa), orient(pb)
    poly = pa
    for i in range(4):
        if not poly:
            break
        poly = clip(poly, pb[i], pb[(i + 1) % 4])
    ia = area(poly) if len(poly) >= 3 else 0.0
    ua = area(pa) + area(pb) - ia
    return ia / ua if ua > 0 else (1.0 if pa == pb else 0.0)
