"""Engine output -> the reference's result schema (layout_boxes, markdown, html).

Schema (stored verbatim in extractions.layout_data and consumed by BoundingBoxMatcher and the UI):
  {"type": "word"|"line", "content": str, ["confidence": float,] "polygon": [x1,y1,..,x4,y4], "page_number": int}
  /root/reference/backend/services/ocr_service.py:293-311; fixture /root/reference/azure_debug_output.json:5-166.
  Units: pixels of the processed image (fixture page 2000.0 x 1090.0, :172-173), origin top-left, TL,TR,BR,BL.
A det+rec engine yields line quads; `line` entries are mandatory for matching
(/root/reference/backend/utils/bbox_matcher.py:47, :103-144) and `word` entries feed the union fallback
(:48, :155-208), so words are synthesised by splitting the line text on spaces and interpolating along the quad.
"""
from __future__ import annotations

from typing import Any, Dict, List, Sequence, Tuple

from .ocr_postprocessor import MergedLine, TextBlock, group_into_lines, sort_and_merge_lines


def _lerp(a: Sequence[float], b: Sequence[float], t: float) -> Tuple[float, float]:
    return a[0] + (b[0] - a[0]) * t, a[1] + (b[1] - a[1]) * t


def split_words(quad: Sequence[float], text: str) -> List[Tuple[str, List[float]]]:
    """Proportional split of a line quad into word quads (character counts as widths, spaces included)."""
    tl, tr, br, bl = (quad[0], quad[1]), (quad[2], quad[3]), (quad[4], quad[5]), (quad[6], quad[7])
    n = len(text)
    out: List[Tuple[str, List[float]]] = []
    if n == 0:
        return out
    pos = 0
    for word in text.split(" "):
        if word:
            t0, t1 = pos / n, (pos + len(word)) / n
            a, b = _lerp(tl, tr, t0), _lerp(tl, tr, t1)
            c, d = _lerp(bl, br, t1), _lerp(bl, br, t0)
            out.append((word, [float(round(v)) for v in (*a, *b, *c, *d)]))
        pos += len(word) + 1
    return out


def build_layout_boxes(lines: Sequence[Tuple[Sequence[int], str, float]], page_number: int = 1) -> List[Dict[str, Any]]:
    """lines: (quad 8 ints, text, score) in reading order -> words first, then lines (the order of ocr_service.py:285-311)."""
    words: List[Dict[str, Any]] = []
    line_boxes: List[Dict[str, Any]] = []
    for quad, text, score in lines:
        q = [float(v) for v in quad]
        for w, wq in split_words(q, text):
            words.append({"type": "word", "content": w, "confidence": float(score), "polygon": wq, "page_number": page_number})
        line_boxes.append({"type": "line", "content": text, "polygon": q, "page_number": page_number})
    return words + line_boxes


def build_paragraph_boxes(merged: Sequence[MergedLine], page_number: int = 1, gap_ratio: float = 0.7) -> List[Dict[str, Any]]:
    """`paragraph` entries of the reference schema (/root/reference/backend/services/ocr_service.py:355-367: content cut to 100
    characters + "...", role, polygon, page_number).  Azure's layout model supplies them there; a det+rec engine derives them from
    the reading-order lines: consecutive lines belong to one paragraph while the vertical gap between them (top of the next minus
    bottom of the previous) stays within gap_ratio x their mean height.  Polygon = the axis-aligned hull of the paragraph's line
    quads (TL, TR, BR, BL); role = "title" for a first paragraph whose lines are at least 1.3 x the page's median line height, else
    "text" (the reference's fallback for a paragraph without a role, :362)."""
    rows = []
    for m in merged:
        pts = [pt for b in m.blocks for pt in b.box]
        if not pts or not m.text:
            continue
        xs, ys = [p[0] for p in pts], [p[1] for p in pts]
        rows.append((m.text, min(xs), min(ys), max(xs), max(ys)))
    if not rows:
        return []
    heights = sorted(r[4] - r[2] for r in rows)
    median_h = heights[len(heights) // 2]
    groups: List[List[Tuple[str, float, float, float, float]]] = [[rows[0]]]
    for prev, cur in zip(rows, rows[1:]):
        mean_h = ((prev[4] - prev[2]) + (cur[4] - cur[2])) / 2.0
        if cur[2] - prev[4] <= gap_ratio * mean_h:
            groups[-1].append(cur)
        else:
            groups.append([cur])
    out: List[Dict[str, Any]] = []
    for gi, g in enumerate(groups):
        text = " ".join(r[0] for r in g)
        x0, y0, x1, y1 = min(r[1] for r in g), min(r[2] for r in g), max(r[3] for r in g), max(r[4] for r in g)
        g_h = sum(r[4] - r[2] for r in g) / len(g)
        out.append({"type": "paragraph", "content": text[:100] + "..." if len(text) > 100 else text,
                    "role": "title" if gi == 0 and len(groups) > 1 and g_h >= 1.3 * median_h else "text",
                    "polygon": [float(x0), float(y0), float(x1), float(y0), float(x1), float(y1), float(x0), float(y1)], "page_number": page_number})
    return out


def reading_order(dets: Sequence[Tuple[Sequence[int], str, float]]) -> Tuple[List[MergedLine], List[Tuple[Sequence[int], str, float]]]:
    """Order detections with the reference's reading-order rules; returns (merged lines, detections in reading order)."""
    blocks = [TextBlock(text=t, confidence=float(s), box=[[float(q[0]), float(q[1])], [float(q[2]), float(q[3])],
                                                           [float(q[4]), float(q[5])], [float(q[6]), float(q[7])]]) for q, t, s in dets]
    index = {id(b): d for b, d in zip(blocks, dets)}
    merged = sort_and_merge_lines(group_into_lines(blocks)) if blocks else []
    ordered = [index[id(b)] for m in merged for b in m.blocks]
    return merged, ordered


def page_markdown(merged: Sequence[MergedLine]) -> str:
    """combined_markdown is fed verbatim to the LLM step and must be non-blank for a non-empty page
    (/root/reference/backend/services/extraction_service.py:290-295, :658-662): one reading-order line per row."""
    return "\n".join(m.text for m in merged if m.text)


def html_from_markdown(markdown_text: str) -> str:
    """ocr_service.py:378-392."""
    return f"<div class='ocr-content'>\n{markdown_text}\n</div>"


def combine_markdown(pages: Sequence[Any]) -> str:
    """ocr_service.py:737-746."""
    multi = len(pages) > 1
    parts = [(f"## Page {p.page_number}\n\n{p.markdown}" if multi else p.markdown) for p in pages if p.markdown]
    return "\n\n---\n\n".join(parts)


def combine_html(pages: Sequence[Any]) -> str:
    """ocr_service.py:748-757."""
    multi = len(pages) > 1
    parts = [(f'<section data-page="{p.page_number}">\n{p.html}\n</section>' if multi else p.html) for p in pages if p.html]
    return "\n<hr>\n".join(parts)


def validate_layout_boxes(boxes: Sequence[Dict[str, Any]]) -> List[str]:
    """Schema check against the reference fixture's shape; returns a list of problems (empty == valid)."""
    problems = []
    for i, b in enumerate(boxes):
        if b.get("type") not in ("word", "line", "selection_mark", "table", "table_cell", "paragraph"):
            problems.append(f"{i}: bad type {b.get('type')!r}")
        poly = b.get("polygon")
        if not isinstance(poly, list) or len(poly) != 8 or not all(isinstance(v, float) for v in poly):
            problems.append(f"{i}: polygon must be 8 floats")
        if not isinstance(b.get("page_number"), int) or b["page_number"] < 1:
            problems.append(f"{i}: page_number must be int >= 1")
        if b.get("type") in ("word", "line") and not isinstance(b.get("content"), str):
            problems.append(f"{i}: content must be str")
        if b.get("type") == "word" and not isinstance(b.get("confidence"), float):
            problems.append(f"{i}: word confidence must be float")
        if b.get("type") == "paragraph" and not (isinstance(b.get("content"), str) and len(b["content"]) <= 103 and isinstance(b.get("role"), str)):
            problems.append(f"{i}: paragraph needs content (<= 100 characters + '...') and role")
    return problems
