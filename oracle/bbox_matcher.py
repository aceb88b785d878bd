"""Oracle (test infrastructure): restatement of the downstream consumer of layout_boxes,
/root/reference/backend/utils/bbox_matcher.py (normalize :52-63, find_match :77-153, word union :155-208,
single word :210-238, union polygon :240-268, find_key_value_pair :270-290).  Used to check that the engine's
layout_boxes are consumable exactly as the reference consumes them. PINNED by tests/golden/bbox_matcher.json."""
from __future__ import annotations

import re
from difflib import SequenceMatcher
from typing import Any, Dict, List, Optional

FUZZY = 0.85


def norm(t: str) -> str:
    return re.sub(r"\s+", " ", t.strip().lower()) if t else ""


def ratio(a: str, b: str) -> float:
    return SequenceMatcher(None, a, b).ratio() if a and b else 0.0


def _hit(box, text, conf):
    return {"polygon": box.get("polygon", []), "matched_text": text, "confidence": conf, "page": box.get("page_number", 1)}


def find_match(layout: List[Dict[str, Any]], target: str, page: Optional[int] = None):
    if not target or not target.strip():
        return None
    lines = [b for b in layout if b.get("type") == "line" and not (page and b.get("page_number") != page)]
    words = [b for b in layout if b.get("type") == "word" and not (page and b.get("page_number") != page)]
    t = norm(target)
    for ln in lines:
        if norm(ln.get("content", "")) == t:
            return _hit(ln, ln.get("content", ""), 1.0)
    best, score = None, 0.0
    for ln in lines:
        c = norm(ln.get("content", ""))
        s = ratio(t, c)
        if t in c or c in t:
            if s > score:
                best, score = ln, max(s, 0.9)
        elif s >= FUZZY and s > score:
            best, score = ln, s
    if best is not None:
        return _hit(best, best.get("content", ""), score)
    parts = target.split()
    if len(parts) == 1:
        w0 = norm(parts[0])
        for w in words:
            c = norm(w.get("content", ""))
            if c == w0:
                return _hit(w, w.get("content", ""), 1.0)
            if ratio(c, w0) >= FUZZY:
                return _hit(w, w.get("content", ""), ratio(c, w0))
        return None
    if len(parts) < 2:
        return None
    got = []
    for part in parts:
        pn = norm(part)
        if len(pn) < 2:
            continue
        for w in words:
            c = norm(w.get("content", ""))
            if c == pn or ratio(c, pn) >= 0.9:
                got.append(w)
                break
    if not got or len(got) < len(parts) * 0.5:
        return None
    xs = [v for b in got for v in b.get("polygon", [])[0::2]]
    ys = [v for b in got for v in b.get("polygon", [])[1::2]]
    poly = [min(xs), min(ys), max(xs), min(ys), max(xs), max(ys), min(xs), max(ys)] if xs and ys else []
    return {"polygon": poly, "matched_text": " ".join(b.get("content", "") for b in got), "confidence": min(len(got) / len(parts), 0.95),
            "page": got[0].get("page_number", 1)}


def find_key_value_pair(layout, key, value, page=None):
    return find_match(layout, key, page), find_match(layout, value, page)
