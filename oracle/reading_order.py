"""Oracle (test infrastructure): independent restatement of the reference's reading-order rules
(/root/reference/backend/utils/ocr_postprocessor.py:26-39 geometry, :101-143 grouping, :146-182 merge).
PINNED by tests/golden/reading_order.json (produced by the reference module)."""
from __future__ import annotations

from typing import List, Sequence, Tuple


def order_lines(items: Sequence[Tuple[Sequence[Sequence[float]], str, float]], ratio: float = 0.5):
    """items: (box 4x2, text, score) -> list of dict(text, confidence, y_position, members=[item indices])."""
    if not items:
        return []
    yc = [(it[0][0][1] + it[0][2][1]) / 2 for it in items]
    ht = [abs(it[0][2][1] - it[0][0][1]) for it in items]
    xl = [min(p[0] for p in it[0]) for it in items]
    order = sorted(range(len(items)), key=lambda i: yc[i])  # stable, like sorted() in the reference
    tol = (sum(ht[i] for i in order) / len(order)) * ratio
    groups: List[List[int]] = [[order[0]]]
    cur = yc[order[0]]
    for i in order[1:]:
        if abs(yc[i] - cur) <= tol:
            groups[-1].append(i)
            cur = sum(yc[j] for j in groups[-1]) / len(groups[-1])
        else:
            groups.append([i])
            cur = yc[i]
    out = []
    for g in groups:
        g2 = sorted(g, key=lambda i: xl[i])
        out.append(dict(text=" ".join(str(items[i][1]) for i in g2), confidence=sum(float(items[i][2]) for i in g2) / len(g2),
                        y_position=sum(yc[i] for i in g2) / len(g2), members=g2))
    out.sort(key=lambda d: d["y_position"])
    return out


def formatted(items) -> str:
    return "\n".join(f"{i:02d}. {ln['text']}" for i, ln in enumerate(order_lines(items), 1))
