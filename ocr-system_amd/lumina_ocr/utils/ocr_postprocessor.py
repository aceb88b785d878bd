"""Reading order for det+rec output: blocks -> lines -> merged text.

Host-side mirror of /root/reference/backend/utils/ocr_postprocessor.py (same public names and
results; pinned by tests/golden/reading_order.json, which that module produced):
  TextBlock :19-39, MergedLine :42-48, parse_rapidocr_output :51-98, group_into_lines :101-143,
  sort_and_merge_lines :146-182, process_ocr_result :185-213, format_merged_output :216-226,
  extract_text_ordered :233-243.
Rules: blocks are visited by ascending vertical centre ((y of point 0 + y of point 2) / 2); a block joins
the open line while |centre - running mean of the line's centres| <= ratio * mean block height (height =
|y2 - y0|); inside a line blocks go left to right by their smallest x and are joined with one space; the
line confidence / y are plain means; lines are finally ordered by y.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Iterable, List, Sequence


@dataclass
class TextBlock:
    text: str
    confidence: float
    box: List[List[float]]  # four [x, y] points, TL TR BR BL

    @property
    def y_center(self) -> float:
        return (self.box[0][1] + self.box[2][1]) / 2

    @property
    def x_left(self) -> float:
        return min(pt[0] for pt in self.box)

    @property
    def height(self) -> float:
        return abs(self.box[2][1] - self.box[0][1])


@dataclass
class MergedLine:
    text: str
    confidence: float
    y_position: float
    blocks: List[TextBlock] = field(default_factory=list)


def parse_rapidocr_output(result: Any) -> List[TextBlock]:
    """Accepts [box, text, score] triples or objects with .box/.text/.score (or a holder with .ocr_result)."""
    if result is None:
        return []
    rows = getattr(result, "ocr_result", result)
    blocks: List[TextBlock] = []
    for row in rows or []:
        try:
            if all(hasattr(row, a) for a in ("box", "text", "score")):
                b = row.box
                blocks.append(TextBlock(row.text, row.score, b if isinstance(b, list) else b.tolist()))
            elif isinstance(row, (list, tuple)) and len(row) >= 3:
                b = row[0]
                blocks.append(TextBlock(str(row[1]), float(row[2]), b.tolist() if hasattr(b, "tolist") else b))
        except Exception:  # a malformed row is skipped, never fatal (reference :94-96)
            continue
    return blocks


def group_into_lines(blocks: Sequence[TextBlock], y_tolerance_ratio: float = 0.5) -> List[List[TextBlock]]:
    if not blocks:
        return []
    ordered = sorted(blocks, key=lambda b: b.y_center)
    tol = (sum(b.height for b in ordered) / len(ordered)) * y_tolerance_ratio
    lines: List[List[TextBlock]] = [[ordered[0]]]
    ref_y = ordered[0].y_center
    for blk in ordered[1:]:
        if abs(blk.y_center - ref_y) <= tol:
            lines[-1].append(blk)
            ref_y = sum(b.y_center for b in lines[-1]) / len(lines[-1])
        else:
            lines.append([blk])
            ref_y = blk.y_center
    return lines


def sort_and_merge_lines(lines: Iterable[List[TextBlock]], space_threshold_ratio: float = 2.0) -> List[MergedLine]:
    merged = []
    for members in lines:
        row = sorted(members, key=lambda b: b.x_left)
        n = len(row)
        merged.append(MergedLine(text=" ".join(b.text for b in row), confidence=sum(b.confidence for b in row) / n,
                                 y_position=sum(b.y_center for b in row) / n, blocks=row))
    merged.sort(key=lambda m: m.y_position)
    return merged


def process_ocr_result(result: Any, y_tolerance_ratio: float = 0.5, merge_lines: bool = True) -> List[MergedLine]:
    blocks = parse_rapidocr_output(result)
    return sort_and_merge_lines(group_into_lines(blocks, y_tolerance_ratio)) if blocks else []


def format_merged_output(merged_lines: Sequence[MergedLine], show_confidence: bool = False) -> str:
    if show_confidence:
        return "\n".join(f"{i:02d}. [{m.confidence:.2f}] {m.text}" for i, m in enumerate(merged_lines, 1))
    return "\n".join(f"{i:02d}. {m.text}" for i, m in enumerate(merged_lines, 1))


def extract_text_ordered(result: Any, y_tolerance: float = 0.5) -> str:
    return format_merged_output(process_ocr_result(result, y_tolerance_ratio=y_tolerance))
