"""CPU: host logic (reading order, layout schema, provider contract) vs the reference's golden vectors."""
import asyncio
import dataclasses
import json
from pathlib import Path

import numpy as np
import pytest

from lumina_ocr.services import ocr_service as svc
from lumina_ocr.utils import layout
from lumina_ocr.utils import ocr_postprocessor as pp
from oracle import bbox_matcher, reading_order

G = Path(__file__).parent / "golden"


def test_reading_order_matches_reference_vectors():
    for c in json.loads((G / "reading_order.json").read_text()):
        for got in (pp.process_ocr_result(c["items"]), None):
            if got is None:
                rows = reading_order.order_lines(c["items"])
                texts, conf, ys = [r["text"] for r in rows], [r["confidence"] for r in rows], [r["y_position"] for r in rows]
                fmt = reading_order.formatted(c["items"])
            else:
                texts, conf, ys = [m.text for m in got], [m.confidence for m in got], [m.y_position for m in got]
                fmt = pp.extract_text_ordered(c["items"])
                assert [len(m.blocks) for m in got] == [ln["n_blocks"] for ln in c["lines"]]
            assert texts == [ln["text"] for ln in c["lines"]]
            assert conf == [ln["confidence"] for ln in c["lines"]] and ys == [ln["y_position"] for ln in c["lines"]]
            assert fmt == c["formatted"]


def test_parse_accepts_objects_and_skips_malformed():
    class Row:
        def __init__(self, box, text, score):
            self.box, self.text, self.score = np.asarray(box, float), text, score
    box = [[0, 0], [10, 0], [10, 5], [0, 5]]
    blocks = pp.parse_rapidocr_output([Row(box, "a", 0.5), [box, "b", 0.25], ("bad",), None])
    assert [b.text for b in blocks] == ["a", "b"] and blocks[0].box == box
    assert pp.parse_rapidocr_output(None) == [] and pp.process_ocr_result([]) == []


def test_bbox_matcher_restatement_matches_reference_vectors():
    d = json.loads((G / "bbox_matcher.json").read_text())
    for c in d["cases"]:
        k, v = bbox_matcher.find_key_value_pair(d["layout"], c["key"], c["value"], c["page"])
        assert k == c["key_bbox"] and v == c["value_bbox"]


def test_layout_boxes_follow_the_reference_fixture_schema():
    fx = json.loads((G / "azure_debug_output.json").read_text())
    ref_keys = set(fx["combined_layout_boxes_sample"][0].keys())
    dets = [([385, 148, 1400, 176, 1398, 215, 381, 183], "SPRINGFIELD UNIVERSITY - UNDERGRADUATE", 0.97),
            ([100, 300, 700, 300, 700, 340, 100, 340], "Applicant Name: Jordan Whitfield", 0.88)]
    merged, ordered = layout.reading_order(dets)
    boxes = layout.build_layout_boxes(ordered, page_number=1)
    assert layout.validate_layout_boxes(boxes) == []
    assert layout.validate_layout_boxes(fx["combined_layout_boxes_sample"]) == []   # the validator accepts the reference's own sample
    words = [b for b in boxes if b["type"] == "word"]
    assert set(words[0].keys()) == ref_keys and [w["content"] for w in words[:3]] == ["SPRINGFIELD", "UNIVERSITY", "-"]
    assert words[0]["polygon"][0] == 385.0 and isinstance(words[0]["polygon"][0], float)
    # consumable by the reference's matcher: exact line hit, and the word-union fallback
    k, v = bbox_matcher.find_key_value_pair(boxes, "Applicant Name: Jordan Whitfield", "UNDERGRADUATE SPRINGFIELD")
    assert k["confidence"] == 1.0 and k["polygon"] == [100.0, 300.0, 700.0, 300.0, 700.0, 340.0, 100.0, 340.0]
    assert v is not None and v["matched_text"] == "UNDERGRADUATE SPRINGFIELD"
    assert layout.page_markdown(merged) == "SPRINGFIELD UNIVERSITY - UNDERGRADUATE\nApplicant Name: Jordan Whitfield"


def test_result_types_mirror_the_reference_fields():
    # /root/reference/backend/services/ocr_service.py:48-63 and :82-92
    assert [f.name for f in dataclasses.fields(svc.OCROutput)] == [
        "markdown", "html", "json_output", "processing_time_ms", "success", "error", "page_number", "image_width", "image_height",
        "layout_boxes", "processed_image_bytes", "page_width_inches", "page_height_inches"]
    assert [f.name for f in dataclasses.fields(svc.DocumentOCRResult)] == [
        "pages", "total_pages", "total_processing_time_ms", "success", "error", "combined_markdown", "combined_html", "combined_layout_boxes"]
    o = svc.OCROutput(processed_image_bytes=b"x")
    assert "processed_image_bytes" not in o.to_dict() and set(o.to_dict()) == {f.name for f in dataclasses.fields(o)} - {"processed_image_bytes"}
    d = svc.DocumentOCRResult(pages=[o], total_pages=1)
    assert list(d.to_dict()) == ["pages", "total_pages", "total_processing_time_ms", "success", "error", "combined_markdown", "combined_html",
                                 "combined_layout_boxes"]


def test_multi_page_join_rules():
    p1, p2 = svc.OCROutput(markdown="a", html="<a>", page_number=1), svc.OCROutput(markdown="b", html="<b>", page_number=2)
    assert layout.combine_markdown([p1, p2]) == "## Page 1\n\na\n\n---\n\n## Page 2\n\nb" and layout.combine_markdown([p1]) == "a"
    assert layout.combine_html([p1, p2]) == '<section data-page="1">\n<a>\n</section>\n<hr>\n<section data-page="2">\n<b>\n</section>'
    assert layout.html_from_markdown("x") == "<div class='ocr-content'>\nx\n</div>"


def test_provider_errors_are_data(tmp_path):
    s = svc.OCRService()
    assert s is svc.OCRService() is svc.ocr_service          # singleton
    r = asyncio.run(s.process_document(tmp_path / "missing.png", "png"))
    assert not r.success and r.error.startswith("File not found")
    f = tmp_path / "a.xyz"
    f.write_bytes(b"0")
    r = asyncio.run(s.process_document(f, ".XYZ"))
    assert not r.success and r.error == "Unsupported file type: xyz"
    st = asyncio.run(svc.get_ocr_status())
    assert "engine" in st and st["max_dimension"] == 2000
    node = asyncio.run(svc.ocr_node({"file_type": "png"}))
    assert node["ocr_success"] is False and node["ocr_error"] == "No document_path in state" and node["ocr_time_ms"] == 0


def test_provider_without_gpu_fails_loudly_as_data(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from PIL import Image
    p = tmp_path / "page.png"
    Image.new("RGB", (64, 48), (255, 255, 255)).save(p)
    s = svc.OCRService()
    s.cleanup()
    s._allow_synthetic = True     # weights are not the problem here: the missing GPU is
    try:
        r = asyncio.run(s.process_document(p, "png"))
    finally:
        s._allow_synthetic = False
    assert r.success is False and r.pages[0].success is False and "ROCm" in r.error   # no CPU fallback exists
    assert r.pages[0].image_width == 64 and r.pages[0].image_height == 48


def test_paragraph_boxes_follow_the_reference_schema():
    """`paragraph` entries (/root/reference/backend/services/ocr_service.py:355-367): lines grouped by vertical gaps, content cut to
    100 characters + "...", role "text" unless a taller first paragraph (title), polygon = hull of the lines, accepted by the validator."""
    from lumina_ocr.utils import layout
    def q(x0, y0, x1, y1):
        return [x0, y0, x1, y0, x1, y1, x0, y1]
    dets = [(q(100, 40, 900, 100), "A BIG TITLE", 0.9),                                    # 60 px tall
            (q(100, 200, 800, 230), "first line of the body " * 4, 0.9), (q(100, 236, 700, 266), "second line", 0.8),
            (q(100, 272, 760, 302), "third line", 0.8),                                    # 6 px gaps: one paragraph
            (q(100, 400, 500, 430), "a separate paragraph", 0.7)]                          # 98 px gap: a new one
    merged, ordered = layout.reading_order(dets)
    paras = layout.build_paragraph_boxes(merged, page_number=2)
    assert [p["role"] for p in paras] == ["title", "text", "text"]
    assert paras[0]["content"] == "A BIG TITLE" and paras[0]["polygon"] == [100.0, 40.0, 900.0, 40.0, 900.0, 100.0, 100.0, 100.0]
    body = paras[1]
    assert body["content"].endswith("...") and len(body["content"]) == 103 and body["polygon"] == [100.0, 200.0, 800.0, 200.0, 800.0, 302.0, 100.0, 302.0]
    assert paras[2]["content"] == "a separate paragraph" and all(p["page_number"] == 2 and p["type"] == "paragraph" for p in paras)
    assert layout.validate_layout_boxes(layout.build_layout_boxes(ordered, 2) + paras) == []
    assert layout.build_paragraph_boxes([], 1) == []
