"""CPU: host logic of the provider above the engine slot, with the engine replaced by a stand-in that returns fixed detections
(the arithmetic is covered by the gpu tests; tests/test_gpu_provider.py runs the same scenarios on the real engine).
  * PDF path: /root/reference/backend/services/ocr_service.py:604-660 (+ :737-757 joins) — page numbering, "## Page n" joins,
    combined_layout_boxes in page order, one JPEG per page, also when pages of different sizes are batched separately;
  * caller contract: the fields run_ocr_task copies (/root/reference/backend/services/extraction_service.py:226-252);
  * the debug harness's JSON keys (/root/reference/debug_azure_output.py:93-111; fixture tests/golden/azure_debug_output.json)."""
import asyncio
import contextlib
import json
import sys
from pathlib import Path

import numpy as np
import pytest
from PIL import Image

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))


class _FakePipeline:
    """Returns, per page, two lines whose text names the page size — enough to see which page went where."""
    charset = None

    def run(self, pages, enhance=True, deskew=False):
        from lumina_ocr.pipeline import PageDetections
        b, h, w, _ = pages.shape
        self.seen_deskew = deskew
        dets = []
        for i in range(b):
            tag = int(pages[i, 0, 0, 0])                      # the test paints the page index into pixel (0, 0)
            quads = np.array([[10, 10, 110, 10, 110, 30, 10, 30], [10, 50, 90, 50, 90, 70, 10, 70]], np.int32)
            dets.append(PageDetections(quads, ["page%d %dx%d" % (tag, w, h), "second line"], np.array([0.9, 0.8], np.float32),
                                       np.array([0.7, 0.6], np.float32), w, h))
        return dets, pages


@pytest.fixture
def service(monkeypatch):
    import torch
    from lumina_ocr.services import ocr_service as svc
    s = svc.OCRService()
    s.cleanup()
    fake = _FakePipeline()
    monkeypatch.setattr(s, "_ensure_engine", lambda: None)
    monkeypatch.setattr(s, "_pipeline", fake)
    monkeypatch.setattr(s, "_upload", lambda staged: staged.clone())
    monkeypatch.setattr(s, "_device_ctx", contextlib.nullcontext)
    monkeypatch.setattr(s._pre, "compress_for_azure_device",
                        lambda processed, **kw: [b"\xff\xd8" + bytes([int(processed[i, 0, 0, 0])]) for i in range(processed.shape[0])])
    yield s, fake
    monkeypatch.undo()
    s.cleanup()


def _page(idx, w, h):
    a = np.full((h, w, 3), 255, np.uint8)
    a[0, 0] = idx
    return Image.fromarray(a)


def test_pdf_path_numbers_joins_and_batches_pages(service, monkeypatch, tmp_path):
    s, _ = service
    pages = [_page(1, 300, 400), _page(2, 500, 350), _page(3, 300, 400)]           # two sizes: pages 1 and 3 form one engine batch
    monkeypatch.setattr(s._pre, "pdf_to_images", lambda path, dpi=None: pages)
    pdf = tmp_path / "doc.pdf"
    pdf.write_bytes(b"%PDF-1.4 stand-in")
    r = asyncio.run(s.process_document(pdf, "pdf"))
    assert r.success and r.error is None and r.total_pages == 3
    assert [p.page_number for p in r.pages] == [1, 2, 3]
    assert [p.markdown.split("\n")[0] for p in r.pages] == ["page1 300x400", "page2 500x350", "page3 300x400"]
    assert [(p.image_width, p.image_height) for p in r.pages] == [(300, 400), (500, 350), (300, 400)]
    assert [p.processed_image_bytes for p in r.pages] == [b"\xff\xd8\x01", b"\xff\xd8\x02", b"\xff\xd8\x03"]   # each page its own JPEG
    assert r.combined_markdown == "\n\n---\n\n".join("## Page %d\n\n%s" % (p.page_number, p.markdown) for p in r.pages)  # :737-746
    assert r.combined_html == "\n<hr>\n".join('<section data-page="%d">\n%s\n</section>' % (p.page_number, p.html) for p in r.pages)
    assert [b["page_number"] for b in r.combined_layout_boxes] == sorted(b["page_number"] for b in r.combined_layout_boxes)
    assert r.combined_layout_boxes == [b for p in r.pages for b in p.layout_boxes]                                     # :635-637
    # errors are data: a missing file, an empty PDF, a rasteriser failure
    assert not asyncio.run(s.process_document(tmp_path / "nope.pdf", "pdf")).success
    monkeypatch.setattr(s._pre, "pdf_to_images", lambda path, dpi=None: [])
    e = asyncio.run(s.process_document(pdf, "pdf"))
    assert not e.success and e.error == "No pages found in PDF"
    def boom(path, dpi=None):
        raise ImportError("pdf2image not installed")
    monkeypatch.setattr(s._pre, "pdf_to_images", boom)
    e = asyncio.run(s.process_document(pdf, "pdf"))
    assert not e.success and "pdf2image" in e.error


def test_run_ocr_task_contract_fields(service, tmp_path):
    """Rebuild OCRTaskOutput exactly as the caller does (extraction_service.py:226-252) from the provider's result."""
    from lumina_ocr.utils import layout
    s, fake = service
    p = tmp_path / "page.png"
    _page(7, 640, 480).save(p)
    result = asyncio.run(s.process_document(str(p), "png"))
    assert result.success
    processed_images, page_dimensions = [], []
    for page in result.pages:
        if page.processed_image_bytes:
            processed_images.append(page.processed_image_bytes)
        page_dimensions.append({"page_number": page.page_number, "width_inches": page.page_width_inches,
                                "height_inches": page.page_height_inches, "image_width_px": page.image_width,
                                "image_height_px": page.image_height})
    out = dict(markdown=result.combined_markdown, html=result.combined_html, total_pages=result.total_pages, success=True,
               layout_boxes=result.combined_layout_boxes, processed_images=processed_images, page_dimensions=page_dimensions)
    assert out["markdown"].strip()                                         # extraction fails on blank markdown (:290-295)
    assert out["html"].startswith("<div class='ocr-content'>")
    assert out["total_pages"] == 1 and len(out["processed_images"]) == 1 and isinstance(out["processed_images"][0], bytes)
    d = out["page_dimensions"][0]
    assert d == {"page_number": 1, "width_inches": 640.0, "height_inches": 480.0, "image_width_px": 640, "image_height_px": 480}
    assert isinstance(d["width_inches"], float) and isinstance(d["image_width_px"], int)
    assert layout.validate_layout_boxes(out["layout_boxes"]) == []
    assert {b["type"] for b in out["layout_boxes"]} == {"line", "word", "paragraph"}   # lines for the matcher, words for its union fallback, paragraphs as in ocr_service.py:355-367
    assert fake.seen_deskew is True                                        # settings.OCR_APPLY_DESKEW default (config.py:85)
    json.dumps(result.to_dict())                                           # stored as JSONB: must serialise, bytes excluded
    assert "processed_image_bytes" not in result.to_dict()["pages"][0]


def test_dump_harness_keys_match_the_reference_fixture(service, tmp_path):
    import dump_ocr
    s, _ = service
    p = tmp_path / "page.png"
    _page(1, 640, 480).save(p)
    result = asyncio.run(s.process_document(str(p), "png"))
    got = dump_ocr.dump_dict(result)
    ref = json.loads((ROOT / "tests" / "golden" / "azure_debug_output.json").read_text())
    assert list(got) == list(ref)
    assert list(got["pages"][0]) == list(ref["pages"][0])
    assert set(got["combined_layout_boxes_sample"][0]) <= {"type", "content", "confidence", "polygon", "page_number"}
    assert got["combined_layout_boxes_count"] == len(result.combined_layout_boxes) and got["pages"][0]["has_processed_image"] is True


def test_unconfigured_provider_is_an_error_not_synthetic_text(monkeypatch, tmp_path):
    """No weights and no LUMINA_OCR_ALLOW_SYNTHETIC: success=False with the reason (errors are data, ocr_service.py:464-475) —
    checked before any GPU is touched, so it also holds in this CPU container."""
    from lumina_ocr.services import ocr_service as svc
    s = svc.OCRService()
    s.cleanup()
    monkeypatch.setattr(s, "_allow_synthetic", False)
    monkeypatch.setattr(s, "_det_weights", "")
    p = tmp_path / "page.png"
    _page(1, 64, 48).save(p)
    r = asyncio.run(s.process_document(str(p), "png"))
    assert not r.success and "weights not configured" in r.error and r.combined_markdown == ""
    monkeypatch.setattr(s, "_det_weights", str(p)); monkeypatch.setattr(s, "_rec_weights", str(p)); monkeypatch.setattr(s, "_rec_dict", "")
    r = asyncio.run(s.process_document(str(p), "png"))
    assert not r.success and "LUMINA_OCR_REC_DICT" in r.error


def test_dictionary_file_round_trip(tmp_path):
    from lumina_ocr import arch
    cs = arch.devanagari_charset()
    f = tmp_path / "dict.txt"
    arch.save_charset(f, cs)
    assert arch.load_charset(f) == cs
    dec = arch.TextDecoder(cs)
    ids = np.array([[1, 2, len(cs) - 1, 3, -1, -1]], np.int32)
    assert dec.decode(ids, np.array([4])) == [cs[1] + cs[2] + " " + cs[3]]
    multi = arch.TextDecoder(["\x00", "ab", "c"])                          # a multi-code-point symbol: per-line join
    assert multi.decode(np.array([[1, 2, 1]], np.int32), np.array([3])) == ["abcab"]
    (tmp_path / "dup.txt").write_text("a\na\n")
    with pytest.raises(ValueError):
        arch.load_charset(tmp_path / "dup.txt")
