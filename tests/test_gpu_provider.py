"""GPU: the provider's multi-page and threading behaviour on the real engine (the CPU twin with a stand-in engine is
tests/test_provider_contract.py).  PDF path: /root/reference/backend/services/ocr_service.py:604-660; callers enter through
asyncio.to_thread worker threads (:674-677) after preload on the main thread (:832-837)."""
import asyncio
import io
import threading

import numpy as np
import pytest
from PIL import Image

from lumina_ocr import synth
from lumina_ocr.utils import layout

pytestmark = pytest.mark.gpu


@pytest.fixture
def service():
    from lumina_ocr.services import ocr_service as svc
    s = svc.OCRService()
    s.cleanup()
    s._allow_synthetic = True          # LUMINA_OCR_ALLOW_SYNTHETIC=1: no trained weights ship offline
    yield s
    s.cleanup()


def test_pdf_pages_of_two_sizes_through_the_engine(service, monkeypatch, tmp_path):
    s = service
    a = synth.synth_page(700, 500, 1, n_lines=10)[0]
    b = synth.synth_page(500, 700, 2, n_lines=8)[0]
    c = synth.synth_page(700, 500, 3, n_lines=10)[0]
    pages = [Image.fromarray(x) for x in (a, b, c)]
    monkeypatch.setattr(s._pre, "pdf_to_images", lambda path, dpi=None: pages)
    pdf = tmp_path / "doc.pdf"
    pdf.write_bytes(b"%PDF-1.4 stand-in")
    r = asyncio.run(s.process_document(pdf, "pdf"))
    assert r.success, r.error
    assert r.total_pages == 3 and [p.page_number for p in r.pages] == [1, 2, 3]
    assert [(p.image_width, p.image_height) for p in r.pages] == [(500, 700), (700, 500), (500, 700)]
    for p, im in zip(r.pages, pages):
        assert p.success and p.markdown.strip()
        assert Image.open(io.BytesIO(p.processed_image_bytes)).size == im.size          # each page its own JPEG, its own pixel grid
        assert (p.page_width_inches, p.page_height_inches) == (float(im.size[0]), float(im.size[1]))
        assert all(bx["page_number"] == p.page_number for bx in p.layout_boxes)
    assert r.combined_markdown == "\n\n---\n\n".join("## Page %d\n\n%s" % (p.page_number, p.markdown) for p in r.pages)
    assert r.combined_layout_boxes == [bx for p in r.pages for bx in p.layout_boxes]
    assert layout.validate_layout_boxes(r.combined_layout_boxes) == []
    # batching pages 1 and 3 together must not change them: page 1 alone gives the same boxes and the same JPEG
    solo = s.process_image_sync(pages[0], 1)
    assert solo.layout_boxes == r.pages[0].layout_boxes and solo.processed_image_bytes == r.pages[0].processed_image_bytes


def test_provider_from_a_fresh_thread_after_preload(service):
    """preload_model() on the main thread, requests from worker threads (what asyncio.to_thread does): the worker must be bound
    to the engine's device for uploads, streams and every C-ABI call."""
    s = service
    s.preload_model()
    page = Image.fromarray(synth.synth_page(480, 640, 5, n_lines=8)[0])
    main = s.process_image_sync(page)
    box = {}
    t = threading.Thread(target=lambda: box.setdefault("r", s.process_image_sync(page)))
    t.start(); t.join(120)
    assert "r" in box and box["r"].success, getattr(box.get("r"), "error", "no result")
    assert box["r"].layout_boxes == main.layout_boxes and box["r"].processed_image_bytes == main.processed_image_bytes
    assert s.get_status()["weights"] == "seeded-synthetic" and s.get_status()["apply_deskew"] is True


def test_deskew_flag_is_honoured(service):
    """OCR_APPLY_DESKEW (config.py:85): a page skewed by 3 degrees comes back upright in processed_image_bytes' grid when on."""
    s = service
    page = Image.fromarray(synth.synth_page(600, 800, 9, n_lines=12)[0]).rotate(3.0, resample=Image.BICUBIC, fillcolor=(255, 255, 255))
    s.apply_deskew = True
    on = s.process_image_sync(page)
    s.apply_deskew = False
    off = s.process_image_sync(page)
    s.apply_deskew = True
    assert on.success and off.success

    def slope(o):   # mean |dy/dx| of the line boxes' top edges
        v = [abs((b["polygon"][3] - b["polygon"][1]) / max(b["polygon"][2] - b["polygon"][0], 1.0)) for b in o.layout_boxes if b["type"] == "line"]
        return float(np.mean(v)) if v else 1.0
    assert slope(on) < 0.5 * slope(off) or slope(on) < 0.01, (slope(on), slope(off))


@pytest.mark.parametrize("size", [(1, 1), (5, 7), (7, 300), (1000, 3), (31, 33), (257, 4099), (2001, 5)])
@pytest.mark.parametrize("mode", ["RGB", "L"])
def test_degenerate_page_sizes_are_processed(service, size, mode):
    """One-pixel pages, strips thinner than any kernel tile, a strip longer than the 2000-px cap, grayscale input (the reference
    passes 'L' images through load_image, image_preprocessing.py:57-75): the provider answers with a page, never with an error —
    with de-skew on, as in the reference's default configuration."""
    w, h = size
    rng = np.random.default_rng(w * 7 + h)
    a = rng.integers(0, 256, (h, w, 3) if mode == "RGB" else (h, w), dtype=np.uint8)
    service.apply_deskew = True
    out = service.process_image_sync(Image.fromarray(a, mode))
    assert out.success, out.error
    assert (out.image_width, out.image_height) == (w, h)
    assert out.page_width_inches >= 1 and out.page_height_inches >= 1 and max(out.page_width_inches, out.page_height_inches) <= 2000
    assert out.processed_image_bytes[:2] == b"\xff\xd8" and out.processed_image_bytes[-2:] == b"\xff\xd9"
    im = Image.open(io.BytesIO(out.processed_image_bytes))
    assert im.size == (int(out.page_width_inches), int(out.page_height_inches))


def test_jpeg_inputs_are_decoded_on_the_device_with_the_same_result(service, tmp_path):
    """A .jpg input (path, bytes, through process_document): the provider decodes baseline JPEGs on the device (lumina_ocr_jpeg_decode,
    byte-identical to the Image.open of the reference, image_preprocessing.py:57-75) — same boxes, same markdown, same processed JPEG as the
    host-decoded path; a progressive file, an EXIF-rotated file and a PNG still take the reference's own path."""
    s = service
    page = synth.synth_page(700, 1000, 9, n_lines=12)[0]
    f = tmp_path / "scan.jpg"
    Image.fromarray(page).save(f, format="JPEG", quality=92)
    s._ensure_engine()
    calls = []
    real = s._engine.jpeg_decode
    s._engine.jpeg_decode = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    s.device_jpeg = False
    host = s.process_image_sync(f)
    assert host.success and not calls
    s.device_jpeg = True
    dev_path = s.process_image_sync(f)
    dev_bytes = s.process_image_sync(f.read_bytes())
    doc = asyncio.run(s.process_document(f, "jpg"))
    assert len(calls) == 3
    for r in (dev_path, dev_bytes, doc.pages[0]):
        assert r.success and r.layout_boxes == host.layout_boxes and r.markdown == host.markdown
        assert r.processed_image_bytes == host.processed_image_bytes and (r.image_width, r.image_height) == (1000, 700)
    # an EXIF-rotated file: decoded AND re-oriented on the device (lumina_ocr_exif_transpose = ImageOps.exif_transpose): same result as the host path
    rot = tmp_path / "rot.jpg"
    ex = Image.Exif()
    ex[0x0112] = 6
    Image.fromarray(page).save(rot, format="JPEG", quality=92, exif=ex)
    n0 = len(calls)
    r_rot = s.process_image_sync(rot)
    s.device_jpeg = False
    r_rot_host = s.process_image_sync(rot)
    s.device_jpeg = True
    assert len(calls) == n0 + 1 and r_rot.success and r_rot_host.success
    assert r_rot.layout_boxes == r_rot_host.layout_boxes and r_rot.processed_image_bytes == r_rot_host.processed_image_bytes
    assert (r_rot.image_width, r_rot.image_height) == (1000, 700)          # original size as stored (ocr_service.py:406), boxes on the rotated page
    assert r_rot.page_width_inches == 700.0 and r_rot.page_height_inches == 1000.0
    # files the device decoder does not take: same results through Pillow, no device decode call
    prog = tmp_path / "prog.jpg"
    Image.fromarray(page).save(prog, format="JPEG", quality=92, progressive=True)
    png = tmp_path / "page.png"
    Image.fromarray(page).save(png)
    n0 = len(calls)
    r_prog, r_png = s.process_image_sync(prog), s.process_image_sync(png)
    assert len(calls) == n0 and r_prog.success and r_png.success
