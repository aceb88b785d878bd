#!/usr/bin/env python3
"""Drive the provider end to end on one file and write the JSON the reference's debug harness writes
(/root/reference/debug_azure_output.py:93-111 -> azure_debug_output.json): success, total_pages, combined_layout_boxes_count,
combined_layout_boxes_sample[:10], pages[{page_number, layout_boxes_count, has_processed_image, page_width_inches,
page_height_inches}].  Usage: python tools/dump_ocr.py path/to/page.png [out.json]   (needs the GPU; without trained weights
set LUMINA_OCR_ALLOW_SYNTHETIC=1)."""
import asyncio
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "ocr-system_amd"))


# the harness's JSON contract, as a schema: key -> how it is read off the provider's result objects (order = the order of the keys in
# /root/reference/azure_debug_output.json, the fixture tests/golden/azure_debug_output.json holds)
SAMPLE_BOXES = 10
DOC_KEYS = (("success", lambda r: r.success), ("total_pages", lambda r: r.total_pages),
            ("combined_layout_boxes_count", lambda r: len(r.combined_layout_boxes)),
            ("combined_layout_boxes_sample", lambda r: list(r.combined_layout_boxes[:SAMPLE_BOXES])))
PAGE_KEYS = (("page_number", lambda pg: pg.page_number), ("layout_boxes_count", lambda pg: len(pg.layout_boxes)),
             ("has_processed_image", lambda pg: pg.processed_image_bytes is not None),
             ("page_width_inches", lambda pg: pg.page_width_inches), ("page_height_inches", lambda pg: pg.page_height_inches))


def dump_dict(result) -> dict:
    """DocumentOCRResult -> the harness's output (same keys, same order)."""
    out = {key: read(result) for key, read in DOC_KEYS}
    out["pages"] = [{key: read(pg) for key, read in PAGE_KEYS} for pg in result.pages]
    return out


async def main(path: str, out: str) -> int:
    from lumina_ocr.services.ocr_service import OCRService
    result = await OCRService().process_document(path, Path(path).suffix.lstrip(".") or "png")
    if not result.success:
        print("OCR failed: %s" % result.error, file=sys.stderr)
        return 1
    Path(out).write_text(json.dumps(dump_dict(result), indent=2, ensure_ascii=False))
    print("pages %d, layout boxes %d, %d ms -> %s" % (result.total_pages, len(result.combined_layout_boxes), result.total_processing_time_ms, out))
    return 0


if __name__ == "__main__":
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    sys.exit(asyncio.run(main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "lumina_debug_output.json")))
