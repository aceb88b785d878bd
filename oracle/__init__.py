"""CPU oracle for the Lumina OCR hot path — TEST INFRASTRUCTURE, not product code.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product (ocr-system_amd/) never does, and fails loudly when its HIP library is
missing.

Parity status:
  * pre/post-processing restatements (resize, enhance, reading order, box matcher, schema)
    are PINNED by golden vectors captured from the importable reference modules
    (tests/golden/, generator tools/make_golden.py).
  * the JPEG hand-off (jpeg.py / csrc/jpeg_oracle.c) restates the encoder behind the reference's
    image.save(..., 'JPEG', quality=q, optimize=True) — Pillow 12.2.0 / libjpeg-turbo, a third-party
    dependency absent from /root/reference — and is PINNED by the digests of the reference's own
    compress_for_azure output (tests/golden/jpeg_digests.json) and by live comparison with Pillow.
  * det+rec numerics (DBNet, DB post-process, crop, CRNN, CTC) have NO reference
    implementation, weights, dictionaries or test images offline: "parity unpinned".
    The restatement below is the definition the HIP path is checked against.
"""
import sys as _sys
from pathlib import Path as _Path

_pkg = str(_Path(__file__).resolve().parent.parent / "ocr-system_amd")
if _pkg not in _sys.path:
    _sys.path.insert(0, _pkg)
