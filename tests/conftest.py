import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "ocr-system_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def engine():
    from lumina_ocr.engine import Engine
    eng = Engine(0)
    yield eng
    eng.close()


@pytest.fixture(scope="session")
def det_weights():
    from lumina_ocr import arch
    return arch.make_det_weights(1234)


@pytest.fixture(scope="session")
def dense_det_weights():
    """Seeded weights WITHOUT the hand-set text path: every row of every layer is dense, so the probability map (and every tap)
    depends on every channel of every layer — what the bit-identity tests of the fused / ring / LDS-DMA kernels need."""
    from lumina_ocr import arch
    return arch.make_det_weights(1234, text_path=False)


@pytest.fixture(scope="session", params=["text_path", "dense"])
def any_det_weights(request, det_weights, dense_det_weights):
    return det_weights if request.param == "text_path" else dense_det_weights


@pytest.fixture(scope="session")
def rec_weights():
    from lumina_ocr import arch
    return arch.make_rec_weights(4321)


@pytest.fixture(scope="session")
def code_rec_weights():
    """The seeded recogniser WITH the hand-set exact code path (arch._install_code_path): trained-like arg-max margins (>= 14 logits
    on every step), so strings are compared for EQUALITY; the plain seeded set above stays for dense-operand coverage."""
    from lumina_ocr import arch
    return arch.make_rec_weights(4321, code_path=True)


def close_stats(got: np.ndarray, ref: np.ndarray, dtype: str = "bf16"):
    """Storage-type-aware comparison: share within 1 / 4 ulps OF THAT TYPE (bf16: 8 significant bits, fp16: 11), max / mean abs error."""
    got = np.asarray(got, np.float32).ravel()
    ref = np.asarray(ref, np.float32).ravel()
    err = np.abs(got - ref)
    assert dtype in ("bf16", "f16")
    ulp = np.maximum(np.abs(ref), 2.0 ** -6) * (2.0 ** -7 if dtype == "bf16" else 2.0 ** -10)  # one ulp at |ref| (floored; an upper bound within the binade)
    return dict(within1=float((err <= ulp).mean()), within4=float((err <= 4 * ulp).mean()), max_abs=float(err.max()),
                mean_abs=float(err.mean()), ref_mean_abs=float(np.abs(ref).mean()))
