"""CPU: the tracked PMC file bench.py reads `roofline.traffic` from must belong to the kernel sources in the tree.
bench.py reports null (never a stale number) when the hashes differ; this test makes the staleness itself a failure, so that a
kernel edit without a re-collected `tools/record_set.sh` cannot reach the end of a round unnoticed."""
import hashlib
import importlib.util
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_pmc_file_matches_the_kernel_sources():
    bench = _bench()
    pmc = json.loads((ROOT / "profiles" / bench.PMC_FILE).read_text())
    assert pmc["source_sha16"], "no source hashes in the PMC file"
    for name, sha in pmc["source_sha16"].items():
        cur = hashlib.sha256((ROOT / "ocr-system_amd" / "csrc" / name).read_bytes()).hexdigest()[:16]
        assert cur == sha, "%s changed since profiles/%s was collected: run tools/record_set.sh on the GPU box and commit the file" % (name, bench.PMC_FILE)


def test_every_ring_instantiation_the_engine_can_name_resolves_in_the_pmc_file():
    """bench.py matches the dominant kernel's name (as the engine spells it) against the profiler's spelling."""
    bench = _bench()
    pmc = json.loads((ROOT / "profiles" / bench.PMC_FILE).read_text())
    names = [bench.flat_kernel_name(k) for k in pmc["kernels"]]
    for want in ("conv_ring_kernel<1,false,false,4>", "conv_ring_kernel<0,false,false,4>", "conv_ring_kernel<0,false,false,8>"):
        assert any(want in n for n in names), want
