"""CPU: `python bench.py --gpus N` must launch N ranks ITSELF (the driver may call it without torchrun), rank 0 prints one
JSON line with n_gpus == N and the rank count the collective backend saw; a failing rank makes the launcher exit non-zero.
Driven with --backend gloo --dry-engine (fake recogniser outputs; the launcher, the env hand-off, the process group, the
capacity all-reduce and the device-tensor all-gather are the real code).  Shard unit: pages
(/root/reference/backend/services/ocr_service.py:620-637)."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _run(extra, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), "--backend", "gloo", "--dry-engine", "--steps", "3", "--warmup", "1",
                           "--pages", "5"] + extra, capture_output=True, text=True, timeout=300, env=e)


def test_bench_gpus_2_launches_two_ranks_itself():
    r = _run(["--gpus", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # exactly one JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["ranks"] == 2 and out["config"]["global_batch"] == 10
    assert out["config"]["pages_gathered_last_step"] == 10  # every rank's pages arrived, in one buffer
    assert out["scaling"] == "weak" and out["steps"] == 3 and out["warmup"] == 1
    assert out["value"] is None and "DRY" in out["data"]    # a rehearsal never produces a number


def test_bench_gpus_1_stays_single_process():
    r = _run(["--gpus", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["config"]["ranks"] == 1 and out["config"]["collective_backend"] is None


def test_bench_under_an_external_launcher_uses_its_ranks():
    """torchrun-style: RANK/WORLD_SIZE in the environment -> no second launcher, the process is a rank."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = _run(["--gpus", "1"], env=dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["config"]["collective_backend"] == "gloo"


def test_launcher_propagates_a_failing_rank():
    r = _run(["--gpus", "2", "--pages", "-3"])             # every rank raises (negative shard size)
    assert r.returncode != 0


def test_launcher_ends_the_survivors_when_one_rank_dies():
    """Only rank 1 raises; rank 0 is then blocked in its first collective (gloo would wait forever): the launcher must notice,
    stop rank 0 and return non-zero well inside the timeout."""
    import time
    t0 = time.time()
    r = _run(["--gpus", "2", "--fail-rank", "1"])
    assert r.returncode != 0
    assert "rank 1 exited" in r.stderr and time.time() - t0 < 120


def test_launcher_preflight_refuses_a_node_with_too_few_gpus():
    """No GPU in the CPU container: `--gpus 2` with the real engine must stop with ONE message before any rank starts."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this node has the GPUs")
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, env=e)
    assert r.returncode == 2 and "visible GPU" in r.stderr and "Traceback" not in r.stderr
