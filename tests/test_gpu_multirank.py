"""GPU: the N > 1 path with the REAL engine.  The GPU box has one MI355X and RCCL refuses two ranks on one device, so this is the
rehearsal `bench.py --share-device` offers: two rank processes (one process per rank, as in production), both on cuda:0, every page
through det + rec on the device, the per-rank results packed on the device (dist.PageGather) and exchanged with ONE all-gather per
step — over gloo instead of RCCL.  What it pins: launcher -> ranks -> engine per rank -> gather -> global page order.
(Shard unit: /root/reference/backend/services/ocr_service.py:620-637; CPU twin with a stand-in engine: tests/test_bench_launcher.py.)"""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _clean_env():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _bench(extra, env):
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "2", "--warmup", "1", "--pages", "4", "--det-sub-batch", "4",
                        "--no-cpu-baseline"] + extra, capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_two_ranks_share_one_gpu_and_gather_every_page():
    env = _clean_env()
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--share-device", "--steps", "2", "--warmup", "1", "--pages", "4",
                        "--det-sub-batch", "4", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["ranks"] == 2 and line["config"]["collective_backend"] == "gloo"
    assert line["config"]["pages_per_gpu"] == 4 and line["config"]["global_batch"] == 8
    assert line["config"]["pages_gathered_last_step"] == 8            # rank 0 holds the pages of BOTH ranks after the gather
    assert line["config"]["lines_last_step"] > 8 * 20                  # ... with their recognised lines (synthetic pages: ~50 per page)
    assert line["value"] > 0 and "REHEARSAL" in line["config"]["parallelism"]
    # global page order: pages 0..3 are rank 0's (seed 2024), pages 4..7 rank 1's (seed 3024) — compared, page by page, with each
    # rank's pages run alone in a single process (crc32 of the page's recognised lines)
    alone = [_bench(["--gpus", "1", "--seed-rank", str(k)], env)["config"]["page_digests_last_step"] for k in (0, 1)]
    assert len(alone[0]) == 4 and alone[0] != alone[1]
    assert line["config"]["page_digests_last_step"] == alone[0] + alone[1]


def test_one_rank_under_an_external_launcher_runs_the_rccl_gather():
    """The production collective path — RCCL process group, capacity all-reduce + all_gather_into_tensor on a side stream, pinned
    copy back — at world size 1 (all a one-GPU box can offer): bench.py as a rank under torchrun-style environment variables,
    default backend.  Same pages, same lines as the run without a process group."""
    import socket
    env = _clean_env()
    plain = _bench(["--gpus", "1"], env)
    assert plain["config"]["collective_backend"] is None
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    line = _bench(["--gpus", "1"], env)
    assert line["config"]["collective_backend"] == "nccl (RCCL)" and line["config"]["ranks"] == 1
    for k in ("pages_gathered_last_step", "lines_last_step", "page_digests_last_step"):
        assert line["config"][k] == plain["config"][k], k
    assert line["config"]["pages_gathered_last_step"] == 4 and line["value"] > 0
