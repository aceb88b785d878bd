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


def test_two_ranks_share_one_gpu_and_gather_every_page():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--share-device", "--steps", "2", "--warmup", "1", "--pages", "4",
                        "--det-sub-batch", "4", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["ranks"] == 2 and line["config"]["collective_backend"] == "gloo"
    assert line["config"]["pages_per_gpu"] == 4 and line["config"]["global_batch"] == 8
    assert line["config"]["pages_gathered_last_step"] == 8            # rank 0 holds the pages of BOTH ranks after the gather
    assert line["config"]["lines_last_step"] > 8 * 20                  # ... with their recognised lines (synthetic pages: ~50 per page)
    assert line["value"] > 0 and "REHEARSAL" in line["config"]["parallelism"]
