"""GPU: the bench lines the judge reads.  `bench.py` (config 4) carries the secondary figures of the same run — value_with_deskew (the
reference provider's default, /root/reference/backend/config.py:85) and value_with_h2d (pages start in pinned host memory) — and
`--config 2 / 3 / 5` emit one line per remaining BASELINE configuration, each with its own roofline object."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _run(extra):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + extra,
                       capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_config4_line_carries_deskew_and_h2d_figures():
    line = _run(["--pages", "8", "--det-sub-batch", "8"])
    assert line["unit"] == "pages/sec" and line["value"] > 0
    assert 0 < line["value_with_deskew"] <= line["value"] * 1.05        # de-skew adds work
    assert 0 < line["value_with_h2d"] <= line["value"] * 1.10
    rf = line["roofline"]
    assert rf["bound"] == "mfma" and rf["kernel"].startswith("conv_") and 0 < rf["frac"] < 1
    assert rf["family"]["frac"] > 0 and "traffic" in rf and "traffic_over_algorithmic" in rf


@pytest.mark.parametrize("cfg,unit,bound,kernel", [(2, "pages/sec", "mfma", "conv_"), (3, "crops/sec", "hbm", ""), (5, "crops/sec", "hbm", "svtr_")])
def test_stage_lines(cfg, unit, bound, kernel):
    line = _run(["--config", str(cfg)])
    assert line["unit"] == unit and line["value"] > 0 and line["n_gpus"] == 1
    rf = line["roofline"]
    assert rf["bound"] == bound and rf["kernel"].startswith(kernel) and 0 < rf["frac"] < 1.2, rf
    assert rf["unit"] == ("TFLOP/s" if bound == "mfma" else "GB/s") and rf["all_timed_launches"]["launches_per_step"] > 5
    if cfg == 5:
        assert line["dtype"] == "f16" and "SVTR" in line["config"]["workload"]
