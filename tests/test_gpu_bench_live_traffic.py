"""GPU: the bench line's roofline.traffic is measured by the run itself (bench.py live_traffic: two rocprofv3 --pmc
child passes of the timed step, FETCH_SIZE calibrated on k_encode's known read volume in the same pass) — a fresh child
process, as the driver starts it."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_line_carries_a_live_pmc_measurement():
    import shutil
    if not (shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3")):
        pytest.skip("no rocprofv3 on this box")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BGN_BENCH_SPAWN")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extra"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    roof = line["roofline"]
    assert roof["traffic_source"].startswith("measured in this run"), roof["traffic_source"]
    alg = roof["algorithmic_bytes_per_pairing"] * line["config"]["batch_per_gpu"]
    assert 50 * alg < roof["traffic"] < 1000 * alg, roof["traffic"] / alg          # 300 x in round 6: scratch + re-reads
    passes = roof["traffic_passes"]
    assert passes["launches"] >= 1 and 1.5 < passes["fetch_calibration"]["factor"] < 2.5, passes
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06_bench_live_traffic_line.json"), "w") as f:
        json.dump(line, f, indent=1)
