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


def test_live_passes_cover_the_secondary_kernels():
    """With the extras on, the same two child passes also launch the secondary kernels whose traffic the line quotes
    (--pmc-extras): the level-1 and level-2 Add of 2^20 and Decrypt's lifts at 2^16 and 2^20 come out of THIS run's
    counters, next to the committed figures they replace (same command, another box: within 15 %)."""
    import importlib.util
    import shutil
    if not (shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3")):
        pytest.skip("no rocprofv3 on this box")
    spec = importlib.util.spec_from_file_location("bench_live", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    env_keys = [k for k in os.environ if k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BGN_BENCH_SPAWN")]
    saved = {k: os.environ.pop(k) for k in env_keys}
    try:
        live = bench.live_traffic(timeout_s=300, extras=True)
    finally:
        os.environ.update(saved)
    assert live is not None, "the PMC child passes failed"
    sec = live["secondary"]
    assert set(sec) == {"eadd_l1", "eadd_l2", "decrypt_lift_2^16", "decrypt_lift_k_pairing_1"}, sec
    alg = 3 * 258 * (1 << 20)
    assert 0.9 * alg < sec["eadd_l2"] < 1.3 * alg, sec["eadd_l2"] / alg                # one launch, wire to wire
    assert 1.5 * alg < sec["eadd_l1"] < 2.6 * alg, sec["eadd_l1"] / alg                # two passes over the slices + the prefix products
    assert sec["decrypt_lift_k_pairing_1"] > 8 * sec["decrypt_lift_2^16"] > 0
    for key, field in (("eadd_l1", "hbm_bytes_per_call"), ("eadd_l2", "hbm_bytes_per_launch"),
                       ("decrypt_lift_2^16", "hbm_bytes_per_launch"), ("decrypt_lift_k_pairing_1", "hbm_bytes_per_launch")):
        node, _ = bench.committed_traffic(key)
        if node and node.get(field):
            assert abs(sec[key] / node[field] - 1) < 0.15, (key, sec[key], node[field])
    with open(os.path.join(ROOT, "gpurun_out", "r06_live_traffic_secondary.json"), "w") as f:
        json.dump(live, f, indent=1)
