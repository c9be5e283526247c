"""CPU: the launch contract of bench.py (no GPU here: everything checked happens before the first GPU call).
`bench.py --gpus N` either finds itself inside a launcher's world of N ranks or spawns N fresh ranks of itself
before touching the GPU; a mismatch is an error, not a silent one-GPU run; without a GPU the script fails loudly."""
import importlib.util
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _load():
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_gpus_must_equal_the_launchers_world_size():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "--gpus 2 but WORLD_SIZE=3" in (r.stderr + r.stdout)


def _children(monkeypatch, bench, script):
    """Real child processes in place of the rank processes: each runs `script` (a Python source reading RANK)."""
    monkeypatch.setattr(bench, "rank_command", lambda argv: [sys.executable, "-c", script] + list(argv))


def test_spawn_starts_n_fresh_ranks_with_the_rendezvous_environment(monkeypatch, tmp_path):
    bench = _load()
    script = (
        "import os, sys, json\n"
        "keys = ['RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'HSA_ENABLE_IPC_MODE_LEGACY']\n"
        "d = {k: os.environ.get(k) for k in keys}\n"
        "d['argv'] = sys.argv[1:]\n"
        "open(os.path.join(%r, 'rank%%s.json' %% os.environ['RANK']), 'w').write(json.dumps(d))\n" % str(tmp_path))
    _children(monkeypatch, bench, script)
    assert bench.spawn_ranks(4, ["--gpus", "4", "--steps", "2"], timeout_s=60) == 0
    import json
    ports = set()
    for rank in range(4):
        env = json.load(open(tmp_path / ("rank%d.json" % rank)))
        assert env["argv"] == ["--gpus", "4", "--steps", "2"]
        assert env["RANK"] == env["LOCAL_RANK"] == str(rank) and env["WORLD_SIZE"] == env["LOCAL_WORLD_SIZE"] == "4"
        assert env["MASTER_ADDR"] == "127.0.0.1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        ports.add(env["MASTER_PORT"])
    assert len(ports) == 1


def test_rank_command_is_this_script():
    bench = _load()
    cmd = bench.rank_command(["--gpus", "2"])
    assert cmd[0] == sys.executable and os.path.abspath(cmd[1]) == BENCH and cmd[2:] == ["--gpus", "2"]


def test_a_failed_rank_ends_the_others_within_seconds(monkeypatch, tmp_path):
    """Rank 1 dies at once while rank 0 sits in its 'barrier' (a long sleep): the run must return non-zero within
    seconds and rank 0 must be gone — real processes, the situation of a peer lost inside an RCCL collective."""
    import time
    bench = _load()
    script = (
        "import os, sys, time\n"
        "open(os.path.join(%r, 'pid%%s' %% os.environ['RANK']), 'w').write(str(os.getpid()))\n"
        "if os.environ['RANK'] == '1':\n"
        "    sys.exit(3)\n"
        "time.sleep(120)\n" % str(tmp_path))
    _children(monkeypatch, bench, script)
    t0 = time.monotonic()
    rc = bench.spawn_ranks(2, [], timeout_s=100)
    assert rc == 3
    assert time.monotonic() - t0 < 20
    pid0 = int(open(tmp_path / "pid0").read())
    with pytest.raises(ProcessLookupError):
        os.kill(pid0, 0)                          # reaped by spawn_ranks: no such process


def test_the_wall_clock_cap_ends_a_hung_run(monkeypatch, tmp_path):
    import time
    bench = _load()
    script = (
        "import os, time\n"
        "open(os.path.join(%r, 'pid%%s' %% os.environ['RANK']), 'w').write(str(os.getpid()))\n"
        "time.sleep(120)\n" % str(tmp_path))
    _children(monkeypatch, bench, script)
    t0 = time.monotonic()
    assert bench.spawn_ranks(2, [], timeout_s=1.5) == 124
    assert time.monotonic() - t0 < 20
    for r in (0, 1):
        with pytest.raises(ProcessLookupError):
            os.kill(int(open(tmp_path / ("pid%d" % r)).read()), 0)


def test_the_watchdog_of_a_rank_exits_nonzero(tmp_path):
    """Inside a rank the cap is a daemon timer that ends the process with 124 (a rank hung in a collective under
    an external launcher)."""
    code = ("import importlib.util, time\n"
            "spec = importlib.util.spec_from_file_location('b', %r)\n"
            "b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
            "b.start_watchdog(); time.sleep(60)\n" % BENCH)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BGN_BENCH_TIMEOUT_S="1"), capture_output=True,
                       text=True, timeout=60)
    assert r.returncode == 124 and "BGN_BENCH_TIMEOUT_S" in r.stderr


def test_no_gpu_is_an_error_not_a_cpu_run():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, BENCH, "--no-extra", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr or "No HIP GPUs" in r.stderr or "BGN_E_HIP" in r.stderr or "hip" in r.stderr.lower()


def test_eight_ranks_one_dying_mid_gather_and_one_overrunning_the_cap(monkeypatch, tmp_path):
    """The driver's 8-GPU shape with real child processes: (1) eight ranks that all finish: rendezvous environment of
    every rank, one shared port, exit 0; (2) rank 5 dies while the other seven sit in their 'all-gather' (a long
    sleep): non-zero exit within seconds, all seven gone; (3) rank 6 never finishes: the wall-clock cap ends all eight."""
    import json
    import time
    bench = _load()
    # (1)
    script = (
        "import os, json\n"
        "d = {k: os.environ.get(k) for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'HSA_ENABLE_IPC_MODE_LEGACY')}\n"
        "open(os.path.join(%r, 'rank%%s.json' %% os.environ['RANK']), 'w').write(json.dumps(d))\n" % str(tmp_path))
    _children(monkeypatch, bench, script)
    assert bench.spawn_ranks(8, ["--gpus", "8"], timeout_s=120) == 0
    envs = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(8)]
    assert [e["RANK"] for e in envs] == [str(r) for r in range(8)] and {e["WORLD_SIZE"] for e in envs} == {"8"}
    assert len({e["MASTER_PORT"] for e in envs}) == 1 and {e["MASTER_ADDR"] for e in envs} == {"127.0.0.1"}
    assert {e["HSA_ENABLE_IPC_MODE_LEGACY"] for e in envs} == {"0"}
    # (2)
    script = (
        "import os, sys, time\n"
        "open(os.path.join(%r, 'pid%%s' %% os.environ['RANK']), 'w').write(str(os.getpid()))\n"
        "if os.environ['RANK'] == '5':\n"
        "    time.sleep(1.0)\n"
        "    sys.exit(7)\n"
        "time.sleep(120)\n" % str(tmp_path))
    _children(monkeypatch, bench, script)
    t0 = time.monotonic()
    assert bench.spawn_ranks(8, [], timeout_s=100) == 7
    assert time.monotonic() - t0 < 30
    for r in range(8):
        if r != 5:
            with pytest.raises(ProcessLookupError):
                os.kill(int(open(tmp_path / ("pid%d" % r)).read()), 0)
    # (3)
    script = (
        "import os, time\n"
        "open(os.path.join(%r, 'pid%%s' %% os.environ['RANK']), 'w').write(str(os.getpid()))\n"
        "time.sleep(120 if os.environ['RANK'] == '6' else 0.2)\n" % str(tmp_path))
    _children(monkeypatch, bench, script)
    t0 = time.monotonic()
    assert bench.spawn_ranks(8, [], timeout_s=3.0) == 124
    assert time.monotonic() - t0 < 30
    with pytest.raises(ProcessLookupError):
        os.kill(int(open(tmp_path / "pid6").read()), 0)


def test_live_traffic_falls_back_quietly(monkeypatch, tmp_path):
    """bench.py measures roofline.traffic itself in two rocprofv3 --pmc child passes; where that cannot work — no GPU
    here, so the child bench exits non-zero; no rocprofv3; a profiler already around this process; a pass that hangs
    — it returns None (the line then quotes the committed summary and says so) and leaves nothing behind in /tmp."""
    import glob
    bench = _load()
    before = set(glob.glob("/tmp/bgn_pmc_*"))
    monkeypatch.setenv("ROCPROFILER_SOMETHING", "1")                  # being profiled: no nested passes, nothing started
    assert bench.live_traffic(timeout_s=5) is None
    monkeypatch.delenv("ROCPROFILER_SOMETHING")
    import shutil
    if shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3"):
        assert bench.live_traffic(timeout_s=120) is None              # the child finds no GPU and exits non-zero
    # a pass that never ends: a stand-in for rocprofv3 that sleeps; the whole process group is ended at the timeout
    fake = tmp_path / "rocprofv3"
    fake.write_text("#!/bin/sh\nsleep 600\n")
    fake.chmod(0o755)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    import time
    t0 = time.time()
    assert bench.live_traffic(timeout_s=2) is None
    assert time.time() - t0 < 30
    assert set(glob.glob("/tmp/bgn_pmc_*")) == before
