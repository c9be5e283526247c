"""CPU: the launch contract of bench.py (no GPU here: everything checked happens before the first GPU call).
`bench.py --gpus N` either finds itself inside a launcher's world of N ranks or spawns N fresh ranks of itself
before touching the GPU; a mismatch is an error, not a silent one-GPU run; without a GPU the script fails loudly."""
import importlib.util
import os
import subprocess
import sys

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


def test_spawn_starts_n_fresh_ranks_with_the_rendezvous_environment(monkeypatch):
    bench = _load()
    started = []

    class FakeProc:
        returncode = 0

        def wait(self):
            return 0

        def poll(self):
            return 0

    def fake_popen(cmd, env=None, **kw):
        started.append((cmd, env))
        return FakeProc()

    monkeypatch.setattr(bench.subprocess, "Popen", fake_popen)
    assert bench.spawn_ranks(4, ["--gpus", "4", "--steps", "2"]) == 0
    assert len(started) == 4
    ports = set()
    for rank, (cmd, env) in enumerate(started):
        assert cmd[0] == sys.executable and os.path.abspath(cmd[1]) == BENCH and cmd[2:] == ["--gpus", "4", "--steps", "2"]
        assert env["RANK"] == env["LOCAL_RANK"] == str(rank) and env["WORLD_SIZE"] == env["LOCAL_WORLD_SIZE"] == "4"
        assert env["MASTER_ADDR"] == "127.0.0.1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        ports.add(env["MASTER_PORT"])
    assert len(ports) == 1


def test_a_failed_rank_fails_the_run(monkeypatch):
    bench = _load()
    killed = []

    class Proc:
        def __init__(self, rc, hangs):
            self.returncode, self.hangs = rc, hangs

        def wait(self):
            return self.returncode

        def poll(self):
            return None if self.hangs else self.returncode

        def kill(self):
            killed.append(self)

    procs = [Proc(1, False), Proc(0, True)]
    monkeypatch.setattr(bench.subprocess, "Popen", lambda cmd, env=None, **kw: procs.pop(0))
    assert bench.spawn_ranks(2, []) != 0
    assert len(killed) == 1                       # the rank left at the barrier is ended


def test_no_gpu_is_an_error_not_a_cpu_run():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, BENCH, "--no-extra", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr or "No HIP GPUs" in r.stderr or "BGN_E_HIP" in r.stderr or "hip" in r.stderr.lower()
