"""CPU: the combiner of concurrent small calls (bgn_amd/csrc/combiner.hpp) on a host-memory backend — 48 threads of
mixed kinds of requests against a fake launch, built with ThreadSanitizer: every caller gets its own results and
status, kinds are never mixed, a failing kind fails alone, concurrent callers are merged, a lone caller is not delayed.
The GPU side of the same code: tests/test_gpu_concurrent.py."""
import os
import subprocess

from conftest import ROOT


def test_combiner_group_commit_under_thread_sanitizer():
    src = os.path.join(ROOT, "tests", "cpp", "combiner_test.cpp")
    out = os.path.join(ROOT, "tests", "cpp", "_build", "combiner_test")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", src, "-o", out])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    r = subprocess.run([out, "48", "60"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "combiner ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
