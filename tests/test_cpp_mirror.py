"""The C++ host mirror (include/bgn_amd.hpp): compiles against the C ABI on CPU, fails loudly without a
GPU, and reproduces the reference CLI's truth table on the GPU."""
import os
import subprocess

import pytest

from conftest import ROOT, load_fixture

BIN = os.path.join(ROOT, "tests", "cpp", "_build", "truth_table")


def build():
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    lib = os.path.join(ROOT, "bgn_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "truth_table.cpp"), "-L" + lib, "-lbgn_amd",
                           "-Wl,-rpath," + lib, "-o", BIN])


def args(fx):
    h = lambda s: s[2:] if s.startswith("0x") else s
    return [BIN, h(fx["p"]), h(fx["n"]), str(fx["l"]), fx["P"], fx["Q"], h(fx["q1"]), str(fx["msg_space"])]


def test_cpp_mirror_builds_and_fails_loudly_without_gpu():
    import torch
    build()
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run(args(load_fixture("toy64")), capture_output=True, text=True)
    assert r.returncode == 3 and "engine error -3" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_mirror_truth_table():
    build()
    r = subprocess.run(args(load_fixture("k512")), capture_output=True, text=True)
    assert r.returncode == 0 and "truth table ok" in r.stdout, r.stdout + r.stderr
