"""The C ABI from plain C, in the call shape of the cgo binding: tests/cpp/cgo_shape.c is
compiled with gcc -std=c99 against include/bgn_amd.h (cgo compiles C, not C++) — on the CPU it must build, link and fail
loudly without a GPU; on the GPU 24 pthreads of single-element Mult / Add / MultConst / Decrypt calls on one context
return the bytes of the batch calls, and the device-array chain of go/bgn_amd.go's *Dev methods (bgn_dev_alloc, Mult -> Add
-> Decrypt through the `_dev` entry points, validate, calibrate) gives the plaintext of the host-buffer calls."""
import os
import subprocess

import pytest

from conftest import ROOT, engine_key, load_fixture

BIN = os.path.join(ROOT, "tests", "cpp", "_build", "cgo_shape")


def build():
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    lib = os.path.join(ROOT, "bgn_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", "-pthread",
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "cgo_shape.c"), "-L" + lib,
                           "-lbgn_amd", "-Wl,-rpath," + lib, "-o", BIN])


def h(s):
    s = s[2:] if s.startswith("0x") else s
    return s if len(s) % 2 == 0 else "0" + s


def test_header_is_c99_and_the_c_caller_fails_loudly_without_gpu():
    import torch
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c",
                           os.path.join(ROOT, "include", "bgn_amd.h")])
    build()
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    fx = load_fixture("toy64")
    z = "00" * (2 * fx["fp_bytes"])
    r = subprocess.run([BIN, h(fx["p"]), h(fx["n"]), str(fx["l"]), fx["P"], fx["Q"], h(fx["q1"]), str(fx["msg_space"]),
                        z, z, "05", z, z, z, "0", "0"], capture_output=True, text=True)
    assert r.returncode == 3 and "no HIP device" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_c_callers_in_the_cgo_call_shape():
    build()
    fx = load_fixture("k512")
    pk, sk = engine_key(fx)
    pk.SetupDecryption(sk)
    eng = pk.engine
    a = bytes.fromhex(fx["encrypt"][3]["ct"])
    b = bytes.fromhex(fx["encrypt"][7]["ct"])                   # 5 * 36 * 2 stays inside the message space of 1021
    k = 0x1234567
    want_mult = eng.mult(a, b).tobytes().hex()
    want_add = eng.add(1, a, b).tobytes().hex()
    want_mc = eng.multconst(1, a, [k]).tobytes().hex()
    m, st = eng.decrypt(1, a)
    assert int(st[0]) == 0
    # the device-resident chain of the C caller: Mult -> Add on level 2 (the product with itself) -> Decrypt
    prod = eng.mult(a, b).tobytes()
    m2, st2 = eng.decrypt(2, eng.add(2, prod, prod).tobytes())
    assert int(st2[0]) == 0
    r = subprocess.run([BIN, h(fx["p"]), h(fx["n"]), str(fx["l"]), fx["P"], fx["Q"], h(fx["q1"]), str(fx["msg_space"]),
                        a.hex(), b.hex(), "%08x" % k, want_mult, want_add, want_mc, str(int(m[0])), str(int(m2[0]))],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "cgo shape ok" in r.stdout, r.stdout + r.stderr
