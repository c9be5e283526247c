#!/usr/bin/env python3
"""Throughput of the reference's own call shape: T host threads, each calling a single-element host-buffer entry
point on ONE context in a loop (poly.go:139-153: one goroutine per coefficient pair around pk.Mult; bgn_test.go:97-140:
one op per call), with the combiner of concurrent small calls (csrc/combiner.hpp) on and off, and the latency of a
lone caller either way.  ctypes releases the GIL inside the call, so the threads are concurrent inside the library.

Two caller pools: `native` — tools/concurrent_callers.cpp, std::thread callers of the C ABI, what a Go host's
goroutines look like to the library (run as a child process on inputs this script prepares) — and `python`, the same
loop from Python threads (CC_PYTHON=1), where the interpreter lock spreads the callers' re-submissions over
milliseconds.

    python tools/concurrent_callers.py [k1024] > profiles/r04_concurrent_callers.csv
"""
import ctypes as C
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
from conftest import load_fixture  # noqa: E402
import bgn_amd  # noqa: E402


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def native_rows(fx, key, eng, N, A, B, L2, K, xs, want_mult, want_add, seconds, threads_list, ops):
    """Write the inputs, build the native caller pool and run it as a child process (it creates its own context)."""
    import struct
    import subprocess
    import tempfile
    src = os.path.join(ROOT, "tools", "concurrent_callers.cpp")
    exe = os.path.join(ROOT, "tools", "_build", "concurrent_callers")
    lib = os.path.join(ROOT, "bgn_amd", "lib")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-I" + os.path.join(ROOT, "include"), src, "-L" + lib,
                               "-lbgn_amd", "-Wl,-rpath," + lib, "-o", exe])
    want_mc = eng.multconst(1, A.tobytes(), [int.from_bytes(bytes(k), "big") for k in K])
    ib = lambda v: int(v).to_bytes((int(v).bit_length() + 7) // 8, "big")
    p, n, q1 = ib(int(fx["p"], 16)), ib(int(fx["n"], 16)), ib(int(fx["q1"], 16))
    kb = key.encode()
    blob = b"BGNCC1\0\0" + struct.pack("<IIQQIIII", eng.L, N, fx["l"], fx["msg_space"], len(p), len(n), len(q1), len(kb))
    blob += p + n + q1 + bytes.fromhex(fx["P"]) + bytes.fromhex(fx["Q"])
    blob += A.tobytes() + B.tobytes() + L2.tobytes() + K.tobytes() + want_mult.tobytes() + want_add.tobytes() + want_mc.tobytes()
    blob += struct.pack("<%dq" % N, *xs[:N]) + kb
    with tempfile.NamedTemporaryFile(suffix=".bgncc", delete=False) as f:
        f.write(blob)
        path = f.name
    try:
        sys.stdout.flush()
        subprocess.check_call([exe, path, str(seconds), ",".join(str(t) for t in threads_list), ",".join(ops)])
    finally:
        os.unlink(path)


def main():
    key = sys.argv[1] if len(sys.argv) > 1 else "k1024"
    seconds = float(os.environ.get("CC_SECONDS", "4"))
    threads_list = [int(x) for x in os.environ.get("CC_THREADS", "1,16,64,256").split(",")]
    fx = load_fixture(key)
    pk = bgn_amd.PublicKey(int(fx["p"], 16), int(fx["n"], 16), fx["l"], bytes.fromhex(fx["P"]), bytes.fromhex(fx["Q"]),
                           fx["msg_space"], True, fx["poly_base"])
    pk.SetupDecryption(bgn_amd.SecretKey(int(fx["q1"], 16)))
    eng = pk.engine
    lib, h, E = eng._lib, eng._h, eng.elem_bytes
    nn = int(fx["n"], 16)
    N = 512
    rng = np.random.default_rng(3)
    xs = [int(v) for v in rng.integers(0, min(fx["msg_space"], 1 << 40), 2 * N)]
    rs = [int.from_bytes(rng.bytes(len(fx["n"]) // 2), "big") % nn for _ in range(2 * N)]
    cts = eng.encrypt(xs, rs)
    A, B = cts[:N].copy(), cts[N:].copy()
    L2 = eng.make_l2(A.tobytes()).copy()
    K = np.frombuffer(b"".join(int(rng.integers(1, 1 << 40)).to_bytes(5, "big") for _ in range(N)), dtype=np.uint8).reshape(N, 5).copy()
    want_mult = eng.mult(A.tobytes(), B.tobytes())
    want_add = eng.add(1, A.tobytes(), B.tobytes())

    def make_call(op, i):
        """A closure over pre-built ctypes arguments: the loop body is one foreign call."""
        out = np.zeros(E, dtype=np.uint8)
        m = np.zeros(1, dtype=np.int64)
        st = np.zeros(1, dtype=np.uint8)
        a, b, l2, k = A[i], B[i], L2[i], K[i]
        if op == "mult":
            args = (h, 1, P(a), P(b), None, 0, P(out))
            fn = lib.bgn_mult_batch
        elif op == "add_l1":
            args = (h, 1, 1, P(a), P(b), None, 0, P(out))
            fn = lib.bgn_add_batch
        elif op == "add_l2":
            args = (h, 1, 2, P(l2), P(l2), None, 0, P(out))
            fn = lib.bgn_add_batch
        elif op == "decrypt_l1":
            args = (h, 1, 1, P(a), P(m), P(st))
            fn = lib.bgn_decrypt_batch
        elif op == "decrypt_l2":
            args = (h, 1, 2, P(l2), P(m), P(st))
            fn = lib.bgn_decrypt_batch
        elif op == "multconst_l1_k40":
            args = (h, 1, 1, P(a), P(k), 5, None, 0, P(out))
            fn = lib.bgn_multconst_batch
        else:
            raise ValueError(op)
        keep = (out, m, st, a, b, l2, k)
        return fn, args, keep

    ops = os.environ.get("CC_OPS", "mult,add_l1,decrypt_l1,multconst_l1_k40,add_l2,decrypt_l2").split(",")
    native_rows(fx, key, eng, N, A, B, L2, K, xs, want_mult, want_add, seconds, threads_list, ops)
    if os.environ.get("CC_PYTHON") != "1":
        return
    for op in ops:
        for combine in (1, 0):
            eng.set_option("combine", combine)
            for T in threads_list:
                if combine == 0 and T not in (1, max(threads_list)):
                    continue
                calls = [make_call(op, t % N) for t in range(T)]
                for fn, args, _ in calls[: min(T, 4)]:          # warm-up (workspace, tables)
                    assert fn(*args) == 0
                counts = [0] * T
                stop = time.perf_counter() + seconds
                start = threading.Barrier(T + 1)
                errs = []

                def worker(t):
                    fn, args, _ = calls[t]
                    start.wait()
                    n = 0
                    while time.perf_counter() < stop:
                        if fn(*args) != 0:
                            errs.append(t)
                            break
                        n += 1
                    counts[t] = n

                s0 = eng.combiner_stats()
                th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
                for t in th:
                    t.start()
                start.wait()
                t0 = time.perf_counter()
                stop = t0 + seconds
                for t in th:
                    t.join()
                dt = time.perf_counter() - t0
                s1 = eng.combiner_stats()
                assert not errs, errs
                total = sum(counts)
                ok = "-"
                if op == "mult":
                    ok = all(calls[t][2][0].tobytes() == want_mult[t % N].tobytes() for t in range(T))
                elif op == "add_l1":
                    ok = all(calls[t][2][0].tobytes() == want_add[t % N].tobytes() for t in range(T))
                elif op == "decrypt_l1":
                    ok = all(int(calls[t][2][1][0]) == xs[t % N] and int(calls[t][2][2][0]) == 0 for t in range(T))
                print("%s,%s,python,%d,%d,%.2f,%d,%.1f,%.3f,%d,%d,%s" % (
                    key, op, T, combine, dt, total, total / dt, dt / max(1, total) * T * 1e3,
                    s1["groups"] - s0["groups"], s1["max_group"], ok), flush=True)
    eng.set_option("combine", 1)


if __name__ == "__main__":
    main()
