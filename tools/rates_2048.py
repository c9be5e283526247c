#!/usr/bin/env python3
"""Rough rates of the 2048-bit (72-limb) path, host buffers, best of two: python tools/rates_2048.py > profiles/r03_rates_2048.csv"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_fixture  # noqa: E402
import bgn_amd  # noqa: E402


def best(fn, reps=2):
    b = None
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        dt = time.perf_counter() - t0
        b = dt if b is None or dt < b else b
    return b, r


def main():
    fx = load_fixture("k2048")
    pk = bgn_amd.PublicKey(int(fx["p"], 16), int(fx["n"], 16), fx["l"], bytes.fromhex(fx["P"]), bytes.fromhex(fx["Q"]),
                           fx["msg_space"], True, fx["poly_base"])
    pk.engine.set_memory_budget(96 << 30)
    pk.SetupDecryption(bgn_amd.SecretKey(int(fx["q1"], 16)))
    eng = pk.engine
    rng = random.Random(1)
    n = int(fx["n"], 16)
    N = 4096
    xs = [rng.randrange(fx["msg_space"]) for _ in range(2 * N)]
    rs = [rng.randrange(n) for _ in range(2 * N)]
    print("op,count,ms,ops_per_s,kernel")
    eng.encrypt(xs[:64], rs[:64])
    dt, cts = best(lambda: eng.encrypt(xs, rs))
    print("encrypt,%d,%.2f,%.1f,%s" % (2 * N, dt * 1e3, 2 * N / dt, eng.last_kernel_name()))
    with eng.options(quad_max_enc=0):                    # the 72-limb chain kernels (functional) beside it
        dt, cts0 = best(lambda: eng.encrypt(xs[:N], rs[:N]), reps=1)
        assert cts0.tobytes() == cts[:N].tobytes()
        print("encrypt (chain kernels: the functional fallback),%d,%.2f,%.1f,%s" % (N, dt * 1e3, N / dt, eng.last_kernel_name()))
    a, b = cts[:N].tobytes(), cts[N:].tobytes()
    ks = [rng.randrange(n) for _ in range(N)]
    dt, mc = best(lambda: eng.multconst(1, a, ks))
    print("multconst_l1 (2048-bit k),%d,%.2f,%.1f,%s" % (N, dt * 1e3, N / dt, eng.last_kernel_name()))
    with eng.options(quad_max_mc=0):
        dt, mc0 = best(lambda: eng.multconst(1, a[: 64 * eng.elem_bytes], ks[:64]), reps=1)
        assert mc0.tobytes() == mc[:64].tobytes()
        print("multconst_l1 (lane kernel: the functional fallback),%d,%.2f,%.1f,%s" % (64, dt * 1e3, 64 / dt, eng.last_kernel_name()))
    dt, _ = best(lambda: eng.add(1, a, b))
    print("add_l1,%d,%.2f,%.1f,%s" % (N, dt * 1e3, N / dt, eng.last_kernel_name()))
    for cnt in (256, 4096):
        dt, out = best(lambda: eng.mult(a[: cnt * eng.elem_bytes], b[: cnt * eng.elem_bytes]))
        print("mult,%d,%.2f,%.1f,%s" % (cnt, dt * 1e3, cnt / dt, eng.last_kernel_name()))
    dt, _ = best(lambda: eng.make_l2(a))
    print("make_l2,%d,%.2f,%.1f,%s" % (N, dt * 1e3, N / dt, eng.last_kernel_name()))
    dt, (m, st) = best(lambda: eng.decrypt(1, a))
    assert m.tolist() == xs[:N] and not st.any()
    print("decrypt_l1,%d,%.2f,%.1f,%s" % (N, dt * 1e3, N / dt, eng.last_aux_kernel_name()))
    eng.set_option("quad_max", 0)
    dt, out2 = best(lambda: eng.mult(a[: 256 * eng.elem_bytes], b[: 256 * eng.elem_bytes]), reps=1)
    assert out2.tobytes() == eng.mult(a[: 256 * eng.elem_bytes], b[: 256 * eng.elem_bytes]).tobytes()
    print("mult (lane kernel: the functional fallback),%d,%.2f,%.1f,%s" % (256, dt * 1e3, 256 / dt, eng.last_kernel_name()))


if __name__ == "__main__":
    main()
