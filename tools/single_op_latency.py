#!/usr/bin/env python3
"""Latency of ONE call of each operation through the host-buffer entry points at count = 1 and 32 — the reference's
own call shape (one pbc.Element operation per Go method call).  python tools/single_op_latency.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401
from conftest import load_fixture  # noqa: E402
import bgn_amd  # noqa: E402


def best(fn, reps=5):
    fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return min(t) * 1e3


def main():
    print("key,op,count,ms")
    for key in ("k512", "k1024"):
        fx = load_fixture(key)
        pk = bgn_amd.PublicKey(int(fx["p"], 16), int(fx["n"], 16), fx["l"], bytes.fromhex(fx["P"]), bytes.fromhex(fx["Q"]),
                               fx["msg_space"], True, fx["poly_base"])
        pk.SetupDecryption(bgn_amd.SecretKey(int(fx["q1"], 16)))
        eng = pk.engine
        n = int(fx["n"], 16)
        for cnt in (1, 32):
            xs = [(7 * i + 3) % 1000 for i in range(2 * cnt)]
            rs = [(123456789 * (i + 1)) % n for i in range(2 * cnt)]
            cts = eng.encrypt(xs, rs)
            a, b = cts[:cnt].tobytes(), cts[cnt:].tobytes()
            l2 = eng.mult(a, b).tobytes()
            pc = eng.encrypt([i % 3 for i in range(20)], [(987654321 * (i + 1)) % n for i in range(20)])
            pa, pb = pc[:10].tobytes(), pc[10:].tobytes()
            ops = {
                "encrypt": lambda: eng.encrypt(xs[:cnt], rs[:cnt]),
                "add_l1": lambda: eng.add(1, a, b),
                "add_l2": lambda: eng.add(2, l2, l2),
                "mult": lambda: eng.mult(a, b),
                "make_l2": lambda: eng.make_l2(a),
                "multconst_l1_k40": lambda: eng.multconst(1, a, [(1 << 39) + i for i in range(cnt)]),
                "multconst_l2_k40": lambda: eng.multconst(2, l2, [(1 << 39) + i for i in range(cnt)]),
                "multpoly_10x10": lambda: eng.poly_mult(1, 10, 10, pa, pb),
                "decrypt_l1": lambda: eng.decrypt(1, a),
                "decrypt_l2": lambda: eng.decrypt(2, l2),
            }
            for name, fn in ops.items():
                print("%s,%s,%d,%.3f" % (key, name, cnt, best(fn)), flush=True)


if __name__ == "__main__":
    main()
