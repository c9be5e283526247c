#!/usr/bin/env python3
"""Encrypt at 2^20 (1024-bit key, 40-bit plaintexts, full-length blinding exponents) with signed and unsigned windows
over Q's 20-bit table (option fixed_signed_q), a fresh context each; bytes compared.

    python tools/encrypt_signed_ab.py [1|0 ...] > profiles/r05_encrypt_signed_windows.csv
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from conftest import load_fixture  # noqa: E402
import bgn_amd  # noqa: E402
import bgn_amd.synthetic as syn  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    fx = load_fixture("k1024")
    n = 1 << 20
    variants = [int(a) for a in sys.argv[1:]] or [1, 0, 1, 0]
    print("key,signed_q,q_window_bits,batch,ms,encrypts_per_s,first_call_s,bytes_equal_to_first_variant")
    ref = None
    for signed in variants:
        pk = bgn_amd.PublicKey(int(fx["p"], 16), int(fx["n"], 16), fx["l"], bytes.fromhex(fx["P"]), bytes.fromhex(fx["Q"]),
                               fx["msg_space"], True, fx["poly_base"])
        eng = pk.engine
        eng.set_option("fixed_signed_q", signed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        xs, rs, cts = syn.config2_ciphertexts(pk, n, seed=1000, device=dev)
        first = time.perf_counter() - t0
        out = torch.empty_like(cts)
        ts = []
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.encrypt_dev(xs, xs.shape[1], rs, rs.shape[1], out, n)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        dt = min(ts)
        if ref is None:
            ref = cts.clone()
        same = bool((out == ref).all().item())
        print("%s,%d,%d,%d,%.2f,%.0f,%.2f,%s" % (fx["name"], signed, 20, n, dt * 1e3, n / dt, first, same), flush=True)
        del xs, rs, cts, out
        eng.close()


if __name__ == "__main__":
    main()
