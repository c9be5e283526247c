#!/usr/bin/env python3
"""Batch sizes that do not fill whole rounds of the lane kernel (65536 lanes, up to sixteen pairings each): wall
time of Mult, makeL2 and Decrypt of both levels under the default dispatch with the cut into whole rounds + remainder
(engine.cpp lane_rounds_head / decrypt_rounds_head) and without it (BGN_SPLIT_ROUNDS=0); device-resident operands,
best of three, results of the two compared.
    python tools/odd_sizes.py [k1024] > profiles/r03_odd_sizes.csv"""
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
    keys = sys.argv[1:] or ["k1024"]
    counts = [int(x) for x in os.environ.get("ODD_COUNTS", "65536,66000,70000,81920,100000,131072,135000,1048576,1049576").split(",")]
    print("key,op,count,split,ms,ops_per_s,last_kernel")
    dev = torch.device("cuda", 0)
    for key in keys:
        fx = load_fixture(key)
        pk = bgn_amd.PublicKey(int(fx["p"], 16), int(fx["n"], 16), fx["l"], bytes.fromhex(fx["P"]), bytes.fromhex(fx["Q"]),
                               fx["msg_space"], True, fx["poly_base"])
        pk.SetupDecryption(bgn_amd.SecretKey(int(fx["q1"], 16)))
        eng = pk.engine
        EB = eng.elem_bytes
        nmax = max(counts)
        _, _, cts = syn.config2_ciphertexts(pk, nmax, seed=3, device=dev)
        b = syn.permuted_copy(cts, EB, seed=4)
        out = torch.empty(nmax * EB, dtype=torch.uint8, device=dev)
        msg = torch.empty(nmax, dtype=torch.int64, device=dev)
        sta = torch.empty(nmax, dtype=torch.uint8, device=dev)
        l2 = torch.empty(nmax * EB, dtype=torch.uint8, device=dev)
        eng.make_l2_dev(cts, l2, nmax)
        for op in ("mult", "make_l2", "decrypt_l1", "decrypt_l2"):
            for n in counts:
                ref = None
                for split in ("1", "0"):
                    eng.set_option("split_rounds", int(split))
                    best = None
                    for rep in range(3 if n < (1 << 19) else 2):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        if op == "mult":
                            eng.mult_dev(cts[: n * EB], b[: n * EB], out[: n * EB], n)
                        elif op == "make_l2":
                            eng.make_l2_dev(cts[: n * EB], out[: n * EB], n)
                        elif op == "decrypt_l1":
                            eng.decrypt_dev(1, cts[: n * EB], msg[:n], sta[:n], n)
                        else:
                            eng.decrypt_dev(2, l2[: n * EB], msg[:n], sta[:n], n)
                        torch.cuda.synchronize()
                        dt = time.perf_counter() - t0
                        best = dt if best is None else min(best, dt)
                    got = (msg[:n].clone(), sta[:n].clone()) if op.startswith("decrypt") else out[: n * EB].clone()
                    if ref is None:
                        ref = got
                    elif op.startswith("decrypt"):
                        assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1]), (key, op, n)
                    else:
                        assert torch.equal(ref, got), (key, op, n)
                    name = eng.last_aux_kernel_name() if op == "decrypt_l1" else eng.last_kernel_name()
                    print("%s,%s,%d,%s,%.2f,%.1f,%s" % (key, op, n, "rounds+remainder" if split == "1" else "one launch", best * 1e3,
                                                      n / best, name), flush=True)
        del pk


if __name__ == "__main__":
    main()
