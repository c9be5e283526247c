#!/usr/bin/env python3
"""MultPoly (poly.go:123-156) wall time by the number of polynomial pairs, d1 x d2 coefficients each, device-resident
operands, best of three: where the time steps.  BGN_POLY_LEVELS=n forces the number of Karatsuba levels (column
karatsuba_levels; "planned" = engine.cpp poly_plan_levels chooses).
    python tools/multpoly_sizes.py [k1024] > profiles/r03_multpoly_sizes.csv"""
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
    shapes = [tuple(int(v) for v in s.split("x")) for s in os.environ.get("MP_SHAPES", "16x16,4x4,10x10").split(",")]
    npolys = [int(x) for x in os.environ.get("MP_NPOLY", "16,64,128,170,200,256,260,300,400,512,600,1024,1100,2048,4096,4200").split(",")]
    if not os.environ.get("MP_NO_HEADER"):
        print("key,d1,d2,npoly,karatsuba_levels,pairs,ms,pairs_per_s,kernel")
    dev = torch.device("cuda", 0)
    for key in keys:
        fx = load_fixture(key)
        pk = bgn_amd.PublicKey(int(fx["p"], 16), int(fx["n"], 16), fx["l"], bytes.fromhex(fx["P"]), bytes.fromhex(fx["Q"]),
                               fx["msg_space"], True, fx["poly_base"])
        eng = pk.engine
        EB = eng.elem_bytes
        for d1, d2 in shapes:
            nmax = max(npolys)
            _, _, a = syn.config2_ciphertexts(pk, nmax * d1, seed=5, device=dev, digits=True)
            _, _, b = syn.config2_ciphertexts(pk, nmax * d2, seed=6, device=dev, digits=True)
            out = torch.empty(nmax * (d1 + d2) * EB, dtype=torch.uint8, device=dev)
            for n in npolys:
                best = None
                for rep in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    eng.poly_mult_dev(n, d1, d2, a[: n * d1 * EB], b[: n * d2 * EB], out[: n * (d1 + d2) * EB])
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                    best = dt if best is None else min(best, dt)
                print("%s,%d,%d,%d,%s,%d,%.2f,%.1f,%s" % (key, d1, d2, n, ("planned" if eng.get_option("poly_levels") < 0 else str(eng.get_option("poly_levels"))), n * d1 * d2, best * 1e3, n * d1 * d2 / best,
                                                      eng.last_kernel_name()), flush=True)


if __name__ == "__main__":
    main()
