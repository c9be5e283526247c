#!/usr/bin/env python3
"""MultConst (per-element scalars, bgn.go:253-291) by batch size: the lane groups (quad/quad_g1.hpp) against one element
per lane, both levels, device-resident operands, best of three; the crossover of engine.cpp quad_mc_limit comes from
this sweep.    python tools/multconst_mid_batch.py [k1024 ...] > profiles/r04_multconst_mid_batch.csv"""
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
    keys = sys.argv[1:] or ["k1024", "k512"]
    counts = [int(x) for x in os.environ.get("MC_COUNTS", "1,256,1024,4096,8192,16384,32768,49152,65536,66000").split(",")]
    kbytes = [int(x) for x in os.environ.get("MC_KBYTES", "5,32,128").split(",")]
    dev = torch.device("cuda", 0)
    print("key,level,scalar_bits,count,kernel,ms,ops_per_s,reported_kernel")
    for key in keys:
        fx = load_fixture(key)
        pk = bgn_amd.PublicKey(int(fx["p"], 16), int(fx["n"], 16), fx["l"], bytes.fromhex(fx["P"]), bytes.fromhex(fx["Q"]),
                               fx["msg_space"], True, fx["poly_base"])
        eng = pk.engine
        EB = eng.elem_bytes
        nmax = max(counts)
        _, _, cts = syn.config2_ciphertexts(pk, nmax, seed=3, device=dev)
        l2 = torch.empty_like(cts)
        eng.make_l2_dev(cts, l2, nmax)
        out = torch.empty_like(cts)
        g = torch.Generator().manual_seed(5)
        for kb in kbytes:
            if kb * 8 > int(fx["n"], 16).bit_length() + 8:
                continue
            ks = torch.randint(0, 256, (nmax, kb), dtype=torch.uint8, generator=g).to(dev)
            ref = {}
            for level, src in ((1, cts), (2, l2)):
                for kernel in ("quad", "lane", "lane_binary"):
                    if kernel == "lane_binary" and (level != 1 or kb >= 16):
                        continue                                      # only short scalars on level 1 have that alternative
                    eng.set_option("quad_max_mc", (1 << 40) if kernel == "quad" else 0)
                    eng.set_option("g1_mul_window_short", 0 if kernel == "lane_binary" else 1)
                    for n in counts:
                        if kernel == "quad" and n > 65536:
                            eng.set_option("quad_max_mc", -1)         # the default cut: lane rounds + a lane-group remainder
                        best = None
                        for rep in range(3):
                            torch.cuda.synchronize()
                            t0 = time.perf_counter()
                            eng._lib.bgn_multconst_batch_dev(eng._h, n, level, src.data_ptr(), ks.data_ptr(), kb, None, 0,
                                                             out.data_ptr(), eng._stream())
                            torch.cuda.synchronize()
                            dt = time.perf_counter() - t0
                            best = dt if best is None or dt < best else best
                        r = out[: min(n, 2048) * EB].clone()
                        kk = (level, n)
                        if kk in ref:
                            assert bool((ref[kk] == r).all().item()), "kernels differ"
                        ref[kk] = r
                        print("%s,%d,%d,%d,%s,%.3f,%.1f,%s" % (key, level, kb * 8, n, kernel if n <= 65536 or kernel != "quad" else "default",
                                                               best * 1e3, n / best, eng.last_kernel_name()), flush=True)
                    eng.set_option("quad_max_mc", -1)
                    eng.set_option("g1_mul_window_short", 1)


if __name__ == "__main__":
    main()
