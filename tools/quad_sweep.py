#!/usr/bin/env python3
"""Mid-size batches: wall time by batch size on the three kernel families — the wave-cooperative one
(coop/coop.hpp), the lane-group one (quad/quad.hpp) and one element per lane — device-resident operands, best of
three, results compared.  Operations (QUAD_SWEEP_OPS, default "mult"): mult; make_l2 and decrypt_l1 (walks over a
key's line table; Decrypt also its power by the secret key).  The crossovers of coop_limit / quad_limit /
quad_table_limit (engine.cpp) are chosen from these CSVs:
    python tools/quad_sweep.py [k1024 ...] > profiles/r03_mid_batch.csv
    QUAD_SWEEP_OPS=make_l2,decrypt_l1 python tools/quad_sweep.py > profiles/r03_mid_batch_table.csv"""
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
    keys = sys.argv[1:] or ["k512", "k1024"]
    counts = [int(x) for x in os.environ.get("QUAD_SWEEP_COUNTS", "256,1024,2048,4096,6144,8192,12288,16384,32768,49152,65536").split(",")]
    kernels = os.environ.get("QUAD_SWEEP_KERNELS", "quad,coop,lane").split(",")
    ops = os.environ.get("QUAD_SWEEP_OPS", "mult").split(",")
    print("key,op,count,kernel,ms,ops_per_s,kernel_name")
    dev = torch.device("cuda", 0)
    for key in keys:
        fx = load_fixture(key)
        pk = bgn_amd.PublicKey(int(fx["p"], 16), int(fx["n"], 16), fx["l"], bytes.fromhex(fx["P"]), bytes.fromhex(fx["Q"]),
                               fx["msg_space"], True, fx["poly_base"])
        eng = pk.engine
        EB = eng.elem_bytes
        nmax = max(counts)
        _, _, cts = syn.config2_ciphertexts(pk, nmax, seed=3, device=dev)
        b = syn.permuted_copy(cts, EB, seed=4)
        out = torch.empty(nmax * EB, dtype=torch.uint8, device=dev)
        ref = {}
        if "decrypt_l1" in ops:
            pk.SetupDecryption(bgn_amd.SecretKey(int(fx["q1"], 16)))
        msg = torch.empty(nmax, dtype=torch.int64, device=dev)
        sta = torch.empty(nmax, dtype=torch.uint8, device=dev)
        big = "100000000"
        for op in ops:
            for kernel in kernels:
                eng.force_kernel(kernel)              # options of the context (bgn_ctx_set_option), not the environment
                for n in counts:
                    if kernel == "coop" and n > 16384:
                        continue
                    best = None
                    for rep in range(3):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        if op == "mult":
                            eng.mult_dev(cts[: n * EB], b[: n * EB], out, n)
                        elif op == "make_l2":
                            eng.make_l2_dev(cts[: n * EB], out, n)
                        else:
                            eng.decrypt_dev(1, cts[: n * EB], msg, sta, n)
                        torch.cuda.synchronize()
                        dt = time.perf_counter() - t0
                        best = dt if best is None or dt < best else best
                    digest = hash(out[: n * EB].cpu().numpy().tobytes()) if op != "decrypt_l1" else \
                        hash(msg[:n].cpu().numpy().tobytes() + sta[:n].cpu().numpy().tobytes())
                    assert ref.setdefault((op, n), digest) == digest, "kernels disagree at %s %d" % (op, n)
                    name = eng.last_aux_kernel_name() if op == "decrypt_l1" else eng.last_kernel_name()
                    print("%s,%s,%d,%s,%.4f,%.1f,%s" % (key, op, n, kernel, best * 1e3, n / best, name), flush=True)


if __name__ == "__main__":
    main()
