#!/usr/bin/env python3
"""Small-batch sweep: Mult / makeL2 / level-1 Decrypt wall time by batch size on the wave-cooperative kernel
(coop/coop.hpp; makeL2 and Decrypt's lift walk the key's line table there) and on the one-pairing-per-lane kernel,
device-resident operands, best of three.  Writes the CSV the engine's crossovers
(coop_limit, engine.cpp) are chosen from:  python tools/small_batch_sweep.py > profiles/r02_small_batch.csv"""
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
    counts = [1, 8, 32, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768]
    print("key,op,count,kernel,ms,ops_per_s,kernel_name")
    dev = torch.device("cuda", 0)
    for key in ("k512", "k1024"):
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
        pk.SetupDecryption(bgn_amd.SecretKey(int(fx["q1"], 16)))
        msg = torch.empty(nmax, dtype=torch.int64, device=dev)
        sta = torch.empty(nmax, dtype=torch.uint8, device=dev)
        for op in ("mult", "make_l2", "decrypt_l1"):
            for kernel in ("coop", "lane"):
                for v in ("coop_max", "coop_max_l2", "coop_max_dec"):     # alone: the cooperative / lane kernel A/B
                    eng.set_option(v, 100000000 if kernel == "coop" else 0)
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
                    assert ref.setdefault((op, n), digest) == digest, "kernels disagree"
                    print("%s,%s,%d,%s,%.4f,%.1f,%s" % (key, op, n, kernel, best * 1e3, n / best,
                                                       eng.last_aux_kernel_name() if op == "decrypt_l1" else eng.last_kernel_name()),
                          flush=True)


if __name__ == "__main__":
    main()
