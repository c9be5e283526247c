#!/usr/bin/env python3
"""Kernel-by-kernel cost of Decrypt at one batch size (1024-bit key, T = 2^40, level 1, the bench's mixed batch), for
rocprofv3 --kernel-trace --stats:

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dec -o dec -- python3 tools/decrypt_breakdown.py 16
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
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cuda", 0)
    fx = load_fixture("k1024")
    pk = bgn_amd.PublicKey(int(fx["p"], 16), int(fx["n"], 16), fx["l"], bytes.fromhex(fx["P"]), bytes.fromhex(fx["Q"]),
                           fx["msg_space"], True, fx["poly_base"])
    eng = pk.engine
    n = 1 << lg
    xs, rs, cts = syn.config2_ciphertexts(pk, n, seed=1000, device=dev)
    pk.SetupDecryption(bgn_amd.SecretKey(int(fx["q1"], 16)))
    mixed, want, want_st = syn.decrypt_mix(pk, fx, cts, xs, dev)
    m = torch.empty(n, dtype=torch.int64, device=dev)
    st = torch.empty(n, dtype=torch.uint8, device=dev)
    eng.decrypt_dev(1, mixed, m, st, n)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        eng.decrypt_dev(1, mixed, m, st, n)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ok = bool((m.cpu() == want).all().item()) and bool((st.cpu() == want_st).all().item())
    print("decrypt 2^%d: best %.2f ms, median %.2f ms, %.4g /s, exact %s" % (lg, min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3, n / min(ts), ok))


if __name__ == "__main__":
    main()
