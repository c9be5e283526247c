"""Neg (one coordinate negated, no field product) on device-resident wire arrays by batch size: the one-launch
kernel k_neg_wire against decode / negate / encode, bytes compared; GB/s of the 2 * 2L algorithmic bytes per element.
   python tools/neg_sweep.py [top_log2=22]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_fixture, engine_key
import bgn_amd.synthetic as syn

fx = load_fixture("k1024")
pk, _ = engine_key(fx)
eng = pk.engine
EB = eng.elem_bytes
dev = torch.device("cuda", 0)
top = int(sys.argv[1]) if len(sys.argv) > 1 else 22
_, _, cts = syn.config2_ciphertexts(pk, 1 << 18, seed=7, device=dev)
pool = cts.view(-1, EB)
print("log2,count,route,call_ms,negs_per_s,kernel_ms,GBps_algorithmic,frac_of_8TBps,same_bytes")
for lg in range(14, top + 1):
    n = 1 << lg
    g = torch.Generator(device="cpu"); g.manual_seed(lg)
    a = pool[torch.randint(0, 1 << 18, (n,), generator=g).to(dev)].contiguous().view(-1)
    outs = {}
    for fused in (1, 0):
        eng.set_option("l1_fused", fused)
        o = torch.empty(n * EB, dtype=torch.uint8, device=dev)
        best, kms = 1e9, 0.0
        for _ in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            assert eng._lib.bgn_neg_batch_dev(eng._h, n, 1, a.data_ptr(), o.data_ptr(), eng._stream()) == 0
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            if dt < best:
                best, kms = dt, (eng.last_kernel_ms() if fused else float("nan"))
        outs[fused] = o
        gb = 2 * EB * n / (kms * 1e-3 if fused else best) / 1e9
        print("%d,%d,%s,%.4f,%.4e,%.4f,%.1f,%.3f,%s" % (lg, n, "k_neg_wire" if fused else "decode+neg+encode", best * 1e3, n / best, kms,
                                                       gb, gb / 8000, "" if fused else str(bool((outs[1] == outs[0]).all().item()))), flush=True)
    eng.set_option("l1_fused", 1)
    del a, outs
