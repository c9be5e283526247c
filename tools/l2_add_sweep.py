"""EAdd / ESub on level-2 ciphertexts (device-resident wire arrays) by batch size: the one-launch route
(k_gt_mul_wire, option l2_fused = 1) against the decode / decode / k_gt_mul / encode pipeline it replaces, on the
same operands — products of Config-2 ciphertexts — with the bytes of the two routes compared.
   python tools/l2_add_sweep.py [top_log2=22]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_fixture, engine_key
import bgn_amd.synthetic as syn

fx = load_fixture(os.environ.get("L2_KEY", "k1024"))
pk, sk = engine_key(fx)
eng = pk.engine
EB = eng.elem_bytes
dev = torch.device("cuda", 0)
top = int(sys.argv[1]) if len(sys.argv) > 1 else 22
npool = 1 << 16
_, _, cts = syn.config2_ciphertexts(pk, npool, seed=7, device=dev)
prod = torch.empty(npool * EB, dtype=torch.uint8, device=dev)
eng.mult_dev(cts, syn.permuted_copy(cts, EB, seed=5), prod, npool)
pool = prod.view(-1, EB)
print("log2,count,route,call_ms,adds_per_s,kernel_ms,GBps_algorithmic,same_bytes")
for lg in range(12, top + 1):
    n = 1 << lg
    g = torch.Generator(device="cpu"); g.manual_seed(lg)
    ia = torch.randint(0, npool, (n,), generator=g).to(dev)
    ib = torch.randint(0, npool, (n,), generator=g).to(dev)
    a = pool[ia].contiguous().view(-1); b = pool[ib].contiguous().view(-1)
    outs = {}
    for fused in (1, 0):
        eng.set_option("l2_fused", fused)
        o = torch.empty(n * EB, dtype=torch.uint8, device=dev)
        best, kms = 1e9, 0.0
        for _ in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.add_dev(2, a, b, o, n)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            if dt < best:
                best, kms = dt, eng.last_kernel_ms()
        outs[fused] = o
        same = "" if fused else str(bool((outs[1] == outs[0]).all().item()))
        print("%d,%d,%s,%.4f,%.4e,%.4f,%.1f,%s" % (lg, n, eng.last_kernel_name(), best * 1e3, n / best, kms,
                                                  3 * EB * n / best / 1e9, same), flush=True)
    eng.set_option("l2_fused", 1)
    del a, b, outs, ia, ib
