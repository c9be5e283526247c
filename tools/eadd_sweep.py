"""EAdd (level 1) on device-resident wire arrays by batch size: the run length per lane — and with it the share of
the batched inversion in every addition — follows the batch (engine.cpp run_for: ceil(count / 65536), at most 64)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_fixture, engine_key
import bgn_amd.synthetic as syn

fx = load_fixture("k1024")
pk, sk = engine_key(fx)
eng = pk.engine
EB = eng.elem_bytes
dev = torch.device("cuda", 0)
top = int(sys.argv[1]) if len(sys.argv) > 1 else 22
_, _, cts = syn.config2_ciphertexts(pk, 1 << 20, seed=7, device=dev)
pool = cts.view(-1, EB)
print("log2,count,run,ms,adds_per_s,k_g1_add_ms,products_per_add,frac_of_one_wave_mad_ceiling,four_launch_ms,same_bytes")
for lg in range(14, top + 1):
    n = 1 << lg
    g = torch.Generator(device="cpu"); g.manual_seed(lg)
    ia = torch.randint(0, 1 << 20, (n,), generator=g).to(dev)
    ib = torch.randint(0, 1 << 20, (n,), generator=g).to(dev)
    a = pool[ia].contiguous().view(-1); b = pool[ib].contiguous().view(-1)
    o = torch.empty(n * EB, dtype=torch.uint8, device=dev)
    best, kms = 1e9, 0.0
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.add_dev(1, a, b, o, n)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if dt < best:
            best, kms = dt, eng.last_kernel_ms()
    ppa = syn.eadd_products(n)
    # the four-launch route (decode, decode, k_g1_add, encode) on the same arrays
    o4 = torch.empty(n * EB, dtype=torch.uint8, device=dev)
    eng.set_option("l1_fused", 0)
    best4 = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.add_dev(1, a, b, o4, n)
        torch.cuda.synchronize(); best4 = min(best4, time.perf_counter() - t0)
    eng.set_option("l1_fused", 1)
    same = bool(torch.equal(o, o4))
    del o4
    print("%d,%d,%d,%.3f,%.4e,%.3f,%.2f,%.3f,%.3f,%s" % (lg, n, max(1, min(64, -(-n // 65536))), best * 1e3, n / best, kms, ppa,
                                                      n / best * syn.mads_from_counts(*syn.eadd_counts(n), nl=36) / 2.55e13, best4 * 1e3, same), flush=True)
    del a, b, o, ia, ib
