"""The fused level-2 Add at 2^20 (device-resident operands: products of Config-2 ciphertexts) a few times — the
command the rocprofv3 passes of tools/pmc_l2_add.sh profile.   python tools/l2_add_one.py [log2=20] [reps=4]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_fixture, engine_key
import bgn_amd.synthetic as syn

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
fx = load_fixture("k1024")
pk, _ = engine_key(fx)
eng = pk.engine
EB = eng.elem_bytes
dev = torch.device("cuda", 0)
npool = 1 << 16
_, _, cts = syn.config2_ciphertexts(pk, npool, seed=7, device=dev)
prod = torch.empty(npool * EB, dtype=torch.uint8, device=dev)
eng.mult_dev(cts, syn.permuted_copy(cts, EB, seed=5), prod, npool)
pool = prod.view(-1, EB)
n = 1 << lg
g = torch.Generator(device="cpu"); g.manual_seed(lg)
a = pool[torch.randint(0, npool, (n,), generator=g).to(dev)].contiguous().view(-1)
b = pool[torch.randint(0, npool, (n,), generator=g).to(dev)].contiguous().view(-1)
o = torch.empty(n * EB, dtype=torch.uint8, device=dev)
ms = []
for _ in range(reps):
    eng.add_dev(2, a, b, o, n)
    torch.cuda.synchronize()
    ms.append(eng.last_kernel_ms())
print("%s n=%d kernel_ms=%s" % (eng.last_kernel_name(), n, ["%.4f" % m for m in ms]))
