"""The fused level-1 Add at 2^20 (device-resident Config-2 ciphertexts) a few times — the command the rocprofv3
passes of tools/pmc_l1_add.sh profile.   python tools/l1_add_one.py [log2=20] [reps=4]"""
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
n = 1 << lg
_, _, cts = syn.config2_ciphertexts(pk, n, seed=7, device=dev)
b = syn.permuted_copy(cts, EB, seed=11)
o = torch.empty(n * EB, dtype=torch.uint8, device=dev)
ms = []
for _ in range(reps):
    eng.add_dev(1, cts, b, o, n)
    torch.cuda.synchronize()
    ms.append(eng.last_kernel_ms())
print("%s n=%d kernel_ms=%s" % (eng.last_kernel_name(), n, ["%.4f" % m for m in ms]))
