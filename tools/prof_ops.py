"""Launch each batch entry point once at a given size (for rocprofv3 --kernel-trace --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from conftest import load_fixture, engine_key
import bgn_amd
fx = load_fixture("k1024")
pk, sk = engine_key(fx)
eng = pk.engine
dev = torch.device("cuda")
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 19
EB = eng.elem_bytes
g = torch.Generator().manual_seed(1)
xs = torch.randint(0, 256, (2 * n, 5), dtype=torch.uint8, generator=g).to(dev)
rs = torch.randint(0, 256, (2 * n, 128), dtype=torch.uint8, generator=g); rs[:, 0] &= 0x3F; rs = rs.to(dev)
cts = torch.empty(2 * n * EB, dtype=torch.uint8, device=dev)
eng.encrypt_dev(xs, 5, rs, 128, cts, 2 * n)
out = torch.empty(n * EB, dtype=torch.uint8, device=dev)
for _ in range(2):
    eng.add_dev(1, cts[: n * EB], cts[n * EB:], out, n)
torch.cuda.synchronize()
print("done", n)
