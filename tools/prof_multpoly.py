"""MultPoly 16x16 over 4096 polynomials, twice (for rocprofv3 --kernel-trace --stats and wall-clock A/B)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from conftest import load_fixture, engine_key
fx = load_fixture("k1024")
pk, sk = engine_key(fx)
eng = pk.engine
dev = torch.device("cuda")
EB = eng.elem_bytes
npoly = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = 2 * npoly * d
g = torch.Generator().manual_seed(1)
xs = torch.randint(0, 256, (n, 5), dtype=torch.uint8, generator=g).to(dev)
rs = torch.randint(0, 256, (n, 128), dtype=torch.uint8, generator=g); rs[:, 0] &= 0x3F; rs = rs.to(dev)
cts = torch.empty(n * EB, dtype=torch.uint8, device=dev)
eng.encrypt_dev(xs, 5, rs, 128, cts, n)
pa, pb = cts[: npoly * d * EB], cts[npoly * d * EB:]
po = torch.empty(npoly * 2 * d * EB, dtype=torch.uint8, device=dev)
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.poly_mult_dev(npoly, d, d, pa, pb, po)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("multpoly %dx%d npoly=%d: %.1f ms wall, %.3e pairs/s, events %.1f ms" % (d, d, npoly, dt * 1e3, npoly * d * d / dt, eng.last_kernel_ms()), flush=True)
