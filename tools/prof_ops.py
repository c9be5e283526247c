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
import time
cts = torch.empty(2 * n * EB, dtype=torch.uint8, device=dev)


def timed(label, units, fn, reps=2):
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("%-22s n=%d  %.2f ms  %.3e /s  (%s %.2f ms)" % (label, units, dt * 1e3, units / dt, eng.last_kernel_name(),
                                                       eng.last_kernel_ms()), flush=True)


timed("encrypt", 2 * n, lambda: eng.encrypt_dev(xs, 5, rs, 128, cts, 2 * n))
out = torch.empty(n * EB, dtype=torch.uint8, device=dev)
timed("add L1", n, lambda: eng.add_dev(1, cts[: n * EB], cts[n * EB:], out, n))
small = 1 << 10
timed("add L1 small", small, lambda: eng.add_dev(1, cts[: small * EB], cts[n * EB: (n + small) * EB], out, small))
m = min(n, 1 << 16)
timed("mult", m, lambda: eng.mult_dev(cts[: m * EB], cts[n * EB: (n + m) * EB], out, m))
# level 2 and blinded (non-deterministic mode) variants: results multiplied by Q^r resp. e(Q,Q)^r
l2 = torch.empty(m * EB, dtype=torch.uint8, device=dev)
eng.mult_dev(cts[: m * EB], cts[n * EB: (n + m) * EB], l2, m)
timed("add L2", m, lambda: eng.add_dev(2, l2, l2, out, m))
rb = rs[:n]
timed("add L1 blinded", n, lambda: eng.add_dev(1, cts[: n * EB], cts[n * EB:], out, n, rb, 128))
timed("add L2 blinded", m, lambda: eng.add_dev(2, l2, l2, out, m, rb, 128), reps=3)
timed("mult blinded", m, lambda: eng.mult_dev(cts[: m * EB], cts[n * EB: (n + m) * EB], out, m, rb, 128))
print("done", n)
