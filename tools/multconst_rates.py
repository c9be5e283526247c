import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, ctypes as C
from conftest import load_fixture, engine_key
from bgn_amd._lib import check
fx = load_fixture("k1024"); pk, sk = engine_key(fx); eng = pk.engine
n = 1 << 16
rng = np.random.default_rng(1)
xs = rng.integers(0, 256, (n, 5), dtype=np.uint8); rs = rng.integers(0, 256, (n, 128), dtype=np.uint8); rs[:,0] &= 0x3F
P = lambda a: a.ctypes.data_as(C.c_void_p)
cts = np.zeros((n, eng.elem_bytes), dtype=np.uint8)
check(eng._lib.bgn_encrypt_batch(eng._h, n, P(xs), 5, P(rs), 128, P(cts)), "enc")
out = np.zeros_like(cts)
for klen in (8, 32, 128):
    k = rng.integers(0, 256, (n, klen), dtype=np.uint8)
    for lvl, src in ((1, cts),):
        for _ in range(2):
            t0 = time.perf_counter(); check(eng._lib.bgn_multconst_batch(eng._h, n, lvl, P(src), P(k), klen, None, 0, P(out)), "mc"); dt = time.perf_counter() - t0
        print("multconst L%d klen=%d: %.1f ms %.3e /s" % (lvl, klen, dt*1e3, n/dt), flush=True)
