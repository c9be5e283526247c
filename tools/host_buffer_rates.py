"""Host-buffer (PCIe-inclusive) rates of the no-suffix entry points at the 1024-bit key: what a cgo caller with
Go-owned slices sees."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import load_fixture, engine_key
import bgn_amd
fx = load_fixture("k1024")
pk, sk = engine_key(fx)
eng = pk.engine
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
rng = np.random.default_rng(1)
xs = rng.integers(0, 256, (n, 5), dtype=np.uint8)
rs = rng.integers(0, 256, (n, 128), dtype=np.uint8); rs[:, 0] &= 0x3F
import ctypes as C
from bgn_amd._lib import check
out = np.zeros((n, eng.elem_bytes), dtype=np.uint8)
P = lambda a: a.ctypes.data_as(C.c_void_p)


def timed(label, fn, units, reps=3):
    """Each measurement with the one-shot staging path (option host_pipe = 0) and with the chunked pipeline."""
    for pipe in ("0", "1"):
        eng.set_option("host_pipe", int(pipe))
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t0)
        print("%-16s pipeline=%s n=%d  %.1f ms  %.3e /s" % (label, pipe, units, best * 1e3, units / best), flush=True)


timed("encrypt", lambda: check(eng._lib.bgn_encrypt_batch(eng._h, n, P(xs), 5, P(rs), 128, P(out)), "enc"), n)
a, b = out[: n // 2].copy(), out[n // 2:].copy()
o2 = np.zeros_like(a)
timed("add L1", lambda: check(eng._lib.bgn_add_batch(eng._h, n // 2, 1, P(a), P(b), None, 0, P(o2)), "add"), n // 2)


def pinned(shape):
    """numpy view of page-locked memory from bgn_host_alloc (what a Go caller would wrap with unsafe.Slice)."""
    nbytes = int(np.prod(shape))
    ptr = eng._lib.bgn_host_alloc(nbytes)
    assert ptr
    return np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr)).reshape(shape)


pa, pb, po = pinned(a.shape), pinned(b.shape), pinned(o2.shape)
pa[:], pb[:] = a, b
timed("add L1 pinned", lambda: check(eng._lib.bgn_add_batch(eng._h, n // 2, 1, P(pa), P(pb), None, 0, P(po)), "add"), n // 2)
assert (po == o2).all()
px, pr, pout = pinned(xs.shape), pinned(rs.shape), pinned(out.shape)
px[:], pr[:] = xs, rs
timed("encrypt pinned", lambda: check(eng._lib.bgn_encrypt_batch(eng._h, n, P(px), 5, P(pr), 128, P(pout)), "enc"), n)
assert (pout == out).all()
m = min(n // 2, 1 << 16)
timed("mult", lambda: check(eng._lib.bgn_mult_batch(eng._h, m, P(a), P(b), None, 0, P(o2)), "mult"), m)
pk.SetupDecryption(sk)
mm = np.zeros(m, dtype=np.int64); st = np.zeros(m, dtype=np.uint8)
timed("decrypt L1", lambda: check(eng._lib.bgn_decrypt_batch(eng._h, m, 1, P(out), P(mm), P(st)), "dec"), m)
