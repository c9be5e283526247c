"""ctypes wrapper of the C oracle (oracle/bgn_oracle.c).

TEST INFRASTRUCTURE ONLY — see the header of bgn_oracle.c.  Used by tests/ as
the fast checker (sizes the pure-Python oracle cannot reach) and by bench.py's
cpu_baseline leg as the reported CPU baseline ("kind": "port").
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import time

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_build", "libbgn_oracle.so")
_lib = None


def available() -> bool:
    return os.path.exists(_PATH)


def _load():
    global _lib
    if _lib is None:
        lib = C.CDLL(_PATH)
        vp, sz, u64 = C.c_void_p, C.c_size_t, C.c_uint64
        lib.orc_create.restype = vp
        lib.orc_create.argtypes = [vp, sz, vp, sz, u64, vp, vp]
        lib.orc_destroy.argtypes = [vp]
        lib.orc_fp_bytes.argtypes = [vp]
        lib.orc_encrypt.argtypes = [vp, sz, vp, sz, vp, sz, vp]
        lib.orc_add.argtypes = [vp, sz, C.c_int, C.c_int, vp, vp, vp]
        lib.orc_mult.argtypes = [vp, sz, vp, vp, vp]
        lib.orc_multconst.argtypes = [vp, sz, C.c_int, vp, vp, sz, vp]
        lib.orc_set_secret.argtypes = [vp, vp, sz]
        lib.orc_setup_decryption.argtypes = [vp, u64]
        lib.orc_setup_decryption_gt.argtypes = [vp, u64]
        lib.orc_decrypt.argtypes = [vp, sz, C.c_int, vp, vp, vp]
        lib.orc_poly_mult.argtypes = [vp, sz, sz, sz, vp, vp, vp]
        _lib = lib
    return _lib


def _ib(v: int) -> bytes:
    return int(v).to_bytes(max(1, (int(v).bit_length() + 7) // 8), "big")


def _pack(vals, length=None) -> tuple:
    vals = [int(v) for v in vals]
    if length is None:
        length = max(1, max(((v.bit_length() + 7) // 8 for v in vals), default=1))
    return b"".join(v.to_bytes(length, "big") for v in vals), length


class Oracle:
    def __init__(self, p: int, n: int, l: int, P_wire: bytes, Q_wire: bytes):
        self.lib = _load()
        pb, nb = _ib(p), _ib(n)
        self.h = self.lib.orc_create(pb, len(pb), nb, len(nb), l, bytes(P_wire), bytes(Q_wire))
        if not self.h:
            raise RuntimeError("orc_create failed")
        self.L = self.lib.orc_fp_bytes(self.h)
        self.E = 2 * self.L

    @classmethod
    def from_fixture(cls, fx) -> "Oracle":
        return cls(int(fx["p"], 16), int(fx["n"], 16), fx["l"], bytes.fromhex(fx["P"]), bytes.fromhex(fx["Q"]))

    def __del__(self):
        try:
            self.lib.orc_destroy(self.h)
        except Exception:
            pass

    def encrypt(self, xs, rs=None) -> bytes:
        xb, xl = _pack(xs)
        out = C.create_string_buffer(len(xs) * self.E)
        if rs is None:
            self.lib.orc_encrypt(self.h, len(xs), xb, xl, None, 0, out)
        else:
            rb, rl = _pack(rs)
            self.lib.orc_encrypt(self.h, len(xs), xb, xl, rb, rl, out)
        return out.raw

    def add(self, level: int, a: bytes, b: bytes, subtract: bool = False) -> bytes:
        n = len(a) // self.E
        out = C.create_string_buffer(n * self.E)
        self.lib.orc_add(self.h, n, level, 1 if subtract else 0, bytes(a), bytes(b), out)
        return out.raw

    def mult(self, a: bytes, b: bytes = None) -> bytes:
        n = len(a) // self.E
        out = C.create_string_buffer(n * self.E)
        self.lib.orc_mult(self.h, n, bytes(a), bytes(b) if b is not None else None, out)
        return out.raw

    def multconst(self, level: int, a: bytes, ks) -> bytes:
        n = len(a) // self.E
        kb, kl = _pack(ks)
        out = C.create_string_buffer(n * self.E)
        self.lib.orc_multconst(self.h, n, level, bytes(a), kb, kl, out)
        return out.raw

    def setup_decryption(self, q1: int, T: int) -> None:
        qb = _ib(q1)
        self.lib.orc_set_secret(self.h, qb, len(qb))
        rc = self.lib.orc_setup_decryption(self.h, T)
        if rc:
            raise RuntimeError(f"orc_setup_decryption -> {rc}")

    def setup_decryption_gt(self, q1: int, T: int) -> None:
        """The GT table only (computeTableGT, gsbs.go:28-37): level-2 decrypt at message spaces whose G1 table this
        port cannot build in bounded time (one Fermat inversion per entry)."""
        qb = _ib(q1)
        self.lib.orc_set_secret(self.h, qb, len(qb))
        rc = self.lib.orc_setup_decryption_gt(self.h, T)
        if rc:
            raise RuntimeError(f"orc_setup_decryption_gt -> {rc}")

    def decrypt(self, level: int, ct: bytes):
        n = len(ct) // self.E
        m = (C.c_int64 * n)()
        st = (C.c_uint8 * n)()
        rc = self.lib.orc_decrypt(self.h, n, level, bytes(ct), m, st)
        if rc:
            raise RuntimeError(f"orc_decrypt -> {rc}")
        return list(m), list(st)

    def poly_mult(self, npoly: int, d1: int, d2: int, a: bytes, b: bytes) -> bytes:
        out = C.create_string_buffer(npoly * (d1 + d2) * self.E)
        self.lib.orc_poly_mult(self.h, npoly, d1, d2, bytes(a), bytes(b), out)
        return out.raw


def host_cores() -> dict:
    """What this process may use of the host: `cores` = min(CPU affinity, cgroup CPU quota rounded up) — the number
    bench.py reports as cpu_baseline.cores and the number of worker threads the legs below start."""
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    quota = None
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()
        if txt[0] != "max":
            quota = float(txt[0]) / float(txt[1])
    except (OSError, ValueError, IndexError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            if q > 0:
                quota = q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except (OSError, ValueError):
            pass
    cores = aff if quota is None else max(1, min(aff, int(-(-quota // 1))))
    return {"cores": cores, "affinity": aff, "cgroup_cpus": quota}


def _run_threads(work, threads: int) -> float:
    """work(i) on `threads` worker threads (ctypes releases the GIL inside the C calls); wall seconds."""
    th = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    t0 = time.time()
    for t in th:
        t.start()
    for t in th:
        t.join()
    return time.time() - t0


def bench_pairings(fx, a_host: bytes, b_host: bytes, gpu_out_host: bytes, seconds: float = 12.0) -> dict:
    """Time orc_mult on the first pairs of the GPU's own batch, one worker thread
    per usable host core over disjoint slices (ctypes releases the GIL), and compare
    the outputs with the GPU's byte for byte."""
    orc = Oracle.from_fixture(fx)
    E = orc.E
    npairs = len(a_host) // E
    hc = host_cores()
    threads = hc["cores"]
    # calibrate on one pairing, then size the sample for ~`seconds` of wall time
    t0 = time.time()
    first = orc.mult(a_host[:E], b_host[:E])
    t1 = max(time.time() - t0, 1e-4)
    per_thread = max(1, min(npairs // threads, int(seconds / t1)))
    total = per_thread * threads
    outs = [None] * threads

    def work(i):
        lo, hi = i * per_thread * E, (i + 1) * per_thread * E
        outs[i] = Oracle.from_fixture(fx).mult(a_host[lo:hi], b_host[lo:hi])

    dt = _run_threads(work, threads)
    got = b"".join(outs)
    ok = got == gpu_out_host[: total * E] and first == gpu_out_host[:E]
    return {"value": total / dt, "unit": "pairings/s", "cores": hc["cores"], "threads": threads, "kind": "port",
            "single_thread_pairings_per_s": 1.0 / t1,
            "sample": f"first {total} pairs of the GPU batch, C restatement oracle/bgn_oracle.c "
                      f"(64-bit limbs, unsigned __int128 CIOS Montgomery, projective Miller loop), "
                      f"{threads} threads x {per_thread} pairings, {dt:.1f} s",
            "matches_gpu_bit_exact": bool(ok)}


def bench_slices(fx, n_items: int, call, want: bytes, out_bytes_per_item: int, unit: str, what: str,
                 seconds: float = 8.0, calibrate: int = 1, max_per_thread: int = 1 << 20) -> dict:
    """The shape shared by bench.py's Encrypt / EAdd / MultPoly CPU legs: `call(orc, lo, hi)` returns the oracle's
    output bytes for items [lo, hi) of the GPU's own batch.  One call of `calibrate` items on one thread gives the
    single-thread rate and sizes the sample (about `seconds` of wall time on every usable core, one thread per
    core over disjoint slices); every output byte is compared with the GPU's (`want`, the same items)."""
    hc = host_cores()
    threads = hc["cores"]
    orc = Oracle.from_fixture(fx)
    calibrate = max(1, min(calibrate, n_items))
    t0 = time.time()
    first = call(orc, 0, calibrate)
    t1 = max(time.time() - t0, 1e-5) / calibrate
    per_thread = max(1, min(n_items // threads, int(seconds / t1), max_per_thread))
    total = per_thread * threads
    outs = [None] * threads

    def work(i):
        outs[i] = call(Oracle.from_fixture(fx), i * per_thread, (i + 1) * per_thread)

    dt = _run_threads(work, threads)
    got = b"".join(outs)
    ok = got == want[: total * out_bytes_per_item] and first == want[: calibrate * out_bytes_per_item]
    return {"value": total / dt, "unit": unit, "cores": hc["cores"], "threads": threads, "kind": "port",
            "single_thread_per_s": 1.0 / t1,
            "sample": f"first {total} {what} of the GPU batch, C restatement oracle/bgn_oracle.c, {threads} threads x "
                      f"{per_thread}, {dt:.1f} s; single-thread figure from {calibrate} of them",
            "matches_gpu_bit_exact": bool(ok)}


def bench_decrypt(fx, ct_host: bytes, want_m, want_status, T: int, max_threads: int = 0) -> dict:
    """bench.py's CPU leg for the second half of BASELINE's metric (BSGS decrypts/s): a bounded sample of the GPU's own
    mixed batch (positives, negatives, out-of-range) decrypted by the reference's algorithm on the host cores.

    The reference decrypts a level-1 ciphertext with a G1 walk (gsbs.go:77-103: one affine subtraction, i.e. one
    field inversion, per giant step, up to 2^20 of them at T = 2^40) — minutes per ciphertext in this plain-C port.
    The leg therefore takes the route that FAVOURS the CPU: lift each sampled ciphertext to level 2 (makeL2 =
    Pair(c, P), bgn.go:316-321, one pairing) and run the reference's level-2 decrypt (power by the secret key, GT
    giant steps of one F_p^2 product each, negative retry, bgn.go:225-242) — the same lift the GPU path makes.  One
    thread per sampled ciphertext over a shared table (read-only); the table build (computeTableGT, 2^20 + 2
    entries) is timed separately, like SetupDecryption on the GPU side."""
    orc = Oracle.from_fixture(fx)
    E = orc.E
    n = len(ct_host) // E
    t0 = time.time()
    orc.setup_decryption_gt(int(fx["q1"], 16), T)
    t_setup = time.time() - t0
    hc = host_cores()
    threads = max(1, min(n, max_threads or hc["cores"]))
    per = (n + threads - 1) // threads
    res = [None] * threads

    def work(i):
        lo, hi = i * per, min(n, (i + 1) * per)
        if lo >= hi:
            res[i] = ([], [])
            return
        l2 = orc.mult(ct_host[lo * E:hi * E])           # makeL2
        res[i] = orc.decrypt(2, l2)

    dt = _run_threads(work, threads)
    m = [v for r in res for v in r[0]]
    st = [v for r in res for v in r[1]]
    ok = m == [int(v) for v in want_m] and st == [int(v) for v in want_status]
    return {"value": n / dt, "unit": "decrypts/s", "cores": hc["cores"], "threads": threads, "kind": "port",
            "sample": f"{n} ciphertexts strided over the GPU's mixed batch ({sum(1 for v in want_m if v < 0)} negative, "
                      f"{sum(1 for v in want_status if v)} out of range), lifted to level 2 (one pairing) and decrypted by "
                      f"the reference's level-2 getDL (up to {int(-(-T ** 0.5 // 1))} GT giant steps) in oracle/bgn_oracle.c — cheaper than "
                      f"the reference's own level-1 G1 walk; {threads} threads, {dt:.1f} s; table build "
                      f"{t_setup:.1f} s not included",
            "table_setup_s": t_setup, "matches_gpu_exact": bool(ok)}
