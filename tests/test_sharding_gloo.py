"""CPU: the N > 1 path with two gloo processes.  What is under test is the sharder the product uses
(bgn_amd/sharding.py: ShardedOps — slice arithmetic, ragged tails, MultPoly sharded by polynomial, gather
order, the packed Decrypt gather); the per-shard compute is stood in for by the oracle behind the Engine's
method signatures (tests may use it; tests/test_multi_gpu.py runs the same code on the HIP engine)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_fixture


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleEngine:
    """The C oracle behind the host-buffer method signatures of bgn_amd.Engine."""

    def __init__(self, fx):
        import oracle_c
        self.o = oracle_c.Oracle.from_fixture(fx)
        self.o.setup_decryption(int(fx["q1"], 16), fx["msg_space"])
        self.elem_bytes = self.o.E

    def _np(self, b):
        return np.frombuffer(b, dtype=np.uint8).reshape(-1, self.elem_bytes)

    def mult(self, a, b):
        return self._np(self.o.mult(a.tobytes(), b.tobytes()))

    def add(self, level, a, b):
        return self._np(self.o.add(level, a.tobytes(), b.tobytes()))

    def poly_mult(self, npoly, d1, d2, a, b):
        return self._np(self.o.poly_mult(npoly, d1, d2, a.tobytes(), b.tobytes()))

    def decrypt(self, level, ct):
        m, st = self.o.decrypt(level, ct.tobytes())
        return np.array(m, dtype=np.int64), np.array(st, dtype=np.uint8)


def sharded_checks(make_engine, fx, total, world, rank, dist_mod):
    """Runs Mult, Add, MultPoly (by polynomial) and Decrypt through ShardedOps on this rank and compares the
    gathered arrays with the oracle's results on the whole batch.  Shared with the GPU tests."""
    import oracle_c
    from bgn_amd.sharding import ShardedOps
    o = oracle_c.Oracle.from_fixture(fx)
    o.setup_decryption(int(fx["q1"], 16), fx["msg_space"])
    E = o.E
    cts = [bytes.fromhex(e["ct"]) for e in fx["encrypt"]]
    a = b"".join(cts[i % len(cts)] for i in range(total))
    b = b"".join(cts[(3 * i + 1) % len(cts)] for i in range(total))
    ta = torch.frombuffer(bytearray(a), dtype=torch.uint8)
    tb = torch.frombuffer(bytearray(b), dtype=torch.uint8)
    ops = ShardedOps(make_engine(), E, world, rank, dist_mod)
    ok = ops.mult(ta, tb).numpy().tobytes() == o.mult(a, b)
    ok &= ops.add(1, ta, tb).numpy().tobytes() == o.add(1, a, b)
    # MultPoly shards by polynomial (BASELINE configs[4]): the unit is one polynomial of d coefficients in and
    # 2d GT coefficients out, so every product's accumulation stays on one rank
    d = 2
    npoly = total // d
    if npoly:
        pa, pb = ta[: npoly * d * E], tb[: npoly * d * E]
        ok &= ops.poly_mult(npoly, d, d, pa, pb).numpy().tobytes() == \
            o.poly_mult(npoly, d, d, pa.numpy().tobytes(), pb.numpy().tobytes())
    m, st = ops.decrypt(1, ta)
    wm, wst = o.decrypt(1, a)
    ok &= m.tolist() == wm and st.tolist() == wst
    return bool(ok)


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bgn_amd.sharding import shard_range
    fx = load_fixture("toy64")
    ok = sharded_checks(lambda: OracleEngine(fx), fx, total, world, rank, dist)
    q.put((rank, ok, shard_range(total, world, rank)))
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [10, 7])
def test_two_rank_sharded_ops_match_single(total):
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    ranges = sorted(r for _, _, r in res)
    assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] == total


class TagEngine:
    """A stand-in behind the Engine's method signatures whose outputs are a cheap, exact function of ONE unit's
    inputs (the unit is a polynomial for poly_mult, an element otherwise) — enough to pin the index arithmetic of a
    sharded call at configs[4]'s size, which the real oracle cannot reach in seconds: a slice that starts one unit
    early or late, or a gather in the wrong order, changes bytes."""

    def __init__(self, elem_bytes):
        self.elem_bytes = elem_bytes

    def poly_mult(self, npoly, d1, d2, a, b):
        E = self.elem_bytes
        pa = np.asarray(a, dtype=np.uint8).reshape(npoly, d1 * E).astype(np.uint32)
        pb = np.asarray(b, dtype=np.uint8).reshape(npoly, d2 * E).astype(np.uint32)
        wa = np.arange(1, d1 * E + 1, dtype=np.uint32)
        wb = np.arange(3, 3 + d2 * E, dtype=np.uint32)
        tag = (pa * wa).sum(axis=1) * np.uint32(2654435761) + (pb * wb).sum(axis=1)      # one word per polynomial
        out = np.empty((npoly, (d1 + d2) * E), dtype=np.uint8)
        k = np.arange((d1 + d2) * E, dtype=np.uint32)
        out[:] = ((tag[:, None] >> (k[None, :] % 24)) + k[None, :]).astype(np.uint8)
        out[:, : d1 * E] ^= pa.astype(np.uint8)                                          # and every input byte
        out[:, d1 * E: (d1 + d2) * E] ^= pb.astype(np.uint8)
        return out.reshape(-1, E)

    def mult(self, a, b):
        n = np.asarray(a).size // self.elem_bytes
        return self.poly_mult(n, 1, 1, a, b).reshape(n, 2, self.elem_bytes)[:, 0, :]


def _worker8(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bgn_amd.sharding import ShardedOps, shard_range
    ok = True
    # (1) the real oracle behind the sharder at world 8: Mult, Add, MultPoly by polynomial, the packed Decrypt gather,
    #     with fewer units than ranks in one case (empty shards) and a ragged total in the other
    fx = load_fixture("toy64")
    for total in (5, 37):
        ok &= sharded_checks(lambda: OracleEngine(fx), fx, total, world, rank, dist)
    # (2) configs[4]'s index arithmetic: 2^14 polynomials of 16x16 coefficients split by polynomial over 8 ranks
    #     (2048 each: the equal-shard all-gather), and 2^14 + 5 (ragged: five ranks own 2049, the padded gather)
    E, d = 16, 16
    tag = TagEngine(E)
    ranges = {}
    for npoly in (1 << 14, (1 << 14) + 5):
        g = torch.Generator().manual_seed(npoly)
        a = torch.randint(0, 256, (npoly * d * E,), dtype=torch.uint8, generator=g)
        b = torch.randint(0, 256, (npoly * d * E,), dtype=torch.uint8, generator=g)
        ops = ShardedOps(tag, E, world, rank, dist)
        got = ops.poly_mult(npoly, d, d, a, b)
        want = tag.poly_mult(npoly, d, d, a.numpy(), b.numpy()).reshape(-1)
        ok &= got.numel() == npoly * 2 * d * E and bool((got.numpy() == want).all())
        lo, hi = shard_range(npoly, world, rank)
        # this rank's slice of the coefficient-pair index space is whole polynomials: [lo*d*d, hi*d*d)
        ok &= (hi - lo) in (npoly // world, npoly // world + 1)
        ranges[npoly] = (lo, hi)
    q.put((rank, bool(ok), ranges))
    dist.destroy_process_group()


def test_eight_rank_sharding_configs4_index_arithmetic():
    """World 8 over gloo (CPU): what the driver's 8-GPU run will do to configs[4] — 2^14 MultPoly instances split by
    polynomial, and the ragged 2^14 + 5 — plus the oracle-backed checks of every sharded operation at world 8."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 8
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    for npoly in (1 << 14, (1 << 14) + 5):
        rs = sorted(r[npoly] for _, _, r in res)
        assert rs[0][0] == 0 and rs[-1][1] == npoly and all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
        assert sum(h - l for l, h in rs) * 256 == npoly * 256            # every coefficient pair owned exactly once


def test_shard_range_properties():
    import ctypes as C
    from bgn_amd import _lib
    from bgn_amd.sharding import shard_range
    lib = _lib.load()
    for total in [0, 1, 5, 64, 1 << 20, (1 << 22) + 3]:
        for world in [1, 2, 3, 4, 8]:
            rs = [shard_range(total, world, r) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == total
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [h - l for l, h in rs]
            assert max(sizes) - min(sizes) <= 1
            for r in range(world):          # the C ABI's split is the same one
                lo, hi = C.c_size_t(), C.c_size_t()
                lib.bgn_shard_range(total, world, r, C.byref(lo), C.byref(hi))
                assert (lo.value, hi.value) == rs[r]
    lo, hi = C.c_size_t(7), C.c_size_t(7)
    lib.bgn_shard_range(10, 0, 0, C.byref(lo), C.byref(hi))     # bad world: empty range
    assert (lo.value, hi.value) == (0, 0)


def test_mctx_rejects_bad_device_lists_before_touching_the_gpu():
    import ctypes as C
    from bgn_amd import _lib
    lib = _lib.load()
    fx = load_fixture("toy64")
    pb = int(fx["p"], 16).to_bytes(fx["fp_bytes"], "big")
    nb = int(fx["n"], 16).to_bytes(8, "big")
    h = C.c_void_p()
    rc = lib.bgn_mctx_create(C.byref(h), pb, len(pb), nb, len(nb), fx["l"], bytes.fromhex(fx["P"]),
                             bytes.fromhex(fx["Q"]), 1, None, 0)
    assert rc == -1 and not h.value and b"device list" in lib.bgn_last_error()
    if not torch.cuda.is_available():       # no CPU fallback: creation fails loudly without a device
        devs = (C.c_int * 2)(0, 0)
        rc = lib.bgn_mctx_create(C.byref(h), pb, len(pb), nb, len(nb), fx["l"], bytes.fromhex(fx["P"]),
                                 bytes.fromhex(fx["Q"]), 1, devs, 2)
        assert rc == -3 and not h.value
