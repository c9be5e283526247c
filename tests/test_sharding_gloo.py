"""CPU: the N > 1 path (contiguous batch sharding + gather of results) with two gloo
processes.  The per-shard compute is stood in for by the oracle (tests may use it);
what is under test is bgn_amd/sharding.py: slice arithmetic, ragged tails, gather order."""
import os
import socket
import sys

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


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_c
    from bgn_amd.sharding import shard_range, sharded_apply
    fx = load_fixture("toy64")
    o = oracle_c.Oracle.from_fixture(fx)
    E = o.E
    cts = [bytes.fromhex(e["ct"]) for e in fx["encrypt"]]
    a = b"".join(cts[i % len(cts)] for i in range(total))
    b = b"".join(cts[(3 * i + 1) % len(cts)] for i in range(total))
    ta = torch.frombuffer(bytearray(a), dtype=torch.uint8)
    tb = torch.frombuffer(bytearray(b), dtype=torch.uint8)

    def op(sa, sb):   # this rank's shard of pk.Mult
        return torch.frombuffer(bytearray(o.mult(sa.numpy().tobytes(), sb.numpy().tobytes())), dtype=torch.uint8)

    got = sharded_apply(op, total, E, E, [ta, tb], world, rank, dist)
    full = o.mult(a, b)
    ok = got.numpy().tobytes() == full
    # MultPoly shards by polynomial (BASELINE configs[4]): the unit is one polynomial of d coefficients in and
    # 2d GT coefficients out, so every product's accumulation stays on one rank
    d = 2
    npoly = total // d
    if npoly:
        pa, pb = ta[: npoly * d * E], tb[: npoly * d * E]

        def pop(sa, sb):
            n_here = sa.numel() // (d * E)
            return torch.frombuffer(bytearray(o.poly_mult(n_here, d, d, sa.numpy().tobytes(), sb.numpy().tobytes())),
                                    dtype=torch.uint8)

        gotp = sharded_apply(pop, npoly, d * E, 2 * d * E, [pa, pb], world, rank, dist)
        ok = ok and gotp.numpy().tobytes() == o.poly_mult(npoly, d, d, pa.numpy().tobytes(), pb.numpy().tobytes())
    lo, hi = shard_range(total, world, rank)
    q.put((rank, ok, (lo, hi)))
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [10, 7])
def test_two_rank_sharded_mult_matches_single(total):
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


def test_shard_range_properties():
    from bgn_amd.sharding import shard_range
    for total in [0, 1, 5, 64, 1 << 20, (1 << 22) + 3]:
        for world in [1, 2, 4, 8]:
            rs = [shard_range(total, world, r) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == total
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [h - l for l, h in rs]
            assert max(sizes) - min(sizes) <= 1
