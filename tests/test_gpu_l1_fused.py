"""GPU: level-1 Add / Sub in one wire-to-wire launch (k_g1_add_wire: the affine additions of g1_add_run with the
dword-stream codec inside, one staged slice per step of a lane's run) — bgn.go:477-483 (Add, level-1 branch),
:414-420 (Sub).  Against the golden vectors, the C oracle and the four-launch route it replaces."""
import random

import numpy as np
import pytest

from conftest import KEYS, engine_key, load_fixture

pytestmark = pytest.mark.gpu


def H(hexes):
    return b"".join(bytes.fromhex(h) for h in hexes)


@pytest.mark.parametrize("name", KEYS + ["k1024b", "k2048"])
def test_l1_add_sub_golden_through_the_fused_kernel(name):
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    eng = pk.engine
    cts = [e["ct"] for e in fx["encrypt"]]
    a, b = H([cts[v["a"]] for v in fx["l1"]]), H([cts[v["b"]] for v in fx["l1"]])
    for fn, key in [(eng.add, "add"), (eng.sub, "sub")]:
        got = fn(1, a, b)
        assert eng.last_kernel_name() == "k_g1_add_wire"
        for row, v in zip(got, fx["l1"]):
            assert bytes(row).hex() == v[key], f"{name}: L1 {key}({v['a']},{v['b']})"


@pytest.mark.parametrize("name,count", [("toy64", 70001), ("toy64", 150001), ("k256", 3000), ("k1024", 700), ("k1024b", 300)])
def test_l1_fused_ragged_runs_vs_c_oracle_and_the_four_launch_route(name, count):
    """Several workgroups, a ragged tail, runs of more than one element per lane (count > 65536: element j*T + t of a
    lane's run is staged with its workgroup's slice at step j), identities, doublings (a + a) and cancellations
    (a - a) sprinkled in: bytes equal to the C oracle's and to the four-launch route's."""
    import oracle_c
    fx = load_fixture(name)
    o = oracle_c.Oracle.from_fixture(fx)
    pk, _ = engine_key(fx)
    eng = pk.engine
    EB = eng.elem_bytes
    rng = random.Random(45)
    n = int(fx["n"], 16)
    pool = [o.encrypt([rng.randrange(1 << 30)], [rng.randrange(n)]) for _ in range(9)]
    pool.append(bytes(EB))                                   # identity
    P = np.frombuffer(b"".join(pool), dtype=np.uint8).reshape(len(pool), EB)
    ia = np.array([rng.randrange(len(pool)) for _ in range(count)])
    ib = np.array([rng.randrange(len(pool)) for _ in range(count)])
    a, b = P[ia].reshape(-1).tobytes(), P[ib].reshape(-1).tobytes()
    want = {(i, k): (o.add(1, pool[i], pool[k]), o.add(1, pool[i], pool[k], True))
            for i in range(len(pool)) for k in range(len(pool))}
    wa = b"".join(want[(int(i), int(k))][0] for i, k in zip(ia, ib))
    ws = b"".join(want[(int(i), int(k))][1] for i, k in zip(ia, ib))
    assert eng.add(1, a, b).tobytes() == wa and eng.last_kernel_name() == "k_g1_add_wire"
    assert eng.sub(1, a, b).tobytes() == ws
    eng.set_option("l1_fused", 0)
    try:
        assert eng.add(1, a, b).tobytes() == wa and eng.last_kernel_name() == "k_g1_add"
        assert eng.sub(1, a, b).tobytes() == ws
    finally:
        eng.set_option("l1_fused", 1)


@pytest.mark.parametrize("name", ["k256", "k1024"])
def test_l1_fused_on_misaligned_device_buffers(name):
    """Operand arrays that start 1, 2 and 3 bytes into a dword are staged at their own misalignment; a result array
    that does not start on a dword takes the four-launch route."""
    import torch
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    eng = pk.engine
    EB = eng.elem_bytes
    cts = [bytes.fromhex(e["ct"]) for e in fx["encrypt"]]
    n = 517
    a = b"".join(cts[i % len(cts)] for i in range(n))
    b = b"".join(cts[(3 * i + 1) % len(cts)] for i in range(n))
    want = eng.add(1, a, b).tobytes()
    dev = torch.device("cuda", 0)
    for ma, mb, mo in [(1, 0, 0), (0, 2, 0), (3, 1, 0), (2, 3, 0), (0, 0, 3), (3, 1, 2)]:
        ta = torch.zeros(n * EB + 8, dtype=torch.uint8, device=dev)
        tb = torch.zeros(n * EB + 8, dtype=torch.uint8, device=dev)
        to = torch.full((n * EB + 8,), 0xEE, dtype=torch.uint8, device=dev)
        ta[ma:ma + n * EB] = torch.frombuffer(bytearray(a), dtype=torch.uint8).to(dev)
        tb[mb:mb + n * EB] = torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
        eng.add_dev(1, ta[ma:ma + n * EB], tb[mb:mb + n * EB], to[mo:mo + n * EB], n)
        torch.cuda.synchronize()
        assert eng.last_kernel_name() == ("k_g1_add_wire" if mo == 0 else "k_g1_add")
        host = to.cpu().numpy().tobytes()
        assert host[mo:mo + n * EB] == want, (ma, mb, mo)
        assert host[:mo] == b"\xee" * mo and host[mo + n * EB:] == b"\xee" * (8 - mo), "bytes outside the result written"


@pytest.mark.parametrize("name", KEYS + ["k1024b"])
def test_neg_in_one_launch_both_levels(name):
    """Neg (bgn.go:436-438) wire to wire in one launch (k_neg_wire): golden vectors of both levels, the identity and a
    real GT element keep their bytes, Neg(Neg(c)) == c on a ragged batch, equal to the decode / negate / encode route,
    and Add(c, Neg(c)) is the identity."""
    import numpy as np
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    eng = pk.engine
    EB = eng.elem_bytes
    cts = [e["ct"] for e in fx["encrypt"]]
    l2 = [v["out"] for v in fx["mult"]]
    a1 = H([cts[v["a"]] for v in fx["l1"]])
    a2 = H([l2[v["a"]] for v in fx["l2"]])
    for lvl, a, rows in ((1, a1, fx["l1"]), (2, a2, fx["l2"])):
        got = eng.neg(lvl, a)
        assert eng.last_kernel_name() == "k_neg_wire"
        for row, v in zip(got, rows):
            assert bytes(row).hex() == v["neg"], f"{name}: L{lvl} neg({v['a']})"
    one = (1).to_bytes(EB // 2, "big") + bytes(EB // 2)
    assert eng.neg(1, bytes(EB)).tobytes() == bytes(EB) and eng.neg(2, one).tobytes() == one
    pool = [bytes.fromhex(c) for c in cts]
    n = 1237
    big = b"".join(pool[(5 * i + 1) % len(pool)] for i in range(n))
    neg = eng.neg(1, big).tobytes()
    assert eng.neg(1, neg).tobytes() == big
    assert eng.add(1, big, neg).tobytes() == bytes(n * EB)
    eng.set_option("l1_fused", 0)
    try:
        assert eng.neg(1, big).tobytes() == neg and eng.last_kernel_name() != "k_neg_wire"
    finally:
        eng.set_option("l1_fused", 1)


@pytest.mark.parametrize("name", ["k256", "k1024"])
def test_one_launch_kernels_in_place(name):
    """The result array may be one of the operand arrays (accumulating into a device array): every workgroup has read
    its slices before it writes them (the level-1 kernel's two passes read a slice again only before that slice's own
    write).  Add / Sub on both levels into the first and into the second operand, Neg into its operand."""
    import torch
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    eng = pk.engine
    EB = eng.elem_bytes
    dev = torch.device("cuda", 0)
    cts = [bytes.fromhex(e["ct"]) for e in fx["encrypt"]]
    l2 = [bytes.fromhex(v["out"]) for v in fx["mult"]]
    n = 70003 if name == "k256" else 1031
    for lvl, pool in ((1, cts), (2, l2)):
        a = b"".join(pool[(2 * i + 1) % len(pool)] for i in range(n))
        b = b"".join(pool[(3 * i) % len(pool)] for i in range(n))
        want = eng.add(lvl, a, b).tobytes()
        want_neg = eng.neg(lvl, a).tobytes()
        for which in (0, 1):
            ta = torch.frombuffer(bytearray(a), dtype=torch.uint8).to(dev)
            tb = torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
            eng.add_dev(lvl, ta, tb, ta if which == 0 else tb, n)
            torch.cuda.synchronize()
            assert (ta if which == 0 else tb).cpu().numpy().tobytes() == want, (lvl, which)
        ta = torch.frombuffer(bytearray(a), dtype=torch.uint8).to(dev)
        eng.neg_dev(lvl, ta, ta, n)
        torch.cuda.synchronize()
        assert ta.cpu().numpy().tobytes() == want_neg, lvl


def test_sub_and_validate_on_device_arrays():
    """The host mirror's sub_dev / validate_dev (bgn_sub_batch_dev, bgn_validate_batch_dev) against the host-buffer forms,
    both levels; a device array shorter than the call needs is refused before any pointer reaches the C side."""
    import torch
    fx = load_fixture("k256")
    pk, _ = engine_key(fx)
    eng = pk.engine
    EB = eng.elem_bytes
    dev = torch.device("cuda", 0)
    pools = {1: [bytes.fromhex(e["ct"]) for e in fx["encrypt"]], 2: [bytes.fromhex(v["out"]) for v in fx["mult"]]}
    n = 777
    for lvl, pool in pools.items():
        a = b"".join(pool[(2 * i + 1) % len(pool)] for i in range(n))
        b = b"".join(pool[(5 * i) % len(pool)] for i in range(n))
        ta = torch.frombuffer(bytearray(a), dtype=torch.uint8).to(dev)
        tb = torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
        to = torch.empty(n * EB, dtype=torch.uint8, device=dev)
        eng.sub_dev(lvl, ta, tb, to, n)
        torch.cuda.synchronize()
        assert to.cpu().numpy().tobytes() == eng.sub(lvl, a, b).tobytes(), lvl
        broken = bytearray(a)
        broken[EB - 1] ^= 1                                          # element 0 leaves the curve / the norm-1 set
        tv = torch.frombuffer(broken, dtype=torch.uint8).to(dev)
        ok = torch.full((n,), 7, dtype=torch.uint8, device=dev)
        eng.validate_dev(lvl, tv, ok, n)
        torch.cuda.synchronize()
        got = ok.cpu().numpy()
        assert got[0] == 0 and got[1:].tolist() == eng.validate(lvl, bytes(broken))[1:].tolist() and got[1:].min() == 1, lvl
        with pytest.raises(ValueError):
            eng.sub_dev(lvl, ta, tb, to[: n * EB - 1], n)
