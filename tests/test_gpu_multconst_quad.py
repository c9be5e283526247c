"""GPU: MultConst with per-element scalars on the lane groups (bgn_amd/csrc/quad/quad_g1.hpp: sixteen lanes per
element; level 1 a windowed Jacobian ladder with the lane kernel as the exact fallback for flagged elements, level 2 a
windowed power in F_p^2) against the golden vectors, the one-element-per-lane kernels and the C oracle
(`res.PowBig(c.C, constant)`, bgn.go:253-291).  Option quad_max_mc moves the range of the kernels (0: never)."""
import random

import numpy as np
import pytest

from conftest import engine_key, load_fixture

pytestmark = pytest.mark.gpu

KEYS = ["k256", "k512", "k1024"]


def H(hexes):
    return b"".join(bytes.fromhex(h) for h in hexes)


@pytest.mark.parametrize("name", KEYS)
@pytest.mark.parametrize("kernel", ["quad", "lane"])
def test_multconst_golden_on_both_kernels(name, kernel, engopts):
    engopts.set("quad_max_mc", (1 << 40) if kernel == "quad" else 0)
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    eng = pk.engine
    cts = [e["ct"] for e in fx["encrypt"]]
    l2 = [v["out"] for v in fx["mult"]]
    for lvl, key, src in [(1, "multconst_l1", cts), (2, "multconst_l2", l2)]:
        out = eng.multconst(lvl, H([src[v["a"]] for v in fx[key]]), [int(v["k"], 16) for v in fx[key]])
        assert ("quad" in eng.last_kernel_name()) == (kernel == "quad"), eng.last_kernel_name()
        for row, v in zip(out, fx[key]):
            assert bytes(row).hex() == v["out"], f"{name}: MultConst L{lvl} k={v['k']} on the {kernel} kernel"


@pytest.mark.parametrize("name,count,kbytes", [("k256", 131, 5), ("k256", 70, 0), ("k512", 49, 16), ("k1024", 37, 0), ("k1024", 200, 5)])
def test_multconst_random_scalars_vs_c_oracle_and_lane_kernel(name, count, kbytes, engopts):
    """Seeded random ciphertexts and scalars (kbytes = 0: full length, some beyond the group order), with the special
    elements the ladder must get right: scalar 0 (identity through the flag), scalar 1, the group order n and 2n (the
    accumulator meets -T: Z = 0, flagged, recomputed by the exact lane kernel: the identity), 16^j - 1 and 0x88..8
    (digits 8 / -1 chains), an identity base; counts that leave the last workgroup and wave ragged.  Both levels."""
    import oracle_c
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    eng = pk.engine
    o = oracle_c.Oracle.from_fixture(fx)
    rng = random.Random(count * 31 + kbytes)
    n = int(fx["n"], 16)
    nb = (n.bit_length() + 7) // 8
    klen = kbytes or nb + 1
    cts = eng.encrypt([rng.randrange(fx["msg_space"]) for _ in range(count)], [rng.randrange(n) for _ in range(count)]).copy()
    cts[6] = 0                                                   # an identity base
    ks = [rng.randrange(1 << (8 * klen)) for _ in range(count)]
    ks[0], ks[1], ks[2] = 0, 1, (16 ** (2 * klen - 1)) - 1
    ks[3] = int("8" * (2 * klen), 16)
    if not kbytes:
        ks[4], ks[5], ks[7] = n, 2 * n, n - 1
    wire = cts.tobytes()
    l2 = eng.make_l2(wire).tobytes()
    res = {}
    for kernel in ("quad", "lane"):
        engopts.set("quad_max_mc", (1 << 40) if kernel == "quad" else 0)
        r1 = eng.multconst(1, wire, ks).tobytes()
        assert ("quad" in eng.last_kernel_name()) == (kernel == "quad")
        r2 = eng.multconst(2, l2, ks).tobytes()
        assert ("quad" in eng.last_kernel_name()) == (kernel == "quad")
        res[kernel] = (r1, r2)
    assert res["quad"][0] == res["lane"][0]
    assert res["quad"][1] == res["lane"][1]
    s = min(count, 24 if name != "k1024" else 10)
    E = eng.elem_bytes
    assert res["quad"][0][: s * E] == o.multconst(1, wire[: s * E], ks[:s])
    assert res["quad"][1][: s * E] == o.multconst(2, l2[: s * E], ks[:s])
    zero = bytes(E)
    assert res["quad"][0][:E] == zero and res["quad"][0][6 * E: 7 * E] == zero       # k = 0; identity base
    if not kbytes:
        assert res["quad"][0][4 * E: 5 * E] == zero and res["quad"][0][5 * E: 6 * E] == zero     # k = n, 2n


def test_multconst_batches_cut_into_lane_rounds_and_a_lane_group_remainder(engopts):
    """65536 + 300 elements: whole rounds of the one-element-per-lane kernel first, the remainder on the lane groups
    (engine.cpp bgn_multconst_batch_dev) — the bytes of the single launch (split_rounds = 0), both levels."""
    fx = load_fixture("k256")
    pk, _ = engine_key(fx)
    eng = pk.engine
    rng = np.random.default_rng(4)
    count = 65536 + 300
    base = eng.encrypt([int(v) for v in rng.integers(0, fx["msg_space"], 64)], [int(v) + 7 for v in rng.integers(0, 1 << 60, 64)])
    idx = rng.integers(0, 64, count)
    wire = base[idx].tobytes()
    ks = [int(v) for v in rng.integers(0, 1 << 62, count)]
    l2 = eng.make_l2(base.tobytes())[idx].tobytes()
    got = {}
    for split in (1, 0):
        engopts.set("split_rounds", split)
        got[split] = (eng.multconst(1, wire, ks).tobytes(), eng.multconst(2, l2, ks).tobytes())
        assert "quad" not in eng.last_kernel_name()             # the head piece is the lane kernel's
    assert got[1] == got[0]


def test_multconst_lane_groups_at_every_size_run_in_pieces(engopts):
    """Where the lane groups take every batch size (the default at 72 limbs; here quad_max_mc = 2^28 on a small key)
    they take it in pieces of 2^17 elements, so that the ladder's per-element tables stay a bounded workspace: 2^17 +
    300 elements are two pieces, bytes equal to the one-element-per-lane kernels', both levels."""
    fx = load_fixture("k256")
    pk, _ = engine_key(fx)
    eng = pk.engine
    rng = np.random.default_rng(9)
    count = (1 << 17) + 300
    base = eng.encrypt([int(v) for v in rng.integers(0, fx["msg_space"], 64)], [int(v) + 3 for v in rng.integers(0, 1 << 60, 64)])
    idx = rng.integers(0, 64, count)
    wire = base[idx].tobytes()
    ks = [int(v) for v in rng.integers(0, 1 << 62, count)]
    l2 = eng.make_l2(base.tobytes())[idx].tobytes()
    got = {}
    for lim in (1 << 28, 0):
        engopts.set("quad_max_mc", lim)
        before = eng.memory_bytes()
        got[lim] = (eng.multconst(1, wire, ks).tobytes(), eng.multconst(2, l2, ks).tobytes())
        assert ("quad" in eng.last_kernel_name()) == (lim != 0)
        if lim:
            # the workspace grew by what 2^17 elements need (1.7 KB of table each at 10 limbs), not by the batch
            assert eng.memory_bytes() - before < (1 << 17) * 8192
    assert got[1 << 28] == got[0]


def test_multconst_default_dispatch():
    """No overrides: a single element and a few thousand go to the lane groups."""
    fx = load_fixture("k256")
    pk, _ = engine_key(fx)
    eng = pk.engine
    assert eng.get_option("quad_max_mc") == -1
    ct = bytes.fromhex(fx["encrypt"][3]["ct"])
    eng.multconst(1, ct, [12345])
    assert "k_g1_mul_quad" in eng.last_kernel_name()
    eng.multconst(2, bytes.fromhex(fx["mult"][0]["out"]), [12345])
    assert "k_gt_pow_quad_each" in eng.last_kernel_name()
    with eng.options(combine=0):
        eng.multconst(1, ct * 3000, list(range(1, 3001)))
        assert "k_g1_mul_quad" in eng.last_kernel_name()
