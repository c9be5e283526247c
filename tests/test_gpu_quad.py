"""GPU: the lane-group pairing kernel (bgn_amd/csrc/quad/: sixteen lanes per pairing, a field element over the four
lanes of a quad) against the golden vectors, the other two pairing kernels and the C oracle.  The engine picks the
kernel by batch size; options quad_min / quad_max move the range of the lane-group kernel (quad_max = 0 disables
it), so every kernel is driven through the same C-ABI calls here."""
import random

import numpy as np
import pytest

from conftest import engine_key, load_fixture

pytestmark = pytest.mark.gpu

QUAD_KEYS = ["k256", "k512", "k1024"]        # limb counts 10, 19, 36: 3, 5, 9 limbs per lane


def H(hexes):
    return b"".join(bytes.fromhex(h) for h in hexes)


def force(engopts, kernel):
    """kernel: 'quad', 'coop' or 'lane' for every batch size."""
    engopts.set("quad_min", "0")
    engopts.set("quad_max", "100000000" if kernel == "quad" else "0")
    engopts.set("coop_max", "100000000" if kernel == "coop" else "0")


@pytest.mark.parametrize("name", QUAD_KEYS)
def test_mult_golden_on_the_lane_group_kernel(name, engopts):
    force(engopts, "quad")
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    eng = pk.engine
    cts = [e["ct"] for e in fx["encrypt"]]
    out = eng.mult(H([cts[v["a"]] for v in fx["mult"]]), H([cts[v["b"]] for v in fx["mult"]]))
    assert "quad" in eng.last_kernel_name()
    for row, v in zip(out, fx["mult"]):
        assert bytes(row).hex() == v["out"], f"{name}: Mult({v['a']},{v['b']}) on the lane-group kernel"


@pytest.mark.parametrize("name,count", [("k256", 131), ("k512", 65), ("k1024", 49), ("k1024", 16)])
def test_quad_random_pairs_vs_c_oracle_and_the_other_kernels(name, count, engopts):
    """Seeded random ciphertext pairs (Encrypt outputs with full-length randomness, two identities; counts that
    leave the last workgroup and the last wave ragged): the three pairing kernels and the C oracle give the same
    bytes."""
    import oracle_c
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    eng = pk.engine
    o = oracle_c.Oracle.from_fixture(fx)
    rng = random.Random(77)
    n = int(fx["n"], 16)
    xs = [rng.randrange(0, fx["msg_space"]) for _ in range(2 * count)]
    rs = [rng.randrange(0, n) for _ in range(2 * count)]
    cts = eng.encrypt(xs, rs).copy()
    cts[5] = 0                      # identity operands (2L zero bytes)
    cts[count + 9] = 0
    a, b = cts[:count].tobytes(), cts[count:].tobytes()
    got = {}
    for kernel in ("quad", "coop", "lane"):
        force(engopts, kernel)
        got[kernel] = eng.mult(a, b).tobytes()
        assert (kernel in eng.last_kernel_name()) == (kernel != "lane"), eng.last_kernel_name()
    assert got["quad"] == got["lane"] == got["coop"]
    assert got["quad"] == o.mult(a, b)
    # the lane groups' two Miller loops: over the width-5 NAF with the per-pairing table (the default) and the plain NAF
    force(engopts, "quad")
    assert eng.get_option("quad_window") == -1
    for w in (0, 1):
        engopts.set("quad_window", w)
        assert eng.mult(a, b).tobytes() == got["lane"], "quad_window = %d" % w
    E = eng.elem_bytes
    one = (1).to_bytes(E // 2, "big") + bytes(E // 2)
    assert got["quad"][5 * E: 6 * E] == one and got["quad"][9 * E: 10 * E] == one


@pytest.mark.parametrize("width", [3, 4, 0])
def test_quad_windowed_loop_at_other_widths(width, engopts, monkeypatch):
    """A context made with miller_window = 3 or 4 (read at bgn_ctx_create: BGN_MILLER_WINDOW) holds a narrower NAF: the
    lane groups' table then has one resp. three multiples; 0 = no width-w NAF at all: the plain loop.  Golden Mult
    vectors on the lane-group and the lane kernel."""
    import bgn_amd
    monkeypatch.setenv("BGN_MILLER_WINDOW", str(width))
    fx = load_fixture("k256")
    pk = bgn_amd.PublicKey(int(fx["p"], 16), int(fx["n"], 16), fx["l"], bytes.fromhex(fx["P"]), bytes.fromhex(fx["Q"]),
                           fx["msg_space"], True, fx["poly_base"])
    monkeypatch.delenv("BGN_MILLER_WINDOW")
    eng = pk.engine
    engopts.register(eng)
    assert eng.get_option("miller_window") == width
    cts = [e["ct"] for e in fx["encrypt"]]
    a, b = H([cts[v["a"]] for v in fx["mult"]]), H([cts[v["b"]] for v in fx["mult"]])
    for kernel in ("quad", "lane"):
        force(engopts, kernel)
        out = eng.mult(a, b)
        assert ("quad" in eng.last_kernel_name()) == (kernel == "quad")
        for row, v in zip(out, fx["mult"]):
            assert bytes(row).hex() == v["out"], "miller_window = %d on the %s kernel" % (width, kernel)


def test_quad_large_batch_runs_in_pieces(engopts):
    """The lane-group launcher cuts a batch into pieces of 196 608 pairings that reuse one workspace (kern_quad.hip
    quad_piece): 196 608 + 300 pairs at a small key, with and without the width-5 loop's tables, equal the lane kernel's
    bytes; the workspace stays that of one piece."""
    fx = load_fixture("k256")
    pk, _ = engine_key(fx)
    eng = pk.engine
    rng = np.random.default_rng(21)
    count = 196608 + 300
    base = eng.encrypt([int(v) for v in rng.integers(0, fx["msg_space"], 96)], [int(v) + 5 for v in rng.integers(0, 1 << 60, 96)])
    a = base[rng.integers(0, 96, count)].tobytes()
    b = base[rng.integers(0, 96, count)].tobytes()
    force(engopts, "lane")
    want = eng.mult(a, b).tobytes()
    force(engopts, "quad")
    for w in (1, 0):
        engopts.set("quad_window", w)
        assert eng.mult(a, b).tobytes() == want, "quad_window = %d" % w
        assert "quad" in eng.last_kernel_name()
    # the walk over the key's line table (makeL2) goes through the same launcher
    res = {}
    for kernel in ("lane", "quad"):
        force_table(engopts, kernel)
        res[kernel] = eng.make_l2(a).tobytes()
        assert ("quad" in eng.last_kernel_name()) == (kernel == "quad")
    assert res["quad"] == res["lane"]


@pytest.mark.parametrize("name,npoly,d1,d2", [("k256", 3, 4, 3), ("k512", 1, 9, 13), ("k1024", 2, 4, 4)])
def test_multpoly_direct_pairs_on_the_lane_group_kernel(name, npoly, d1, d2, engopts):
    """MultPoly's coefficient pairs (poly.go:139-146) on the lane-group kernel: same coefficients as the lane
    kernels (line tables, Karatsuba levels) and the C oracle."""
    import oracle_c
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    eng = pk.engine
    o = oracle_c.Oracle.from_fixture(fx)
    rng = random.Random(d1 * 100 + d2)
    n = int(fx["n"], 16)
    ca = eng.encrypt([rng.randrange(3) for _ in range(npoly * d1)], [rng.randrange(n) for _ in range(npoly * d1)]).copy()
    cb = eng.encrypt([rng.randrange(3) for _ in range(npoly * d2)], [rng.randrange(n) for _ in range(npoly * d2)]).copy()
    ca[1] = 0                                                        # an identity coefficient
    a, b = ca.tobytes(), cb.tobytes()
    force(engopts, "quad")
    got = eng.poly_mult(npoly, d1, d2, a, b).tobytes()
    force(engopts, "lane")
    lane = eng.poly_mult(npoly, d1, d2, a, b).tobytes()
    assert got == lane == o.poly_mult(npoly, d1, d2, a, b)


def test_default_dispatch_uses_the_lane_group_kernel_between_the_crossovers():
    """At a 1024-bit key a batch of 12 000 Mults is above the cooperative kernel's crossover and far below one
    pairing per lane filling the chip: the engine picks the lane-group kernel by itself; 64 pairs go to the
    cooperative kernel.  Both give the bytes of the golden vectors (the batch repeats them)."""
    fx = load_fixture("k1024")
    pk, sk = engine_key(fx)
    eng = pk.engine
    cts = [e["ct"] for e in fx["encrypt"]]
    reps = 12000 // len(fx["mult"]) + 1
    a = H([cts[v["a"]] for v in fx["mult"]]) * reps
    b = H([cts[v["b"]] for v in fx["mult"]]) * reps
    out = eng.mult(a, b)
    assert "quad" in eng.last_kernel_name()
    want = [v["out"] for v in fx["mult"]]
    for i in (0, 1, len(want) - 1, len(want), 5000, len(out) - 1):
        assert bytes(out[i]).hex() == want[i % len(want)]
    out = eng.mult(a[: 64 * eng.elem_bytes], b[: 64 * eng.elem_bytes])
    assert "coop" in eng.last_kernel_name()
    # makeL2 and Decrypt of 5 000 ciphertexts walk the key's line tables on the lane-group kernel by default; 200 go
    # to the cooperative kernel, 2^16 to the lane kernels
    pk.SetupDecryption(sk)
    five = a[: 5000 * eng.elem_bytes]
    l2 = eng.make_l2(five)
    assert "quad" in eng.last_kernel_name()
    want_l2 = {v["a"]: v["out"] for v in fx["make_l2"]}
    for i, v in enumerate(fx["mult"][:4]):
        if v["a"] in want_l2:
            assert bytes(l2[i]).hex() == want_l2[v["a"]]
    m, st = eng.decrypt(1, five)
    assert "quad" in eng.last_aux_kernel_name()
    m200, st200 = eng.decrypt(1, five[: 200 * eng.elem_bytes])
    assert "coop" in eng.last_aux_kernel_name()
    assert m[:200].tolist() == m200.tolist() and st[:200].tolist() == st200.tolist()
    big = a[: 65536 * eng.elem_bytes] if len(a) >= 65536 * eng.elem_bytes else (a * (65536 * eng.elem_bytes // len(a) + 1))[: 65536 * eng.elem_bytes]
    mb, sb = eng.decrypt(1, big)
    assert "quad" not in eng.last_aux_kernel_name() and "coop" not in eng.last_aux_kernel_name()
    assert mb[:200].tolist() == m200.tolist() and sb[:200].tolist() == st200.tolist()


@pytest.mark.parametrize("name", ["k512", "k1024"])
def test_zero_norm_yields_the_identity_on_every_kernel(name, engopts):
    """An operand that is not on the curve can drive f to zero: A = (a, 0) doubles to Z3 = 2YZ = 0 and its tangent
    at phi(B), B = (-a, y), is (3a^2 + 1)(xB + a) + 0i = 0, so N(f) = 0 and its inverse is 0.  The lane kernel maps
    such a pairing to the identity (as PBC's SetBytes maps an invalid point to O); the cooperative and the
    lane-group kernel must return the same bytes, and the neighbours of the bad pair their own values."""
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    eng = pk.engine
    p = int(fx["p"], 16)
    L = eng.elem_bytes // 2
    cts = [e["ct"] for e in fx["encrypt"]]
    good = [v for v in fx["mult"] if int(cts[v["a"]], 16) and int(cts[v["b"]], 16)][:2]
    bad_a = (5).to_bytes(L, "big") + bytes(L)
    bad_b = (p - 5).to_bytes(L, "big") + (7).to_bytes(L, "big")
    a = bytes.fromhex(cts[good[0]["a"]]) + bad_a + bytes.fromhex(cts[good[1]["a"]])
    b = bytes.fromhex(cts[good[0]["b"]]) + bad_b + bytes.fromhex(cts[good[1]["b"]])
    one = (1).to_bytes(L, "big") + bytes(L)
    for kernel in ("lane", "coop", "quad"):
        force(engopts, kernel)
        out = eng.mult(a, b)
        assert (kernel in eng.last_kernel_name()) == (kernel != "lane")
        assert bytes(out[0]).hex() == good[0]["out"] and bytes(out[2]).hex() == good[1]["out"], kernel
        assert bytes(out[1]) == one, kernel


def force_table(engopts, kernel):
    """The walks over a key's line table (makeL2, Decrypt's lift) and Decrypt's power on 'quad', 'coop' or 'lane'."""
    big = "100000000"
    engopts.set("quad_min", "0")
    for v in ("quad_max_l2", "quad_max_dec", "quad_max_pow"):
        engopts.set(v, big if kernel == "quad" else "0")
    for v in ("coop_max_l2", "coop_max_dec"):
        engopts.set(v, big if kernel == "coop" else "0")


@pytest.mark.parametrize("name", QUAD_KEYS)
def test_make_l2_golden_on_the_lane_group_table_walk(name, engopts):
    force_table(engopts, "quad")
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    eng = pk.engine
    cts = [e["ct"] for e in fx["encrypt"]]
    out = eng.make_l2(H([cts[v["a"]] for v in fx["make_l2"]]))
    assert "quad" in eng.last_kernel_name()
    for row, v in zip(out, fx["make_l2"]):
        assert bytes(row).hex() == v["out"], f"{name}: makeL2({v['a']}) on the lane-group kernel"


@pytest.mark.parametrize("name,count", [("k256", 70), ("k512", 41), ("k1024", 33), ("k1024b", 17)])
def test_table_walk_and_power_on_the_lane_groups_match_the_other_kernels(name, count, engopts):
    """makeL2 and level-1 / level-2 Decrypt of a batch with an identity, a negative and an out-of-range value: the
    lane-group kernels (table walk over P's table / over the secret order's table, power by the secret key), the
    cooperative kernels and the lane kernels give the same bytes, plaintexts and statuses; makeL2 also equals the C
    oracle.  k1024b: 37 limbs (10 per lane, the top limb in lane 3 at index 6)."""
    import oracle_c
    import bgn_amd
    fx = load_fixture(name)
    if name == "k1024b":
        pk = bgn_amd.PublicKey(int(fx["p"], 16), int(fx["n"], 16), fx["l"], bytes.fromhex(fx["P"]), bytes.fromhex(fx["Q"]),
                               fx["msg_space"], True, fx["poly_base"])
        sk = bgn_amd.SecretKey(int(fx["q1"], 16))
        pk.engine.set_memory_budget(40 << 30)
        engopts.register(pk.engine)
    else:
        pk, sk = engine_key(fx)
    pk.SetupDecryption(sk)
    eng = pk.engine
    o = oracle_c.Oracle.from_fixture(fx)
    rng = random.Random(count)
    n = int(fx["n"], 16)
    T = fx["msg_space"]
    ms = [rng.randrange(T) for _ in range(count)]
    ms[4] = 3 * T + 11                                                # no discrete log in either direction
    cts = eng.encrypt(ms, [rng.randrange(n) for _ in ms]).copy()
    cts[2] = 0                                                       # an identity
    cts[3] = eng.neg(1, cts[3:4])[0]
    wire = cts.tobytes()
    res = {}
    for kernel in ("quad", "coop", "lane"):
        force_table(engopts, kernel)
        l2 = eng.make_l2(wire).tobytes()
        assert (kernel in eng.last_kernel_name()) == (kernel != "lane"), eng.last_kernel_name()
        m1, s1 = eng.decrypt(1, wire)
        assert (kernel in eng.last_aux_kernel_name()) == (kernel != "lane"), eng.last_aux_kernel_name()
        m2, s2 = eng.decrypt(2, l2)
        res[kernel] = (l2, m1.tolist(), s1.tolist(), m2.tolist(), s2.tolist())
    assert res["quad"] == res["lane"] == res["coop"]
    assert res["quad"][0] == o.mult(wire)                              # b = None: e(., P), makeL2
    want = list(ms)
    want[2], want[3] = 0, -ms[3]
    st = res["quad"][2]
    assert st[4] == 1 and not any(st[:4]) and not any(st[5:])
    assert [v for i, v in enumerate(res["quad"][1]) if i != 4] == [v for i, v in enumerate(want) if i != 4]
