"""GPU: the wave-cooperative small-batch pairing kernel (bgn_amd/csrc/coop/) against the golden vectors, the
one-pairing-per-lane kernel and the C oracle.  The engine picks the kernel by batch size (options coop_max /
coop_max_l2 override the crossovers; 0 disables the cooperative kernel), so both kernels are driven through
the same C-ABI calls here."""
import random

import numpy as np
import pytest

import bgn_ref as R
from conftest import engine_key, load_fixture, oracle_key, KEYS

pytestmark = pytest.mark.gpu


def H(hexes):
    return b"".join(bytes.fromhex(h) for h in hexes)


@pytest.mark.parametrize("name", KEYS)
@pytest.mark.parametrize("kernel", ["coop", "lane"])
def test_mult_and_make_l2_golden_on_both_kernels(name, kernel, engopts):
    lim = "1000000" if kernel == "coop" else "0"
    engopts.set("coop_max", lim)
    engopts.set("coop_max_l2", lim)
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    eng = pk.engine
    cts = [e["ct"] for e in fx["encrypt"]]
    out = eng.mult(H([cts[v["a"]] for v in fx["mult"]]), H([cts[v["b"]] for v in fx["mult"]]))
    assert ("coop" in eng.last_kernel_name()) == (kernel == "coop")
    for row, v in zip(out, fx["mult"]):
        assert bytes(row).hex() == v["out"], f"{name}: Mult({v['a']},{v['b']}) on the {kernel} kernel"
    out = eng.make_l2(H([cts[v["a"]] for v in fx["make_l2"]]))
    assert ("coop" in eng.last_kernel_name()) == (kernel == "coop")
    for row, v in zip(out, fx["make_l2"]):
        assert bytes(row).hex() == v["out"]


@pytest.mark.parametrize("name,count", [("toy64", 333), ("k256", 130), ("k512", 64), ("k1024", 48)])
def test_coop_random_pairs_vs_c_oracle_and_lane_kernel(name, count, engopts):
    """Seeded random ciphertext pairs (Encrypt outputs with full-length randomness, a few identities): the
    cooperative kernel, the lane kernel and the C oracle give the same bytes."""
    import oracle_c
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    eng = pk.engine
    o = oracle_c.Oracle.from_fixture(fx)
    rng = random.Random(99)
    n = int(fx["n"], 16)
    xs = [rng.randrange(0, fx["msg_space"]) for _ in range(2 * count)]
    rs = [rng.randrange(0, n) for _ in range(2 * count)]
    cts = eng.encrypt(xs, rs).copy()
    cts[5] = 0                      # identity operands (2L zero bytes)
    cts[count + 9] = 0
    a, b = cts[:count].tobytes(), cts[count:].tobytes()
    engopts.set("coop_max", "1000000")
    got = eng.mult(a, b).tobytes()
    assert "coop" in eng.last_kernel_name()
    engopts.set("coop_max", "0")
    lane = eng.mult(a, b).tobytes()
    assert "coop" not in eng.last_kernel_name()
    assert got == lane
    assert got == o.mult(a, b)
    E = eng.elem_bytes
    one = (1).to_bytes(E // 2, "big") + bytes(E // 2)
    assert got[5 * E: 6 * E] == one and got[9 * E: 10 * E] == one


def test_coop_single_pairing_count_one(engopts):
    """count = 1, the reference's own call shape (bgn_test.go:127-140: one Mult per iteration)."""
    fx = load_fixture("k1024")
    pk, _ = engine_key(fx)
    cts = [e["ct"] for e in fx["encrypt"]]
    v = fx["mult"][0]
    out = pk.engine.mult(bytes.fromhex(cts[v["a"]]), bytes.fromhex(cts[v["b"]]))
    assert "coop" in pk.engine.last_kernel_name()
    assert bytes(out[0]).hex() == v["out"]


@pytest.mark.parametrize("name", ["k256", "k512", "k1024"])
def test_decrypt_small_batch_lifts_with_the_cooperative_kernel(name, engopts):
    """Decrypt of a few level-1 ciphertexts lifts with the cooperative kernel (e(C, P) in full) instead of the
    lane kernel's walk over the secret order's table: same plaintexts and statuses, negatives and an
    out-of-range value included (bgn.go:218-250)."""
    fx = load_fixture(name)
    pk, sk = engine_key(fx)
    pk.SetupDecryption(sk)
    eng = pk.engine
    T = fx["msg_space"]
    rng = random.Random(8)
    n = int(fx["n"], 16)
    ms = [0, 1, T - 1, 5 % T, 3 * T + 11] + [rng.randrange(T) for _ in range(20)]
    cts = eng.encrypt(ms, [rng.randrange(n) for _ in ms])
    cts[3] = eng.neg(1, cts[3:4])[0]
    engopts.set("coop_max_dec", "100000")
    m1, s1 = eng.decrypt(1, cts.tobytes())
    assert "coop" in eng.last_aux_kernel_name()
    engopts.set("coop_max_dec", "0")
    m0, s0 = eng.decrypt(1, cts.tobytes())
    assert "coop" not in eng.last_aux_kernel_name()
    assert m1.tolist() == m0.tolist() and s1.tolist() == s0.tolist()
    want = list(ms)
    want[3] = -want[3]
    # level 2: the power by the secret key alone is cooperative (k_gt_pow_coop)
    l2 = eng.make_l2(cts.tobytes()).tobytes()
    engopts.set("coop_max_dec", "100000")
    m2, s2 = eng.decrypt(2, l2)
    engopts.set("coop_max_dec", "0")
    m3, s3 = eng.decrypt(2, l2)
    assert m2.tolist() == m3.tolist() == m0.tolist() and s2.tolist() == s3.tolist() == s0.tolist()
    for got, st, w in zip(m1.tolist(), s1.tolist(), want):
        B = int(T ** 0.5) + (0 if int(T ** 0.5) ** 2 == T else 1)
        if abs(w) > B * B + B + 2:
            assert st == 1
        else:
            assert st == 0 and got == w


@pytest.mark.parametrize("name,npoly,d1,d2", [("k256", 3, 4, 3), ("k512", 1, 9, 13), ("k1024", 2, 4, 4)])
def test_multpoly_small_products_on_both_kernels(name, npoly, d1, d2, engopts):
    """MultPoly of a few short polynomials (the reference's own call: ONE product of ~10 x 10 coefficients,
    poly_test.go:173-189) pairs directly on the cooperative kernel; the lane kernels (line tables, Karatsuba
    levels) and the C oracle give the same coefficients."""
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
    engopts.set("coop_max", "1000000")
    got = eng.poly_mult(npoly, d1, d2, a, b).tobytes()
    engopts.set("coop_max", "0")
    lane = eng.poly_mult(npoly, d1, d2, a, b).tobytes()
    assert got == lane == o.poly_mult(npoly, d1, d2, a, b)


@pytest.mark.parametrize("name,count", [("toy64", 70), ("k256", 40), ("k512", 24), ("k1024", 12)])
def test_table_walk_on_the_waves_matches_the_general_program_and_the_lane_kernel(name, count, engopts):
    """makeL2 and the level-1 decryption lift of a small batch walk the key's normalised line table on the
    cooperative kernel (TD / TDA segments of tools/coop/gen_prog.py: 6 / 10 products per step in 2 / 3 rounds,
    coefficients prefetched one segment ahead).  Same bytes as the general cooperative program
    (option coop_table = 0), as the lane kernel's table loop and as the C oracle; identities in the batch; Decrypt's
    plaintexts and statuses equal on all three paths."""
    import oracle_c
    fx = load_fixture(name)
    pk, sk = engine_key(fx)
    pk.SetupDecryption(sk)
    eng = pk.engine
    o = oracle_c.Oracle.from_fixture(fx)
    rng = random.Random(count)
    n = int(fx["n"], 16)
    T = fx["msg_space"]
    ms = [rng.randrange(T) for _ in range(count)]
    cts = eng.encrypt(ms, [rng.randrange(n) for _ in ms])
    cts[2] = 0                                                       # an identity
    wire = cts.tobytes()
    res = {}
    for label, lim, tab in (("table", "1000000", "1"), ("general", "1000000", "0"), ("lane", "0", "1")):
        engopts.set("coop_max_l2", lim)
        engopts.set("coop_max_dec", lim)
        engopts.set("coop_table", tab)
        l2 = eng.make_l2(wire).tobytes()
        assert ("coop" in eng.last_kernel_name()) == (label != "lane")
        m, st = eng.decrypt(1, wire)
        assert ("coop" in eng.last_aux_kernel_name()) == (label != "lane")
        res[label] = (l2, m.tolist(), st.tolist())
    assert res["table"] == res["general"] == res["lane"]
    E = eng.elem_bytes
    s = min(count, 8)
    assert res["table"][0][: s * E] == o.mult(wire[: s * E])          # makeL2 = Pair(c, P), bgn.go:316-321
    want = [0 if i == 2 else m for i, m in enumerate(ms)]
    assert res["table"][1] == want and not any(res["table"][2])


def test_decrypt_default_dispatch_by_batch_size():
    """The default dispatch of Decrypt, no overrides, at the 256-bit key (cooperative crossover 512 there): 300
    ciphertexts lift and power on the cooperative kernels, 1000 on the lane-group kernels (table walk over the secret
    order's line table + square-and-multiply by the secret key); plaintexts and statuses as encrypted, and equal to
    the all-lane path."""
    fx = load_fixture("k256")
    pk, sk = engine_key(fx)
    pk.SetupDecryption(sk)
    eng = pk.engine
    for v, dflt in (("coop_max_dec", -1), ("coop_table", 1), ("quad_max_dec", -1), ("quad_min", -1)):
        assert eng.get_option(v) == dflt
    rng = random.Random(12)
    n, T = int(fx["n"], 16), fx["msg_space"]
    ms = [rng.randrange(T) for _ in range(1000)]
    cts = eng.encrypt(ms, [rng.randrange(n) for _ in ms]).tobytes()
    m, st = eng.decrypt(1, cts)
    assert "quad" in eng.last_aux_kernel_name()
    assert m.tolist() == ms and not st.any()
    ms3, st3 = eng.decrypt(1, cts[: 300 * eng.elem_bytes])
    assert "coop" in eng.last_aux_kernel_name()
    assert ms3.tolist() == ms[:300] and not st3.any()
    with eng.options(coop_max_dec=0):
        m0, st0 = eng.decrypt(1, cts)
        assert "coop" not in eng.last_aux_kernel_name() and "quad" not in eng.last_aux_kernel_name()
    assert m0.tolist() == ms and st0.tolist() == st.tolist()
