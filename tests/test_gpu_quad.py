"""GPU: the lane-group pairing kernel (bgn_amd/csrc/quad/: sixteen lanes per pairing, a field element over the four
lanes of a quad) against the golden vectors, the other two pairing kernels and the C oracle.  The engine picks the
kernel by batch size; BGN_QUAD_MIN / BGN_QUAD_MAX move the range of the lane-group kernel (BGN_QUAD_MAX=0 disables
it), so every kernel is driven through the same C-ABI calls here."""
import random

import pytest

from conftest import engine_key, load_fixture

pytestmark = pytest.mark.gpu

QUAD_KEYS = ["k256", "k512", "k1024"]        # limb counts 10, 19, 36: 3, 5, 9 limbs per lane


def H(hexes):
    return b"".join(bytes.fromhex(h) for h in hexes)


def force(monkeypatch, kernel):
    """kernel: 'quad', 'coop' or 'lane' for every batch size."""
    monkeypatch.setenv("BGN_QUAD_MIN", "0")
    monkeypatch.setenv("BGN_QUAD_MAX", "100000000" if kernel == "quad" else "0")
    monkeypatch.setenv("BGN_COOP_MAX", "100000000" if kernel == "coop" else "0")


@pytest.mark.parametrize("name", QUAD_KEYS)
def test_mult_golden_on_the_lane_group_kernel(name, monkeypatch):
    force(monkeypatch, "quad")
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    eng = pk.engine
    cts = [e["ct"] for e in fx["encrypt"]]
    out = eng.mult(H([cts[v["a"]] for v in fx["mult"]]), H([cts[v["b"]] for v in fx["mult"]]))
    assert "quad" in eng.last_kernel_name()
    for row, v in zip(out, fx["mult"]):
        assert bytes(row).hex() == v["out"], f"{name}: Mult({v['a']},{v['b']}) on the lane-group kernel"


@pytest.mark.parametrize("name,count", [("k256", 131), ("k512", 65), ("k1024", 49), ("k1024", 16)])
def test_quad_random_pairs_vs_c_oracle_and_the_other_kernels(name, count, monkeypatch):
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
        force(monkeypatch, kernel)
        got[kernel] = eng.mult(a, b).tobytes()
        assert (kernel in eng.last_kernel_name()) == (kernel != "lane"), eng.last_kernel_name()
    assert got["quad"] == got["lane"] == got["coop"]
    assert got["quad"] == o.mult(a, b)
    E = eng.elem_bytes
    one = (1).to_bytes(E // 2, "big") + bytes(E // 2)
    assert got["quad"][5 * E: 6 * E] == one and got["quad"][9 * E: 10 * E] == one


@pytest.mark.parametrize("name,npoly,d1,d2", [("k256", 3, 4, 3), ("k512", 1, 9, 13), ("k1024", 2, 4, 4)])
def test_multpoly_direct_pairs_on_the_lane_group_kernel(name, npoly, d1, d2, monkeypatch):
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
    force(monkeypatch, "quad")
    got = eng.poly_mult(npoly, d1, d2, a, b).tobytes()
    force(monkeypatch, "lane")
    lane = eng.poly_mult(npoly, d1, d2, a, b).tobytes()
    assert got == lane == o.poly_mult(npoly, d1, d2, a, b)


def test_default_dispatch_uses_the_lane_group_kernel_between_the_crossovers():
    """At a 1024-bit key a batch of 12 000 Mults is above the cooperative kernel's crossover and far below one
    pairing per lane filling the chip: the engine picks the lane-group kernel by itself; 64 pairs go to the
    cooperative kernel.  Both give the bytes of the golden vectors (the batch repeats them)."""
    fx = load_fixture("k1024")
    pk, _ = engine_key(fx)
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


@pytest.mark.parametrize("name", ["k512", "k1024"])
def test_zero_norm_yields_the_identity_on_every_kernel(name, monkeypatch):
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
        force(monkeypatch, kernel)
        out = eng.mult(a, b)
        assert (kernel in eng.last_kernel_name()) == (kernel != "lane")
        assert bytes(out[0]).hex() == good[0]["out"] and bytes(out[2]).hex() == good[1]["out"], kernel
        assert bytes(out[1]) == one, kernel
