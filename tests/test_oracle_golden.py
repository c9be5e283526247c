"""CPU: the pure-Python oracle against the committed golden vectors, plus the
algebraic identities the reference's own tests rely on."""
import random

import pytest

import bgn_ref as R
from conftest import load_fixture, oracle_key

FAST = ["toy64", "k256"]


def W(pk, e):
    return R.elem_to_bytes(e, pk.p).hex()


def cts_of(pk, fx):
    return [R.Ciphertext(R.elem_from_bytes(bytes.fromhex(e["ct"]), pk.p) if int(e["ct"], 16) else None, False)
            for e in fx["encrypt"]]


@pytest.mark.parametrize("name", FAST + ["k512"])
def test_encrypt_vectors(name):
    fx = load_fixture(name)
    pk, _ = oracle_key(fx)
    for e in fx["encrypt"]:
        assert W(pk, pk.EncryptWithRandomness(int(e["x"], 16), int(e["r"], 16)).C) == e["ct"]
    # first vector is x = 0, r = 0: the identity, encoded as zero bytes
    assert int(fx["encrypt"][0]["ct"], 16) == 0


@pytest.mark.parametrize("name", FAST)
def test_l1_and_mult_vectors(name):
    fx = load_fixture(name)
    pk, _ = oracle_key(fx)
    cts = cts_of(pk, fx)
    for v in fx["l1"]:
        a, b = cts[v["a"]], cts[v["b"]]
        assert W(pk, pk.Add(a, b).C) == v["add"]
        assert W(pk, pk.Sub(a, b).C) == v["sub"]
        assert W(pk, pk.Neg(a).C) == v["neg"]
    for v in fx["mult"]:
        assert W(pk, pk.Mult(cts[v["a"]], cts[v["b"]]).C) == v["out"]


@pytest.mark.parametrize("name", FAST)
def test_pairing_properties(name):
    fx = load_fixture(name)
    pk, _ = oracle_key(fx)
    p, n = pk.p, pk.n
    g = pk.e(pk.P, pk.P)
    assert g != R.F2_ONE                                   # e(P,P) != 1
    assert R.f2_pow(g, n, p) == R.F2_ONE                   # order divides n
    assert (g[0] * g[0] + g[1] * g[1]) % p == 1            # norm-1 subgroup: inverse = conjugate
    rng = random.Random(3)
    a, b = rng.randrange(1, n), rng.randrange(1, n)
    A, B = R.pt_mul(pk.P, a, p), R.pt_mul(pk.P, b, p)
    assert pk.e(A, B) == R.f2_pow(g, a * b % n, p)         # bilinear
    assert pk.e(A, B) == pk.e(B, A)                        # symmetric (distortion map)
    assert pk.e(None, A) == R.F2_ONE and pk.e(A, None) == R.F2_ONE
    # two-step final exponentiation == one big power
    f = R.miller(A, B, n, p)
    two = R.f2_pow(R.f2_mul((f[0], (-f[1]) % p), R.f2_inv(f, p), p), pk.l, p)
    assert two == pk.e(A, B)
    # Q has order q1: e(Q, .)^q1 = 1
    q1 = int(fx["q1"], 16)
    assert R.f2_pow(pk.e(pk.Q, A), q1, p) == R.F2_ONE


@pytest.mark.parametrize("name", FAST)
def test_decrypt_vectors_and_truth_table(name):
    fx = load_fixture(name)
    pk, sk = oracle_key(fx)
    pk.SetupDecryption(sk)
    for d in fx["decrypt"]:
        e = R.elem_from_bytes(bytes.fromhex(d["ct"]), pk.p)
        if d["level"] == 1 and int(d["ct"], 16) == 0:
            e = None
        assert sk.Decrypt(R.Ciphertext(e, d["level"] == 2), pk) == d["expect"]
    # cmd/main.go:79-104 truth table (deterministic mode)
    zero, one = pk.EncryptWithRandomness(0, 5), pk.EncryptWithRandomness(1, 7)
    D = lambda c: sk.DecryptFailSafe(c, pk)
    assert [D(pk.Add(zero, zero)), D(pk.Add(zero, one)), D(pk.Add(one, one))] == [0, 1, 2]
    assert [D(pk.Mult(zero, one)), D(pk.Mult(one, one)), D(pk.Mult(one, pk.Neg(one)))] == [0, 1, -1]
    assert D(pk.Add(zero, pk.Neg(one))) == -1 and D(pk.Mult(pk.Neg(one), pk.Neg(one))) == 1


@pytest.mark.parametrize("name", FAST)
def test_aggregate_identity(name):
    """gadgets_test.go:24-46: Add(Enc(v1,r1),Enc(v2,r2)) == Enc(v1+v2, r1+r2), v,r < N."""
    fx = load_fixture(name)
    pk, _ = oracle_key(fx)
    rng = random.Random(9)
    v1, r1, v2, r2 = (rng.randrange(pk.n) for _ in range(4))
    lhs = pk.Add(pk.EncryptWithRandomness(v1, r1), pk.EncryptWithRandomness(v2, r2))
    assert lhs.C == pk.EncryptWithRandomness(v1 + v2, r1 + r2).C


@pytest.mark.parametrize("name", FAST)
def test_poly_vectors(name):
    fx = load_fixture(name)
    pk, sk = oracle_key(fx)
    pk.SetupDecryption(sk)
    po = fx["poly"]
    dec = lambda h: None if int(h, 16) == 0 else R.elem_from_bytes(bytes.fromhex(h), pk.p)
    ea = [R.Ciphertext(dec(h), False) for h in po["a"]]
    eb = [R.Ciphertext(dec(h), False) for h in po["b"]]
    prod = pk.MultPoly(ea, eb)
    assert [W(pk, c.C) for c in prod] == po["out"]
    # decrypting the product gives the plaintext convolution (poly_test.go:172-189)
    conv = [0] * (po["d1"] + po["d2"])
    for i, x in enumerate(po["ca"]):
        for k, y in enumerate(po["cb"]):
            conv[i + k] += x * y
    assert [sk.Decrypt(c, pk) for c in prod] == conv
