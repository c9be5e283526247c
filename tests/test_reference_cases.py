"""The reference's OWN test inputs, verbatim, through the HIP engine.

The reference holds no byte-level vectors (every test draws a fresh key); what it does hold are behavioural
pins with literal values.  Each test below is one case of
    /root/reference/poly_test.go:68-189   (9.123; 0.1 + 4.2; 50.1 + 41.2 at level 2; 9.13 x 4.12 at both levels;
                                           1.1 x 40.2 — each compared as "%.1f"),
    /root/reference/gadgets_test.go:8-105 (decryption proofs, aggregate proofs, proofs of plaintext knowledge),
    /root/reference/cmd/main.go:24-104    (the arithmetic check and the 0 / 1 / -1 truth table)
at the constants of /root/reference/bgn_test.go:8-13: KEYBITS 512, MSGSPACE 1021, POLYBASE 3, FPSCALEBASE 3,
FPPREC 0.0001, DET true (the k512 fixture key).  The fixed-point encoding (plaintext.go) stays on the CPU side of
the boundary: tests/plaintext_go.py."""
import random

import pytest

from conftest import engine_key, load_fixture
from plaintext_go import Encoding, f1

KEYBITS, POLYBASE, MSGSPACE, FPSCALEBASE, FPPREC, DET = 512, 3, 1021, 3, 0.0001, True      # bgn_test.go:8-13


@pytest.fixture(scope="module")
def keys():
    fx = load_fixture("k512")
    assert int(fx["n"], 16).bit_length() == KEYBITS and fx["msg_space"] == MSGSPACE and fx["poly_base"] == POLYBASE
    pk, sk = engine_key(fx)
    assert pk.Deterministic == DET
    pk.SetupDecryption(sk)
    return pk, sk, Encoding(POLYBASE, FPSCALEBASE, FPPREC)


def enc(pk, p):
    return pk.EncryptPoly(p.Coefficients, p.ScaleFactor)


def dec(pk, sk, E, ct):
    return E.decrypted(sk.DecryptPoly(ct, pk), ct.ScaleFactor)


# ---- plaintext.go alone (no GPU): poly_test.go:68-90
def test_encode_balanced_poly():
    E = Encoding(POLYBASE, FPSCALEBASE, FPPREC)
    assert f1(E.NewPolyPlaintext(9.123).PolyEval()) == f1(9.123) == "9.1\n"


def test_encode_unbalanced_poly():
    E = Encoding(POLYBASE, FPSCALEBASE, FPPREC)
    assert f1(E.NewUnbalancedPlaintext(9.123).PolyEval()) == f1(9.123)


# ---- poly_test.go:92-189
@pytest.mark.gpu
def test_encode_encrypt_decrypt_poly(keys):                         # poly_test.go:92-104
    pk, sk, E = keys
    p1 = E.NewPolyPlaintext(9.123)
    actual = dec(pk, sk, E, enc(pk, p1)).PolyEval()
    assert f1(actual) == f1(9.123) == "9.1\n"


@pytest.mark.gpu
def test_add_poly(keys):                                            # poly_test.go:106-124
    pk, sk, E = keys
    p1, p2 = E.NewPolyPlaintext(0.1), E.NewPolyPlaintext(4.2)
    r1 = pk.AddPoly(enc(pk, p1), enc(pk, p2))
    actual = dec(pk, sk, E, r1).PolyEval()
    expected = p1.PolyEval() + p2.PolyEval()
    assert f1(actual) == f1(expected) == "4.3\n"


@pytest.mark.gpu
def test_add_poly_l2(keys):                                         # poly_test.go:126-146
    pk, sk, E = keys
    p1, p2 = E.NewPolyPlaintext(50.1), E.NewPolyPlaintext(41.2)
    c1, c2 = pk.MakePolyL2(enc(pk, p1)), pk.MakePolyL2(enc(pk, p2))
    assert c1.L2 and c2.L2
    actual = dec(pk, sk, E, pk.AddPoly(c1, c2)).PolyEval()
    expected = p1.PolyEval() + p2.PolyEval()
    assert f1(actual) == f1(expected) == "91.3\n"


@pytest.mark.gpu
def test_mult_const_poly(keys):                                     # poly_test.go:148-171
    pk, sk, E = keys
    p1, p2 = E.NewPolyPlaintext(9.13), E.NewPolyPlaintext(4.12)
    c1 = enc(pk, p1)
    const = E.NewUnbalancedPlaintext(4.12)                          # what MultConstPoly makes of its *big.Float, poly.go:78
    expected = p1.PolyEval() * p2.PolyEval()
    actual = dec(pk, sk, E, pk.MultConstPoly(c1, const)).PolyEval()
    assert f1(actual) == f1(expected) == "37.6\n", "[L1]"
    c1 = pk.MakePolyL2(c1)
    actual = dec(pk, sk, E, pk.MultConstPoly(c1, const)).PolyEval()
    assert f1(actual) == f1(expected), "[L2]"


@pytest.mark.gpu
def test_mult_poly(keys):                                           # poly_test.go:173-189
    pk, sk, E = keys
    p1, p2 = E.NewPolyPlaintext(1.1), E.NewPolyPlaintext(40.2)
    r1 = pk.MultPoly(enc(pk, p1), enc(pk, p2))
    actual = dec(pk, sk, E, r1).PolyEval()
    expected = p1.PolyEval() * p2.PolyEval()
    assert f1(actual) == f1(expected) == "44.2\n"


# ---- cmd/main.go:24-72 runPolyArithmeticCheck
@pytest.mark.gpu
def test_cli_poly_arithmetic_check(keys):
    pk, sk, E = keys
    m1, m2, m3, m4 = (E.NewPolyPlaintext(v) for v in (0.0111, 9.1, 2.75, 2.99))
    c1, c2, c3, c4 = (enc(pk, m) for m in (m1, m2, m3, m4))
    c6 = pk.NegPoly(c4)
    ev = lambda ct: dec(pk, sk, E, ct).PolyEval()
    for c, m in ((c1, m1), (c2, m2), (c3, m3), (c4, m4)):
        assert f1(ev(c)) == f1(m.PolyEval())
    assert f1(ev(pk.AddPoly(c1, c4))) == f1(m1.PolyEval() + m4.PolyEval()) == "3.0\n"          # [Add]
    r2 = pk.MultConstPoly(c2, E.NewUnbalancedPlaintext(10.0))
    assert f1(ev(r2)) == f1(m2.PolyEval() * 10.0) == "91.0\n"                                # [MultConst]
    r3 = pk.MultPoly(c3, c4)
    dr3 = ev(r3)
    assert f1(dr3) == f1(m3.PolyEval() * m4.PolyEval()) == "8.2\n"                           # [Mult]
    r4 = pk.MultConstPoly(r3, E.NewUnbalancedPlaintext(0.5))
    assert f1(ev(r4)) == f1(dr3 * 0.5) == "4.1\n"                                            # [MultConst] on level 2
    assert f1(ev(pk.AddPoly(r3, r3))) == f1(dr3 + dr3) == "16.4\n"                           # [Add] on level 2
    assert f1(ev(pk.AddPoly(c1, c6))) == f1(m1.PolyEval() - m4.PolyEval()) == "-3.0\n"       # [Add] with Neg


# ---- cmd/main.go:74-104 runSimpleCheck
@pytest.mark.gpu
def test_cli_truth_table(keys):
    pk, sk, _ = keys
    zero, one = pk.Encrypt(0), pk.Encrypt(1)
    negone = pk.Encrypt(-1)                    # Encrypt(big.NewInt(-1)): P^(x mod n) here (documented deviation)
    D = lambda ct: sk.DecryptFailSafe(ct, pk)
    N = pk.Neg
    assert [D(pk.Add(zero, zero)), D(pk.Add(zero, one)), D(pk.Add(one, one)), D(pk.Add(one, zero))] == [0, 1, 2, 1]
    assert [D(pk.Mult(zero, zero)), D(pk.Mult(zero, one)), D(pk.Mult(one, zero)), D(pk.Mult(one, one))] == [0, 0, 0, 1]
    assert [D(pk.Add(zero, N(zero))), D(pk.Add(zero, N(one))), D(pk.Add(zero, negone)), D(pk.Add(one, N(one))),
            D(pk.Add(one, N(zero)))] == [0, -1, -1, 0, 1]
    assert [D(pk.Mult(zero, N(zero))), D(pk.Mult(zero, N(one))), D(pk.Mult(one, N(zero))), D(pk.Mult(one, N(one))),
            D(pk.Mult(N(one), N(one)))] == [0, 0, 0, -1, 1]


# ---- gadgets_test.go:8-105 (newCryptoRandom(pk.N): seeded here)
@pytest.mark.gpu
def test_decryption_proof_valid(keys):                              # gadgets_test.go:8-22
    import bgn_amd
    pk, _, _ = keys
    rng = random.Random(501)
    r, v = rng.randrange(pk.N), rng.randrange(pk.N)
    ct = pk.EncryptWithRandomness(v, r)
    assert pk.CheckDecryptionProof(ct, bgn_amd.NewDecryptionProof(v, r))


@pytest.mark.gpu
def test_decryption_proof_aggregate_valid(keys):                    # gadgets_test.go:24-46
    import bgn_amd
    pk, _, _ = keys
    rng = random.Random(502)
    r1, v1, r2, v2 = (rng.randrange(pk.N) for _ in range(4))
    ct3 = pk.Add(pk.EncryptWithRandomness(v1, r1), pk.EncryptWithRandomness(v2, r2))
    assert pk.CheckDecryptionProof(ct3, bgn_amd.NewDecryptionProof(v1 + v2, r1 + r2))      # sums exceed N


@pytest.mark.gpu
def test_decryption_proof_bad(keys):                                # gadgets_test.go:48-68
    import bgn_amd
    pk, _, _ = keys
    rng = random.Random(503)
    r, r2, v = (rng.randrange(pk.N) for _ in range(3))
    ct = pk.EncryptWithRandomness(v, r)
    assert not pk.CheckDecryptionProof(ct, bgn_amd.NewDecryptionProof(v, r2))               # wrong randomness
    assert not pk.CheckDecryptionProof(ct, bgn_amd.NewDecryptionProof(r2, r))               # wrong value


@pytest.fixture(scope="module")
def keys_with_r():
    """The prover needs SecretKey.R (gadgets.go:46), which the committed fixture key does not record: a fresh
    seeded 512-bit key from the oracle's restatement of NewKeyGen (bgn.go:65-138), same constants."""
    import bgn_amd
    import bgn_ref as R
    opk, osk = R.NewKeyGen(KEYBITS, MSGSPACE, POLYBASE, DET, 2025)
    pk = bgn_amd.PublicKey(opk.p, opk.n, opk.l, R.elem_to_bytes(opk.P, opk.p), R.elem_to_bytes(opk.Q, opk.p), MSGSPACE,
                           DET, POLYBASE)
    return pk, bgn_amd.SecretKey(osk.Key, osk.R)


@pytest.mark.gpu
def test_proof_of_plaintext_knowledge_valid(keys_with_r):           # gadgets_test.go:70-84
    pk, sk = keys_with_r
    rng = random.Random(504)
    r, v = rng.randrange(pk.N), rng.randrange(pk.N)
    ct = pk.EncryptWithRandomness(v, r)
    proof = pk.NewProofOfPlaintextKnowledge(sk, v, r)
    assert pk.CheckProofOfPlaintextKnoewledge(ct, proof)


@pytest.mark.gpu
def test_proof_of_plaintext_knowledge_bad(keys_with_r):             # gadgets_test.go:85-105
    pk, sk = keys_with_r
    rng = random.Random(505)
    r, r2, v = (rng.randrange(pk.N) for _ in range(3))
    ct = pk.EncryptWithRandomness(v, r)
    assert not pk.CheckProofOfPlaintextKnoewledge(ct, pk.NewProofOfPlaintextKnowledge(sk, v, r2))   # wrong randomness
    assert not pk.CheckProofOfPlaintextKnoewledge(ct, pk.NewProofOfPlaintextKnowledge(sk, r2, r))   # wrong value
