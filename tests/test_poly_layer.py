"""poly.go beyond MultPoly — MultConstPoly, EvalPoly, AddPoly with scale alignment, SubPoly, MakePolyL2
(SURVEY.md section 8(f) rank 3).

CPU tests: the plaintext encoders (plaintext.go, CPU side of the boundary), the oracle's MultConstPoly
through decryption (the homomorphic pins of poly_test.go:92-189), and the device lane program of
polyops.hpp in the host emulator against the oracle.  GPU tests: the engine's entry points against the
oracle byte for byte, and the Go-shaped host mirror end to end."""
import os
import random
import sys

import pytest

import bgn_ref as R
from conftest import ROOT, engine_key, load_fixture, oracle_key

sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))


# ---------------------------------------------------------------- CPU: encoders
def test_plaintext_encoders_match_and_evaluate_back():
    from bgn_amd.api import unbalanced_encode
    for m in list(range(0, 400)) + [3 ** 20, 3 ** 20 - 1, 2 * 3 ** 15 + 7, 10 ** 12]:
        a = R.unbalancedEncode(m, 3)
        assert a == unbalanced_encode(m, 3), m
        assert R.poly_eval_plain(a, 3) == m and all(c in (0, 1, 2) for c in a)
        assert len(a) == 1 if m == 0 else a[-1] == 0          # the reference keeps one zero above the top digit
    for m in range(-60, 400):
        a = R.balancedEncode(m, 3)
        assert R.poly_eval_plain(a, 3) == m and all(c in (-1, 0, 1) for c in a)
    for x in [0.5, 0.25, 1 / 3, 0.1, 0.7, 0.999]:
        num, scale = R.rationalize(x, 3, 1e-4)
        assert abs(num / 3 ** scale - x) <= 1e-4
    assert R.NewUnbalancedPlaintext(7.0, 3, 3, 1e-4) == ([1, 2, 0], 0)
    digits, scale = R.NewUnbalancedPlaintext(2.5, 3, 3, 1e-4)
    assert abs(R.poly_eval_plain(digits, 3) / 3 ** scale - 2.5) <= 1e-4


def _enc_poly(opk, rng, coeffs):
    return opk.EncryptPolyCoeffs(coeffs, [rng.randrange(opk.n) for _ in coeffs])


def test_oracle_multconstpoly_decrypts_to_product():
    """Dec(MultConstPoly(Enc m, k)) evaluates to m * k (poly_test.go:127-148), level 1 and level 2."""
    fx = load_fixture("toy64")
    opk, osk = oracle_key(fx)
    opk.SetupDecryption(osk)
    rng = random.Random(5)
    for m, k in [(7, 5), (-11, 8), (0, 4), (13, 0), (26, 26)]:
        ct = _enc_poly(opk, rng, R.balancedEncode(m, 3))
        digits = R.unbalancedEncode(k, 3)
        for l2 in (False, True):
            cts = [opk.makeL2(c) for c in ct] if l2 else ct
            out = opk.MultConstPoly(cts, l2, digits)
            assert len(out) == len(ct) + len(digits)
            dec = [osk.DecryptFailSafe(c, opk) for c in out]
            assert R.poly_eval_plain(dec, 3) == m * k, (m, k, l2, dec)


# ---------------------------------------------------------------- CPU: kernel logic in the emulator
@pytest.fixture(scope="module", params=["toy64", "k256"])
def ctx(request):
    import emu
    fx = load_fixture(request.param)
    return fx, emu.Emu.from_fixture(fx)


def _wire(opk, cts):
    return [R.elem_to_bytes(c.C, opk.p) for c in cts]


@pytest.mark.parametrize("l2", [False, True])
def test_emu_poly_lin_matches_oracle(ctx, l2):
    fx, E = ctx
    opk, _ = oracle_key(fx)
    rng = random.Random(17 + l2)
    n = opk.n
    ct = _enc_poly(opk, rng, [1, -1, 0, 2, 1])
    ct[2] = opk.encryptZero()                    # identity coefficient
    ct[4] = ct[0]                                # equal operands: the accumulator meets its own addend
    if l2:
        ct = [opk.makeL2(c) for c in ct]
    level = 2 if l2 else 1
    for digits in ([1, 0], [2, 2, 1, 0], [0], [2], [1, 2, 0, 1, 2, 2, 0], [5, n - 1, 3]):
        want = _wire(opk, opk.MultConstPoly(ct, l2, digits))
        assert E.poly_lin(level, _wire(opk, ct), digits, len(digits)) == want, digits
    # EvalPoly: Horner == dot product with the powers of the base
    want = _wire(opk, [opk.EvalPoly(ct)])
    assert E.poly_lin(level, _wire(opk, ct), [opk.PolyBase ** i for i in range(len(ct))], 0) == want
    # a + (-a): the sum passes through the identity and leaves it again
    neg = opk.Neg(ct[1])
    pair = [ct[1], neg, ct[3]]
    want = _wire(opk, [opk.Add(opk.Add(pair[0], pair[1]), pair[2])])
    assert E.poly_lin(level, _wire(opk, pair), [1, 1, 1], 0) == want


# ---------------------------------------------------------------- GPU: engine entry points
@pytest.mark.gpu
@pytest.mark.parametrize("name,npoly", [("k256", 7), ("k512", 2)])
@pytest.mark.parametrize("l2", [False, True])
def test_gpu_poly_multconst_and_eval_vs_oracle(name, npoly, l2):
    fx = load_fixture(name)
    opk, _ = oracle_key(fx)
    pk, _ = engine_key(fx)
    rng = random.Random(31 + npoly + l2)
    d, level = 4, 2 if l2 else 1
    polys = []
    for q in range(npoly):
        ct = _enc_poly(opk, rng, [rng.choice([-1, 0, 1, 2]) for _ in range(d)])
        if q == 1:
            ct[0] = opk.encryptZero()
            ct[3] = ct[1]
        polys.append([opk.makeL2(c) for c in ct] if l2 else ct)
    flat = b"".join(b"".join(_wire(opk, ct)) for ct in polys)
    E = pk.engine.elem_bytes
    # one constant for every polynomial
    digits = [2, 0, 1, 0]
    out = pk.engine.poly_multconst(npoly, d, level, flat, digits).tobytes()
    want = b"".join(b"".join(_wire(opk, opk.MultConstPoly(ct, l2, digits))) for ct in polys)
    assert out == want
    # a constant per polynomial, multi-byte scalars
    per = [[rng.choice([0, 1, 2, 300, opk.n - 2]) for _ in range(3)] for _ in range(npoly)]
    out = pk.engine.poly_multconst(npoly, d, level, flat, per, shared=False).tobytes()
    want = b"".join(b"".join(_wire(opk, opk.MultConstPoly(ct, l2, dg))) for ct, dg in zip(polys, per))
    assert out == want
    # EvalPoly
    out = pk.engine.poly_eval(npoly, d, level, flat, fx["poly_base"]).tobytes()
    want = b"".join(_wire(opk, [opk.EvalPoly(ct) for ct in polys]))
    assert out == want and len(out) == npoly * E


@pytest.mark.gpu
def test_gpu_poly_layer_mirror_end_to_end():
    """The poly_test.go pins through the host mirror: MultConstPoly, AddPoly across scale factors and levels,
    SubPoly, MakePolyL2, EvalPoly — decrypted and evaluated."""
    fx = load_fixture("k256")
    pk, sk = engine_key(fx)
    pk.SetupDecryption(sk)
    ev = lambda pct: R.poly_eval_plain(sk.DecryptPoly(pct, pk), pk.PolyBase)
    a = pk.EncryptPoly(R.balancedEncode(9, 3))
    b = pk.EncryptPoly(R.balancedEncode(-4, 3))
    assert ev(pk.MultConstPoly(a, 6)) == 54
    assert ev(pk.MultConstPoly(b, -5)) == 20                     # negative constant: NegPoly of the product
    assert ev(pk.MultConstPoly(pk.MultPoly(a, b), 2)) == -72     # level-2 operand
    assert ev(pk.AddPoly(a, b)) == 5 and ev(pk.SubPoly(a, b)) == 13
    l2 = pk.MakePolyL2(a)
    assert l2.L2 and ev(l2) == 9
    assert ev(pk.AddPoly(pk.MultPoly(a, b), a)) == -36 + 9       # mixed levels lift the level-1 operand
    # scale alignment (poly.go:209-226): b at scale 2 means -4 / 3^2; a is brought to the same scale
    b2 = pk.EncryptPoly(R.balancedEncode(-4, 3), scale=2)
    s = pk.AddPoly(a, b2)
    assert s.ScaleFactor == 2 and ev(s) == 9 * 9 - 4
    # EvalPoly == direct evaluation, on both levels
    m, st = pk.engine.decrypt(1, pk.EvalPoly(a).C)
    assert not st.any() and int(m[0]) == 9
    m, st = pk.engine.decrypt(2, pk.EvalPoly(pk.MultPoly(a, b)).C)
    assert not st.any() and int(m[0]) == -36


@pytest.mark.gpu
def test_gpu_nondeterministic_key_blinds_poly_products():
    """A key with Deterministic == false (the reference's production mode, bgn.go:37): MultPoly's coefficients are
    blinded with fresh randomness — different bytes on every call, the same plaintext product."""
    import bgn_amd
    fx = load_fixture("k256")
    det, sk = engine_key(fx)
    pk = bgn_amd.PublicKey(int(fx["p"], 16), int(fx["n"], 16), fx["l"], bytes.fromhex(fx["P"]), bytes.fromhex(fx["Q"]),
                           fx["msg_space"], False, fx["poly_base"])
    pk.SetupDecryption(sk)
    ev = lambda pct: R.poly_eval_plain(sk.DecryptPoly(pct, pk), pk.PolyBase)
    a = pk.EncryptPoly(R.balancedEncode(7, 3))
    b = pk.EncryptPoly(R.balancedEncode(-5, 3))
    p1, p2 = pk.MultPoly(a, b), pk.MultPoly(a, b)
    assert ev(p1) == -35 and ev(p2) == -35
    assert [c.C for c in p1.Coefficients] != [c.C for c in p2.Coefficients]
    # the deterministic product of the same operands is the unblinded value: equal plaintext, different bytes
    p0 = det.MultPoly(a, b)
    assert ev(p0) == -35 and [c.C for c in p0.Coefficients] != [c.C for c in p1.Coefficients]
    assert ev(pk.AddPoly(p1, a)) == -28 and ev(pk.SubPoly(a, b)) == 12
    # MultConstPoly and EvalPoly are blinded the same way
    c1, c2 = pk.MultConstPoly(a, 4), pk.MultConstPoly(a, 4)
    assert ev(c1) == 28 and [c.C for c in c1.Coefficients] != [c.C for c in c2.Coefficients]
    e1, e2 = pk.EvalPoly(a), pk.EvalPoly(a)
    assert e1.C != e2.C and sk.Decrypt(e1, pk) == 7 and sk.Decrypt(e2, pk) == 7
