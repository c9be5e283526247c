"""CPU: the device lane programs (bgn_amd/csrc/{fpmont,pairing,ops,codec}.hpp) compiled
for the host by the emulation harness (tests/emu) and checked against the golden
vectors / oracle.  This is a test of kernel *logic* (slot programs, exception
paths, bounds); it is not a product path — the product runs the same headers
through hipcc on the GPU (tests -m gpu)."""
import os
import random
import sys

import pytest

import bgn_ref as R
from conftest import ROOT, load_fixture

sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
import emu  # noqa: E402

NAMES = ["toy64", "k256"]


@pytest.fixture(scope="module", params=NAMES)
def ctx(request):
    fx = load_fixture(request.param)
    return fx, emu.Emu.from_fixture(fx)


def test_emu_pairing(ctx):
    fx, E = ctx
    cts = [bytes.fromhex(e["ct"]) for e in fx["encrypt"]]
    for v in fx["mult"][:4]:
        assert E.pairing(cts[v["a"]], cts[v["b"]]).hex() == v["out"]


@pytest.mark.parametrize("w", [3, 4, 5])
def test_emu_windowed_miller_loop(ctx, w):
    """The windowed Miller loop (digits 0, +-1, +-3 [, +-5, +-7] of n; dA and f_d precomputed per pairing) gives
    the Mult golden vectors, whatever the top window digit of the group order."""
    fx, E = ctx
    cts = [bytes.fromhex(e["ct"]) for e in fx["encrypt"]]
    digits = emu.wnaf(int(fx["n"], 16), w)
    assert all(abs(d) < (1 << (w - 1)) and (d == 0 or d % 2) for d in digits) and digits[-1] > 0
    E.set_window(w)
    rows = [v for v in fx["mult"] if any(cts[v["a"]]) and any(cts[v["b"]])]
    for v in rows[:4]:
        assert E.pairing_w3(cts[v["a"]], cts[v["b"]]).hex() == v["out"]
    E.set_window(5)


@pytest.mark.parametrize("window", [0, 4, 2])
def test_emu_scalar_mult_exceptional_cases(ctx, window):
    """acc == +-base inside the ladder: k = n, n+-1, n+2 for P; multiples of q1 (+-1, +2) for Q of order q1; a base
    of order 2 and of order 4 (points of the curve outside the ciphertext subgroup), the identity — for the binary
    ladder and for the 4-bit and the 2-bit windows over a per-element table of multiples."""
    fx, E = ctx
    p, n, q1 = int(fx["p"], 16), int(fx["n"], 16), int(fx["q1"], 16)
    Pw, Qw = bytes.fromhex(fx["P"]), bytes.fromhex(fx["Q"])
    Pp, Qp = R.elem_from_bytes(Pw, p), R.elem_from_bytes(Qw, p)
    rng = random.Random(1)
    L = (n.bit_length() + 7) // 8 + 1
    for k in [0, 1, 2, 3, 4, 5, 7, 15, 16, 17, 0xF0F0, n - 1, n, n + 1, n + 2, 2 * n + 5, 16 * n + 3, rng.randrange(n)]:
        assert E.g1_mul(Pw, k, L, window) == R.elem_to_bytes(R.pt_mul(Pp, k, p), p), k
    for k in [q1 - 1, q1, q1 + 1, q1 + 2, 2 * q1, 2 * q1 + 2, 3 * q1 + 2]:
        assert E.g1_mul(Qw, k, L, window) == R.elem_to_bytes(R.pt_mul(Qp, k, p), p), k
    assert E.g1_mul(bytes(2 * E.L), 5, 2, window) == bytes(2 * E.L)      # identity base
    # (0, 0) is the identity's encoding; a point of order 4 exists when l = (p+1)/n is a multiple of 4 (it is, by
    # construction): x = 1 gives y^2 = 2 — take any point T of order n*l and multiply up instead
    T = None
    for x in range(2, 200):
        rhs = (x * x * x + x) % p
        y = pow(rhs, (p + 1) // 4, p)
        if y * y % p == rhs:
            T = R.pt_mul((x, y), n * (fx["l"] // 4), p)       # order divides 4
            if T is not None and R.pt_mul(T, 2, p) is not None:
                break
            T = None
    if T is not None:
        Tw = R.elem_to_bytes(T, p)
        for k in [1, 2, 3, 4, 5, 6, 7, 8, 0x1234, 0x4444]:
            assert E.g1_mul(Tw, k, 2, window) == R.elem_to_bytes(R.pt_mul(T, k, p), p), ("order-4 base", k)


def test_emu_g1_add_run(ctx):
    """The whole l1 vector list as ONE lane's batched-inversion run (identities, doubling, cancellation inside)."""
    fx, E = ctx
    cts = [bytes.fromhex(e["ct"]) for e in fx["encrypt"]]
    a = [cts[v["a"]] for v in fx["l1"]]
    b = [cts[v["b"]] for v in fx["l1"]]
    assert [x.hex() for x in E.g1_add(a, b)] == [v["add"] for v in fx["l1"]]
    assert [x.hex() for x in E.g1_add(a, b, True)] == [v["sub"] for v in fx["l1"]]
    # the representation EAdd / ESub run in: plain residues in and out, no Montgomery conversions
    assert [x.hex() for x in E.g1_add(a, b, plain=True)] == [v["add"] for v in fx["l1"]]
    assert [x.hex() for x in E.g1_add(a, b, True, plain=True)] == [v["sub"] for v in fx["l1"]]


def test_emu_gt_ops(ctx):
    fx, E = ctx
    n = int(fx["n"], 16)
    l2 = [bytes.fromhex(v["out"]) for v in fx["mult"]]
    for v in fx["l2"]:
        assert E.gt_mul(l2[v["a"]], l2[v["b"]]).hex() == v["add"]
        assert E.gt_mul(l2[v["a"]], l2[v["b"]], True).hex() == v["sub"]
        assert E.gt_mul(l2[v["a"]], l2[v["b"]], plain_a=True).hex() == v["add"]          # wire-to-wire form
        assert E.gt_mul(l2[v["a"]], l2[v["b"]], True, plain_a=True).hex() == v["sub"]
    for v in fx["multconst_l2"]:
        assert E.gt_pow(l2[v["a"]], int(v["k"], 16), (n.bit_length() + 7) // 8).hex() == v["out"]


@pytest.mark.parametrize("wbits", [8, 16])
def test_emu_gt_fixed_base_blinding(wbits):
    """e(Q,Q)^r from the GT window table (level-2 blinding, bgn.go:302-311) == square-and-multiply, alone and
    multiplied into a result; zero digits, r = 0, odd scalar lengths, r >= n."""
    from conftest import oracle_key
    fx = load_fixture("toy64")
    E = emu.Emu.from_fixture(fx)
    opk, _ = oracle_key(fx)
    p, n = opk.p, opk.n
    g = opk.e(opk.Q, opk.Q)
    W = (n.bit_length() + wbits - 1) // wbits + 1
    KB = (W * wbits + 7) // 8
    tab = E.gt_table(R.elem_to_bytes(g, p), wbits, W)
    rng = random.Random(12)
    other = R.f2_pow(opk.e(opk.P, opk.Q), 777, p)
    for r, klen in [(0, 1), (1, 1), (0x10000, 3), (0xFFFF, 2), (n - 1, KB), (n + 5, KB), (rng.randrange(n), KB),
                    (rng.randrange(1 << 40), 5)]:
        want = R.f2_pow(g, r, p)
        assert E.gt_fixed(tab, wbits, r, klen) == R.elem_to_bytes(want, p), r
        assert E.gt_fixed(tab, wbits, r, klen, R.elem_to_bytes(other, p)) == R.elem_to_bytes(R.f2_mul(want, other, p), p), r


def test_emu_norm1_power_ladder(ctx):
    """x^k on the norm-1 group by the real-part ladder == the F_p^2 power, including k = 0, 1, the secret key, k >= n,
    and the bases 1 and -1 whose imaginary part is zero."""
    from conftest import oracle_key
    fx, E = ctx
    opk, osk = oracle_key(fx)
    p, n = opk.p, opk.n
    cts = [R.elem_from_bytes(bytes.fromhex(e["ct"]), p) for e in fx["encrypt"] if int(e["ct"], 16)]
    xs = [opk.e(cts[0], cts[1]), opk.e(cts[2], opk.P), (1, 0), (p - 1, 0)]
    rng = random.Random(9)
    for x in xs:
        assert (x[0] * x[0] + x[1] * x[1]) % p == 1
        for k in [0, 1, 2, 3, osk.Key, n - 1, n + 7, rng.randrange(n)]:
            assert E.gt_pow_norm1(R.elem_to_bytes(x, p), k) == R.elem_to_bytes(R.f2_pow(x, k, p), p), (x, k)


def test_emu_bsgs_ranges_and_signs(ctx):
    """Accept range [1, Mmax] with Mmax = B*B+B+2 (gsbs.go:77-105), zero short-cut, negative retry,
    for several baby/giant splits (the result must not depend on the split)."""
    import math
    from conftest import oracle_key
    fx, E = ctx
    opk, osk = oracle_key(fx)
    p, T = opk.p, fx["msg_space"]
    g = R.f2_pow(opk.e(opk.P, opk.P), osk.Key, p)
    B = int(math.ceil(math.sqrt(T)))
    Mmax = B * B + B + 2
    ms = [0, 1, 2, B, Mmax - 1, Mmax, Mmax + 1, -1, -2, -Mmax, -Mmax - 1, 500, -777, 2 * Mmax]
    xs = [R.elem_to_bytes(R.f2_pow(g, m % opk.n, p), p) for m in ms]
    for S in [None, 2, 4, 16, 64]:      # S = 2, 4: more than 256 giant steps -> the per-element range split is exercised
        m, st = E.bsgs(R.elem_to_bytes(g, p), T, xs, S)
        for want, got, s in zip(ms, m, st):
            if abs(want) <= Mmax:
                assert s == 0 and got == want, (S, want, got, s)
            else:
                assert s == 1, (S, want, got, s)


def test_emu_poly_accumulation(ctx):
    from conftest import oracle_key
    fx, E = ctx
    opk, _ = oracle_key(fx)
    po = fx["poly"]
    dec = lambda h: None if int(h, 16) == 0 else R.elem_from_bytes(bytes.fromhex(h), opk.p)
    ea, eb = [dec(h) for h in po["a"]], [dec(h) for h in po["b"]]
    Ew = [R.elem_to_bytes(opk.e(a, b), opk.p) for a in ea for b in eb]
    assert [o.hex() for o in E.poly_acc(Ew, po["d1"], po["d2"])] == po["out"]


@pytest.mark.parametrize("wbits", [8, 11, 16])
def test_emu_fixed_base_encrypt(wbits):
    """Window-table Encrypt (P^x * Q^r, one table entry per window, no doublings) incl. zero digits, x = 0,
    r = 0, odd scalar lengths and scalars >= n; the table itself is built by the engine's round scheme and
    checked against the oracle entry by entry."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    import oracle_c
    fx = load_fixture("toy64")
    E = emu.Emu.from_fixture(fx)
    o = oracle_c.Oracle.from_fixture(fx)
    n = int(fx["n"], 16)
    W = (n.bit_length() + wbits - 1) // wbits + 1
    KB = (W * wbits + 7) // 8          # bytes of a scalar that fills the table
    zero = bytes(2 * E.L)
    enc = lambda k, base_is_q: o.encrypt([0], [k]) if base_is_q else o.encrypt([k], None)

    def table(base_is_q):
        tab = E.build_table(wbits, W, [enc(1 << i, base_is_q) for i in range(W * wbits)])
        rng = random.Random(3 + base_is_q)
        probes = [(w, d) for w in (0, W - 1) for d in (1, 2, 3, (1 << wbits) - 1)]
        probes += [(rng.randrange(W), rng.randrange(1, 1 << wbits)) for _ in range(24)]
        for w, d in probes:
            want, _ = E.decode(enc(d << (wbits * w), base_is_q))
            i = (w << wbits) + d
            assert list(tab[2 * E.nl * i:2 * E.nl * (i + 1)]) == list(want), (w, d)
        return tab

    tP, tQ = table(False), table(True)
    rng = random.Random(8)
    cases = [(0, 0), (0, 5), (7, 0), (1 << 40, 1 << 56), (n - 1, n - 1), (n + 3, n + 1), (256, 65536)]
    cases += [(rng.randrange(1 << 40), rng.randrange(n)) for _ in range(6)]
    for x, r in cases:
        assert E.g1_fixed(tP, tQ, wbits, x, KB, r, KB) == o.encrypt([x], [r]), (x, r)
    assert E.g1_fixed(tP, tQ, wbits, 77, 2, None, 0) == o.encrypt([77], None)
    assert E.g1_fixed(tP, tQ, wbits, 0x12345, 3, 0x6789ABCDEF, 5) == o.encrypt([0x12345], [0x6789ABCDEF])
    assert E.g1_fixed(tP, tQ, wbits, 0, 1, None, 0) == zero


def test_emu_signed_window_recoding():
    """ops.hpp scalar_window_digit: every window recoded on its own gives digits in (-2^wbits, 2^wbits] that sum to
    the scalar, the table index is |digit| mod 2^wbits, and scalar_windows() windows are enough — random scalars and
    the ones that stress the carry rule (windows equal to 2^wbits, runs of all-ones windows, lengths that are a
    multiple of the window width)."""
    fx = load_fixture("toy64")
    E = emu.Emu.from_fixture(fx)
    rng = random.Random(21)
    for wbits, klen in [(4, 5), (7, 8), (7, 16), (11, 9), (16, 17), (20, 21), (20, 128), (22, 23), (22, 128)]:
        s = wbits + 1
        H = 1 << wbits
        W = E.scalar_windows(klen, wbits, s)
        assert W == (8 * klen) // s + 1
        top = 1 << (8 * klen)
        full = (8 * klen + s - 1) // s
        ks = [0, 1, top - 1, H, H + 1, H - 1, sum(H << (s * w) for w in range(full)) % top,
              (sum(H << (s * w) for w in range(full)) + 1) % top, sum(((1 << s) - 1) << (s * w) for w in range(0, full, 2)) % top,
              (H << s) | H, ((H + 1) << (2 * s)) | (H << s) | H, ((H - 1) << (2 * s)) | (H << s) | (H + 1)]
        ks += [rng.randrange(top) for _ in range(12)]
        for k in ks:
            k %= top
            total = 0
            for w in range(W):
                d, idx = E.window_digit(k, klen, wbits, s, w)
                assert -H < d <= H, (wbits, klen, hex(k), w, d)
                assert idx == abs(d) % H
                total += d << (s * w)
            assert total == k, (wbits, klen, hex(k))
        # unsigned windows are the digits themselves
        k = rng.randrange(top)
        assert sum(E.window_digit(k, klen, wbits, wbits, w)[0] << (wbits * w) for w in range(E.scalar_windows(klen, wbits, wbits))) == k


@pytest.mark.parametrize("wbits", [8, 11])
def test_emu_fixed_base_encrypt_signed_windows(wbits):
    """Encrypt over Q's table with signed windows (wbits + 1 scalar bits per window, index 0 = 2^wbits * 2^(s*w) * Q,
    negative digits add the negated entry): table entries and products against the oracle, with blinding exponents
    that hit the digit 2^wbits, its negative neighbour, zero digits after a carry, and the carry out of the top window."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    import oracle_c
    fx = load_fixture("toy64")
    E = emu.Emu.from_fixture(fx)
    o = oracle_c.Oracle.from_fixture(fx)
    n = int(fx["n"], 16)
    s = wbits + 1
    H = 1 << wbits
    nbytes = (n.bit_length() + 7) // 8
    WP = (n.bit_length() + wbits - 1) // wbits + 1
    WQ = E.scalar_windows(nbytes, wbits, s)
    tP = E.build_table(wbits, WP, [o.encrypt([1 << i], None) for i in range(WP * wbits)])
    tQ = E.build_table(wbits, WQ, [o.encrypt([0], [1 << i]) for i in range(WQ * s)], sbits=s)
    rng = random.Random(5)
    probes = [(w, d) for w in (0, WQ - 1) for d in (0, 1, 2, H - 1)] + [(rng.randrange(WQ), rng.randrange(H)) for _ in range(16)]
    for w, d in probes:
        want, _ = E.decode(o.encrypt([0], [(d or H) << (s * w)]))
        i = (w << wbits) + d
        assert list(tQ[2 * E.nl * i:2 * E.nl * (i + 1)]) == list(want), (w, d)
    top = 1 << (8 * nbytes)
    rs = [0, 1, H, H + 1, H - 1, (1 << s) - 1, 1 << s, (H << s) | H, ((H + 1) << s) | H, n - 1, n + 1, top - 1,
          sum(H << (s * w) for w in range(WQ)) % top, sum(((1 << s) - 1) << (s * w) for w in range(WQ)) % top]
    rs += [rng.randrange(n) for _ in range(6)]
    for r in rs:
        x = rng.randrange(1 << 16)
        assert E.g1_fixed(tP, tQ, wbits, x, 2, r, nbytes, sbits_q=s) == o.encrypt([x], [r]), hex(r)
    assert E.g1_fixed(tP, tQ, wbits, 9, 2, 0x6789ABCDEF, 5, sbits_q=s) == o.encrypt([9], [0x6789ABCDEF])
    # the same with the emulation's range checks on (carry-outs, and the bounds the canonical stores rely on: x3 < 4p,
    # y3 < 3p) for products without an exceptional addition: every digit non-zero, so no lane carries don't-care values
    E.lib.emu_set_g1_fixed_checks(1)
    try:
        for _ in range(12):
            r = sum(rng.choice([rng.randrange(1, H), rng.randrange(H + 1, 1 << s)]) << (s * w) for w in range(WQ - 1))
            x = sum(rng.randrange(1, 1 << wbits) << (wbits * w) for w in range(2))
            assert E.g1_fixed(tP, tQ, wbits, x, (2 * wbits + 7) // 8, r, nbytes, sbits_q=s) == o.encrypt([x], [r]), hex(r)
    finally:
        E.lib.emu_set_g1_fixed_checks(0)


def test_emu_fixed_base_chain_kernels():
    """Encrypt on the chain kernels (k_g1_fixed_chain over four accumulation chains + the two chain-sum additions, the
    launch sequence of engine.cpp fixed_base_product): 70 elements on one workgroup, so a lane owns runs of several
    virtual elements and the first pass's requests one and two elements ahead, the abscissa-only loads and the
    digit bytes fetched ahead in the second pass are all exercised — signed and unsigned windows of Q, zero scalars and
    zero windows (identity operands) among them — against the oracle."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    import oracle_c
    fx = load_fixture("toy64")
    E = emu.Emu.from_fixture(fx)
    o = oracle_c.Oracle.from_fixture(fx)
    n = int(fx["n"], 16)
    nbytes = (n.bit_length() + 7) // 8
    wb = 5
    WP = (n.bit_length() + wb - 1) // wb + 1
    tP = E.build_table(wb, WP, [o.encrypt([1 << i], None) for i in range(WP * wb)])
    rng = random.Random(77)
    count = 70
    xs = [rng.randrange(1 << 16) for _ in range(count)]
    for sbits in (wb + 1, wb):
        WQ = E.scalar_windows(nbytes, wb, sbits)
        tQ = E.build_table(wb, WQ, [o.encrypt([0], [1 << i]) for i in range(WQ * sbits)], sbits=sbits)
        rs = [rng.randrange(n) for _ in range(count)]
        xs2 = list(xs)
        xs2[0], rs[0] = 0, 0                     # the identity
        xs2[1] = 0
        rs[2] = 0
        rs[3] = (1 << (8 * nbytes)) - 1
        rs[4] = sum((1 << wb) << (sbits * w) for w in range(WQ - 1)) % (1 << (8 * nbytes))
        got = E.g1_fixed_chains(tP, tQ, wb, wb, sbits, xs2, 2, rs, nbytes)
        E_bytes = 2 * E.L
        want = o.encrypt(xs2, rs)
        assert b"".join(got) == want, sbits
        assert got[0] == bytes(E_bytes)


def test_emu_fixed_argument_pairing(ctx):
    """e(P, C) over the precomputed line table == makeL2 golden vectors (= e(C, P): the pairing is symmetric)."""
    fx, E = ctx
    tab = E.fixed_table(bytes.fromhex(fx["P"]))
    cts = [bytes.fromhex(e["ct"]) for e in fx["encrypt"]]
    for v in fx["make_l2"]:
        assert E.pairing_fixed(tab, cts[v["a"]]).hex() == v["out"]
    # the same table with every line divided by its c: one product less per step, same pairing values
    E.fixed_normalize(tab)
    for v in fx["make_l2"]:
        assert E.pairing_fixed(tab, cts[v["a"]], normalized=True).hex() == v["out"]


def test_emu_per_coefficient_line_tables(ctx):
    """MultPoly's shared first arguments: tables of several ciphertexts as columns of one limb-major table
    (fixedpair.hpp) give the Mult golden vectors, in either argument order (the pairing is symmetric)."""
    fx, E = ctx
    cts = [bytes.fromhex(e["ct"]) for e in fx["encrypt"]]
    # identity operands are overridden by the kernel around the loop, not by the table
    rows = [v for v in fx["mult"] if any(cts[v["a"]]) and any(cts[v["b"]])][:3]
    ts, tab = 5, None
    for col, v in enumerate(rows):
        tab = E.fixed_table(cts[v["a"]], ts, col + 1, tab)           # columns 1..3 of 5
    for col, v in enumerate(rows):
        assert E.pairing_fixed(tab, cts[v["b"]], ts, col + 1).hex() == v["out"]
    v = rows[0]
    tb = E.fixed_table(cts[v["b"]], 2, 1)
    assert E.pairing_fixed(tb, cts[v["a"]], 2, 1).hex() == v["out"]


def test_emu_multi_pairing_output_coefficients():
    """MultPoly as a multi-pairing (fixedpair.hpp miller_loop_fixed_multi): every output coefficient of a 3 x 3 product —
    prod_{i+j=s} e(a_i, b_j) from ONE Miller loop with a shared f^2 and one final exponentiation, tables and operands
    coefficient-major — equals the product of the oracle's single pairings, in a column other than 0, with identity
    coefficients on either side (their terms contribute the factor 1) and for the single-term coefficients s = 0, 2d-2."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    import oracle_c
    from conftest import oracle_key
    fx = load_fixture("toy64")
    E = emu.Emu.from_fixture(fx)
    o = oracle_c.Oracle.from_fixture(fx)
    opk, _ = oracle_key(fx)
    d, Qp, q = 3, 4, 2
    rng = random.Random(31)
    A = [o.encrypt([rng.randrange(1000)], [rng.randrange(1 << 40)]) for _ in range(d)]
    B = [o.encrypt([rng.randrange(1000)], [rng.randrange(1 << 40)]) for _ in range(d)]
    ts = d * Qp
    dec = lambda w: R.elem_from_bytes(w, opk.p)

    def want(a_list, b_list, s):
        acc = None
        for i in range(d):
            j = s - i
            if 0 <= j < d and a_list[i] is not None and b_list[j] is not None:
                e = opk.e(dec(a_list[i]), dec(b_list[j]))
                acc = e if acc is None else R.f2_mul(acc, e, opk.p)
        if acc is None:
            return (1).to_bytes(E.L, "big") + bytes(E.L)
        return R.elem_to_bytes(acc, opk.p)

    def table(a_list):
        tab = None
        for i, w in enumerate(a_list):
            tab = E.fixed_table(w if w is not None else A[0], ts, i * Qp + q, tab)     # (an identity's column: any table)
        return tab

    tab = table(A)
    for s in range(2 * d - 1):
        assert E.pairing_fixed_multi(tab, ts, Qp, q, A, B, s) == want(A, B, s), s
    A2 = [A[0], None, A[2]]
    B2 = [None, B[1], B[2]]
    tab2 = table(A2)
    for s in range(2 * d - 1):
        assert E.pairing_fixed_multi(tab2, ts, Qp, q, A2, B2, s) == want(A2, B2, s), ("identities", s)


@pytest.mark.parametrize("nl", [3, 10, 19, 36, 37])
def test_emu_dword_codec_matches_the_byte_codec(nl):
    """codec.hpp's dword forms (wire_element_dw: two-dword reads + byte permutes, odd lanes two bytes into a dword
    when L is odd; limbs_to_wire_dw when an element is a whole number of dwords) against Python integers for every
    L the limb count serves: limbs of random values incl. 0, 1, 2^(8L) - 1 patterns, and the re-encoded bytes
    equal the input."""
    import ctypes as C
    lib = emu.Emu.from_fixture(load_fixture("toy64")).lib
    rng = random.Random(nl)
    lmax = (emu.LIMB * nl - 9 + 7) // 8
    lmin = max(4, (emu.LIMB * (nl - 1) + 7) // 8 - 6)
    for L in range(lmin, lmax + 1):
        n = 9
        top = min(8 * L, emu.LIMB * nl)
        vals = [0, 1, (1 << top) - 1, (1 << (top - 1)) + 1] + [rng.getrandbits(top) for _ in range(2 * n - 4)]
        wire = b"".join(v.to_bytes(L, "big") for v in vals)
        limbs_out = (C.c_uint32 * (2 * nl * n))()
        wire_out = C.create_string_buffer(2 * L * n)
        assert lib.emu_codec_dw(nl, wire, L, n, limbs_out, wire_out) == 0
        for i, v in enumerate(vals):
            got = sum(int(limbs_out[i * nl + j]) << (emu.LIMB * j) for j in range(nl))
            assert all(int(limbs_out[i * nl + j]) < (1 << emu.LIMB) for j in range(nl))
            assert got == v, (nl, L, i, hex(v), hex(got))
        assert wire_out.raw == wire, (nl, L)


@pytest.mark.parametrize("name", ["k1024", "k1024b", "k2048"])
def test_emu_pairing_at_36_and_37_limbs(name):
    """1024-bit keys: 36 limbs (p of 1031 bits) and 37 (1037 bits) — the products that flush their accumulators half
    way (fpmont.hpp fp_flush; the segmented square in front of its third segment): one Mult golden vector each through
    the general and the windowed Miller loop."""
    fx = load_fixture(name)
    E = emu.Emu.from_fixture(fx)
    assert E.nl == {"k1024": 36, "k1024b": 37, "k2048": 72}[name]        # 72: four flush intervals, squarings by fp_mul
    cts = [bytes.fromhex(e["ct"]) for e in fx["encrypt"]]
    v = [v for v in fx["mult"] if any(cts[v["a"]]) and any(cts[v["b"]])][0]
    assert E.pairing(cts[v["a"]], cts[v["b"]]).hex() == v["out"]
    E.set_window(5)
    assert E.pairing_w3(cts[v["a"]], cts[v["b"]]).hex() == v["out"]


@pytest.mark.parametrize("name", ["k512", "k1024", "k1024b", "k2048"])
def test_emu_step_programs_at_the_product_limb_counts(name):
    """The Miller step programs at 19, 36, 37 and 72 limbs — the limb counts whose products flush their accumulators
    (fpmont.hpp: one flush at 36 / 37 limbs, three at 72; a sum of two products adds three product units per row) —
    with the emulation's range checks on (BGN_CHECK: accumulator capacity, carry-outs, negative differences): the
    plain NAF loop, the windowed loop at two widths, the walk over a key's line table (plain and normalized) and over
    a per-coefficient table, against the golden vectors."""
    fx = load_fixture(name)
    E = emu.Emu.from_fixture(fx)
    cts = [bytes.fromhex(e["ct"]) for e in fx["encrypt"]]
    rows = [v for v in fx["mult"] if any(cts[v["a"]]) and any(cts[v["b"]])][:2]
    for v in rows:
        assert E.pairing(cts[v["a"]], cts[v["b"]]).hex() == v["out"]
        for w in (3, 5):
            E.set_window(w)
            assert E.pairing_w3(cts[v["a"]], cts[v["b"]]).hex() == v["out"]
    if name == "k2048":
        return
    tab = E.fixed_table(bytes.fromhex(fx["P"]))
    for v in fx["make_l2"][:2]:
        assert E.pairing_fixed(tab, cts[v["a"]]).hex() == v["out"]
    E.fixed_normalize(tab)
    for v in fx["make_l2"][:2]:
        assert E.pairing_fixed(tab, cts[v["a"]], normalized=True).hex() == v["out"]
    v = rows[0]
    tb = E.fixed_table(cts[v["a"]], 2, 1)
    assert E.pairing_fixed(tb, cts[v["b"]], 2, 1).hex() == v["out"]


@pytest.mark.parametrize("name", ["toy64", "k256", "k512", "k1024", "k1024b", "k2048"])
def test_emu_products_at_extreme_limbs(name):
    """fp_mul, fp_sqr and fp_mul2 on operands whose limbs are all ones (every column of the schoolbook at its
    maximum), alternating, and a lone top limb — the patterns that decide whether a flush interval, the partial
    mid-product flush of fp_mul (nine accumulators at 36 limbs) and the peeled first row are right.  The emulation
    aborts on an accumulator that would wrap; the values are compared with Python integers."""
    fx = load_fixture(name)
    E = emu.Emu.from_fixture(fx)
    p, nl = E.p, E.nl
    R = 1 << (emu.LIMB * nl)
    Rinv = pow(R, -1, p)
    ones = (1 << (p.bit_length() - 1)) - 1                                # below p, every limb under its top bit all ones
    full = (1 << p.bit_length()) - 1 if (1 << p.bit_length()) - 1 < 2 * p else 2 * p - 1    # the same one bit longer: < 2p
    stripes = sum(emu.MASK << (emu.LIMB * j) for j in range(0, nl, 2))
    alt = ones & stripes
    lone = 1 << (p.bit_length() - 1)
    rng = random.Random(nl)
    pats = [full, ones, alt, ones & ~stripes, lone, p - 1, 2 * p - 1, 1, 0, rng.randrange(p)]
    assert all(0 <= v < 2 * p for v in pats)
    for a in pats:
        for b in pats[:7]:
            c, d = b, a
            m, sq, m2 = E.fp_products(a, b, c, d)
            assert m % p == a * b * Rinv % p and m < 2 * p, (name, "fp_mul")
            assert sq % p == a * a * Rinv % p and sq < 2 * p, (name, "fp_sqr")
            assert m2 % p == (a * b + c * d) * Rinv % p and m2 < 2 * p, (name, "fp_mul2")


@pytest.mark.parametrize("nl", [3, 10, 19, 36, 37, 72])
def test_emu_stream_codec_every_length_and_alignment(nl):
    """codec.hpp's dword-stream forms (wire_to_limbs_stream: per-lane alignment folded into one v_perm_b32 selector,
    any misalignment of the slice; limbs_to_wire_stream: whole aligned words plus, with L odd, one half word per
    element) against Python integers for every L the limb count serves: limbs of edge patterns and random values, the
    re-encoded stage equal to the input bytes, and not one byte written outside the elements."""
    import ctypes as C
    lib = emu.Emu.from_fixture(load_fixture("toy64")).lib
    rng = random.Random(100 + nl)
    lmax = (emu.LIMB * nl - 9 + 7) // 8
    lmin = max(4, (emu.LIMB * (nl - 1) + 7) // 8 - 6)
    for L in range(lmin, lmax + 1):
        n = 7
        top = min(8 * L, emu.LIMB * nl)
        vals = [0, 1, (1 << top) - 1, (1 << (top - 1)) + 1, int.from_bytes(bytes((i % 255) + 1 for i in range(L)), "big") & ((1 << top) - 1)]
        vals += [rng.getrandbits(top) for _ in range(2 * n - len(vals))]
        wire = b"".join(v.to_bytes(L, "big") for v in vals)
        stage_bytes = (2 * L * n + 3) // 4 * 4 + 16
        for mis in range(4):
            limbs_out = (C.c_uint32 * (2 * nl * n))()
            stage = C.create_string_buffer(stage_bytes)
            assert lib.emu_codec_stream(nl, wire, L, n, mis, limbs_out, stage, stage_bytes) == 0
            for i, v in enumerate(vals):
                got = sum(int(limbs_out[i * nl + j]) << (emu.LIMB * j) for j in range(nl))
                assert all(int(limbs_out[i * nl + j]) < (1 << emu.LIMB) for j in range(nl))
                assert got == v, (nl, L, mis, i, hex(v), hex(got))
            assert stage.raw[:2 * L * n] == wire, (nl, L)
            assert stage.raw[2 * L * n:] == b"\x5a" * (stage_bytes - 2 * L * n), (nl, L, "bytes past the slice written")


@pytest.mark.parametrize("name", ["toy64", "k256", "k512", "k1024", "k1024b"])
def test_emu_barrett_product_of_plain_residues(name):
    """barrett.hpp: T mod p for edge values of T (0, p - 1, p, p^2, 2 p^2, the largest number of its domain, numbers one
    off a multiple of p) and random ones; the fused level-2 Add / Sub (a*b and a*conj(b) in F_p^2 on plain residues,
    operands incl. 0, 1, p - 1) against Python integers and the fixture's L2 add / sub rows."""
    fx = load_fixture(name)
    E = emu.Emu.from_fixture(fx)
    p, nl = E.p, E.nl
    assert p.bit_length() >= emu.LIMB * (nl - 2) + 2, "the fused kernel's precondition (engine.cpp build_barrett checks the same)"
    rng = random.Random(7)
    top = p << (emu.LIMB * nl)          # barrett_reduce's domain: T < p * B^NL (the quotient then has NL limbs)
    ts = [0, 1, p - 1, p, p + 1, p * p, 2 * p * p, top - 1, top - p, (top // p) * p - 1, (top // p - 1) * p, 3 * p - 1, 4 * p - 1,
          (1 << (emu.LIMB * nl)) - 1]
    ts += [rng.randrange(top) for _ in range(20)] + [rng.randrange(p) * rng.randrange(p) for _ in range(20)]
    for t in ts:
        assert E.barrett(t) == t % p, (name, hex(t))
    Lb = E.L
    enc = lambda re, im: re.to_bytes(Lb, "big") + im.to_bytes(Lb, "big")
    edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2]
    cases = [(a0, a1, b0, b1) for a0 in edge[:4] for a1 in edge[:4] for b0 in (0, 1, p - 1) for b1 in (0, 1, p - 1)]
    cases += [tuple(rng.randrange(p) for _ in range(4)) for _ in range(24)]
    for a0, a1, b0, b1 in cases:
        want = enc((a0 * b0 - a1 * b1) % p, (a0 * b1 + a1 * b0) % p)
        assert E.gt_mul_plain(enc(a0, a1), enc(b0, b1)) == want
        want = enc((a0 * b0 + a1 * b1) % p, (a1 * b0 - a0 * b1) % p)
        assert E.gt_mul_plain(enc(a0, a1), enc(b0, b1), True) == want
    l2 = [bytes.fromhex(v["out"]) for v in fx["mult"]]
    for v in fx["l2"][:6]:
        assert E.gt_mul_plain(l2[v["a"]], l2[v["b"]]).hex() == v["add"]
        assert E.gt_mul_plain(l2[v["a"]], l2[v["b"]], True).hex() == v["sub"]
