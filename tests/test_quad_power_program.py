"""CPU: the lane-group kernels for MultConst with per-element scalars (bgn_amd/csrc/quad/quad_g1.hpp) — their step
programs (tools/coop/gen_prog.py build_quad_g1_programs) and their controllers (tests/quad_power_model.py) on Python
integers and on the lane-level model of the kernel's arithmetic, against the oracle's scalar multiplication and
F_p^2 power (bgn.go:253-291).  tests/test_gpu_multconst_quad.py compares the HIP kernels on the GPU."""
import os
import random

import pytest

from conftest import ROOT, load_fixture

import quad_model as qm
import quad_power_model as qpm


def _base(fx):
    import bgn_ref as R
    p = int(fx["p"], 16)
    return p, R.elem_from_bytes(bytes.fromhex(fx["encrypt"][3]["ct"]), p)


def test_signed_window_recoding():
    rng = random.Random(1)
    for klen in (1, 2, 5, 32):
        for k in [0, 1, 8, 9, 15, 16, 0x88, 0x89, (1 << (8 * klen)) - 1] + [rng.randrange(1 << (8 * klen)) for _ in range(50)]:
            k &= (1 << (8 * klen)) - 1
            d = qpm.recode_w4(k, klen)
            assert len(d) == 2 * klen + 1 and all(-7 <= x <= 8 for x in d)


@pytest.mark.parametrize("key", ["toy64x", "k256", "k512"])
def test_g1_ladder_on_integers_matches_the_oracle(key):
    """k * B for scalars of every shape the API sees: small constants in a long field (leading zero windows), digits
    8 and -7, full-length ones, scalars beyond the group order; with the dummy additions of a neighbour's window."""
    import bgn_ref as R
    fx = load_fixture("k256" if key == "toy64x" else key)
    p, B = _base(fx)
    n = int(fx["n"], 16)
    nl = qm.nl_for(p)
    rng = random.Random(7)
    nb = (n.bit_length() + 7) // 8
    cases = [(1, 1), (2, 1), (8, 1), (9, 1), (0x78, 1), (0x88, 2), (3 ** 20, 5), (rng.randrange(n), nb), (n - 1, nb), (n + 5, nb + 1)]
    if key == "k512":
        cases = cases[6:8]
    if key == "toy64x":
        cases = [(rng.randrange(1 << 40), 5), (rng.randrange(n), nb)]
    for k, klen in cases:
        m = qpm.PowerValueMachine(p, nl)
        got = m.g1_mul(B[0], B[1], k, klen, force_dummy_adds=(key != "toy64x"))
        want = R.pt_mul(B, k, p)
        assert got == ("pt", want[0], want[1]), (key, hex(k))


def test_g1_ladder_identity_and_exceptional_cases():
    """k = 0 gives the identity through the flag; a scalar that makes the accumulator meet -T (k = n: the last
    addition is -d*B + d*B) leaves Z = 0, which the single test at the end turns into the fallback flag."""
    fx = load_fixture("k256")
    p, B = _base(fx)
    n = int(fx["n"], 16)
    nl = qm.nl_for(p)
    nb = (n.bit_length() + 7) // 8
    assert qpm.PowerValueMachine(p, nl).g1_mul(B[0], B[1], 0, 4) == ("inf",)
    assert qpm.PowerValueMachine(p, nl).g1_mul(B[0], B[1], n, nb) == ("exc",)
    assert qpm.PowerValueMachine(p, nl).g1_mul(B[0], B[1], 2 * n, nb + 1) == ("exc",)


def test_g1_ladder_lane_model_matches_the_oracle():
    import bgn_ref as R
    fx = load_fixture("k256")
    p, B = _base(fx)
    nl = qm.nl_for(p)
    k = random.Random(3).randrange(1 << 72) | 0x8F        # digits 8 and -1 at the bottom
    m = qpm.PowerLaneMachine(p, nl)
    got = m.g1_mul(B[0], B[1], k, 9)
    want = R.pt_mul(B, k, p)
    assert got == ("pt", want[0], want[1])
    assert m.max_limb < (1 << qm.LIMB) + (1 << 12)


@pytest.mark.parametrize("key", ["k256", "k512"])
def test_gt_power_on_integers_matches_the_oracle(key):
    import bgn_ref as R
    fx = load_fixture(key)
    p = int(fx["p"], 16)
    g = R.elem_from_bytes(bytes.fromhex(fx["mult"][0]["out"]), p)
    nl = qm.nl_for(p)
    rng = random.Random(5)
    cases = [(0, 2), (1, 1), (15, 1), (16, 1), (0x1001, 3), (rng.randrange(1 << 64), 8), (rng.randrange(1 << 250), 32)]
    for k, klen in cases[: (7 if key == "k256" else 6)]:
        m = qpm.PowerValueMachine(p, nl)
        assert m.gt_pow(g[0], g[1], k, klen, wave_top=(2 * klen - 1 if k % 3 == 0 else None)) == R.f2_pow(g, k, p), hex(k)
    m = qpm.PowerLaneMachine(p, nl)
    k = rng.randrange(1 << 40)
    assert m.gt_pow(g[0], g[1], k, 5) == R.f2_pow(g, k, p)


def _window_table(R, B, p, wbits, windows):
    """tab[w][d] = d * 2^(wbits*w) * B as affine points (None: the identity)."""
    tab = []
    for w in range(windows):
        base = R.pt_mul(B, 1 << (wbits * w), p)
        row = [None]
        for d in range(1, 1 << wbits):
            row.append(R.pt_mul(base, d, p))
        tab.append([None if (pt is None or pt == (0, 0) or pt == "O") else pt for pt in row])
    return tab


def test_fixed_base_product_on_integers_matches_the_oracle_and_resolves_the_exceptional_cases():
    """P^x * Q^r over window tables (k_g1_fixed_quad's controller): random scalars; then tables in which the
    accumulator MUST meet the entry — Q = P with x = r (the sum is a doubling: the state becomes the entry, one GDBL) —
    and its negative — Q = -P with x = r (the sum is the identity)."""
    import bgn_ref as R
    fx = load_fixture("k256")
    p = int(fx["p"], 16)
    Pp = R.elem_from_bytes(bytes.fromhex(fx["P"]), p)
    Qp = R.elem_from_bytes(bytes.fromhex(fx["Q"]), p)
    nl = qm.nl_for(p)
    wbits, xlen, rlen = 4, 1, 2
    tp = _window_table(R, Pp, p, wbits, 4)
    tq = _window_table(R, Qp, p, wbits, 4)
    rng = random.Random(11)
    for x, r in [(0x5a, 0x1234), (0, 0x00f0), (0x30, 0), (rng.randrange(256), rng.randrange(65536))]:
        m = qpm.PowerValueMachine(p, nl)
        got = m.g1_fixed(tp, tq, wbits, x, xlen, r, rlen)
        want = R.pt_add(R.pt_mul(Pp, x, p), R.pt_mul(Qp, r, p), p)
        assert got == ("pt", want[0], want[1]), (hex(x), hex(r))
        assert m.doublings == 0
    assert qpm.PowerValueMachine(p, nl).g1_fixed(tp, tq, wbits, 0, xlen, 0, rlen) == ("inf",)
    # deterministic Encrypt: no r at all
    m = qpm.PowerValueMachine(p, nl)
    want = R.pt_mul(Pp, 0x77, p)
    assert m.g1_fixed(tp, tq, wbits, 0x77, 1, None, 0) == ("pt", want[0], want[1])
    # acc == entry: Q's table is P's, x = r = one digit in window 0
    m = qpm.PowerValueMachine(p, nl)
    want = R.pt_mul(Pp, 2 * 0x05, p)
    assert m.g1_fixed(tp, tp, wbits, 0x05, 1, 0x05, 1) == ("pt", want[0], want[1]) and m.doublings == 1
    # acc == -entry: Q = -P
    tn = _window_table(R, R.pt_neg(Pp, p), p, wbits, 4)
    m = qpm.PowerValueMachine(p, nl)
    assert m.g1_fixed(tp, tn, wbits, 0x05, 1, 0x05, 1) == ("inf",)
    # ... and the accumulator goes on from the identity: x = 0x35, r = 0x05 -> (5 + 48 - 5) P
    m = qpm.PowerValueMachine(p, nl)
    want = R.pt_mul(Pp, 0x30, p)
    assert m.g1_fixed(tp, tn, wbits, 0x35, 1, 0x05, 1) in (("pt", want[0], want[1]),)
    # the lane-level model on one random product
    m = qpm.PowerLaneMachine(p, nl)
    want = R.pt_add(R.pt_mul(Pp, 0xa7, p), R.pt_mul(Qp, 0x0c31, p), p)
    assert m.g1_fixed(tp, tq, wbits, 0xa7, xlen, 0x0c31, rlen) == ("pt", want[0], want[1])


def test_quad_power_schedules():
    """Three rounds per doubling, five per addition, seventeen value slots: five row blocks of LDS, three workgroups
    per CU like the Miller loop; every round at most four micro-ops that read only earlier rounds' values."""
    G, A = qpm.g1_programs()
    seg = dict(G.segments)
    assert len(seg["GDBL"]) == 3 and len(seg["GADD"]) == 5 and len(seg["GZZZ"]) == 1 and G.nslots <= 20
    assert [G.phys[s] for s in qpm.STATE] == [0, 1, 2, 3] and [G.phys[s] for s in qpm.ENTRY] == [4, 5, 6, 7, 8]
    assert len(dict(A.segments)["AFF"]) == 4 and A.nslots <= 8
    for P in (G, A):
        for name, rounds in P.segments:
            written = {}
            for r, us in enumerate(rounds):
                assert 1 <= len(us) <= 4
                for u in us:
                    for s in u.reads():
                        assert written.get(s, -1) < r, (name, s)
                for u in us:
                    written[u.dst] = r
                assert len({P.phys[u.dst] for u in us}) == len(us)


def test_generated_quad_g1_table_is_current():
    path = os.path.join(ROOT, "bgn_amd", "csrc", "quad", "quad_g1_prog.inc")
    have = open(path).read()
    tmp = path + ".check"
    try:
        qm.gen_prog.emit_quad_g1(tmp, verbose=False)
        assert open(tmp).read() == have
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
