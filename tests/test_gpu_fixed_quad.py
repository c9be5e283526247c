"""GPU: the fixed-base products P^x * Q^r on the lane groups (bgn_amd/csrc/quad/quad_g1.hpp k_g1_fixed_quad: Encrypt and
the blinding terms, bgn.go:340-353, :488-495; the default at 2048-bit keys, option quad_max_enc elsewhere) against
the golden vectors, the chain kernels and the C oracle — and with keys whose tables force the additions' exceptional
cases (Q = P: the accumulator meets the entry, a doubling; Q = -P: its negative, the identity)."""
import random

import pytest

from conftest import engine_key, load_fixture

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["k256", "k512", "k1024"])
def test_encrypt_golden_on_the_lane_groups(name, engopts):
    engopts.set("quad_max_enc", 1 << 40)
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    eng = pk.engine
    enc = fx["encrypt"]
    det = [e for e in enc if e.get("r") is None]
    rnd = [e for e in enc if e.get("r") is not None]
    if det:
        out = eng.encrypt([int(e["x"], 16) for e in det])
        assert "k_g1_fixed_quad" in eng.last_kernel_name()
        for row, e in zip(out, det):
            assert bytes(row).hex() == e["ct"], (name, e["x"])
    if rnd:
        out = eng.encrypt([int(e["x"], 16) for e in rnd], [int(e["r"], 16) for e in rnd])
        assert "k_g1_fixed_quad" in eng.last_kernel_name()
        for row, e in zip(out, rnd):
            assert bytes(row).hex() == e["ct"], (name, e["x"], e["r"])
    assert det or rnd


@pytest.mark.parametrize("name,count", [("k256", 203), ("k1024", 70)])
def test_fixed_base_products_vs_chains_and_c_oracle(name, count, engopts):
    """Random messages and full-length randomness, zero scalars among them (x = 0, r = 0, both: the identity), counts
    that leave the last workgroup ragged: the lane groups, the chain kernels and the C oracle give the same bytes;
    so do the blinded Add and MultConst that draw Q^r from the same tables."""
    import oracle_c
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    eng = pk.engine
    o = oracle_c.Oracle.from_fixture(fx)
    rng = random.Random(count)
    n = int(fx["n"], 16)
    xs = [rng.randrange(fx["msg_space"]) for _ in range(count)]
    rs = [rng.randrange(n) for _ in range(count)]
    xs[0], rs[0] = 0, 0
    xs[1] = 0
    rs[2] = 0
    xs[3], rs[3] = n, 5                        # x = n: the windows of x sum to the identity
    got = {}
    for path in ("quad", "chains"):
        engopts.set("quad_max_enc", (1 << 40) if path == "quad" else 0)
        ct = eng.encrypt(xs, rs)
        assert ("k_g1_fixed_quad" in eng.last_kernel_name()) == (path == "quad"), eng.last_kernel_name()
        det = eng.encrypt(xs)
        a, b = ct[: count // 2].tobytes(), ct[count // 2: 2 * (count // 2)].tobytes()
        blinded = eng.add(1, a, b, rs[: count // 2])
        got[path] = (ct.tobytes(), det.tobytes(), blinded.tobytes())
    assert got["quad"] == got["chains"]
    s = min(count, 16)
    assert got["quad"][0][: s * eng.elem_bytes] == o.encrypt(xs[:s], rs[:s])
    assert got["quad"][1][: s * eng.elem_bytes] == o.encrypt(xs[:s])
    assert got["quad"][0][: eng.elem_bytes] == bytes(eng.elem_bytes)            # P^0 * Q^0


@pytest.mark.parametrize("which", ["Q = P", "Q = -P"])
def test_exceptional_additions_are_resolved_in_the_kernel(which, engopts):
    """A key whose second generator is P itself (or -P): Encrypt(x, r) with equal scalars makes the accumulator meet the
    table entry (the same point: a doubling; its negative: the identity) in every window of r.  Never a real key — Q has
    order q1 — but legal points, and PBC's element_pow would add them without complaint."""
    import bgn_amd
    import oracle_c
    fx = load_fixture("k256")
    p = int(fx["p"], 16)
    Pw = bytes.fromhex(fx["P"])
    L = len(Pw) // 2
    if which == "Q = P":
        Qw = Pw
    else:
        y = int.from_bytes(Pw[L:], "big")
        Qw = Pw[:L] + ((p - y) % p).to_bytes(L, "big")
    pk = bgn_amd.PublicKey(p, int(fx["n"], 16), fx["l"], Pw, Qw, fx["msg_space"], True, fx["poly_base"])
    eng = engopts.register(pk.engine)
    o = oracle_c.Oracle(p, int(fx["n"], 16), fx["l"], Pw, Qw)
    xs = [5, 0x10005, 0xabcdef, 7, 0x7fff0001, 3]
    rs = [5, 0x10005, 0xabcdef, 9, 0x7fff0001, 0]
    res = {}
    for path in ("quad", "chains"):
        engopts.set("quad_max_enc", (1 << 40) if path == "quad" else 0)
        res[path] = eng.encrypt(xs, rs).tobytes()
    assert res["quad"] == res["chains"] == o.encrypt(xs, rs)
    if which == "Q = -P":
        E = eng.elem_bytes
        for i in (0, 1, 2, 4):
            assert res["quad"][i * E: (i + 1) * E] == bytes(E)                     # x P - x P


def _stress_exponents(n_bytes, wbits, rng, n):
    """Blinding exponents that stress the signed recoding of ops.hpp scalar_window_digit (s = wbits + 1 bits a window):
    windows equal to 2^wbits (the carry is decided further down), just above and below it, runs of all-ones windows
    (a carry turns them into zero digits), the top of the range, and random ones."""
    s, H, top = wbits + 1, 1 << wbits, 1 << (8 * n_bytes)
    full = (8 * n_bytes + s - 1) // s
    rs = [0, 1, H, H + 1, H - 1, (1 << s) - 1, 1 << s, (H << s) | H, ((H + 1) << s) | H, ((H - 1) << (2 * s)) | (H << s) | (H + 1),
          sum(H << (s * w) for w in range(full)), sum(((1 << s) - 1) << (s * w) for w in range(full)),
          sum(((1 << s) - 1) << (s * w) for w in range(0, full, 2)), top - 1, n - 1, n, n + 1]
    rs += [rng.randrange(n) for _ in range(15)]
    return [r % top for r in rs]


@pytest.mark.parametrize("name,wbits_q", [("k256", 11), ("k512", 16), ("k1024", 0)])
def test_signed_windows_of_q_on_every_kernel_family(name, wbits_q, engopts):
    """Q's table with signed windows (the default: one scalar bit more per window over the same entries, index 0 holding
    2^wbits) against unsigned ones and the C oracle: the lane groups, the chain kernels and the one-launch-per-window
    kernel each recode every window on their own and give the same bytes, with exponents built to hit the carry rule."""
    import bgn_amd
    import oracle_c
    fx = load_fixture(name)
    o = oracle_c.Oracle.from_fixture(fx)
    n = int(fx["n"], 16)
    n_bytes = (n.bit_length() + 7) // 8
    rng = random.Random(wbits_q + 1)
    eff = wbits_q or 20
    rs = _stress_exponents(n_bytes, eff, rng, n)
    xs = [rng.randrange(fx["msg_space"]) for _ in rs]
    want = o.encrypt(xs, rs)
    for signed in (1, 0):
        pk = bgn_amd.PublicKey(int(fx["p"], 16), n, fx["l"], bytes.fromhex(fx["P"]), bytes.fromhex(fx["Q"]), fx["msg_space"], True,
                               fx["poly_base"])
        eng = engopts.register(pk.engine)
        eng.set_option("fixed_signed_q", signed)
        if wbits_q:
            eng.set_option("fixed_window_bits_q", wbits_q)
        for path in ("quad", "chains", "steps"):
            eng.set_option("quad_max_enc", (1 << 40) if path == "quad" else 0)
            eng.set_option("fixed_chains", 1 if path == "steps" else 4)
            got = eng.encrypt(xs, rs).tobytes()
            kern = eng.last_kernel_name()
            assert ("k_g1_fixed_quad" in kern) == (path == "quad") and ("k_g1_fixed_step" in kern) == (path == "steps"), kern
            assert got == want, (name, signed, path)
        # exponents longer than the table serves (two more bytes than n has) take the general scalar multiplication
        long_rs = [r + (5 << (8 * n_bytes + 3)) for r in rs[:6]]
        assert eng.encrypt(xs[:6], long_rs).tobytes() == o.encrypt(xs[:6], long_rs), (name, signed, "long exponents")
        with pytest.raises(bgn_amd.BgnError):
            eng.set_option("fixed_signed_q", 1 - signed)              # shapes a table that exists now
        eng.close()
