"""GPU: HIP engine (through the C ABI) against the golden vectors and the oracle."""
import random

import numpy as np
import pytest

import bgn_ref as R
from conftest import engine_key, load_fixture, oracle_key, KEYS

pytestmark = pytest.mark.gpu


def H(fx_hex_list):
    return b"".join(bytes.fromhex(h) for h in fx_hex_list)


@pytest.mark.parametrize("name", KEYS)
def test_mult_golden(name):
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    cts = [e["ct"] for e in fx["encrypt"]]
    a = H([cts[v["a"]] for v in fx["mult"]])
    b = H([cts[v["b"]] for v in fx["mult"]])
    out = pk.engine.mult(a, b)
    for row, v in zip(out, fx["mult"]):
        assert bytes(row).hex() == v["out"], f"{name}: Mult({v['a']},{v['b']})"


@pytest.mark.parametrize("name", KEYS)
def test_make_l2_golden(name):
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    cts = [e["ct"] for e in fx["encrypt"]]
    out = pk.engine.make_l2(H([cts[v["a"]] for v in fx["make_l2"]]))
    for row, v in zip(out, fx["make_l2"]):
        assert bytes(row).hex() == v["out"]


@pytest.mark.parametrize("name,count", [("toy64", 700), ("k256", 300), ("k512", 40), ("k1024", 8)])
def test_mult_random_vs_oracle(name, count):
    """Seeded random G1 pairs, ragged count (not a multiple of the block size)."""
    fx = load_fixture(name)
    opk, _ = oracle_key(fx)
    pk, _ = engine_key(fx)
    rng = random.Random(1234)
    # a small pool of oracle points, paired up pseudo-randomly (oracle cost stays in seconds)
    pool = [R.pt_mul(opk.P, rng.randrange(1, opk.n), opk.p) for _ in range(6)]
    pool.append(None)
    ia = [rng.randrange(len(pool)) for _ in range(count)]
    ib = [rng.randrange(len(pool)) for _ in range(count)]
    cache = {}
    a = b"".join(R.elem_to_bytes(pool[i], opk.p) for i in ia)
    b = b"".join(R.elem_to_bytes(pool[i], opk.p) for i in ib)
    out = pk.engine.mult(a, b)
    for row, i, k in zip(out, ia, ib):
        if (i, k) not in cache:
            cache[(i, k)] = R.elem_to_bytes(opk.e(pool[i], pool[k]), opk.p)
        assert bytes(row) == cache[(i, k)]


def test_mult_empty_batch():
    fx = load_fixture("toy64")
    pk, _ = engine_key(fx)
    assert pk.engine.mult(b"", b"").shape == (0, pk.engine.elem_bytes)


@pytest.mark.parametrize("name", ["toy64", "k1024"])
def test_every_entry_point_takes_an_empty_batch(name):
    """count = 0 is a valid call on every entry point, host and device arrays: success, nothing launched with an empty
    grid, nothing written (the one-launch Add / Sub / Neg kernels and the 2-bit-window MultConst of round 6 included),
    and the context works afterwards."""
    import numpy as np
    import torch
    fx = load_fixture(name)
    pk, sk = engine_key(fx)
    pk.SetupDecryption(sk)
    eng = pk.engine
    EB = eng.elem_bytes
    shape = (0, EB)
    for lvl in (1, 2):
        assert eng.add(lvl, b"", b"").shape == shape and eng.sub(lvl, b"", b"").shape == shape
        assert eng.neg(lvl, b"").shape == shape
        assert eng.multconst(lvl, b"", []).shape == shape
        m, st = eng.decrypt(lvl, b"")
        assert len(m) == 0 and len(st) == 0
        assert len(eng.validate(lvl, b"")) == 0
    assert eng.encrypt([], []).shape == shape and eng.make_l2(b"").shape == shape
    assert eng.poly_mult(0, 2, 2, b"", b"").shape == shape
    dev = torch.device("cuda", 0)
    guard = torch.full((64,), 0xEE, dtype=torch.uint8, device=dev)
    k = torch.zeros(8, dtype=torch.uint8, device=dev)
    m = torch.zeros(1, dtype=torch.int64, device=dev)
    for lvl in (1, 2):
        eng.add_dev(lvl, guard, guard, guard, 0)
        eng.sub_dev(lvl, guard, guard, guard, 0)
        eng.neg_dev(lvl, guard, guard, 0)
        eng.multconst_dev(lvl, guard, k, 5, guard, 0)
        eng.validate_dev(lvl, guard, guard, 0)
        eng.decrypt_dev(lvl, guard, m, guard, 0)
    eng.mult_dev(guard, guard, guard, 0)
    eng.make_l2_dev(guard, guard, 0)
    eng.encrypt_dev(k, 5, None, 0, guard, 0)
    torch.cuda.synchronize()
    assert bool((guard == 0xEE).all().item())
    cts = [bytes.fromhex(e["ct"]) for e in fx["encrypt"]]
    assert bytes(eng.add(1, cts[0], cts[1])[0]).hex() == next(v["add"] for v in fx["l1"] if v["a"] == 0 and v["b"] == 1) \
        if any(v["a"] == 0 and v["b"] == 1 for v in fx["l1"]) else True


# ---------------------------------------------------------------------------
# Encrypt / Add / Sub / Neg / MultConst against the golden vectors
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", KEYS)
def test_encrypt_golden(name):
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    xs = [int(e["x"], 16) for e in fx["encrypt"]]
    rs = [int(e["r"], 16) for e in fx["encrypt"]]
    out = pk.engine.encrypt(xs, rs)
    for row, e in zip(out, fx["encrypt"]):
        assert bytes(row).hex() == e["ct"], f"{name}: Encrypt(x={e['x']}, r={e['r']})"
    # EncryptDeterministic (bgn.go:325-331) == r = 0
    det = pk.engine.encrypt(xs[:4], None)
    ref = pk.engine.encrypt(xs[:4], [0, 0, 0, 0])
    assert (det == ref).all()
    assert not det[0].any()                      # P^0 = identity = zero bytes


@pytest.mark.parametrize("name", KEYS)
def test_l1_add_sub_neg_golden(name):
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    cts = [e["ct"] for e in fx["encrypt"]]
    a, b = H([cts[v["a"]] for v in fx["l1"]]), H([cts[v["b"]] for v in fx["l1"]])
    for got, key in [(pk.engine.add(1, a, b), "add"), (pk.engine.sub(1, a, b), "sub"), (pk.engine.neg(1, a), "neg")]:
        for row, v in zip(got, fx["l1"]):
            assert bytes(row).hex() == v[key], f"{name}: L1 {key}({v['a']},{v['b']})"


@pytest.mark.parametrize("name", KEYS)
def test_l2_add_sub_neg_golden(name):
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    l2 = [v["out"] for v in fx["mult"]]
    a, b = H([l2[v["a"]] for v in fx["l2"]]), H([l2[v["b"]] for v in fx["l2"]])
    for got, key in [(pk.engine.add(2, a, b), "add"), (pk.engine.sub(2, a, b), "sub"), (pk.engine.neg(2, a), "neg")]:
        for row, v in zip(got, fx["l2"]):
            assert bytes(row).hex() == v[key], f"{name}: L2 {key}({v['a']},{v['b']})"


@pytest.mark.parametrize("name", KEYS)
def test_multconst_golden(name):
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    cts = [e["ct"] for e in fx["encrypt"]]
    l2 = [v["out"] for v in fx["mult"]]
    for lvl, key, src in [(1, "multconst_l1", cts), (2, "multconst_l2", l2)]:
        out = pk.engine.multconst(lvl, H([src[v["a"]] for v in fx[key]]), [int(v["k"], 16) for v in fx[key]])
        for row, v in zip(out, fx[key]):
            assert bytes(row).hex() == v["out"], f"{name}: MultConst L{lvl} k={v['k']}"


@pytest.mark.parametrize("name", ["toy64", "k256"])
def test_blinded_ops_vs_oracle(name):
    """Non-deterministic mode with explicit randomness (bgn.go:302-311, 466-474, 488-495, 260-288)."""
    fx = load_fixture(name)
    opk, _ = oracle_key(fx)
    opk.Deterministic = False
    pk, _ = engine_key(fx)
    rng = random.Random(77)
    dec = lambda h: None if int(h, 16) == 0 else R.elem_from_bytes(bytes.fromhex(h), opk.p)
    cts = [R.Ciphertext(dec(e["ct"]), False) for e in fx["encrypt"]]
    w = lambda c: R.elem_to_bytes(c.C, opk.p)
    idx = [(1, 2), (3, 4), (0, 5), (6, 6)]
    rs = [rng.randrange(opk.n) for _ in idx]
    A = b"".join(w(cts[i]) for i, _ in idx)
    B = b"".join(w(cts[k]) for _, k in idx)
    got = pk.engine.add(1, A, B, rs)
    for row, (i, k), r in zip(got, idx, rs):
        assert bytes(row) == w(opk.Add(cts[i], cts[k], r))
    got = pk.engine.mult(A, B, rs)
    l2 = []
    for row, (i, k), r in zip(got, idx, rs):
        c = opk.Mult(cts[i], cts[k], r)
        l2.append(c)
        assert bytes(row) == w(c)
    A2 = b"".join(w(c) for c in l2)
    got = pk.engine.sub(2, A2, A2[::-1][: len(A2)][::-1], rs)          # a - a, blinded
    for row, c, r in zip(got, l2, rs):
        assert bytes(row) == w(opk.Sub(c, c, r))
    got = pk.engine.multconst(1, A, [3, 0, 77, 5], rs)
    for row, (i, _), k, r in zip(got, idx, [3, 0, 77, 5], rs):
        assert bytes(row) == w(opk.MultConst(cts[i], k, r))
    got = pk.engine.multconst(2, A2, [3, 0, 77, 5], rs)
    for row, c, k, r in zip(got, l2, [3, 0, 77, 5], rs):
        assert bytes(row) == w(opk.MultConst(c, k, r))
    opk.Deterministic = True


def test_aggregate_identity_gpu():
    """gadgets_test.go:24-46 on the engine: Add(Enc(v1,r1),Enc(v2,r2)) == Enc(v1+v2, r1+r2) with v, r < N
    (so the exponents of the right-hand side exceed N)."""
    fx = load_fixture("k512")
    pk, _ = engine_key(fx)
    n = pk.N
    rng = random.Random(5)
    v1, r1, v2, r2 = (rng.randrange(n) for _ in range(4))
    c1, c2 = pk.EncryptWithRandomness(v1, r1), pk.EncryptWithRandomness(v2, r2)
    assert pk.Add(c1, c2).C == pk.EncryptWithRandomness(v1 + v2, r1 + r2).C


@pytest.mark.parametrize("name,count", [("toy64", 70001), ("k256", 3000)])
def test_l1_add_large_ragged_vs_c_oracle(name, count):
    """More than one element per lane (batched inversion runs > 1) and a ragged tail,
    with identities, doublings and cancellations sprinkled in."""
    import oracle_c
    fx = load_fixture(name)
    o = oracle_c.Oracle.from_fixture(fx)
    pk, _ = engine_key(fx)
    E = pk.engine.elem_bytes
    rng = random.Random(42)
    pool = [o.encrypt([rng.randrange(1 << 30)], [rng.randrange(int(fx["n"], 16))]) for _ in range(12)]
    pool.append(bytes(E))                                   # identity
    ia = [rng.randrange(len(pool)) for _ in range(count)]
    ib = [rng.randrange(len(pool)) for _ in range(count)]
    a = b"".join(pool[i] for i in ia)
    b = b"".join(pool[i] for i in ib)
    assert pk.engine.add(1, a, b).tobytes() == o.add(1, a, b)
    assert pk.engine.sub(1, a, b).tobytes() == o.add(1, a, b, True)


@pytest.mark.parametrize("name,count", [("toy64", 900), ("k512", 130)])
def test_encrypt_random_vs_c_oracle(name, count):
    import oracle_c
    fx = load_fixture(name)
    o = oracle_c.Oracle.from_fixture(fx)
    pk, _ = engine_key(fx)
    rng = random.Random(7)
    n = int(fx["n"], 16)
    xs = [rng.randrange(1 << 40) % n for _ in range(count)]
    rs = [rng.randrange(n) for _ in range(count)]
    assert pk.engine.encrypt(xs, rs).tobytes() == o.encrypt(xs, rs)


@pytest.mark.parametrize("name", ["toy64", "k256", "k512"])
def test_g1_runs_longer_than_one(name):
    """Batches above 65536 elements give every lane a run of several elements sharing one inversion (EAdd), and
    the window tables are built by such runs: check a 150k-element EAdd against the oracle's distinct sums and
    Encrypt of scalars whose digits reach the last-built table entries."""
    import numpy as np
    import oracle_c
    fx = load_fixture(name)
    o = oracle_c.Oracle.from_fixture(fx)
    pk, _ = engine_key(fx)
    eng = pk.engine
    EB = eng.elem_bytes
    pool = [bytes.fromhex(e["ct"]) for e in fx["encrypt"]][:7]
    count = 150001
    ia = np.arange(count) % len(pool)
    ib = (np.arange(count) * 3 + 1) % len(pool)
    P = np.frombuffer(b"".join(pool), dtype=np.uint8).reshape(len(pool), EB)
    got = np.asarray(eng.add(1, P[ia].reshape(-1), P[ib].reshape(-1))).reshape(count, EB)
    want = {}
    for i in range(len(pool)):
        j = (i * 3 + 1) % len(pool)
        want[i] = np.frombuffer(o.add(1, pool[i], pool[j]), dtype=np.uint8)
    W = np.stack([want[i] for i in range(len(pool))])
    assert (got == W[ia]).all()
    n = int(fx["n"], 16)
    xs = [0xFFFF, 0xFFFE, 0x8001, 0xFFFF0000FFFF, 4095, 5000, 32767, (1 << 40) - 1, n - 1]
    rs = [n - 1, 0xFFFF, 1, 0xFFFFFFFF, 2, 3, 4, 5, 0xFFFF7FFF]
    assert eng.encrypt(xs, rs).tobytes() == o.encrypt(xs, rs)


# ---------------------------------------------------------------------------
# Decrypt (BSGS) and MultPoly
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", KEYS)
def test_decrypt_golden(name):
    """Fixture decryptions: zero, positive, boundary B*B+B+2, out of range (error), negatives via retry; L1 and L2."""
    fx = load_fixture(name)
    pk, sk = engine_key(fx)
    pk.SetupDecryption(sk)
    for lvl in (1, 2):
        items = [d for d in fx["decrypt"] if d["level"] == lvl]
        m, st = pk.engine.decrypt(lvl, H([d["ct"] for d in items]))
        for d, mi, si in zip(items, m, st):
            if d["expect"] is None:
                assert si == 1, f"{name}: m={d['m']} must be out of bounds"
            else:
                assert si == 0 and int(mi) == d["expect"], f"{name}: Decrypt(m={d['m']}) -> {mi}, status {si}"


def test_truth_table_like_cmd_main():
    """cmd/main.go:79-104 on the engine (deterministic mode, 512-bit key like the reference's tests)."""
    fx = load_fixture("k512")
    pk, sk = engine_key(fx)
    pk.SetupDecryption(sk)
    zero, one = pk.Encrypt(0), pk.Encrypt(1)
    D = lambda c: sk.DecryptFailSafe(c, pk)
    assert [D(pk.Add(zero, zero)), D(pk.Add(zero, one)), D(pk.Add(one, one)), D(pk.Add(one, zero))] == [0, 1, 2, 1]
    assert [D(pk.Mult(zero, zero)), D(pk.Mult(zero, one)), D(pk.Mult(one, zero)), D(pk.Mult(one, one))] == [0, 0, 0, 1]
    assert D(pk.Add(zero, pk.Neg(zero))) == 0 and D(pk.Add(zero, pk.Neg(one))) == -1
    assert D(pk.Add(one, pk.Neg(one))) == 0 and D(pk.Add(one, pk.Neg(zero))) == 1
    assert D(pk.Mult(zero, pk.Neg(one))) == 0 and D(pk.Mult(one, pk.Neg(one))) == -1
    assert D(pk.Mult(pk.Neg(one), pk.Neg(one))) == 1
    # mixed-level Add lifts the L1 operand (bgn.go:447-453)
    assert D(pk.Add(pk.Mult(one, one), one)) == 2
    import bgn_amd
    with pytest.raises(bgn_amd.api.DecryptError):
        sk.Decrypt(pk.Encrypt(5000), pk)            # > B*B+B+2 for T = 1021


@pytest.mark.parametrize("name", KEYS)
def test_poly_mult_golden(name):
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    po = fx["poly"]
    out = pk.engine.poly_mult(1, po["d1"], po["d2"], H(po["a"]), H(po["b"]))
    assert [bytes(r).hex() for r in out] == po["out"]


def test_poly_mult_many_vs_c_oracle_and_decrypt():
    """Several polynomials in one call (sharding unit = polynomial), then DecryptPoly == plaintext convolution
    (poly_test.go:172-189)."""
    import oracle_c
    fx = load_fixture("k256")
    o = oracle_c.Oracle.from_fixture(fx)
    pk, sk = engine_key(fx)
    pk.SetupDecryption(sk)
    rng = random.Random(3)
    n = int(fx["n"], 16)
    npoly, d1, d2 = 5, 4, 3
    ca = [[rng.choice([-1, 0, 1]) for _ in range(d1)] for _ in range(npoly)]
    cb = [[rng.choice([-1, 0, 1]) for _ in range(d2)] for _ in range(npoly)]
    flat = lambda cs: [x % n for row in cs for x in row]
    ea = o.encrypt(flat(ca), [rng.randrange(n) for _ in range(npoly * d1)])
    eb = o.encrypt(flat(cb), [rng.randrange(n) for _ in range(npoly * d2)])
    out = pk.engine.poly_mult(npoly, d1, d2, ea, eb)
    assert out.tobytes() == o.poly_mult(npoly, d1, d2, ea, eb)
    m, st = pk.engine.decrypt(2, out.tobytes())
    assert not st.any()
    for q in range(npoly):
        conv = [0] * (d1 + d2)
        for i, x in enumerate(ca[q]):
            for k, y in enumerate(cb[q]):
                conv[i + k] += x * y
        assert [int(v) for v in m[q * (d1 + d2):(q + 1) * (d1 + d2)]] == conv


T1 = {"poly_tables": "1"}                    # the table path whatever the size (by default products of up to 65536
                                                 # coefficient pairs pair directly: one pairing's latency)


@pytest.mark.parametrize("d1,d2,env", [
    (4, 3, T1),                                  # tables on the second polynomial (fewer coefficients)
    (2, 5, T1),                                  # tables on the first
    (3, 3, {**T1, "poly_table_max_mb": "3"}),      # table budget of 3 MB = 64 columns: 21 polynomials per chunk
    # whole rounds of tables first, the remainder pairs directly (rounds of 64 / 16 lanes instead of 65536 here):
    (3, 3, {**T1, "poly_table_max_mb": "3", "poly_round": "64"}),                # 70 = 3 x 21 + 7: 63 pairs direct
    (2, 2, {**T1, "poly_round": "32", "npoly": "37"}),                               # one chunk of 32 + 5 direct
    (2, 2, {**T1, "poly_table_max_mb": "3", "poly_round": "16", "npoly": "43"}), # 32, then 8 by tables + 3 direct
    (4, 3, {"poly_tables": "0"}),            # direct d1*d2 full pairings
    (4, 3, {}), (2, 5, {}),                      # the default dispatch at this size
    (1, 4, T1), (4, 1, T1), (1, 1, {}),          # degenerate shapes
    (2, 2, T1),                                  # square, below the Karatsuba threshold
    (4, 4, T1), (8, 8, T1),                      # Karatsuba: one and two levels down to 2 x 2
    (6, 6, T1),                                  # one level, odd leaves 3 x 3
    (8, 8, {**T1, "poly_karatsuba": "0"}),   # the same product without it
    (8, 8, {**T1, "poly_levels": "1"}),      # one level where two are possible (engine.cpp poly_plan_levels picks by cost)
    (16, 16, {"poly_levels": "2", "npoly": "5"}), (16, 16, {"npoly": "5"}),
    (4, 4, {"poly_tables": "0"}),            # Karatsuba over direct pairings at the leaves
    # square leaves run as multi-pairings by default (one lane per output coefficient, fixedpair.hpp
    # miller_loop_fixed_multi): the one-lane-per-pair walk behind option poly_multi = 0 stays covered
    (4, 4, {**T1, "poly_multi": "0"}), (8, 8, {**T1, "poly_multi": "0"}), (3, 3, {**T1, "poly_multi": "0"}),
    (3, 3, T1), (5, 5, {**T1, "npoly": "67"}),   # odd square leaves as multi-pairings; a ragged last group of 64 products
])
def test_poly_mult_table_paths_vs_c_oracle(d1, d2, env, engopts):
    """MultPoly over per-coefficient line tables (fixedpair.hpp) == the oracle's d1*d2 full pairings +
    accumulation, for either table side, chunked tables, whole rounds + direct remainder, identity coefficients
    (Enc(0) deterministic) and the direct path."""
    import oracle_c
    env = dict(env)
    npoly = int(env.pop("npoly", 70 if env.get("poly_table_max_mb") else 9))
    for k, v in env.items():
        engopts.set(k, v)
    fx = load_fixture("k256")
    o = oracle_c.Oracle.from_fixture(fx)
    pk, _ = engine_key(fx)
    rng = random.Random(100 * d1 + d2)
    n = int(fx["n"], 16)
    xa = [rng.choice([0, 1, 2, n - 1]) for _ in range(npoly * d1)]
    xb = [rng.choice([0, 1, 2, n - 1]) for _ in range(npoly * d2)]
    # r = 0 with x = 0 is the identity of G1 (encryptZero in deterministic mode, bgn.go:562-564)
    ea = o.encrypt(xa, [rng.choice([0, rng.randrange(n)]) for _ in xa])
    eb = o.encrypt(xb, [rng.choice([0, rng.randrange(n)]) for _ in xb])
    out = pk.engine.poly_mult(npoly, d1, d2, ea, eb)
    assert out.tobytes() == o.poly_mult(npoly, d1, d2, ea, eb)
    if env.get("poly_tables") == "1" and d1 * d2 >= 2 and "poly_round" not in env:
        assert "fixedpair" in pk.engine.last_kernel_name()


def test_decrypt_large_message_space_1024():
    """T = 2^40 (BASELINE configs[3]): uniform messages incl. negatives, through Encrypt -> Decrypt on the engine;
    exercises the re-balanced 2^26-entry HBM table."""
    fx = load_fixture("k1024")
    pk, sk = engine_key(fx)
    pk.SetupDecryption(sk)
    rng = random.Random(11)
    T = fx["msg_space"]
    msgs = [0, 1, T - 1, T, (1 << 20) ** 2 + (1 << 20) + 2, -1, -(T - 1)] + [rng.randrange(T) for _ in range(20)] + \
           [-rng.randrange(T) for _ in range(5)]
    n = pk.N
    cts = pk.engine.encrypt([m % n for m in msgs], [rng.randrange(n) for _ in msgs])
    m, st = pk.engine.decrypt(1, cts)
    assert [int(v) for v in m] == msgs and not st.any()
    # one beyond the reference's reach -> error
    m2, st2 = pk.engine.decrypt(1, pk.engine.encrypt([(1 << 40) + (1 << 20) + 3], [5]))
    assert st2[0] == 1


def test_repeated_setup_keeps_key_tables_alive():
    """Regression: a second SetupDecryption must not invalidate the fixed-base tables used by Encrypt."""
    fx = load_fixture("k256")
    pk, sk = engine_key(fx)
    for _ in range(3):
        pk.SetupDecryption(sk)
        c = pk.EncryptWithRandomness(7, 123456789)
        assert sk.Decrypt(c, pk) == 7


@pytest.mark.parametrize("split", ["0", "1"])
def test_mult_runs_with_shared_inversion(split, engopts):
    """count > 2*65536 makes every lane own a run of pairings that share one F_p inversion
    (Montgomery's trick in the final exponentiation); ragged tail; identities inside the runs.
    split = 0: one launch, the last run ragged; 1 (the default): three whole rounds, then the 77 on their own."""
    engopts.set("split_rounds", split)
    fx = load_fixture("toy64")
    opk, _ = oracle_key(fx)
    pk, _ = engine_key(fx)
    rng = random.Random(99)
    pool = [R.pt_mul(opk.P, rng.randrange(1, opk.n), opk.p) for _ in range(6)] + [None]
    wires = [R.elem_to_bytes(x, opk.p) for x in pool]
    count = 3 * 65536 + 77           # run = 3, ragged last run
    ia = np.array([rng.randrange(len(pool)) for _ in range(count)])
    ib = np.array([rng.randrange(len(pool)) for _ in range(count)])
    tab = np.frombuffer(b"".join(wires), dtype=np.uint8).reshape(len(pool), -1)
    out = pk.engine.mult(tab[ia].tobytes(), tab[ib].tobytes())
    exp = np.zeros((len(pool), len(pool), out.shape[1]), dtype=np.uint8)
    for i in range(len(pool)):
        for k in range(len(pool)):
            exp[i, k] = np.frombuffer(R.elem_to_bytes(opk.e(pool[i], pool[k]), opk.p), dtype=np.uint8)
    assert (out == exp[ia, ib]).all()


def test_encrypt_runs_with_shared_inversion():
    """count > 2*65536: every lane owns a run of encryptions sharing one Jacobian->affine inversion; zero
    plaintext/randomness (identity results) inside the runs; ragged tail."""
    import oracle_c
    fx = load_fixture("toy64")
    o = oracle_c.Oracle.from_fixture(fx)
    pk, _ = engine_key(fx)
    rng = random.Random(17)
    n = int(fx["n"], 16)
    count = 2 * 65536 + 77
    xs = [rng.randrange(1 << 20) for _ in range(count)]
    rs = [rng.randrange(n) for _ in range(count)]
    for i in range(0, count, 1000):
        xs[i], rs[i] = 0, 0                       # identity
    assert pk.engine.encrypt(xs, rs).tobytes() == o.encrypt(xs, rs)


def test_argument_errors_are_reported_not_fatal():
    import ctypes
    from bgn_amd import _lib
    fx = load_fixture("toy64")
    pk, _ = engine_key(fx)
    lib, h = pk.engine._lib, pk.engine._h
    buf = ctypes.create_string_buffer(4 * pk.engine.elem_bytes)
    assert lib.bgn_add_batch(h, 2, 3, buf, buf, None, 0, buf) == _lib.BGN_E_ARG          # bad level
    assert b"level" in lib.bgn_last_error()
    assert lib.bgn_mult_batch(h, 2, None, buf, None, 0, buf) == _lib.BGN_E_ARG           # null operand
    assert lib.bgn_mult_batch_dev(h, (1 << 28) + 1, buf, buf, None, 0, buf, None) == _lib.BGN_E_ARG
    assert b"too large" in lib.bgn_last_error()
    fresh = load_fixture("k256")
    import bgn_amd
    pk2 = bgn_amd.PublicKey(int(fresh["p"], 16), int(fresh["n"], 16), fresh["l"], bytes.fromhex(fresh["P"]),
                            bytes.fromhex(fresh["Q"]), fresh["msg_space"])
    m = (ctypes.c_int64 * 1)()
    st = (ctypes.c_uint8 * 1)()
    rc = pk2.engine._lib.bgn_decrypt_batch(pk2.engine._h, 1, 1, buf, m, st)              # no secret key yet
    assert rc == _lib.BGN_E_STATE
    pk2.engine.set_secret(int(fresh["q1"], 16))
    rc = pk2.engine._lib.bgn_decrypt_batch(pk2.engine._h, 1, 1, buf, m, st)              # gsbs.go:56-58
    assert rc == _lib.BGN_E_STATE and b"DL tables not computed!" in lib.bgn_last_error()
    pk2.engine.close()


@pytest.mark.parametrize("name", ["k256", "k1024"])
def test_validate_batch(name):
    """bgn_validate_batch: range and curve / norm-1 membership of untrusted encodings (the reference accepts any
    bytes, ciphertext.go:100; PBC maps an invalid point to O silently)."""
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    p = int(fx["p"], 16)
    L = fx["fp_bytes"]
    cts = [bytes.fromhex(e["ct"]) for e in fx["encrypt"]]
    good = [c for c in cts if any(c)][:4]
    ident = bytes(2 * L)
    off_curve = bytearray(good[0]); off_curve[-1] ^= 1
    x = int.from_bytes(good[1][:L], "big")
    y = int.from_bytes(good[1][L:], "big")
    too_big = None
    if x + p < 1 << (8 * L):                              # a non-canonical representative of a valid point
        too_big = (x + p).to_bytes(L, "big") + good[1][L:]
    neg = good[1][:L] + (p - y).to_bytes(L, "big")        # -A is on the curve
    rows = good + [ident, bytes(off_curve), neg] + ([too_big] if too_big else [])
    want = [1] * len(good) + [1, 0, 1] + ([0] if too_big else [])
    assert [int(v) for v in pk.engine.validate(1, b"".join(rows))] == want
    l2 = pk.engine.mult(good[0] + good[1], good[2] + good[3])
    one = (1).to_bytes(L, "big") + bytes(L)
    bad = bytearray(bytes(l2[0])); bad[3] ^= 0x40
    rows = [bytes(l2[0]), bytes(l2[1]), one, bytes(bad), ident]
    assert [int(v) for v in pk.engine.validate(2, b"".join(rows))] == [1, 1, 1, 0, 0]


def test_decrypt_with_unusual_secrets_does_not_misbehave():
    """bgn_ctx_set_secret builds the secret-order line table only when the secret divides n; any other value
    falls back to the table of P.  A wrong secret must simply fail to find discrete logs (or find the trivial
    one), never crash; the right one must still decrypt afterwards."""
    import bgn_amd
    fx = load_fixture("k256")
    pk, sk = engine_key(fx)
    n, q1 = int(fx["n"], 16), int(fx["q1"], 16)
    cts = pk.EncryptBatch([5, 0, 9], [11, 0, 13])
    wire = b"".join(c.C for c in cts)
    for wrong in (n // q1, q1 + 2, 1):             # the other prime factor; not a factor; trivial
        pk.SetupDecryption(bgn_amd.SecretKey(wrong))
        m, st = pk.engine.decrypt(1, wire)
        assert len(m) == 3 and int(st[1]) == 0 and int(m[1]) == 0      # the identity decrypts to 0 under any key
    pk.SetupDecryption(sk)
    m, st = pk.engine.decrypt(1, wire)
    assert not st.any() and [int(v) for v in m] == [5, 0, 9]


def test_a_rejected_secret_leaves_the_context_as_it_was():
    """bgn_ctx_set_secret validates before it touches the current key: a zero key is refused (BGN_E_ARG) and the
    decryption state set up before keeps working."""
    from bgn_amd._lib import BGN_E_ARG
    fx = load_fixture("k256")
    pk, sk = engine_key(fx)
    pk.SetupDecryption(sk)
    cts = pk.EncryptBatch([5, 0, 9], [11, 0, 13])
    wire = b"".join(c.C for c in cts)
    eng = pk.engine
    assert eng._lib.bgn_ctx_set_secret(eng._h, bytes(4), 4) == BGN_E_ARG       # four zero bytes: the key 0
    assert b"zero" in eng._lib.bgn_last_error()
    m, st = pk.engine.decrypt(1, wire)                        # same secret, same tables
    assert not st.any() and [int(v) for v in m] == [5, 0, 9]


@pytest.mark.parametrize("name,count", [("toy64", 300), ("k256", 65), ("k512", 5)])
def test_multconst_windowed_vs_oracle(name, count):
    """MultConst on level 1 with scalars of 128 bits and more runs over a per-element table of multiples (4-bit
    windows); ragged counts (the last element has stand-in lanes), scalars >= n, an identity operand."""
    import oracle_c
    fx = load_fixture(name)
    o = oracle_c.Oracle.from_fixture(fx)
    pk, _ = engine_key(fx)
    rng = random.Random(23)
    n = int(fx["n"], 16)
    xs = [rng.randrange(1, 1000) for _ in range(count)]
    rs = [rng.randrange(n) for _ in range(count)]
    xs[1 % count], rs[1 % count] = 0, 0                      # identity
    cts = o.encrypt(xs, rs)
    klen = max(16, (n.bit_length() + 7) // 8 + 1)
    ks = [rng.randrange(1 << (8 * klen)) for _ in range(count)]
    ks[0] = n
    ks[-1] = n + 1
    assert pk.engine.multconst(1, cts, ks).tobytes() == o.multconst(1, cts, ks)
