"""GPU: MultConst on level 1 with plaintext-sized constants (bgn.go:253-268; BenchmarkMultConstant bgn_test.go:112-125
multiplies by 1) on the lane kernel's 2-bit windows (ops.hpp g1_scalarmul_win_lane with G1MulArgs::wbits == 2: a
per-element table of 1*B .. 3*B, two doublings and one mixed addition per window) against the binary ladder it
replaces for scalars below 128 bits, and against the C oracle."""
import random

import pytest

from conftest import engine_key, load_fixture

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,count", [("toy64", 4099), ("k256", 1500), ("k1024", 200), ("k1024b", 130)])
def test_short_scalars_on_the_lane_kernel_vs_binary_ladder_and_oracle(name, count):
    import oracle_c
    fx = load_fixture(name)
    o = oracle_c.Oracle.from_fixture(fx)
    pk, _ = engine_key(fx)
    eng = pk.engine
    n = int(fx["n"], 16)
    rng = random.Random(23)
    pool = [bytes.fromhex(e["ct"]) for e in fx["encrypt"]] + [bytes(eng.elem_bytes)]          # identity among the bases
    a = b"".join(pool[rng.randrange(len(pool))] for _ in range(count))
    eng.set_option("quad_max_mc", 0)                                   # the lane kernel, whatever the batch size
    try:
        for bits in (1, 2, 7, 8, 9, 40, 64, 120):
            ks = [rng.choice([0, 1, 2, 3, (1 << bits) - 1, 1 << (bits - 1), rng.randrange(1 << bits)]) for _ in range(count)]
            ks[0] = (1 << bits) - 1                                    # fixes the scalar length of the call
            got = eng.multconst(1, a, ks).tobytes()
            assert eng.last_kernel_name() == "k_g1_mul"
            assert got == o.multconst(1, a, ks), (name, bits)
            eng.set_option("g1_mul_window_short", 0)
            try:
                assert eng.multconst(1, a, ks).tobytes() == got, (name, bits, "binary ladder")
            finally:
                eng.set_option("g1_mul_window_short", 1)
    finally:
        eng.set_option("quad_max_mc", -1)


def test_short_scalars_with_bases_of_small_order():
    """Points of the curve outside the ciphertext subgroup (order 2 and 4) and scalars that walk the accumulator
    into the base, its negative and the identity inside a window: the exceptional cases of the doublings and the
    mixed additions are exact, as on the 4-bit route."""
    import bgn_ref as R
    fx = load_fixture("k256")
    pk, _ = engine_key(fx)
    eng = pk.engine
    p, n = int(fx["p"], 16), int(fx["n"], 16)
    T = None
    for x in range(2, 400):
        rhs = (x * x * x + x) % p
        y = pow(rhs, (p + 1) // 4, p)
        if y * y % p == rhs:
            T = R.pt_mul((x, y), n * (fx["l"] // 4), p)
            if T is not None and R.pt_mul(T, 2, p) is not None:
                break
            T = None
    assert T is not None
    T2 = R.pt_mul(T, 2, p)                                             # order 2: (0, 0)?  no: y = 0 at a root of x^3 + x
    bases = [T, T2, R.elem_from_bytes(bytes.fromhex(fx["P"]), p)]
    ks = [0, 1, 2, 3, 4, 5, 6, 7, 8, 0x1234, 0x4444, 0xFFFF, 0xAAAA, 0x5555]
    a = b"".join(R.elem_to_bytes(B, p) for B in bases for _ in ks)
    kk = ks * len(bases)
    want = b"".join(R.elem_to_bytes(R.pt_mul(B, k, p), p) for B in bases for k in ks)
    eng.set_option("quad_max_mc", 0)
    try:
        assert eng.multconst(1, a, kk).tobytes() == want
        assert eng.last_kernel_name() == "k_g1_mul"
    finally:
        eng.set_option("quad_max_mc", -1)
