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


@pytest.mark.parametrize("name,count", [("k256", 700), ("k1024", 200)])
def test_small_constants_in_long_scalar_fields(name, count):
    """A wave whose scalars all have leading zero bytes / bits skips them (ops.hpp wave_top_bit): constants below 2^8,
    2^20 and 0 carried in fields of 3, 16, 40 and 130 bytes, waves of zeros beside waves of full-length scalars, on
    both levels of the lane kernel and on the default dispatch — the bytes of the C oracle."""
    import numpy as np
    import oracle_c
    from bgn_amd.api import _as_u8, _ptr, _scalars
    from bgn_amd._lib import check
    fx = load_fixture(name)
    o = oracle_c.Oracle.from_fixture(fx)
    pk, _ = engine_key(fx)
    eng = pk.engine
    rng = random.Random(29)
    pools = {1: [bytes.fromhex(e["ct"]) for e in fx["encrypt"]], 2: [bytes.fromhex(v["out"]) for v in fx["mult"]]}
    for lvl in (1, 2):
        a = b"".join(pools[lvl][rng.randrange(len(pools[lvl]))] for _ in range(count))
        A = _as_u8(a, eng.elem_bytes)
        for klen in (3, 16, 40, 130):
            for shape in ("bytes", "bits20", "zeros", "mixed"):
                if shape == "bytes":
                    ks = [rng.randrange(256) for _ in range(count)]
                elif shape == "bits20":
                    ks = [rng.randrange(1 << 20) for _ in range(count)]
                elif shape == "zeros":
                    ks = [0] * count
                else:                                   # the first wave small, the second zero, the rest full length
                    ks = [rng.randrange(16) for _ in range(64)] + [0] * 64 + [rng.randrange(1 << (8 * klen)) for _ in range(count - 128)]
                want = o.multconst(lvl, a, ks)
                K = _scalars(ks, klen)
                for quad_max in (0, -1):
                    eng.set_option("quad_max_mc", quad_max)
                    try:
                        out = np.empty((count, eng.elem_bytes), dtype=np.uint8)
                        check(eng._lib.bgn_multconst_batch(eng._h, count, lvl, _ptr(A), _ptr(K), klen, None, 0, _ptr(out)),
                              "bgn_multconst_batch")
                        assert out.tobytes() == want, (name, lvl, klen, shape, quad_max)
                    finally:
                        eng.set_option("quad_max_mc", -1)
