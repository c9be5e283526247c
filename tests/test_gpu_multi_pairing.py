"""GPU: MultPoly's table rounds as multi-pairings (fixedpair.hpp miller_loop_fixed_multi, k_pairing_multi: one lane per
output coefficient, the f^2 of a doubling step shared by all its terms e(a_i, b_j), i + j = s; poly.go:139-153) at the
limb counts the parity test (a 256-bit key) does not visit — 19, 37 and 72 limbs — against the one-lane-per-pair walk
and the C oracle, with identity coefficients among the operands."""
import random

import pytest

from conftest import engine_key, load_fixture

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,d,npoly", [("k512", 4, 5), ("k1024b", 4, 3), ("k1024", 3, 66), ("k2048", 2, 2)])
def test_multi_pairing_rounds_at_other_key_sizes(name, d, npoly, engopts):
    import oracle_c
    fx = load_fixture(name)
    o = oracle_c.Oracle.from_fixture(fx)
    pk, _ = engine_key(fx)
    eng = pk.engine
    rng = random.Random(7 + d)
    n = int(fx["n"], 16)
    xa = [rng.choice([0, 1, 2, n - 1]) for _ in range(npoly * d)]
    xb = [rng.choice([0, 1, 2, n - 1]) for _ in range(npoly * d)]
    # r = 0 with x = 0 is the identity of G1 (encryptZero in deterministic mode, bgn.go:562-564)
    ea = o.encrypt(xa, [rng.choice([0, rng.randrange(n)]) for _ in xa])
    eb = o.encrypt(xb, [rng.choice([0, rng.randrange(n)]) for _ in xb])
    check = min(npoly, 3)                      # (the oracle's pairings at 1024 and 2048 bits take their time)
    E = eng.elem_bytes
    want = o.poly_mult(check, d, d, ea[: check * d * E], eb[: check * d * E])
    engopts.set("poly_tables", 1)
    engopts.set("poly_karatsuba", 0)
    got = {}
    for multi in (1, 0):
        engopts.set("poly_multi", multi)
        got[multi] = eng.poly_mult(npoly, d, d, ea, eb).tobytes()
        assert ("k_pairing_multi" in eng.last_kernel_name()) == (multi == 1), eng.last_kernel_name()
    assert got[1] == got[0]
    assert got[1][: check * 2 * d * E] == want


def test_multi_pairing_layout_that_does_not_fit_the_table_budget_falls_back(engopts):
    """A table budget below one group of 64 products in the multi-pairing layout (dt * 64 columns): the pass is sized
    for the compact one-lane-per-pair layout and walked that way — same bytes — instead of allocating a table up to 64
    times what the budget was meant for (engine.cpp poly_table_chunk).  With the budget raised to whole groups the
    multi-pairing rounds run in several passes of whole groups."""
    import oracle_c
    fx = load_fixture("k256")
    o = oracle_c.Oracle.from_fixture(fx)
    pk, _ = engine_key(fx)
    eng = pk.engine
    rng = random.Random(99)
    n = int(fx["n"], 16)
    d, npoly = 4, 150
    ea = o.encrypt([rng.randrange(3) for _ in range(npoly * d)], [rng.randrange(n) for _ in range(npoly * d)])
    eb = o.encrypt([rng.randrange(3) for _ in range(npoly * d)], [rng.randrange(n) for _ in range(npoly * d)])
    E = eng.elem_bytes
    want = o.poly_mult(4, d, d, ea[: 4 * d * E], eb[: 4 * d * E])
    engopts.set("poly_tables", 1)
    engopts.set("poly_karatsuba", 0)
    engopts.set("poly_multi", 1)
    # one Miller step of one coefficient is 3 * NL * 4 bytes; a 256-bit key walks ~260 + ~90 steps: ~42 KB per
    # coefficient, 10.7 MB for one group of 64 products of 4 coefficients
    engopts.set("poly_table_max_mb", 4)
    small = eng.poly_mult(npoly, d, d, ea, eb).tobytes()
    assert "k_pairing_multi" not in eng.last_kernel_name(), eng.last_kernel_name()
    engopts.set("poly_table_max_mb", 24)                       # two groups per pass: 128 + 22 products
    groups = eng.poly_mult(npoly, d, d, ea, eb).tobytes()
    assert "k_pairing_multi" in eng.last_kernel_name(), eng.last_kernel_name()
    engopts.set("poly_table_max_mb", 0)
    whole = eng.poly_mult(npoly, d, d, ea, eb).tobytes()
    assert small == whole and groups == whole
    assert whole[: 4 * 2 * d * E] == want
