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
