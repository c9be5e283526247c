"""GPU: MultConst on level-2 ciphertexts (bgn.go:270-288) by the norm-1 ladder (k_gt_pow with a per-wave norm check:
ops.hpp gt_pow_norm1_lane, two field products per scalar bit) against the general square-and-multiply in F_p^2 it
replaces, the C oracle and Python integers — for GT elements, and for bytes that are no ciphertext (norm != 1), which
must still come out as base^k."""
import random

import pytest

from conftest import engine_key, load_fixture

pytestmark = pytest.mark.gpu


def fp2_pow(a, k, p):
    r = (1, 0)
    b = a
    while k:
        if k & 1:
            r = ((r[0] * b[0] - r[1] * b[1]) % p, (r[0] * b[1] + r[1] * b[0]) % p)
        b = ((b[0] * b[0] - b[1] * b[1]) % p, (2 * b[0] * b[1]) % p)
        k >>= 1
    return r


@pytest.mark.parametrize("name,count", [("toy64", 5000), ("k256", 700), ("k1024", 130)])
def test_multconst_l2_ladder_vs_general_power_and_oracle(name, count):
    import oracle_c
    fx = load_fixture(name)
    o = oracle_c.Oracle.from_fixture(fx)
    pk, _ = engine_key(fx)
    eng = pk.engine
    n = int(fx["n"], 16)
    rng = random.Random(3)
    l2 = [bytes.fromhex(v["out"]) for v in fx["mult"]]
    a = b"".join(l2[i % len(l2)] for i in range(count))
    ks = [rng.choice([0, 1, 2, 3, n - 1, n, n + 1, rng.randrange(1 << 40), rng.randrange(n)]) for _ in range(count)]
    got = eng.multconst(2, a, ks).tobytes()
    eng.set_option("multconst_l2_ladder", 0)
    try:
        assert eng.multconst(2, a, ks).tobytes() == got
    finally:
        eng.set_option("multconst_l2_ladder", 1)
    ncheck = min(count, 64)
    EB = eng.elem_bytes
    assert got[: ncheck * EB] == o.multconst(2, a[: ncheck * EB], ks[:ncheck])


@pytest.mark.parametrize("name", ["toy64", "k256", "k1024"])
def test_multconst_l2_on_bytes_that_are_no_ciphertext(name):
    """Bases whose norm is not 1 (and the corner bases 1, -1, i, 0): the wave falls back to the general power; a batch
    that mixes them with GT elements gives base^k for every element."""
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    eng = pk.engine
    p, n = int(fx["p"], 16), int(fx["n"], 16)
    Lb = eng.elem_bytes // 2
    rng = random.Random(8)
    enc = lambda x, y: x.to_bytes(Lb, "big") + y.to_bytes(Lb, "big")
    dec = lambda b: (int.from_bytes(b[:Lb], "big"), int.from_bytes(b[Lb:], "big"))
    l2 = [bytes.fromhex(v["out"]) for v in fx["mult"]]
    bases = [enc(1, 0), enc(p - 1, 0), enc(0, 1), enc(0, 0), enc(2, 3), enc(rng.randrange(p), rng.randrange(p))]
    rows = [bases[i % len(bases)] if i % 3 == 0 else l2[i % len(l2)] for i in range(200)]
    # ... and waves of norm-1 elements only behind them (elements 256 .. 319 and 320 .. 383: they take the ladder),
    # the bases 1 and -1 among them: their imaginary part is 0, which the ladder's last division must survive
    rows += [l2[i % len(l2)] if i % 5 else (enc(1, 0) if i % 10 else enc(p - 1, 0)) for i in range(192)]
    ks = [rng.choice([0, 1, 2, 5, n - 1, rng.randrange(1 << 40), rng.randrange(n)]) for _ in rows]
    got = eng.multconst(2, b"".join(rows), ks)
    for row, base, k in zip(got, rows, ks):
        assert dec(bytes(row)) == fp2_pow(dec(base), k, p), (dec(base), k)


@pytest.mark.parametrize("name,count", [("k256", 30011), ("k1024", 257)])
def test_multconst_in_place_on_device_arrays(name, count):
    """include/bgn_amd.h (device buffers, aliasing): MultConst may write its result over its operand array — the
    codec reads the whole operand before the first result byte is written on every route (lane kernel, lane groups,
    both levels)."""
    import numpy as np
    import torch
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    eng = pk.engine
    dev = torch.device("cuda", 0)
    rng = random.Random(17)
    n = int(fx["n"], 16)
    pools = {1: [bytes.fromhex(e["ct"]) for e in fx["encrypt"]], 2: [bytes.fromhex(v["out"]) for v in fx["mult"]]}
    for lvl in (1, 2):
        pool = pools[lvl]
        for cnt in (count, 33):
            a = b"".join(pool[(3 * i + 1) % len(pool)] for i in range(cnt))
            ks = [rng.choice([0, 1, rng.randrange(1 << 40), rng.randrange(n)]) for _ in range(cnt)]
            want = eng.multconst(lvl, a, ks).tobytes()
            klen = max(1, (max(ks).bit_length() + 7) // 8)
            kb = np.frombuffer(b"".join(k.to_bytes(klen, "big") for k in ks), dtype=np.uint8)
            ta = torch.frombuffer(bytearray(a), dtype=torch.uint8).to(dev)
            tk = torch.from_numpy(kb.copy()).to(dev)
            eng.multconst_dev(lvl, ta, tk, klen, ta, cnt)
            torch.cuda.synchronize()
            assert ta.cpu().numpy().tobytes() == want, (lvl, cnt)
