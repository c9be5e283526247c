"""MultPoly on the table paths for key sizes the parity test does not visit (19, 37 and 72 limbs): the multi-pairing rounds
(option poly_multi) and the one-lane-per-pair walk against the C oracle, identity coefficients among the operands.

    python tools/check_multi_keys.py        (GPU)
"""
import os, sys, random
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from conftest import load_fixture, engine_key
import oracle_c
for name, d, npoly in (("k512", 4, 5), ("k1024b", 4, 3), ("k2048", 2, 2), ("k2048", 4, 1)):
    fx = load_fixture(name)
    o = oracle_c.Oracle.from_fixture(fx)
    pk, _ = engine_key(fx)
    eng = pk.engine
    rng = random.Random(7)
    n = int(fx["n"], 16)
    xa = [rng.choice([0, 1, 2, n - 1]) for _ in range(npoly * d)]
    xb = [rng.choice([0, 1, 2, n - 1]) for _ in range(npoly * d)]
    ea = o.encrypt(xa, [rng.choice([0, rng.randrange(n)]) for _ in xa])
    eb = o.encrypt(xb, [rng.choice([0, rng.randrange(n)]) for _ in xb])
    want = o.poly_mult(npoly, d, d, ea, eb)
    res = {}
    for multi in (1, 0):
        eng.set_option("poly_tables", 1)
        eng.set_option("poly_karatsuba", 0)
        eng.set_option("poly_multi", multi)
        res[multi] = eng.poly_mult(npoly, d, d, ea, eb).tobytes()
        print(name, d, npoly, "poly_multi", multi, "kernel", eng.last_kernel_name(), "equal to the C oracle:", res[multi] == want, flush=True)
    eng.reset_options()
    assert res[1] == want and res[0] == want
print("ok")
