import os, sys, random
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from conftest import load_fixture, engine_key
for name in ["toy64", "k256"]:
    fx = load_fixture(name); pk, sk = engine_key(fx); eng = pk.engine
    rng = random.Random(3)
    n = int(fx["n"], 16)
    for count in [1, 2, 5, 64, 65, 300]:
        xs = [rng.randrange(1, 1000) for _ in range(count)]
        rs = [rng.randrange(n) for _ in range(count)]
        cts = eng.encrypt(xs, rs).tobytes()
        for klen in [8, 16]:
            ks = [rng.randrange(1 << (8 * klen)) for _ in range(count)]
            os.environ["BGN_G1_MUL_WINDOW"] = "0"
            ref = eng.multconst(1, cts, ks)
            os.environ["BGN_G1_MUL_WINDOW"] = "1"
            got = eng.multconst(1, cts, ks)
            bad = [i for i in range(count) if bytes(ref[i]) != bytes(got[i])]
            print(name, "count", count, "klen", klen, "bad", len(bad), bad[:10], flush=True)
