import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
from conftest import load_fixture, engine_key
import oracle_c
name = sys.argv[1]; count = int(sys.argv[2])
fx = load_fixture(name)
o = oracle_c.Oracle.from_fixture(fx)
pk, _ = engine_key(fx)
eng = pk.engine
EB = eng.elem_bytes
pool = [bytes.fromhex(e["ct"]) for e in fx["encrypt"]][:7]
ia = np.arange(count) % len(pool)
ib = (np.arange(count) * 3 + 1) % len(pool)
P = np.frombuffer(b"".join(pool), dtype=np.uint8).reshape(len(pool), EB)
got = np.asarray(eng.add(1, P[ia].reshape(-1), P[ib].reshape(-1))).reshape(count, EB)
W = np.stack([np.frombuffer(o.add(1, pool[i], pool[(i * 3 + 1) % len(pool)]), dtype=np.uint8) for i in range(len(pool))])
ok = (got == W[ia]).all(axis=1)
bad = np.nonzero(~ok)[0]
zero = (got == 0).all(axis=1)
print(name, count, "bad", len(bad), "zero", int(zero.sum()), "first", bad[:10], "last", bad[-5:])
if len(bad):
    lanes = (count + 1) // 2
    print("bad mod 64 hist", np.bincount(bad % 64, minlength=64))
