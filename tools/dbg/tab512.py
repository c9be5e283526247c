import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
from conftest import load_fixture, engine_key
import oracle_c
name = sys.argv[1]; lo = int(sys.argv[2]); hi = int(sys.argv[3]); w = int(sys.argv[4])
fx = load_fixture(name)
o = oracle_c.Oracle.from_fixture(fx)
pk, _ = engine_key(fx)
eng = pk.engine
EB = eng.elem_bytes
xs = [d << (16 * w) for d in range(lo, hi)]
got = np.asarray(eng.encrypt(xs, None)).reshape(len(xs), EB)
want = np.frombuffer(o.encrypt(xs, None), dtype=np.uint8).reshape(len(xs), EB)
ok = (got == want).all(axis=1)
zero = (got == 0).all(axis=1)
bad = np.nonzero(~ok)[0] + lo
print(name, "w", w, "range", lo, hi, "bad", len(bad), "zero", int(zero.sum()))
if len(bad):
    # print as ranges
    r = []; s = bad[0]; p = bad[0]
    for b in bad[1:]:
        if b != p + 1:
            r.append((int(s), int(p))); s = b
        p = b
    r.append((int(s), int(p)))
    print("bad ranges", r[:40])
