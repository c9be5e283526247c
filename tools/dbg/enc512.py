import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
from conftest import load_fixture, engine_key
import oracle_c
name = sys.argv[1] if len(sys.argv) > 1 else "k512"
fx = load_fixture(name)
pk, sk = engine_key(fx)
o = oracle_c.Oracle.from_fixture(fx)
eng = pk.engine
xs = [1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 31, 32, 33, 255, 256, 257, 1000, 4095, 4096, 5000, 32767, 32768, 65535, 65536, 65537, 1 << 20, (1 << 32) + 5]
import numpy as np
got = np.asarray(eng.encrypt(xs, None)).reshape(-1)
EB = eng.elem_bytes
bad = []
for i, x in enumerate(xs):
    w = o.encrypt([x], None)
    g = bytes(got[i * EB:(i + 1) * EB])
    if g != w:
        bad.append((x, "zero" if g == bytes(EB) else "wrong"))
print(name, "bad:", bad)
