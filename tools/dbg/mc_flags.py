import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from conftest import load_fixture
import bgn_amd, oracle_c
for name in ("k256", "k1024"):
    fx = load_fixture(name)
    pk = bgn_amd.PublicKey(int(fx["p"], 16), int(fx["n"], 16), fx["l"], bytes.fromhex(fx["P"]), bytes.fromhex(fx["Q"]), fx["msg_space"], True, fx["poly_base"])
    eng = pk.engine
    o = oracle_c.Oracle.from_fixture(fx)
    rng = random.Random(1)
    n = int(fx["n"], 16)
    cnt = 40
    cts = eng.encrypt([rng.randrange(1000) for _ in range(cnt)], [rng.randrange(n) for _ in range(cnt)]).tobytes()
    ks = [rng.randrange(1 << 64) for _ in range(cnt)]
    want = o.multconst(1, cts, ks)
    eng.set_option("combine", 0)
    eng.set_option("test_mc_fallback", 2)
    got = eng.multconst(1, cts, ks).tobytes()
    print(name, "with fallback ok:", got == want, "flagged:", eng.get_option("test_mc_flagged"), eng.last_kernel_name())
    eng.set_option("test_mc_fallback", 1)
    got = eng.multconst(1, cts, ks).tobytes()
    E = eng.elem_bytes
    bad = [i for i in range(cnt) if got[i*E:(i+1)*E] != want[i*E:(i+1)*E]]
    print(name, "without fallback ok:", got == want, "bad:", bad[:10], len(bad))
    t0 = time.perf_counter(); eng.multconst(1, cts, ks); print("ms", (time.perf_counter() - t0) * 1e3)
