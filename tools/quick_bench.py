import sys, time, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, numpy as np
from conftest import load_fixture, engine_key
fx = load_fixture(sys.argv[1] if len(sys.argv) > 1 else "k1024")
count = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
pk, _ = engine_key(fx)
eng = pk.engine
cts = [bytes.fromhex(e["ct"]) for e in fx["encrypt"][2:]]
a = np.frombuffer(b"".join(cts[i % len(cts)] for i in range(count)), dtype=np.uint8)
b = np.frombuffer(b"".join(cts[(i * 5 + 1) % len(cts)] for i in range(count)), dtype=np.uint8)
da = torch.from_numpy(a.copy()).cuda(); db = torch.from_numpy(b.copy()).cuda(); do = torch.empty_like(da)
for it in range(2):
    torch.cuda.synchronize(); t = time.time()
    eng.mult_dev(da, db, do); torch.cuda.synchronize(); dt = time.time() - t
    print(f"{fx['name']} count={count} wall={dt*1e3:.1f} ms kernel={eng.last_kernel_ms():.1f} ms -> {count/dt:.0f} pairings/s ({eng.last_kernel_name()})", flush=True)
