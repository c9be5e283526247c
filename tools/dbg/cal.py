import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_fixture
import bgn_amd
fx = load_fixture("k1024")
for lg in (24, 31):
    pk = bgn_amd.PublicKey(int(fx["p"], 16), int(fx["n"], 16), fx["l"], bytes.fromhex(fx["P"]), bytes.fromhex(fx["Q"]), fx["msg_space"], True, fx["poly_base"])
    pk.engine.set_option("bsgs_max_log2", lg)
    pk.SetupDecryption(bgn_amd.SecretKey(int(fx["q1"], 16)))
    pk.engine.set_option("test_calibrate_trace", 1)
    print(lg, pk.engine.calibrate(), flush=True)
    pk.engine.close()
