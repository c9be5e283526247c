#!/usr/bin/env python3
"""bgn_ctx_calibrate on this device beside the constants of the committed sweeps: the crossovers (elements per call)
between the cooperative, the lane-group and the one-pairing-per-lane kernel for Mult, makeL2, Decrypt's lift and power,
and how long the probes take.    python tools/calibrate_report.py > profiles/r04_calibrate.csv"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401
from conftest import load_fixture  # noqa: E402
import bgn_amd  # noqa: E402

COMMITTED = {   # engine.cpp coop_limit / quad_limit / quad_table_limit (profiles/r04_mid_batch*.csv)
    "k1024": {"coop": [950, 850, 1300, 1300], "quad": [55000, 45300, 31000, 31000]},
    "k512": {"coop": [815, 640, 740, 740], "quad": [45500, 37700, 26900, 26900]},
}
print("key,run,seconds,operation,coop_up_to_calibrated,coop_up_to_committed,quad_up_to_calibrated,quad_up_to_committed")
for key in sys.argv[1:] or ["k1024", "k512"]:
    fx = load_fixture(key)
    pk = bgn_amd.PublicKey(int(fx["p"], 16), int(fx["n"], 16), fx["l"], bytes.fromhex(fx["P"]), bytes.fromhex(fx["Q"]),
                           fx["msg_space"], True, fx["poly_base"])
    pk.SetupDecryption(bgn_amd.SecretKey(int(fx["q1"], 16)))
    for run in range(2):
        t0 = time.perf_counter()
        xo = pk.engine.calibrate()
        dt = time.perf_counter() - t0
        for i, op in enumerate(("mult", "make_l2", "decrypt_lift", "decrypt_power")):
            print("%s,%d,%.2f,%s,%d,%d,%d,%d" % (key, run, dt, op, xo["coop"][i], COMMITTED[key]["coop"][i], xo["quad"][i],
                                                COMMITTED[key]["quad"][i]), flush=True)
