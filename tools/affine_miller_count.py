#!/usr/bin/env python3
"""Round 6, verdict item 4: what an AFFINE Miller step with the lane run's shared inversion would cost against the
Jacobian steps of pairing.hpp, in 32x32->64 multiply-adds (bgn_amd/synthetic.py prices: a plain product 2 NL^2 + its
flush, a squaring by the segmented square, a sum of two products 3 NL^2 + flush; an inversion by division steps = 55
products).  The pairings of a lane's run (16 at 2^20 per GPU) walk the same schedule — n is the key's — so at every
step their denominators can share ONE inversion by Montgomery's trick: (55 + 3 run) / run product-equivalents per
pairing-step.  The price of that is the run's state: a lane cannot hold sixteen accumulators in registers, so every
step streams (x, y, f0, f1) in and out and the evaluation point in: 10 F_p per pairing-step.

   python tools/affine_miller_count.py [k1024] > profiles/r06_affine_miller_count.csv"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bgn_amd.synthetic as syn  # noqa: E402
from conftest import load_fixture  # noqa: E402

fx = load_fixture(sys.argv[1] if len(sys.argv) > 1 else "k1024")
n, p = int(fx["n"], 16), int(fx["p"], 16)
nl = syn.limbs_for(p)
M, S, SOP = syn.product_mads(nl), syn.square_mads(nl), syn.sop_mads(nl)
dbl, add, big, pre = syn.miller_schedule(n, 5)


def mads(red, sq, sop):
    return (red - sq - sop) * M + sq * S + sop * SOP


# the f-update of a step is the same in both forms: f^2 (g0, F0*F1: two plain products) and f^2 * l (F0, F1: two sums)
F_DBL = (4, 0, 2)
# the f-update of an addition step: f * l only (two sums) — STEP_ADD's F0, F1
F_ADD = (2, 0, 2)
jac_dbl, jac_add = syn.STEP_DBL, syn.STEP_ADD
print("# key %s: NL = %d; product %d, squaring %d, sum of two products %d multiply-adds; width-5 NAF of n: %d doubling steps, "
      "%d addition steps, %d products by f_d" % (fx["name"], nl, M, S, SOP, dbl, add, big))
print("form,run,dbl_reductions,dbl_mads,add_reductions,add_mads,loop_mads,vs_jacobian,state_bytes_per_pairing_step,"
      "state_GB_per_s_at_4.65e5_pairings_per_s")
base = dbl * mads(*jac_dbl) + add * mads(*jac_add) + big * mads(*syn.STEP_MULF)
print("jacobian (pairing.hpp),16,%d,%d,%d,%d,%d,1.000,0,0" % (jac_dbl[0], mads(*jac_dbl), jac_add[0], mads(*jac_add), base))
for run in (16, 32, 64, 256):
    inv = (syn.INVERSION_PRODUCTS + 3 * run) / run
    # doubling: x^2 (S), lambda = (3x^2 + 1) / 2y (M), lambda^2 (S) -> x3, y3 = lambda (x - x3) - y (M),
    # line l = (lambda (xB + x) - y) + i yB: one product, the imaginary part is the evaluation point's ordinate
    a_dbl = (5 + F_DBL[0] + inv, 2 + F_DBL[1], F_DBL[2])
    # addition: lambda = (y2 - y1) / (x2 - x1) (M), lambda^2 (S), y3 (M), line (M)
    a_add = (4 + F_ADD[0] + inv, 1 + F_ADD[1], F_ADD[2])
    loop = dbl * mads(*a_dbl) + add * mads(*a_add) + big * mads(*syn.STEP_MULF)
    state = 10 * nl * 4                      # x, y, f0, f1 in and out, xB, yB in
    print("affine + shared inversion,%d,%.2f,%d,%.2f,%d,%d,%.3f,%d,%.0f" %
          (run, a_dbl[0], mads(*a_dbl), a_add[0], mads(*a_add), loop, loop / base, state,
           state * (dbl + add) * 4.65e5 / 1e9))
