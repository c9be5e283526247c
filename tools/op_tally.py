#!/usr/bin/env python3
"""Where the VALU instructions of the one-element-per-lane kernels go: the dynamic count of field primitives of one
lane (the host emulation of the device headers tallies them, fpmont.hpp BGN_TALLY), priced in instructions per unit
as the gfx950 code objects have them, beside the SQ_INSTS_VALU the counters measured for the same kernel.

    python tools/op_tally.py [k1024] > profiles/r06_op_tally.csv        (CPU only; the emulator is tests/emu)

Prices (instructions per unit, read off the disassembly of kern_nl36): a multiply-add 1; a product row 5 (Montgomery
factor: multiply + mask, the 64-bit row carry: shift + add, the loop's share); an accumulator flushed 3; a limb of a
linear pass or of a product's final carry 4 resp. 3; of a conditional-subtraction round 5 (subtract, mask, shift +
select); of a select 1; of a comparison 2; of an AGPR move 1; LDS and global accesses are not VALU instructions (their
address arithmetic is: 1 per limb of a strided global access).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
sys.path.insert(0, ROOT)
from conftest import load_fixture  # noqa: E402
import emu  # noqa: E402

PRICE = {"mad": 1, "row": 5, "flush": 3, "pass": 4, "final": 3, "reduce": 5, "select": 1, "cmp": 2, "agpr": 1, "lds": 0, "gmem": 1}


def priced(t):
    return {k: t[k] * PRICE[k] for k in PRICE}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "k1024"
    fx = load_fixture(name)
    E = emu.Emu.from_fixture(fx)
    import oracle_c
    o = oracle_c.Oracle.from_fixture(fx)
    a = o.encrypt([5], [12345])
    b = o.encrypt([7], [54321])
    jobs = []

    def run(label, unit, fn, units=1):
        E.tally_reset()
        fn()
        t = E.tally()
        jobs.append((label, unit, units, t))

    E.set_window(5)
    run("k_pairing (Mult): one pairing alone, width-5 Miller loop (its inversions by exponentiation, which a run of 16 shares on the device)", "pairing", lambda: E.pairing_w3(a, b))
    E.set_window(2)
    tab = {}
    run("k_fixedpair_build_batch: one line table (MultPoly)", "table", lambda: tab.setdefault("t", E.fixed_table(a)))
    run("k_pairing<.,1>: one walk over a coefficient's table (MultPoly)", "walk", lambda: E.pairing_fixed(tab["t"], b))
    # a four-term lane of the multi-pairing rounds (MultPoly, leaves of 4 x 4: the coefficient s = 3), per term
    A4 = [o.encrypt([3 + i], [777 + i]) for i in range(4)]
    B4 = [o.encrypt([9 + i], [555 + i]) for i in range(4)]
    Qp, q = 2, 1
    mt = {}

    def multi_tables():
        t = None
        for i, w in enumerate(A4):
            t = E.fixed_table(w, 4 * Qp, i * Qp + q, t)
        mt["t"] = t

    multi_tables()
    run("k_pairing_multi: one output coefficient of four terms (MultPoly, multi-pairing rounds), per term", "term",
        lambda: E.pairing_fixed_multi(mt["t"], 4 * Qp, Qp, q, A4, B4, 3), 4)
    run("k_g1_add: affine additions, one run of 16 (EAdd at 2^20)", "addition",
        lambda: E.g1_add([a] * 16, [b] * 16, plain=True), 16)
    # the fused level-2 Add (barrett.hpp fp2_mul_plain): one F_p^2 product of plain residues, no codec
    l2a, l2b = o.mult(a, b), o.mult(b, b)
    run("k_gt_mul_wire: one level-2 Add (Barrett F_p^2 product of plain residues; the codec around it is not tallied)", "addition",
        lambda: E.gt_mul_plain(l2a, l2b))
    print("# tools/op_tally.py %s: primitives of ONE lane by the host emulation, priced in VALU instructions (see the tool's header)" % name)
    print("kernel,unit," + ",".join("n_" + k for k in PRICE) + ",mad_instructions,other_instructions,other_share," +
          ",".join("instr_" + k for k in PRICE if k != "mad"))
    for label, unit, units, t in jobs:
        pr = priced(t)
        mad = pr["mad"] / units
        other = sum(v for k, v in pr.items() if k != "mad") / units
        print("%s,%s,%s,%.0f,%.0f,%.3f,%s" % (label, unit, ",".join("%.0f" % (t[k] / units) for k in PRICE), mad, other,
                                           other / (mad + other), ",".join("%.0f" % (pr[k] / units) for k in PRICE if k != "mad")))


if __name__ == "__main__":
    main()
