"""Models of the lane-group kernels for per-element powers (bgn_amd/csrc/quad/quad_g1.hpp) for the CPU tests.

MultConst with per-element scalars on sixteen lanes per element: level 1 a fixed-window Jacobian ladder in G1
(tools/coop/gen_prog.py build_quad_g1_programs: GDBL / GADD / GZZZ, then AFF after the inversion of Z), level 2 a
fixed-window power in F_p^2 over the LSQ / LMU / OUT segments of the final exponentiation's program.  The classes
below run the kernels' CONTROLLERS — table build, recoding, the per-window sequence with the suppressed stores of an
element whose digit is zero while its neighbours in the wave add, the single zero test of Z — on the value machine
(Python integers by physical slot) and on the lane machine (4 x M signed limbs, the kernel's arithmetic) of
tests/quad_model.py.  TEST INFRASTRUCTURE: not used by the product.
"""
from __future__ import annotations

import numpy as np

import quad_model as qm
from quad_model import I64, gen_prog

_GPROGRAMS = None
STATE = ("X", "Y", "Z", "ZZ")
ENTRY = ("tx", "ty", "tz", "tzz", "tzzz")


def g1_programs():
    global _GPROGRAMS
    if _GPROGRAMS is None:
        _GPROGRAMS = gen_prog.build_quad_g1_programs()
    return _GPROGRAMS


def recode_w4(k: int, klen: int):
    """k_recode_w4: signed digits in -7 .. 8, least significant window first, 2 * klen + 1 windows."""
    out, carry = [], 0
    for j in range(2 * klen + 1):
        u = ((k >> (4 * j)) & 15) + carry if j < 2 * klen else carry
        if u > 8:
            u -= 16
            carry = 1
        else:
            carry = 0
        out.append(u)
    assert carry == 0 and sum(d << (4 * j) for j, d in enumerate(out)) == k
    return out


class PowerMixin:
    """Controllers of k_g1_mul_quad / k_g1_aff_quad / k_gt_pow_quad_each over a QuadValueMachine-like base."""

    # -- helpers over the current program's slots --
    def run_protected(self, seg, keep, protect):
        """quad_run_p: stores to the physical slots of `protect` are suppressed when keep."""
        rounds = dict(self.P.segments)[seg]
        for us in rounds:
            res = [(u.dst, self.exec_uop(u)) for u in us]
            for k, v in res:
                assert 0 <= self.value(v) < self.P.bound[k] * self.p, "bound of %s exceeded" % k
                if keep and self.P.phys[k] in protect:
                    continue
                self.V[self.P.phys[k]] = v
            self.rounds_run += 1

    def neg_y(self, y):
        return 19 * self.p - self.value(y)

    def g1_mul(self, bx, by, k, klen, force_dummy_adds=True):
        """k * (bx, by) as the kernel computes it.  Returns ("inf",), ("exc",) or ("pt", x, y) with plain residues.
        force_dummy_adds: run the addition also in the windows where this element's digit is zero (its neighbours
        in the wave may have one), with the stores to the state suppressed — the kernel's worst case."""
        G, A = g1_programs()
        self.P, self.V = G, {}
        protect = {G.phys[s] for s in STATE}
        one = self.mont(1)
        self.put("X", self.mont(bx))
        self.put("Y", self.mont(by))
        self.put("Z", one)
        self.put("ZZ", one)
        table = {}

        def store_entry(d):
            self.run("GZZZ")
            table[d] = [self.get(s) for s in STATE] + [self.get("zzz")]

        def load_state(d):
            for s, v in zip(STATE, table[d][:4]):
                self.V[G.phys[s]] = v

        def load_entry(d, neg):
            vals = list(table[d])
            if neg:
                vals[1] = self.store(self.neg_y(vals[1]))
            for s, v in zip(ENTRY, vals):
                self.V[G.phys[s]] = v

        store_entry(1)
        load_entry(1, False)
        self.run("GDBL")
        store_entry(2)
        self.run_protected("GADD", False, protect)
        store_entry(3)
        for src, dst in ((2, 4), (3, 6), (4, 8)):
            load_state(src)
            self.run("GDBL")
            store_entry(dst)
            if dst < 8:
                self.run_protected("GADD", False, protect)
                store_entry(dst + 1)
        digits = recode_w4(k, klen)
        acc_inf = True
        for j in range(len(digits) - 1, -1, -1):
            if not acc_inf:
                for _ in range(4):
                    self.run("GDBL")
            d = digits[j]
            if d == 0 and not force_dummy_adds:
                continue
            load_entry(abs(d) if d else 1, d < 0)
            take = d != 0 and not acc_inf
            self.run_protected("GADD", not take, protect)
            if d != 0 and acc_inf:
                for s, t in zip(STATE, ENTRY):
                    self.V[G.phys[s]] = self.get(t)
                acc_inf = False
        if acc_inf:
            return ("inf",)
        z = self.value(self.get("Z")) % self.p
        if z == 0:
            return ("exc",)
        X, Y = self.get("X"), self.get("Y")
        zi = self.R * self.R * pow(self.value(self.get("Z")) % self.p, -1, self.p) % self.p      # k_coop_invert: R / Z
        self.P, self.V = A, {}
        self.V[A.phys["X"]] = X
        self.V[A.phys["Y"]] = Y
        self.put("zi", zi)
        self.put("raw1", 1)
        self.run("AFF")
        return ("pt", self.value(self.get("out0")) % self.p, self.value(self.get("out1")) % self.p)

    def g1_fixed(self, tab_p, tab_q, wbits, x, xlen, r, rlen, force_dummy_adds=True):
        """P^x * Q^r as k_g1_fixed_quad computes it from window tables tab[w][d] = affine point d * 2^(wbits*w) * B (None:
        the identity; d = 0 never read).  Returns ("inf",) or ("pt", x, y), plain residues."""
        G, A = g1_programs()
        self.P, self.V = G, {}
        protect = {G.phys[s] for s in STATE}
        one = self.mont(1)
        for s_ in STATE + ENTRY:
            self.put(s_, one)
        acc_inf = True
        wx = (8 * xlen + wbits - 1) // wbits if x is not None else 0
        wr = (8 * rlen + wbits - 1) // wbits if r is not None else 0
        dbls = 0
        for i in range(wx + wr):
            isx = i < wx
            lw = i if isx else i - wx
            k = x if isx else r
            d = (k >> (wbits * lw)) & ((1 << wbits) - 1)
            ent = (tab_p if isx else tab_q)[lw][d] if d else None
            use = d != 0 and ent is not None
            if not use and not force_dummy_adds:
                continue
            src = ent if use else (tab_p[0][1])
            self.put("tx", self.mont(src[0]))
            self.put("ty", self.mont(src[1]))
            took = use and not acc_inf
            if use and acc_inf:
                for s_, t in zip(STATE, ENTRY):
                    self.V[G.phys[s_]] = self.get(t)
                acc_inf = False
            self.run_protected("GADM", not took, protect)
            if took and self.value(self.get("Z")) % self.p == 0:
                if self.value(self.get("X")) % self.p == 0:                     # acc == entry: double the entry
                    for s_, t in zip(STATE, ENTRY):
                        self.V[G.phys[s_]] = self.get(t)
                    self.run_protected("GDBL", False, protect)
                    dbls += 1
                    if self.value(self.get("Z")) % self.p == 0:
                        acc_inf = True
                else:
                    acc_inf = True                                              # acc == -entry
        self.doublings = dbls
        if acc_inf:
            return ("inf",)
        X, Y = self.get("X"), self.get("Y")
        zi = self.R * self.R * pow(self.value(self.get("Z")) % self.p, -1, self.p) % self.p
        self.P, self.V = A, {}
        self.V[A.phys["X"]] = X
        self.V[A.phys["Y"]] = Y
        self.put("zi", zi)
        self.put("raw1", 1)
        self.run("AFF")
        return ("pt", self.value(self.get("out0")) % self.p, self.value(self.get("out1")) % self.p)

    def gt_pow(self, g0, g1, k, klen, wave_top=None):
        """(g0 + i g1)^k with plain residues in and out, as k_gt_pow_quad_each computes it.  wave_top: the highest
        window the wave starts from (another element's exponent may be longer)."""
        F = qm.programs()[1]
        self.P, self.V = F, {}
        for s, v in (("h0", g0), ("h1", g1), ("r0", g0), ("r1", g1)):
            self.put(s, self.mont(v))
        self.put("raw1", 1)
        table = {1: (self.get("r0"), self.get("r1"))}
        for d in range(2, 16):
            self.run("LMU")
            table[d] = (self.get("r0"), self.get("r1"))
        one = (self.store(self.mont(1)), self.store(0))

        def entry(d):
            return table[d] if d else one

        nib = [(k >> (4 * j)) & 15 for j in range(2 * klen)]
        top = 2 * klen - 1
        while top > 0 and nib[top] == 0:
            top -= 1
        if wave_top is not None:
            top = max(top, wave_top)
        self.V[F.phys["r0"]], self.V[F.phys["r1"]] = entry(nib[top])
        for j in range(top - 1, -1, -1):
            for _ in range(4):
                self.run("LSQ")
            self.V[F.phys["h0"]], self.V[F.phys["h1"]] = entry(nib[j])
            self.run("LMU")
        self.run("OUT")
        return self.value(self.get("out0")) % self.p, self.value(self.get("out1")) % self.p


class PowerValueMachine(PowerMixin, qm.QuadValueMachine):
    pass


class PowerLaneMachine(PowerMixin, qm.QuadLaneMachine):
    def neg_y(self, y):
        """The kernel's own negation: 19 p - y limb by limb, one carry pass."""
        return self.normalize(I64(19) * self.p_q - y.astype(I64))

    def store(self, v):
        return v if isinstance(v, np.ndarray) else qm.to_quad(v, self.nl)
