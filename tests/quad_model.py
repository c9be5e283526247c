"""Models of the lane-group pairing kernel (bgn_amd/csrc/quad/) for the CPU tests.

The kernel gives one pairing to 16 lanes of a wave: four quads, each running one micro-op of a round of the tables
tools/coop/gen_prog.py schedules for four workers (build_quad_programs: the formulas of the wave-cooperative
kernel, the Miller state updated in place, one program per launch).  Inside a quad a field element is split over
the four lanes, M = ceil(NL / 4) limbs of 29 bits per lane (lane s holds limbs s*M .. s*M + M - 1), and a
Montgomery product is NL rows of {broadcast one limb of a inside the quad, multiply-add into the lane's M
accumulators, quotient digit from lane 0, multiply-add of p, retire the lowest accumulator: its low 29 bits go to
the lane below, the rest to the next accumulator}.

  * QuadValueMachine — the tables on Python integers, values kept by PHYSICAL slot and every round's reads done
    before its writes, as the kernel does: checks formulas, schedule, in-place updates, slot allocation, bounds.
  * QuadLaneMachine  — slots hold 4 x M signed 32-bit limbs (numpy) and every step is the kernel's instruction-level
    arithmetic with 32 / 64-bit wrap-around: what bgn_amd/csrc/quad/quad.hpp implements.
TEST INFRASTRUCTURE: not used by the product.
"""
from __future__ import annotations

import numpy as np

import coop_model as cm
from coop_model import LIMB, MASK, gen_prog, naf

I64 = np.int64
I32 = np.int32

_QPROGRAMS = None


def programs():
    """(Miller-loop program, final-exponentiation program, Miller loop over a key's line table)."""
    global _QPROGRAMS
    if _QPROGRAMS is None:
        _QPROGRAMS = gen_prog.build_quad_programs()
    return _QPROGRAMS


def wnaf(n: int, w: int):
    """Width-w NAF, least significant digit first (hostbig.hpp wnaf): odd digits below 2^(w-1) in absolute value."""
    d, k = [], n
    while k:
        if k & 1:
            z = k % (1 << w)
            if z >= 1 << (w - 1):
                z -= 1 << w
            k -= z
        else:
            z = 0
        d.append(z)
        k >>= 1
    return d


def nl_for(p: int) -> int:
    need = (p.bit_length() + 9 + LIMB - 1) // LIMB
    return next(x for x in (10, 19, 36, 37, 72) if x >= need)


class QuadValueMachine:
    def __init__(self, p: int, nl: int):
        self.p, self.nl = p, nl
        self.R = 1 << (LIMB * nl)
        self.pinvR = (-pow(p, -1, self.R)) % self.R
        self.PM, self.PF, self.PT = programs()
        self.P = self.PM
        self.V = {}                      # physical slot -> value
        self.rounds_run = 0
        self.products = 0

    def mont(self, x):
        return x * self.R % self.p

    # -- storage by physical slot --
    def put(self, name, v):
        self.V[self.P.phys[name]] = self.store(v)

    def get(self, name):
        return self.V[self.P.phys[name]]

    def store(self, v):
        return v

    def value(self, x):
        return x

    def _combo(self, form, K):
        v = sum(c * self.value(self.get(s)) for s, c in form.items()) + K * self.p
        assert v >= 0, "negative operand"
        return v

    def _mul(self, A, B):
        assert A * B < self.R * self.p, "Montgomery input condition violated"
        Q = (A * B * self.pinvR) % self.R
        self.products += 1
        return (A * B + Q * self.p) // self.R

    def exec_uop(self, u):
        if u.kind == "mul":
            v = self._mul(self._combo(u.A, u.KA), self._combo(u.B, u.KB))
            if u.E:
                v += self._combo(u.E, u.KE)
        else:
            v = self._combo(u.E, u.KE)
        return v

    def run(self, seg):
        rounds = dict(self.P.segments)[seg]
        for us in rounds:
            assert len(us) <= self.P.w
            res = [(u.dst, self.exec_uop(u)) for u in us]          # all reads of a round precede its writes
            assert len({self.P.phys[k] for k, _ in res}) == len(res), "two micro-ops of a round write one slot"
            for k, v in res:
                assert 0 <= self.value(v) < self.P.bound[k] * self.p, "bound of %s exceeded" % k
                self.V[self.P.phys[k]] = v
            self.rounds_run += 1

    def miller(self, ax, ay, bx, by, n):
        """Launch 1 as the kernel's controller sequences it; returns the parked (F0^2, F1^2, F0*F1)."""
        self.P = self.PM
        self.V = {}
        for k, v in (("ax", ax), ("ay", ay), ("bx", bx), ("by", by), ("X", ax), ("Y", ay), ("Z", 1), ("ZZ", 1), ("W", 1),
                     ("v0", 1), ("v2", 1)):
            self.put(k, self.mont(v))
        self.put("v1", 0)
        d = naf(n)
        i = len(d) - 2
        while i >= 0:
            if d[i] and i != 0:
                self.run("DAP" if d[i] > 0 else "DAM")                    # doubling + addition of +-A, one segment
                i -= 1
            elif i >= 1 and (d[i - 1] == 0 or i - 1 == 0):
                self.run("DBL2")                                          # two plain doublings: nine rounds instead of ten
                i -= 2
            else:
                self.run("DBL")
                i -= 1
        self.run("NORM")
        return [self.get(k) for k in ("n1", "n2", "fm")]

    # -- the width-w loop (pairing.hpp miller_loop_w): quad.hpp k_pairing_quad_wtab stages A and B, the table made
    # affine and the windowed controller of k_pairing_quad<NL, 1> --
    def canon(self, x):
        """What the kernel stores in a table: the canonical representative of a value below 2p."""
        return self.store(self.value(x) % self.p)

    def inv_of(self, z):
        """k_coop_invert: R / Z for the Montgomery value z = Z R."""
        return self.store(self.R * self.R * pow(self.value(z) % self.p, -1, self.p) % self.p)

    def miller_w(self, ax, ay, bx, by, n, w):
        self.P = self.PM
        self.V = {}
        one = self.mont(1)
        A = (self.mont(ax), self.mont(ay))

        def state(x, y, f0=None, f1=None):
            self.put("X", x)
            self.put("Y", y)
            for k in ("Z", "ZZ", "W"):
                self.put(k, one)
            if f0 is None:
                self.put("v0", one)
                self.put("v1", 0)
                self.put("v2", one)
            else:                                                         # F0 = v0 - v1, F1 = v2 - v0 - v1
                self.put("v0", f0)
                self.put("v1", 0)
                self.put("v2", self.add2(f0, f1))

        def operands(pt):
            self.put("ax", pt[0])
            self.put("ay", pt[1])

        self.put("bx", self.mont(bx))
        self.put("by", self.mont(by))
        self.put("one", one)
        d = wnaf(n, w)
        maxd = (1 << (w - 1)) - 1
        npts = (maxd - 1) // 2
        # stage A: (2A, f_2) by one doubling step from (A, 1)
        operands(A)
        state(*A)
        self.run("DBL")
        X2, Y2, Z2 = self.get("X"), self.get("Y"), self.get("Z")
        self.run("FOUT")
        f2 = (self.canon(self.get("axo")), self.canon(self.get("ayo")))
        # stage B: 2A affine, then (2k+1)A = (2k-1)A + 2A with f_(2k+1) = f_(2k-1) * l * f_2
        self.put("X", X2)
        self.put("Y", Y2)
        self.put("zi", self.inv_of(Z2))
        self.run("AFM")
        A2 = (self.canon(self.get("axo")), self.canon(self.get("ayo")))
        state(*A)
        jac, fd, pzs = {}, {}, {0: self.store(one)}
        for k in range(1, npts + 1):
            operands(A2)
            self.run("ADDP")
            operands(f2)
            self.run("FMP")
            jac[k] = (self.get("X"), self.get("Y"), self.canon(self.get("Z")))
            self.put("pz", pzs[k - 1])                                    # the running product of the Z (Montgomery's trick)
            self.run("FOUZ")
            fd[k] = (self.canon(self.get("axo")), self.canon(self.get("ayo")))
            pzs[k] = self.canon(self.get("pzo"))
        # the table made affine (the Miller launch's prologue) from ONE inverse per pairing: that of Z_1 .. Z_npts
        aff = {0: A}
        self.put("zi", self.inv_of(pzs[npts]))
        for k in range(npts, 0, -1):
            self.put("X", jac[k][0])
            self.put("Y", jac[k][1])
            self.put("pzp", pzs[k - 1])
            self.put("zk", jac[k][2])
            self.run("AFZ")
            aff[k] = (self.canon(self.get("axo")), self.canon(self.get("ayo")))
        # the loop: from the top digit (1, 3, ..., maxd)
        top = d[-1]
        assert top > 0 and top & 1 and top <= maxd
        if top == 1:
            state(*A)
        else:
            state(*aff[top // 2], *fd[top // 2])
        i = len(d) - 2
        while i >= 0:
            di = d[i]
            if di and i != 0:
                operands(aff[abs(di) // 2])
                self.run("DAP" if di > 0 else "DAM")
            elif i >= 1 and (d[i - 1] == 0 or i - 1 == 0) and not (abs(di) > 1):
                self.run("DBL2")
                i -= 1
                di = d[i]                                                 # (i == 0: its addition is skipped, not its f_d)
            else:
                self.run("DBL")
            if abs(di) > 1:
                operands(fd[abs(di) // 2])
                self.run("FMP" if di > 0 else "FMM")
            i -= 1
        self.run("NORM")
        return [self.get(k) for k in ("n1", "n2", "fm")]

    def add2(self, a, b):
        return self.store(self.value(a) + self.value(b))

    def miller_table(self, xc, yc, table, n):
        """Launch 1 in its table form: e(K, C) over the normalised line table of the key point K (TD / TDA segments;
        the coefficients of a segment's steps are put in their slots before it, as the kernel's prefetch does)."""
        self.P = self.PT
        self.V = {}
        for k, v in (("ax", xc), ("ay", yc), ("v0", 1), ("v2", 1)):
            self.put(k, self.mont(v))
        self.put("v1", 0)
        s = 0
        d = naf(n)
        for i in range(len(d) - 2, -1, -1):
            both = bool(d[i]) and i != 0
            self.put("ta1", self.mont(table[s][0]))
            self.put("tb1", self.mont(table[s][1]))
            if both:
                self.put("ta2", self.mont(table[s + 1][0]))
                self.put("tb2", self.mont(table[s + 1][1]))
            self.run("TDA" if both else "TD")
            s += 2 if both else 1
        assert s == len(table)
        self.run("NORM")
        return [self.get(k) for k in ("n1", "n2", "fm")]

    def final(self, parked, inv, l):
        """Launch 2: parked values and inv = R^2 / N(f) mod p (what k_coop_invert writes) -> the two output slots."""
        self.P = self.PF
        self.V = {}
        for k, v in zip(("n1", "n2", "fm"), parked):
            self.V[self.P.phys[k]] = v
        self.put("inv", inv)
        self.put("raw1", 1)
        self.run("H")
        for i in range(l.bit_length() - 2, -1, -1):
            self.run("LSQ")
            if (l >> i) & 1:
                self.run("LMU")
        self.run("OUT")
        return self.get("out0"), self.get("out1")

    def _finish(self, parked, l):
        N = (self.value(parked[0]) + self.value(parked[1])) % self.p
        inv = self.R * self.R * pow(N, -1, self.p) % self.p
        o0, o1 = self.final(parked, inv, l)
        return self.value(o0) % self.p, self.value(o1) % self.p

    def pairing(self, ax, ay, bx, by, n, l):
        """e(A, B); plain residues in and out."""
        return self._finish(self.miller(ax, ay, bx, by, n), l)

    def pairing_w(self, ax, ay, bx, by, n, l, w):
        """e(A, B) with the width-w Miller loop."""
        return self._finish(self.miller_w(ax, ay, bx, by, n, w), l)

    def pairing_table(self, xc, yc, table, n, l):
        """e(K, C) from K's line table (coop_model.line_table)."""
        return self._finish(self.miller_table(xc, yc, table, n), l)


def to_quad(v: int, nl: int) -> np.ndarray:
    """4 x M tight limbs of a non-negative value below 2^(29 NL)."""
    m = (nl + 3) // 4
    out = np.zeros((4, m), dtype=I32)
    assert 0 <= v < 1 << (LIMB * nl)
    for pos in range(nl):
        out[pos // m, pos % m] = (v >> (LIMB * pos)) & MASK
    return out


def from_quad(x: np.ndarray) -> int:
    m = x.shape[1]
    return sum(int(x[s, j]) << (LIMB * (s * m + j)) for s in range(4) for j in range(m))


class QuadLaneMachine(QuadValueMachine):
    """Same controller and slot handling; a slot holds 4 x M lane limbs."""

    def __init__(self, p: int, nl: int):
        super().__init__(p, nl)
        self.m = (nl + 3) // 4
        self.jtop = nl - 1 - 3 * self.m            # the top limb (position NL - 1) sits in lane 3 at this index
        assert self.m >= 2 and 0 <= self.jtop < self.m
        self.p_q = to_quad(p, nl).astype(I64)
        self.pinv = (-pow(p, -1, 1 << LIMB)) % (1 << LIMB)
        self.max_limb = 0

    def store(self, v):
        return v if isinstance(v, np.ndarray) else to_quad(v, self.nl)

    def value(self, x):
        return from_quad(x)

    # -- quad_perm moves: lane s reads lane (s + 1) & 3 / (s - 1) & 3 --
    @staticmethod
    def rot_down(x):
        return np.roll(x, -1, axis=0)

    @staticmethod
    def rot_up(x):
        return np.roll(x, 1, axis=0)

    def normalize(self, acc):
        """One carry pass (quad_normalize): exact inside a lane, the lane's carry-out added lazily to the two lowest
        limbs of the lane above.  Lane 3 keeps everything at position NL - 1 and holds zeros beyond it."""
        m, jt = self.m, self.jtop
        acc = acc.astype(I64)
        x = np.zeros((4, m), dtype=I64)
        cy = np.zeros(4, dtype=I64)
        is3 = np.array([0, 0, 0, 1], dtype=bool)
        for j in range(m):
            t = acc[:, j] + cy
            lo = t & I64(MASK)
            cy = t >> I64(LIMB)
            if j == jt:
                lo = np.where(is3, t, lo)            # unmasked top limb (fits 32 bits: checked below)
                cy = np.where(is3, 0, cy)
            elif j > jt:
                assert t[3] == 0, "lane 3 beyond the top limb"
            x[:, j] = lo
        cin = self.rot_up(cy)                        # lane 0 takes lane 3's carry-out: zero
        assert cin[0] == 0
        t0 = x[:, 0] + cin
        if jt == 0:
            x[:, 0] = np.where(is3, t0, t0 & I64(MASK))
            c1 = np.where(is3, 0, t0 >> I64(LIMB))
        else:
            x[:, 0] = t0 & I64(MASK)
            c1 = t0 >> I64(LIMB)
        x[:, 1] = x[:, 1] + c1
        assert np.all(np.abs(x) < (1 << 31))
        self.max_limb = max(self.max_limb, int(np.abs(x).max()))
        return x.astype(I32)

    def combo(self, form, K):
        acc = np.zeros((4, self.m), dtype=I64)
        for s, c in sorted(form.items(), key=lambda kv: self.P.phys[kv[0]]):
            acc = acc + I64(c) * self.get(s).astype(I64)                     # v_mad_i64_i32
        return acc + I64(K) * self.p_q

    def mul(self, a, b):
        """a, b: 4 x M signed limbs (one pass normalised).  Returns the unnormalised int64 accumulators of
        a*b/R + (multiple of p)/R, accumulator j of lane s at position s*M + j."""
        m = self.m
        a64, b64 = a.astype(I64), b.astype(I64)
        acc = np.zeros((4, m), dtype=I64)
        for i in range(self.nl):
            ai = a64[i // m, i % m]                                          # v_mov_b32_dpp quad_perm broadcast
            acc = acc + ai * b64                                             # M x v_mad_i64_i32
            t0 = int(acc[0, 0]) & 0xFFFFFFFF
            q = (t0 * self.pinv) & MASK                                      # v_mul_lo_u32, v_and_b32, broadcast from lane 0
            acc = acc + I64(q) * self.p_q                                    # M x v_mad_u64_u32
            c = acc[:, 0] >> I64(LIMB)                                       # v_ashrrev_i64
            lo = acc[:, 0] & I64(MASK)
            assert lo[0] == 0
            up = self.rot_down(lo)                                           # lane s takes the low bits of lane s + 1
            acc = np.concatenate([acc[:, 1:], up.reshape(4, 1)], axis=1)
            acc[:, 0] += c
            if m > 15:                                                       # quad_row: one mid-life carry per accumulator
                f = m // 2 - 1
                acc[:, f + 1] += (acc[:, f] >> I64(32)) * I64(1 << (32 - LIMB))   # upper dword: v_mad_i64_i32
                acc[:, f] &= I64(0xFFFFFFFF)
            assert np.all(np.abs(acc) < (1 << 62) + (1 << 61)), "accumulator headroom"
        return acc

    def exec_uop(self, u):
        if u.kind == "mul":
            a = self.normalize(self.combo(u.A, u.KA))
            b = self.normalize(self.combo(u.B, u.KB))
            t = self.mul(a, b)
            if u.E:
                t = t + self.combo(u.E, u.KE)
            return self.normalize(t)
        return self.normalize(self.combo(u.E, u.KE))

    def tight(self, x):
        """Exact carry resolution (quad_tight): four passes, each exact inside the lanes and handing the carries one
        lane up."""
        m, jt = self.m, self.jtop
        x = x.astype(I64)
        is3 = np.array([0, 0, 0, 1], dtype=bool)
        for _ in range(4):
            cy = np.zeros(4, dtype=I64)
            for j in range(m):
                t = x[:, j] + cy
                lo = t & I64(MASK)
                cy = t >> I64(LIMB)
                if j == jt:
                    lo = np.where(is3, t, lo)
                    cy = np.where(is3, 0, cy)
                x[:, j] = lo
            x[:, 0] += self.rot_up(cy)
        return x

    def canonical(self, x):
        """Tight limbs of the representative in [0, p) of a value in [0, 2p)."""
        a = self.tight(x)
        d = self.tight(a - self.p_q)
        neg = d[3, self.jtop] < 0                                            # broadcast from lane 3
        r = a if neg else d
        assert np.all(r >= 0) and np.all(r <= MASK)
        return r

    def canon(self, x):
        return self.canonical(x)                                  # quad_canonical: tight limbs of the value mod p

    def add2(self, a, b):
        return (a.astype(I64) + b.astype(I64)).astype(I32)        # limb by limb, as the kernel adds f_d's components

    def pairing_w(self, ax, ay, bx, by, n, l, w):
        return self._finish_parked(self.miller_w(ax, ay, bx, by, n, w), l)

    def pairing(self, ax, ay, bx, by, n, l):
        return self._finish_parked(self.miller(ax, ay, bx, by, n), l)

    def _finish_parked(self, parked, l):
        nt = self.tight(parked[0].astype(I64) + parked[1].astype(I64))       # what launch 1 hands to the inversion kernel
        N = from_quad(nt)
        assert np.all(nt >= 0) and np.all(nt <= MASK) and 0 <= N < 4 * self.p
        inv = self.R * self.R * pow(N, -1, self.p) % self.p
        o0, o1 = self.final(parked, inv, l)
        return from_quad(self.canonical(o0)), from_quad(self.canonical(o1))
