"""Models of the wave-cooperative pairing kernel (bgn_amd/csrc/coop/) for the CPU tests.

Two levels, both driven by the micro-op tables of tools/coop/gen_prog.py:
  * ValueMachine — slots hold Python integers; a micro-op is evaluated exactly as the kernel defines it
    (operands made non-negative with K*p, Montgomery product (A*B + Q*p)/R with the unique Q < R).  Checks the
    program: formulas, schedule (a round reads every operand before any write), slot allocation, bounds.
  * LaneMachine — slots hold 64 lanes of 32-bit words (numpy), one 29-bit limb per lane, and every step is the
    kernel's own instruction-level arithmetic with its 32/64-bit wrap-around: signed multiply-add, quotient digit
    from lane 0, lane shifts, one-pass normalisation.  Checks the arithmetic the HIP code implements.
TEST INFRASTRUCTURE: not used by the product.
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "coop"))
import gen_prog  # noqa: E402

LIMB = 29                      # bgn_amd/csrc/consts.hpp LIMB_BITS
MASK = (1 << LIMB) - 1
_PROGRAM = None


def program():
    global _PROGRAM
    if _PROGRAM is None:
        _PROGRAM = gen_prog.build_program()
    return _PROGRAM


def nl_for(p: int) -> int:
    need = (p.bit_length() + 9 + LIMB - 1) // LIMB
    return next(x for x in (3, 10, 19, 36, 37) if x >= need)


def naf(n: int):
    d = []
    while n:
        if n & 1:
            z = 2 - (n & 3)
            d.append(z)
            n -= z
        else:
            d.append(0)
        n >>= 1
    return d


class ValueMachine:
    def __init__(self, p: int, nl: int):
        self.p, self.nl = p, nl
        self.R = 1 << (LIMB * nl)
        self.pinvR = (-pow(p, -1, self.R)) % self.R
        self.P = program()
        self.V = {}
        self.rounds_run = 0
        self.products = 0

    def mont(self, x):
        return x * self.R % self.p

    def _combo(self, form, K):
        v = sum(c * self.V[s] for s, c in form.items()) + K * self.p
        assert v >= 0, "negative operand"
        return v

    def _mul(self, A, B):
        assert A * B < self.R * self.p, "Montgomery input condition violated"
        Q = (A * B * self.pinvR) % self.R
        self.products += 1
        return (A * B + Q * self.p) // self.R

    def run(self, seg):
        rounds = dict(self.P.segments)[seg]
        for us in rounds:
            res = []
            for u in us:
                if u.kind == "mul":
                    v = self._mul(self._combo(u.A, u.KA), self._combo(u.B, u.KB))
                    if u.E:
                        v += self._combo(u.E, u.KE)
                else:
                    v = self._combo(u.E, u.KE)
                assert v < self.P.bound[u.dst] * self.p, "bound of %s exceeded" % u.dst
                res.append((u.dst, v))
            for k, v in res:          # all reads of a round precede its writes
                self.V[k] = v
            self.rounds_run += 1

    def pairing(self, ax, ay, bx, by, n, l):
        """e(A, B) as the kernel's controller sequences it; inputs/outputs are plain residues."""
        V = self.V
        V.update({"ax": self.mont(ax), "ay": self.mont(ay), "bx": self.mont(bx), "by": self.mont(by),
                  "one": self.mont(1), "raw1": 1, "zero": 0})
        V.update({"X@0": V["ax"], "Y@0": V["ay"], "Z@0": V["one"], "ZZ@0": V["one"], "W@0": V["one"],
                  "v0@0": V["one"], "v1@0": 0, "v2@0": V["one"]})
        par = 0
        d = naf(n)
        i = len(d) - 2
        while i >= 0:
            if d[i] and i != 0:
                self.run(("DAP%d" if d[i] > 0 else "DAM%d") % par)       # doubling + addition of +-A, one segment
                i -= 1
            elif i >= 1 and (i == 1 or d[i - 1] == 0):
                self.run("DD%d" % par)                                    # two plain doublings, one segment
                i -= 2
            else:
                self.run("DBL%d" % par)
                i -= 1
            par ^= 1
        return self._finish(par, l)

    def _finish(self, par, l):
        V = self.V
        self.run("NORM%d" % par)
        self.run("INV0")
        ip = 0
        e = self.p - 2
        for i in range(e.bit_length()):                                      # right to left
            self.run(("IMU%d" if (e >> i) & 1 else "ISQ%d") % ip)
            ip ^= 1
        self.run("H%d" % ip)
        lp = 0
        for i in range(l.bit_length() - 2, -1, -1):
            self.run("LSQ%d" % lp)
            lp ^= 1
            if (l >> i) & 1:
                self.run("LMU%d" % lp)
                lp ^= 1
        self.run("OUT%d" % lp)
        return V["out0"] % self.p, V["out1"] % self.p



def line_table(px, py, n, p):
    """(a_s / c_s, b_s / c_s) of every Miller step of f_{n,P}: the per-key table of fixedpair.hpp after
    fixed_normalize_lane, as plain residues (Jacobian doubling / mixed addition along the NAF of n; the line at
    phi(C) is (a*xC + b) + i*c*yC)."""
    X, Y, Z = px, py, 1
    out = []
    d = naf(n)
    for i in range(len(d) - 2, -1, -1):
        ZZ = Z * Z % p
        M = (3 * X * X + ZZ * ZZ) % p
        YY = Y * Y % p
        S = 4 * X * YY % p
        Z3 = 2 * Y * Z % p
        a, b, c = M * ZZ % p, (M * X - 2 * YY) % p, Z3 * ZZ % p
        ci = pow(c, p - 2, p)
        out.append((a * ci % p, b * ci % p))
        X3 = (M * M - 2 * S) % p
        Y3 = (M * (S - X3) - 8 * YY * YY) % p
        X, Y, Z = X3, Y3, Z3
        if d[i] and i != 0:
            ys = py if d[i] > 0 else (-py) % p
            ZZ = Z * Z % p
            rr = (ys * ZZ * Z - Y) % p
            H = (px * ZZ - X) % p
            Z3 = Z * H % p
            a, b, c = rr, (rr * px - Z3 * ys) % p, Z3
            ci = pow(c, p - 2, p)
            out.append((a * ci % p, b * ci % p))
            HH = H * H % p
            HHH = H * HH % p
            XHH = X * HH % p
            X3 = (rr * rr - HHH - 2 * XHH) % p
            Y3 = (rr * (XHH - X3) - Y * HHH) % p
            X, Y, Z = X3, Y3, Z3
    return out


def _pairing_table(self, xc, yc, table, n, l):
    """e(P, C) over P's line table as the kernel's controller sequences it (TD / TDA segments, the coefficients of
    a segment's steps in the slot set of its parity); plain residues in and out."""
    V = self.V
    V.update({"ax": self.mont(xc), "ay": self.mont(yc), "one": self.mont(1), "raw1": 1, "zero": 0})
    V.update({"v0@0": V["one"], "v1@0": 0, "v2@0": V["one"]})
    par, s = 0, 0
    d = naf(n)
    for i in range(len(d) - 2, -1, -1):
        both = bool(d[i]) and i != 0
        V["ta1@%d" % par], V["tb1@%d" % par] = self.mont(table[s][0]), self.mont(table[s][1])
        if both:
            V["ta2@%d" % par], V["tb2@%d" % par] = self.mont(table[s + 1][0]), self.mont(table[s + 1][1])
        self.run(("TDA%d" if both else "TD%d") % par)
        s += 2 if both else 1
        par ^= 1
    assert s == len(table)
    return self._finish(par, l)


ValueMachine.pairing_table = _pairing_table


# ---------------------------------------------------------------------------------------------------------------
# lane level
# ---------------------------------------------------------------------------------------------------------------
U32 = np.uint32
U64 = np.uint64
I64 = np.int64
I32 = np.int32


def to_lanes(v: int, nl: int) -> np.ndarray:
    out = np.zeros(64, dtype=U32)
    for j in range(nl):
        out[j] = (v >> (LIMB * j)) & MASK
    assert v >> (LIMB * nl) == 0
    return out


def from_lanes(x: np.ndarray) -> int:
    """Value of a lane vector of signed 32-bit limbs."""
    return sum(int(np.int32(x[j])) << (LIMB * j) for j in range(64))


class LaneMachine(ValueMachine):
    """Same controller; slots hold lane vectors (uint32 bit patterns of signed limbs)."""

    def __init__(self, p: int, nl: int):
        super().__init__(p, nl)
        self.p_l = to_lanes(p, nl)
        self.pinv = (-pow(p, -1, 1 << LIMB)) % (1 << LIMB)
        lane = np.arange(64)
        self.M = np.where(lane < nl - 1, U32(MASK), U32(0xFFFFFFFF)).astype(U32)      # keep mask of the normalisation
        self.H = np.where(lane < nl - 1, U32(0xFFFFFFFF), U32(0)).astype(U32)         # carry-out mask
        self.L = {}

    # -- the kernel's primitives --
    @staticmethod
    def shr1(x):       # wave_shr:1, bound_ctrl:0 — lane j reads lane j-1, lane 0 reads 0
        return np.concatenate([np.zeros(1, dtype=x.dtype), x[:-1]])

    @staticmethod
    def shl1(x):       # wave_shl:1 — lane j reads lane j+1, lane 63 reads 0
        return np.concatenate([x[1:], np.zeros(1, dtype=x.dtype)])

    def normalize64(self, acc):
        """acc: int64 per lane -> uint32 limbs after one carry pass: lo + (carry of the lane below)."""
        lo = (acc.astype(U64) & U64(0xFFFFFFFF)).astype(U32) & self.M
        hi = ((acc >> I64(LIMB)).astype(U64) & U64(0xFFFFFFFF)).astype(U32) & self.H          # v_alignbit_b32
        return (lo + self.shr1(hi)).astype(U32)

    def combo(self, form, K):
        acc = np.zeros(64, dtype=I64)
        for s, c in sorted(form.items(), key=lambda kv: self.P.phys[kv[0]]):
            acc = acc + I64(c) * self.L[s].astype(I32).astype(I64)                            # v_mad_i64_i32
        acc = acc + I64(K) * self.p_l.astype(I64)
        return acc

    def mul(self, a, b):
        """a, b: uint32 patterns of signed limbs (one pass normalised).  Returns the unnormalised int64 lanes."""
        a64 = a.astype(I32).astype(I64)
        acc = np.zeros(64, dtype=I64)
        for i in range(self.nl):
            bi = I64(np.int32(b[i]))                                                          # v_readlane_b32
            acc = acc + a64 * bi                                                              # v_mad_i64_i32
            t0 = int(acc[0]) & 0xFFFFFFFF                                                     # v_readfirstlane_b32
            q = (t0 * self.pinv) & MASK                                                       # s_mul_i32, s_and_b32
            acc = acc + self.p_l.astype(I64) * I64(q)                                         # v_mad_u64_u32
            lo = (acc.astype(U64) & U64(MASK)).astype(I64)
            hi = (acc >> I64(LIMB))                                                           # fits 32 bits, signed
            assert np.all(np.abs(hi) < (1 << 31))
            acc = self.shl1(lo) + hi                                                          # lane j takes lo of lane j+1
            assert acc[self.nl:].sum() == 0 and np.all(acc[self.nl:] == 0)
        return acc

    def exec_uop(self, u):
        if u.kind == "mul":
            a = self.normalize64(self.combo(u.A, u.KA))
            b = self.normalize64(self.combo(u.B, u.KB))
            assert np.all(np.abs(a.astype(I32).astype(I64)) <= (1 << LIMB) + (1 << 12))
            t = self.mul(a, b)
            if u.E:
                t = t + self.combo(u.E, u.KE)
            return self.normalize64(t)
        return self.normalize64(self.combo(u.E, u.KE))

    def run(self, seg):
        rounds = dict(self.P.segments)[seg]
        for us in rounds:
            res = [(u.dst, self.exec_uop(u)) for u in us]
            for k, v in res:
                self.L[k] = v
                val = from_lanes(v)
                assert 0 <= val < self.P.bound[k] * self.p, "lane value of %s out of its bound" % k
            self.rounds_run += 1

    def canonical(self, x):
        """Tight limbs of the value of x reduced into [0, p): the kernel's final pass (value < 2p, >= 0)."""
        acc = x.astype(I32).astype(I64)
        for _ in range(self.nl):
            acc = self.normalize64(acc).astype(I32).astype(I64)
        d = acc - self.p_l.astype(I64)
        for _ in range(self.nl):
            d = self.normalize64(d).astype(I32).astype(I64)
        neg = d[self.nl - 1] < 0
        return (acc if neg else d)

    def pairing(self, ax, ay, bx, by, n, l):
        for k, v in {"ax": self.mont(ax), "ay": self.mont(ay), "bx": self.mont(bx), "by": self.mont(by),
                     "one": self.mont(1), "raw1": 1, "zero": 0}.items():
            self.L[k] = to_lanes(v, self.nl)
        for k, s in {"X@0": "ax", "Y@0": "ay", "Z@0": "one", "ZZ@0": "one", "W@0": "one", "v0@0": "one", "v1@0": "zero",
                     "v2@0": "one"}.items():
            self.L[k] = self.L[s].copy()
        self.V = _Unused()
        ValueMachine.pairing(self, ax, ay, bx, by, n, l)
        re = from_lanes(self.canonical(self.L["out0"]).astype(np.int64).astype(U32))
        im = from_lanes(self.canonical(self.L["out1"]).astype(np.int64).astype(U32))
        return re, im


class _Unused(dict):
    """The lane machine keeps its state in L; the value controller's bookkeeping writes are dropped."""

    def update(self, *a, **k):
        pass

    def __getitem__(self, k):
        return 0
