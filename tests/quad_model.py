"""Models of the lane-group pairing kernel (bgn_amd/csrc/quad/) for the CPU tests.

The kernel gives one pairing to 16 lanes of a wave: four quads, each running one micro-op of a round of the tables
tools/coop/gen_prog.py schedules for four workers (the step programs of the wave-cooperative kernel, same
formulas).  Inside a quad a field element is split over the four lanes, M = ceil(NL / 4) limbs of 28 bits per lane
(lane s holds limbs s*M .. s*M + M - 1), and a Montgomery product is NL rows of {broadcast one limb of a inside the
quad, multiply-add into the lane's M accumulators, quotient digit from lane 0, multiply-add of p, retire the lowest
accumulator: its low 28 bits go to the lane below, the rest to the next accumulator}.

  * QuadValueMachine — the four-worker tables on Python integers (program, schedule, slot allocation, bounds).
  * QuadLaneMachine  — slots hold 4 x M signed 32-bit limbs (numpy) and every step is the kernel's instruction-level
    arithmetic with 32 / 64-bit wrap-around: what bgn_amd/csrc/quad/quad.hpp implements.
TEST INFRASTRUCTURE: not used by the product.
"""
from __future__ import annotations

import numpy as np

import coop_model as cm
from coop_model import LIMB, MASK, gen_prog

I64 = np.int64
I32 = np.int32
U64 = np.uint64
U32 = np.uint32

_QPROGRAM = None


def program():
    global _QPROGRAM
    if _QPROGRAM is None:
        _QPROGRAM = gen_prog.build_program(gen_prog.QUAD_W, gen_prog.QUAD_W)
    return _QPROGRAM


def nl_for(p: int) -> int:
    need = (p.bit_length() + 9 + 27) // 28
    return next(x for x in (10, 19, 38) if x >= need)


class QuadValueMachine(cm.ValueMachine):
    def __init__(self, p: int, nl: int):
        super().__init__(p, nl)
        self.P = program()


def to_quad(v: int, nl: int) -> np.ndarray:
    """4 x M tight limbs of a non-negative value below 2^(28 NL)."""
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
    """Same controller; slots hold 4 x M lane limbs."""

    def __init__(self, p: int, nl: int):
        super().__init__(p, nl)
        self.m = (nl + 3) // 4
        self.jtop = nl - 1 - 3 * self.m            # the top limb (position NL - 1) sits in lane 3 at this index
        assert self.m >= 2 and 0 <= self.jtop < self.m
        self.p_q = to_quad(p, nl).astype(I64)
        self.pinv = (-pow(p, -1, 1 << LIMB)) % (1 << LIMB)
        self.L = {}
        self.max_limb = 0

    # -- quad_perm moves: lane s reads lane (s + 1) & 3 / (s - 1) & 3 / lane k --
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
            acc = acc + I64(c) * self.L[s].astype(I64)                       # v_mad_i64_i32
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

    def run(self, seg):
        rounds = dict(self.P.segments)[seg]
        for us in rounds:
            res = [(u.dst, self.exec_uop(u)) for u in us]
            for k, v in res:
                self.L[k] = v
                val = from_quad(v)
                assert 0 <= val < self.P.bound[k] * self.p, "lane value of %s out of its bound" % k
            self.rounds_run += 1

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

    def pairing(self, ax, ay, bx, by, n, l):
        for k, v in {"ax": self.mont(ax), "ay": self.mont(ay), "bx": self.mont(bx), "by": self.mont(by),
                     "one": self.mont(1), "raw1": 1, "zero": 0}.items():
            self.L[k] = to_quad(v, self.nl)
        for k, s in {"X@0": "ax", "Y@0": "ay", "Z@0": "one", "ZZ@0": "one", "W@0": "one", "v0@0": "one", "v1@0": "zero",
                     "v2@0": "one"}.items():
            self.L[k] = self.L[s].copy()
        self.V = cm._Unused()
        cm.ValueMachine.pairing(self, ax, ay, bx, by, n, l)
        return from_quad(self.canonical(self.L["out0"])), from_quad(self.canonical(self.L["out1"]))
