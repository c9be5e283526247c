#!/usr/bin/env python3
"""Step programs of the wave-cooperative pairing kernel (bgn_amd/csrc/coop/), as scheduled micro-op tables.

The small-batch kernel gives ONE pairing to a workgroup of W = 8 waves (two per SIMD of a CU).  A field element
is spread over the lanes of a wave, one 28-bit limb per lane, so a value is one VGPR and a Montgomery product is
NL broadcast-multiply-shift steps instead of 2*NL^2 multiply-adds in one lane.  What the waves do is fixed by
this file: every segment of the pairing (Miller doubling step, addition steps, the pieces of the final
exponentiation; formulas of bgn_amd/csrc/pairing.hpp, which restates `Pair` of bgn.go:300) is a DAG of micro-ops

    dst = (sum ca_k * V[ia_k] + KA*p) * (sum cb_k * V[ib_k] + KB*p) / R  +  sum ce_k * V[ie_k] + KE*p     (mul)
    dst =  sum ce_k * V[ie_k] + KE*p                                                                        (lin)

over value slots V[] in LDS.  The generator tracks an upper bound (in units of p) for every slot, chooses the
multiples of p that keep every operand non-negative, checks the Montgomery input condition
bound(A) * bound(B) <= 2^9 <= R/p, schedules the DAG into rounds of at most W independent micro-ops
(one per wave, a workgroup barrier between rounds) and assigns LDS slots by liveness.  Loop-carried state is
ping-ponged between two slot sets so that no slot is read and written in the same round.

Output: bgn_amd/csrc/coop/coop_prog.inc (committed; regenerate with `python tools/coop/gen_prog.py`).
tests/test_coop_program.py executes the same tables on Python integers against the oracle.
"""
from __future__ import annotations

import os
import sys

W = 8                 # waves per pairing: two per SIMD; slots k and k + 4 of a round share a SIMD
BASE = 4              # products a round takes freely (one per SIMD); more only when they are on the critical path
QUAD_W = 4            # lane-group kernel: quads of lanes per pairing (a round costs the same with one product or four)
QUAD_TRIES = 400      # schedules tried per Miller segment of the lane-group kernel (Program.segment)
MAX_TERMS = 4         # terms per linear operand
BOUND_PRODUCT = 512   # bound(A) * bound(B) <= 2^9 <= R/p (engine.cpp chooses NL so)
MUL_WEIGHT, LIN_WEIGHT = 10, 1


class Form(dict):
    """Integer linear form over slot names."""

    def __add__(self, o):
        r = Form(self)
        for k, v in o.items():
            r[k] = r.get(k, 0) + v
            if r[k] == 0:
                del r[k]
        return r

    def __neg__(self):
        return Form({k: -v for k, v in self.items()})

    def __sub__(self, o):
        return self + (-o)

    def __rmul__(self, c):
        return Form({k: c * v for k, v in self.items()}) if c else Form()


def S(name):
    return Form({name: 1})


class UOp:
    def __init__(self, kind, dst, A, B, E):
        self.kind, self.dst, self.A, self.B, self.E = kind, dst, A, B, E
        self.KA = self.KB = self.KE = 0
        self.round = -1

    def reads(self):
        r = set()
        for f in (self.A, self.B, self.E):
            if f:
                r |= set(f)
        return r


class Builder:
    """Accumulates the micro-ops of one segment; `bound` holds the slot bounds (units of p)."""

    def __init__(self, bound, prefix):
        self.bound = bound
        self.prefix = prefix
        self.uops = []
        self.ntemp = 0

    def _combo(self, form):
        """(K, upper bound) of form + K*p with K the smallest multiple of p that makes it non-negative."""
        assert len(form) <= MAX_TERMS, "too many terms: %r" % (form,)
        K = sum(-c * self.bound[s] for s, c in form.items() if c < 0)
        ub = sum(c * self.bound[s] for s, c in form.items() if c > 0) + K
        return K, ub

    def _dst(self, out):
        if out is None:
            out = "%s.t%d" % (self.prefix, self.ntemp)
            self.ntemp += 1
        return out

    def mul(self, A, B, E=None, out=None):
        u = UOp("mul", self._dst(out), Form(A), Form(B), Form(E) if E else None)
        u.KA, ba = self._combo(u.A)
        u.KB, bb = self._combo(u.B)
        assert ba * bb <= BOUND_PRODUCT, "bound product %d * %d too large for %s" % (ba, bb, u.dst)
        bd = 2                                          # (A*B + Q*p)/R < (ba*bb/2^9 + 1) p <= 2p
        if u.E:
            u.KE, be = self._combo(u.E)
            bd += be
        assert u.dst not in self.bound or self.bound[u.dst] >= bd, "bound of %s: %d > declared %d" % (
            u.dst, bd, self.bound.get(u.dst, -1))
        self.bound.setdefault(u.dst, bd)
        self.uops.append(u)
        return S(u.dst)

    def lin(self, E, out=None):
        u = UOp("lin", self._dst(out), None, None, Form(E))
        u.KE, be = self._combo(u.E)
        assert u.dst not in self.bound or self.bound[u.dst] >= be
        self.bound.setdefault(u.dst, be)
        self.uops.append(u)
        return S(u.dst)


# Round-time model of the search below (cycles at a 1024- / 512-bit key averaged): a round costs its fixed part
# (barrier, operand reads, normalisation, store) plus one product, or ROUND_HEAVY products' worth when more than
# `base` products make two waves share a SIMD.
ROUND_FIXED, ROUND_PRODUCT, ROUND_HEAVY = 1000, 1700, 1.33


def schedule(uops, w=W, base=BASE, anti=None, rng=None, jitter=0.0):
    """Scheduling into rounds of at most w micro-ops; a micro-op reads only slots written in earlier rounds (or
    never written in this segment).  Products are taken in order of their longest path to a sink.  Two waves on a
    SIMD share its issue slots, so a round with more than `base` products takes about 1.33x as long (measured) as one with at
    most `base` — but then it may as well hold w of them.  Every round therefore takes either its `base` best ready
    products or up to w, and the choice per round is searched exhaustively against the round-time model above
    (a segment is at most a dozen rounds).  Linear micro-ops cost no product: they fill free waves of the first
    round they are ready in.  Measured against 4 waves with one product each: a doubling 3 rounds instead of 5.
    anti: {writer index: reader indices} — the writer overwrites (in place) a slot those micro-ops read, so it may not
    run in an earlier round than any of them (the same round is fine where a round's reads precede its writes: the
    lane-group kernel)."""
    anti = anti or {}
    writer = {u.dst: i for i, u in enumerate(uops)}
    assert len(writer) == len(uops), "a slot is written twice in one segment"
    deps = [sorted(writer[s] for s in u.reads() if s in writer) for u in uops]
    for i, d in enumerate(deps):
        assert all(j < i for j in d), "program order must be topological"
    users = [[] for _ in uops]
    for i, d in enumerate(deps):
        for j in d:
            users[j].append(i)
    prio = [0] * len(uops)
    for i in reversed(range(len(uops))):
        wgt = MUL_WEIGHT if uops[i].kind == "mul" else LIN_WEIGHT
        prio[i] = wgt + max((prio[j] for j in users[i]), default=0)
    if rng is not None:                       # a perturbed order of equally (or nearly equally) urgent micro-ops: Program.segment
        prio = [p + rng.random() * jitter for p in prio]    # tries several and keeps the one with the fewest temporaries
    n = len(uops)
    best = [None, None]                       # cost, list of rounds (index lists)

    def search(done, rounds, cost):
        if best[0] is not None and cost >= best[0]:
            return
        if len(done) == n:
            best[0], best[1] = cost, [list(r) for r in rounds]
            return
        r = len(rounds)
        ready = [i for i in range(n) if i not in done and all(j in done and done[j] < r for j in deps[i])]
        assert ready
        muls = sorted((i for i in ready if uops[i].kind == "mul"), key=lambda i: (-prio[i], i))
        lins = sorted((i for i in ready if uops[i].kind != "mul"), key=lambda i: (-prio[i], i))

        def pick(k):
            """At most k products and w micro-ops in priority order; an in-place writer only when every reader of
            the old value has run or runs in this round (found as a fixpoint: drop writers whose readers are
            missing, refill)."""
            pool = list(muls) + list(lins)
            while True:
                sel, nm = [], 0
                for i in pool:
                    if len(sel) >= w:
                        break
                    if uops[i].kind == "mul":
                        if nm >= k:
                            continue
                        nm += 1
                    sel.append(i)
                bad = [i for i in sel if not all(j == i or j in done or j in sel for j in anti.get(i, ()))]
                if not bad:
                    return sel
                pool = [i for i in pool if i not in bad]

        options = sorted({min(len(muls), base), min(len(muls), w)}, reverse=True)
        for k in options:
            if anti:
                take = pick(k)
            else:
                take = muls[:k] + lins[: w - k]
            if not take:
                if anti and k == options[-1]:
                    raise AssertionError("in-place constraints leave nothing to run")
                continue
            c = ROUND_FIXED + (0 if k == 0 else ROUND_PRODUCT * (1 if k <= base else ROUND_HEAVY))
            d2 = dict(done)
            for i in take:
                d2[i] = r
            search(d2, rounds + [take], cost + c)

    search({}, [], 0)
    out = []
    for r, take in enumerate(best[1]):
        for i in take:
            uops[i].round = r
        out.append([uops[i] for i in take])
    return out


class Program:
    def __init__(self, w=W, base=BASE, reads_first=False):
        self.w, self.base = w, base
        self.reads_first = reads_first   # a round's reads all precede its writes (lane-group kernel: one wave, no barrier)
        self.globals = {}      # name -> physical slot (never written by a scheduled micro-op of a loop body reading it)
        self.segments = []     # (name, rounds)
        self.bound = {}
        self.phys = {}
        self.nphys = 0

    def fixed(self, name, bound):
        self.phys[name] = self.nphys
        self.nphys += 1
        self.bound[name] = bound
        return S(name)

    def temps_of(self, name, rounds):
        """Temporaries a schedule of segment `name` needs (the count allocate_temps arrives at)."""
        last_read = {}
        for r, us in enumerate(rounds):
            for u in us:
                for s in u.reads():
                    last_read[s] = r
        free, busy, nxt = 0, {}, 0
        for r, us in enumerate(rounds):
            for s in [s for s, lr in busy.items() if lr < r or (self.reads_first and lr == r)]:
                busy.pop(s)
                free += 1
            for u in us:
                if u.dst in self.phys and not u.dst.startswith(name + ".t"):
                    continue
                if free:
                    free -= 1
                else:
                    nxt += 1
                busy[u.dst] = last_read[u.dst]
        return nxt

    def segment(self, name, build, inplace=None, tries=0):
        """inplace: {new name: old name} — the micro-op writing `new` stores into the slot of `old`, which other
        micro-ops of the segment still read as the old value (needs reads_first).
        tries: beside the schedule of the plain priorities, that many with perturbed priorities (seeds 1 .. tries,
        deterministic) are made and the one with the fewest rounds, then the fewest temporaries, is kept — a segment
        packed into whole rounds keeps more values alive, and the LDS has room for so many slots."""
        b = Builder(self.bound, name)
        build(b)
        anti = None
        if inplace:
            assert self.reads_first
            anti = {}
            for i, u in enumerate(b.uops):
                if u.dst in inplace:
                    anti[i] = [j for j, v in enumerate(b.uops) if inplace[u.dst] in v.reads()]
                    self.phys[u.dst] = self.phys[inplace[u.dst]]
        rounds = schedule(b.uops, self.w, self.base, anti)
        if tries:
            import random
            best = ((len(rounds), self.temps_of(name, rounds)), rounds)
            for t in range(1, tries + 1):
                cand = schedule(b.uops, self.w, self.base, anti, rng=random.Random(t), jitter=12.0)
                k = (len(cand), self.temps_of(name, cand))
                if k < best[0]:
                    best = (k, cand)
            rounds = best[1]
            for r, us in enumerate(rounds):
                for u in us:
                    u.round = r
        if inplace:
            rnd = {u.dst: r for r, us in enumerate(rounds) for u in us}
            for r, us in enumerate(rounds):
                for u in us:
                    for new, old in inplace.items():
                        assert old not in u.reads() or new not in rnd or r <= rnd[new], (name, new, u.dst)
            for new, old in inplace.items():       # two generations in one slot: the later one is written in a later round
                assert old not in rnd or new not in rnd or rnd[old] < rnd[new], (name, new, old)
        self.segments.append((name, rounds))
        return rounds

    def allocate_temps(self):
        """Temps are segment-local: a physical slot is free again in the round after its last read."""
        base = self.nphys
        top = base
        for name, rounds in self.segments:
            last_read = {}
            for r, us in enumerate(rounds):
                for u in us:
                    for s in u.reads():
                        last_read[s] = r
            free, busy = [], {}
            nxt = base
            for r, us in enumerate(rounds):
                for s in [s for s, (ph, lr) in busy.items() if lr < r or (self.reads_first and lr == r)]:
                    free.append(busy.pop(s)[0])
                for u in us:
                    if u.dst in self.phys and not u.dst.startswith(name + ".t"):
                        continue
                    assert u.dst in last_read, "dead micro-op %s in %s" % (u.dst, name)
                    if free:
                        ph = free.pop()
                    else:
                        ph = nxt
                        nxt += 1
                    self.phys[u.dst] = ph
                    busy[u.dst] = (ph, last_read[u.dst])
            top = max(top, nxt)
        self.nslots = top


# ---------------------------------------------------------------------------------------------------------------
# The pairing
# ---------------------------------------------------------------------------------------------------------------
# Miller state: V = (X, Y, Z) with ZZ = Z^2 and W = Z^4 carried along (the doubling then needs no product before
# M = 3 X^2 + Z^4 and the line's Z^2 * xB, Z^2 * yB: three rounds deep instead of four), f = v-form
STATE_BOUNDS = {"X": 19, "Y": 19, "Z": 2, "ZZ": 2, "W": 2, "v0": 2, "v1": 2, "v2": 2}


def name(f):
    (k,) = f.keys()
    return k


def f_of(s):      # f = F0 + i F1 from the stored Karatsuba triple
    return s["v0"] - s["v1"], s["v2"] - s["v0"] - s["v1"]


def finish_f(b, F0, F1, cre, cim, so):
    """f <- (F0 + i F1) * (cre + i cim) as the triple (F0*cre, F1*cim, (F0+F1)(cre+cim))."""
    b.mul(F0, cre, out=name(so["v0"]))
    b.mul(F1, cim, out=name(so["v1"]))
    b.mul(F0 + F1, cre + cim, out=name(so["v2"]))


def dbl(b, si, so, O, want_w=True, fold=False):
    """pairing.hpp miller_double: f <- f^2 * l_{V,V}(phi(B)), V <- 2V (Jacobian, a = 1), arranged three products
    deep: X3 = M^2 - 2S is never an operand of a product here (Y3 takes M^2 and X*YY directly), the line takes
    ZZ and W from the state.  O: the operand slots ax, ay, bx, by."""
    bx, by = O["bx"], O["by"]
    X, Y, Z, ZZ, Wq = si["X"], si["Y"], si["Z"], si["ZZ"], si["W"]
    F0, F1 = f_of(si)
    XX = b.mul(X, X)
    YY = b.mul(Y, Y)
    Z3 = b.mul(2 * Y, Z, out=name(so["Z"]))
    g0 = b.mul(F0 + F1, F0 - F1)
    g1h = b.mul(F0, F1)
    ZZxB = b.mul(ZZ, bx)
    ZZyB = b.mul(ZZ, by)
    M = 3 * XX + Wq
    M2 = None if fold else b.mul(M, M)
    XYY = b.mul(X, YY)
    Y4 = b.mul(YY, YY)
    cre = b.mul(M, ZZxB + X, E=-2 * YY)
    cim = b.mul(Z3, ZZyB)                                            # (Z3 ZZ) yB
    ZZ3 = b.mul(Z3, Z3, out=name(so["ZZ"]))
    if fold:
        # four quads: a linear micro-op takes a quad for a whole round, so X3 is made by the product that squares M
        # (one micro-op less: 18 in a doubling, 36 in two or in a doubling with its addition — whole rounds of four)
        X3 = b.mul(M, M, E=-8 * XYY, out=name(so["X"]))              # M^2 - 2S, S = 4 X YY
        b.mul(M, 4 * XYY - X3, E=-8 * Y4, out=name(so["Y"]))         # M (S - X3) - 8 YY^2
    else:
        b.lin(M2 - 8 * XYY, out=name(so["X"]))                       # M^2 - 2S, S = 4 X YY
        b.mul(M, 12 * XYY - M2, E=-8 * Y4, out=name(so["Y"]))        # M (S - X3) - 8 YY^2
    if want_w:
        b.mul(ZZ3, ZZ3, out=name(so["W"]))
    finish_f(b, g0, 2 * g1h, cre, cim, so)


def add(sign, O, fold=False):
    ax, ay, bx, by = O["ax"], O["ay"], O["bx"], O["by"]

    def build(b, si, so):
        """pairing.hpp miller_add: f <- f * l_{V,sA}(phi(B)), V <- V + sA (mixed addition); ZZ comes with the
        state, rr^2 and the line's rr*(xB + xA) are products of their own so that neither X3 nor cre sits on a
        chain, and the new ZZ, W are made for the doubling that follows."""
        X, Y, Z, ZZ = si["X"], si["Y"], si["Z"], si["ZZ"]
        F0, F1 = f_of(si)
        ysA = sign * ay
        ZZZ = b.mul(ZZ, Z)
        xZZ = b.mul(ax, ZZ)
        yZ3 = b.mul(ysA, ZZZ)
        rrr = yZ3 - Y
        H = xZZ - X
        Z3 = b.mul(Z, H, out=name(so["Z"]))
        HH = b.mul(H, H)
        HHH = b.mul(H, HH)
        XHH = b.mul(X, HH)
        if fold:                                                     # (see dbl)
            YH = b.mul(b.mul(Y, H), HH)                              # Y H^3 without waiting for H^3
            X3 = b.mul(rrr, rrr, E=-HHH - 2 * XHH, out=name(so["X"]))
            b.mul(rrr, XHH - X3, E=-YH, out=name(so["Y"]))           # rr (XHH - X3) - Y HHH
        else:
            rr2 = b.mul(rrr, rrr)
            b.lin(rr2 - HHH - 2 * XHH, out=name(so["X"]))
            YH = b.mul(b.mul(Y, H), HH)                              # Y H^3 without waiting for H^3
            b.mul(rrr, 3 * XHH + HHH - rr2, E=-YH, out=name(so["Y"]))    # rr (XHH - X3) - Y HHH
        Z3y = b.mul(Z3, ysA)
        T = b.mul(rrr, bx + ax)
        cim = b.mul(Z3, by)
        ZZ3 = b.mul(Z3, Z3, out=name(so["ZZ"]))
        b.mul(ZZ3, ZZ3, out=name(so["W"]))
        finish_f(b, F0, F1, T - Z3y, cim, so)
    return build


def tab_line(b, t, which, O):
    """Line of a table step at phi(C) = (ax, ay): (a'*xC + b') + i*yC  (fixedpair.hpp); cre < 3."""
    return b.mul(t["ta%d" % which], O["ax"]) + t["tb%d" % which], O["ay"]


def build_program(w=W, base=BASE):
    P = Program(w, base)
    # operands (canonical Montgomery, < p) and constants
    ax, ay, bx, by = (P.fixed(n, 1) for n in ("ax", "ay", "bx", "by"))
    one = P.fixed("one", 1)          # R mod p
    raw1 = P.fixed("raw1", 1)        # the integer 1: a Montgomery product with it divides by R
    zero = P.fixed("zero", 1)
    # values that outlive a loop
    n1, n2, fm = (P.fixed(n, 2) for n in ("n1", "n2", "fm"))            # F0^2, F1^2, F0*F1
    h0, h1 = P.fixed("h0", 2), P.fixed("h1", 2)                          # h = conj(f)^2 / N(f)
    out0, out1 = P.fixed("out0", 2), P.fixed("out1", 2)
    st = [{k: P.fixed("%s@%d" % (k, par), b) for k, b in STATE_BOUNDS.items()} for par in (0, 1)]
    acc = [P.fixed("acc@%d" % par, 4) for par in (0, 1)]                 # inversion: running product
    sq = [P.fixed("sq@%d" % par, 4) for par in (0, 1)]                   # ... and the squaring chain N^(2^i)
    rr = [(P.fixed("r0@%d" % par, 9), P.fixed("r1@%d" % par, 9)) for par in (0, 1)]   # h^l ladder
    # line coefficients (a_s/c_s, b_s/c_s) of the table steps a segment consumes (fixedpair.hpp), canonical; the
    # kernel loads the next segment's into the other set while the current one runs
    tc = [{k: P.fixed("%s@%d" % (k, par), 1) for k in ("ta1", "tb1", "ta2", "tb2")} for par in (0, 1)]

    O = {"ax": ax, "ay": ay, "bx": bx, "by": by}

    # a NAF digit is followed by a zero, so the loop is a sequence of D (doubling) and DA+- (doubling, then the
    # addition of +-A) steps; DA is scheduled as ONE segment so that the addition's first products overlap the
    # doubling's last ones.  The intermediate state of DA lives in temporaries.
    MID_BOUNDS = dict(STATE_BOUNDS)
    for par in (0, 1):
        si, so = st[par], st[1 - par]
        P.segment("DBL%d" % par, lambda b, si=si, so=so: dbl(b, si, so, O))
        def dbl2(b, si=si, so=so, par=par):                      # two doubling steps in one schedule
            mid = {k: S("DD%d.%s" % (par, k)) for k in STATE_BOUNDS}
            for k, f in mid.items():
                b.bound[name(f)] = MID_BOUNDS[k]
            dbl(b, si, mid, O)
            dbl(b, mid, so, O)
        P.segment("DD%d" % par, dbl2)
        for sign, nm in ((1, "DAP"), (-1, "DAM")):
            def dbladd(b, si=si, so=so, sign=sign, nm=nm, par=par):
                mid = {k: S("%s%d.%s" % (nm, par, k)) for k in STATE_BOUNDS}
                for k, f in mid.items():
                    b.bound[name(f)] = MID_BOUNDS[k]
                del mid["W"]                                     # the addition does not read it
                dbl(b, si, mid, O, want_w=False)
                add(sign, O)(b, mid, so)
            P.segment("%s%d" % (nm, par), dbladd)

    # ---- Miller loop over a normalised per-key line table (fixedpair.hpp miller_loop_fixed: makeL2, the level-1
    # decryption lift).  The evaluation point phi(C) sits in (ax, ay); a step's line is (a'*xC + b') + i*yC, so a
    # doubling step is f <- f^2 * l (six products, two rounds) and the addition that follows a non-zero digit one
    # more Karatsuba product with the next table entry (ten products, three rounds together).
    for par in (0, 1):
        si, so, t = st[par], st[1 - par], tc[par]
        def tdbl(b, si=si, so=so, t=t):
            F0, F1 = f_of(si)
            cre, cim = tab_line(b, t, 1, O)
            g0 = b.mul(F0 + F1, F0 - F1)
            g1h = b.mul(F0, F1)
            finish_f(b, g0, 2 * g1h, cre, cim, so)
        P.segment("TD%d" % par, tdbl)
        def tdbladd(b, si=si, so=so, t=t, par=par):
            F0, F1 = f_of(si)
            cre, cim = tab_line(b, t, 1, O)
            cre2, cim2 = tab_line(b, t, 2, O)
            g0 = b.mul(F0 + F1, F0 - F1)
            g1h = b.mul(F0, F1)
            mid = {k: S("TDA%d.%s" % (par, k)) for k in ("v0", "v1", "v2")}
            for k, f in mid.items():
                b.bound[name(f)] = STATE_BOUNDS[k]
            finish_f(b, g0, 2 * g1h, cre, cim, mid)
            M0, M1 = f_of(mid)
            finish_f(b, M0, M1, cre2, cim2, so)
        P.segment("TDA%d" % par, tdbladd)

    # ---- final exponentiation: f^(p-1) = conj(f)^2 / N(f), then ^l (pairing.hpp final_exp_with_inverse) ----
    for par in (0, 1):
        def norm(b, s=st[par], par=par):
            F0, F1 = f_of(s)
            b.mul(F0, F0, out="n1")
            b.mul(F1, F1, out="n2")
            b.mul(F0, F1, out="fm")
        P.segment("NORM%d" % par, norm)
    N = n1 + n2
    # 1/N = N^(p-2) by Fermat, right to left: s_i = N^(2^i) and acc <- acc * s_i for the set bits run on two
    # waves in the same round, so the chain is bits(p) rounds deep instead of 1.5 * bits(p)
    def inv0(b):
        b.lin(N, out="sq@0")
        b.lin(one, out="acc@0")
    P.segment("INV0", inv0)
    for par in (0, 1):
        def isq(b, par=par):            # bit clear: only the squaring; acc moves to the other set unchanged
            b.mul(sq[par], sq[par], out="sq@%d" % (1 - par))
            b.lin(acc[par], out="acc@%d" % (1 - par))
        def imu(b, par=par):            # bit set
            b.mul(sq[par], sq[par], out="sq@%d" % (1 - par))
            b.mul(acc[par], sq[par], out="acc@%d" % (1 - par))
        P.segment("ISQ%d" % par, isq)
        P.segment("IMU%d" % par, imu)
    for par in (0, 1):
        def hseg(b, par=par):
            b.mul(n1 - n2, acc[par], out="h0")
            b.mul(-2 * fm, acc[par], out="h1")
            b.mul(n1 - n2, acc[par], out="r0@0")
            b.mul(-2 * fm, acc[par], out="r1@0")
        P.segment("H%d" % par, hseg)
    for par in (0, 1):
        r0, r1 = rr[par]
        def f2sq(b, r0=r0, r1=r1, par=par):
            b.mul(r0 + r1, r0 - r1, out="r0@%d" % (1 - par))
            b.mul(2 * r0, r1, out="r1@%d" % (1 - par))
        P.segment("LSQ%d" % par, f2sq)
        def f2mu(b, r0=r0, r1=r1, par=par):
            t1 = b.mul(r1, h1)
            t2 = b.mul(r0, h1)
            b.mul(r0, h0, E=-t1, out="r0@%d" % (1 - par))
            b.mul(r1, h0, E=t2, out="r1@%d" % (1 - par))
        P.segment("LMU%d" % par, f2mu)
        def outseg(b, r0=r0, r1=r1):
            b.mul(r0, raw1, out="out0")
            b.mul(r1, raw1, out="out1")
        P.segment("OUT%d" % par, outseg)
    # loop-carried bounds hold (Builder.mul asserted every declared bound)
    P.allocate_temps()
    return P


# ---------------------------------------------------------------------------------------------------------------
# The lane-group kernel's programs (csrc/quad/)
# ---------------------------------------------------------------------------------------------------------------
# Same formulas, scheduled for QUAD_W quads of lanes.  The quads of a pairing share a wave, so a round's reads all
# precede its writes: the Miller state is updated IN PLACE (no second slot set), a temporary's slot is reused in the
# round of its last read, and the two launches (Miller loop + norms; final exponentiation) have their own slot
# numbering.  That brings a pairing's values down to what two workgroups per CU can hold in LDS (32 slots of
# 4 lanes x M limbs each at a 1024-bit key), i.e. two waves per SIMD.
def build_quad_programs(w=QUAD_W):
    # ---- launch 1: Miller loop over the NAF of n, then F0^2, F1^2, F0*F1 ----
    M = Program(w, w, reads_first=True)
    O = {k: M.fixed(k, 1) for k in ("ax", "ay", "bx", "by")}
    st = {k: M.fixed(k, b) for k, b in STATE_BOUNDS.items()}
    new = {k: S(k + "'") for k in STATE_BOUNDS}
    inplace = {k + "'": k for k in STATE_BOUNDS}

    def declare_new(b, keys=STATE_BOUNDS):
        for k in keys:
            b.bound[k + "'"] = STATE_BOUNDS[k]

    def seg_dbl(b):
        declare_new(b)
        dbl(b, st, new, O, fold=True)
    M.segment("DBL", seg_dbl, inplace, tries=QUAD_TRIES)

    # Two plain doublings in one segment: 36 products in nine rounds where two DBL segments take ten (the second
    # step's first products fill the first step's last round).  Two generations per state slot, as in DAP / DAM below.
    def seg_dbl2(b):
        declare_new(b)
        mid = {k: S("DBL2.%s" % k) for k in STATE_BOUNDS}
        for k, f in mid.items():
            b.bound[name(f)] = STATE_BOUNDS[k]
        dbl(b, st, mid, O, fold=True)
        dbl(b, mid, new, O, fold=True)
    chain2 = {}
    for k in STATE_BOUNDS:
        chain2["DBL2.%s" % k] = k
        chain2[k + "'"] = "DBL2.%s" % k
    M.segment("DBL2", seg_dbl2, chain2, tries=QUAD_TRIES)
    # A doubling and the addition after it: ten rounds for the 38 micro-ops, two fewer than DBL and a lone addition
    # take.  The intermediate state (after the doubling) lives in the state's own slots, like the new one after it:
    # two generations per slot in one segment, each written when every reader of the one before it has run.  With
    # that the Miller loop needs 8 temporaries beside the 12 operand / state slots: 20 value slots, five row blocks
    # of LDS — three workgroups per CU instead of two.
    for sign, nm in ((1, "DAP"), (-1, "DAM")):
        def seg_da(b, sign=sign, nm=nm):
            declare_new(b)
            mid = {k: S("%s.%s" % (nm, k)) for k in STATE_BOUNDS}
            for k, f in mid.items():
                b.bound[name(f)] = STATE_BOUNDS[k]
            del mid["W"]
            dbl(b, st, mid, O, want_w=False, fold=True)
            add(sign, O, fold=True)(b, mid, new)
        chain = {}
        for k in STATE_BOUNDS:
            if k == "W":
                chain["W'"] = "W"                       # the doubling half makes no W: the addition writes the new one
            else:
                chain["%s.%s" % (nm, k)] = k
                chain[k + "'"] = "%s.%s" % (nm, k)
        M.segment(nm, seg_da, chain, tries=QUAD_TRIES)

    # the norms go where X, Y, Z were (dead once the loop is over)
    def seg_norm(b):
        for k in ("n1", "n2", "fm"):
            b.bound[k] = 2
        F0, F1 = f_of(st)
        b.mul(F0, F0, out="n1")
        b.mul(F1, F1, out="n2")
        b.mul(F0, F1, out="fm")
    M.segment("NORM", seg_norm, {"n1": "X", "n2": "Y", "fm": "Z"})

    # ---- the width-w Miller loop (pairing.hpp miller_loop_w) on the lane groups: a digit +-d adds +-dA in one step,
    # f <- f * l * f_d^(+-1).  The odd multiples dA (affine) and their Miller values f_d are made per pairing by the
    # table launches (quad.hpp k_pairing_quad_wtab) from these segments: ADDP = a lone addition step V <- V + (ax, ay)
    # with its line; FMP / FMM = f <- f * (ax + i ay) resp. its conjugate (the operand slots are free between two
    # additions: the kernel loads f_d into them); FOUT = F0, F1 as single values (for the table: canonical after the
    # kernel's conditional subtraction); AFM = affine Montgomery coordinates of the state's (X, Y) from R / Z.
    # Other names — and bounds — for slots whose usual value is dead where these segments run:
    def alias(nm, of, bound):
        M.phys[nm] = M.phys[of]
        M.bound[nm] = bound
        return S(nm)
    one = M.fixed("one", 1)                  # R mod p (the twentieth slot)
    zi = alias("zi", "W", 4)                 # R / Z from the inversion kernel
    alias("axo", "ax", 2)
    alias("ayo", "ay", 2)

    def seg_addp(b):
        declare_new(b)
        add(1, O, fold=True)(b, st, new)
    M.segment("ADDP", seg_addp, inplace, tries=QUAD_TRIES)
    fin = {k + "'": k for k in ("v0", "v1", "v2")}
    for sign, nm in ((1, "FMP"), (-1, "FMM")):
        def seg_fm(b, sign=sign):
            declare_new(b, ("v0", "v1", "v2"))
            F0, F1 = f_of(st)
            finish_f(b, F0, F1, O["ax"], sign * O["ay"], new)
        M.segment(nm, seg_fm, fin)

    def seg_fout(b):
        F0, F1 = f_of(st)
        b.mul(F0, one, out="axo")
        b.mul(F1, one, out="ayo")
    M.segment("FOUT", seg_fout)

    def seg_afm(b):
        zi2 = b.mul(zi, zi)
        zi3 = b.mul(zi2, zi)
        b.mul(st["X"], zi2, out="axo")
        b.mul(st["Y"], zi3, out="ayo")
    M.segment("AFM", seg_afm)

    # One inversion per pairing for the seven multiples (Montgomery's trick): the table launch keeps the running product
    # of their Z (FOUZ = FOUT and pz <- pz * Z), only the last product is inverted, and the Miller launch's prologue walks
    # back from the last multiple: 1 / Z_k = I * (Z_1 .. Z_(k-1)), I <- I * Z_k (AFZ = AFM behind those two products).
    # pz: the first temporary's slot (FOUZ has no temporaries; the kernel loads the product before the segment and
    # stores it after); pzp, zk: slots of f's triple, whose values the Miller launch sets after its prologue.
    M.phys["pz"] = M.phys["pzo"] = M.nphys
    M.bound["pz"] = M.bound["pzo"] = 2
    pz = S("pz")
    pzp = alias("pzp", "v0", 2)
    zk = alias("zk", "v1", 2)
    alias("zio", "W", 4)

    def seg_fouz(b):
        F0, F1 = f_of(st)
        b.mul(F0, one, out="axo")
        b.mul(F1, one, out="ayo")
        b.mul(pz, st["Z"], out="pzo")
    M.segment("FOUZ", seg_fouz)

    def seg_afz(b):
        z1 = b.mul(zi, pzp)                                  # 1 / Z_k
        b.mul(zi, zk, out="zio")                             # the inverse of the shorter product
        z2 = b.mul(z1, z1)
        z3 = b.mul(z2, z1)
        b.mul(st["X"], z2, out="axo")
        b.mul(st["Y"], z3, out="ayo")
    M.segment("AFZ", seg_afz, {"zio": "zi"})
    M.allocate_temps()
    # ---- launch 2: h = conj(f)^2 / N(f), g = h^l, division by R ----
    F = Program(w, w, reads_first=True)
    n1, n2, fm = (F.fixed(k, 2) for k in ("n1", "n2", "fm"))
    inv = F.fixed("inv", 4)                                              # R / N(f), from the inversion kernel
    raw1 = F.fixed("raw1", 1)
    h0, h1 = F.fixed("h0", 2), F.fixed("h1", 2)
    r0, r1 = F.fixed("r0", 9), F.fixed("r1", 9)
    out0, out1 = F.fixed("out0", 2), F.fixed("out1", 2)

    def seg_h(b):
        b.mul(n1 - n2, inv, out="h0")
        b.mul(-2 * fm, inv, out="h1")
        b.mul(n1 - n2, inv, out="r0")
        b.mul(-2 * fm, inv, out="r1")
    F.segment("H", seg_h)
    rin = {"r0'": "r0", "r1'": "r1"}

    def seg_lsq(b):
        b.bound["r0'"], b.bound["r1'"] = 9, 9
        b.mul(r0 + r1, r0 - r1, out="r0'")
        b.mul(2 * r0, r1, out="r1'")
    F.segment("LSQ", seg_lsq, rin)

    def seg_lmu(b):
        b.bound["r0'"], b.bound["r1'"] = 9, 9
        t1 = b.mul(r1, h1)
        t2 = b.mul(r0, h1)
        b.mul(r0, h0, E=-t1, out="r0'")
        b.mul(r1, h0, E=t2, out="r1'")
    F.segment("LMU", seg_lmu, rin)

    def seg_out(b):
        b.mul(r0, raw1, out="out0")
        b.mul(r1, raw1, out="out1")
    F.segment("OUT", seg_out)
    F.allocate_temps()
    # ---- launch 1, table form: the Miller loop over a key's normalised line table (fixedpair.hpp; makeL2 and the
    # level-1 decryption lift).  The evaluation point phi(C) sits in (ax, ay); a segment consumes the coefficients
    # (a_s/c_s, b_s/c_s) of one step (TD: f <- f^2 * l, six products in two rounds) or two (TDA: the addition after a
    # non-zero digit, ten in three); the kernel fetches the next segment's coefficients while this one runs and
    # stores them, in place, after its last round.
    T = Program(w, w, reads_first=True)
    OT = {k: T.fixed(k, 1) for k in ("ax", "ay")}
    tc = {k: T.fixed(k, 1) for k in ("ta1", "tb1", "ta2", "tb2")}
    fst = {k: T.fixed(k, STATE_BOUNDS[k]) for k in ("v0", "v1", "v2")}
    fnew = {k: S(k + "'") for k in fst}
    fin = {k + "'": k for k in fst}

    def seg_td(b):
        for k in fst:
            b.bound[k + "'"] = STATE_BOUNDS[k]
        F0, F1 = f_of(fst)
        cre, cim = tab_line(b, tc, 1, OT)
        g0 = b.mul(F0 + F1, F0 - F1)
        g1h = b.mul(F0, F1)
        finish_f(b, g0, 2 * g1h, cre, cim, fnew)
    T.segment("TD", seg_td, fin)

    def seg_tda(b):
        for k in fst:
            b.bound[k + "'"] = STATE_BOUNDS[k]
        F0, F1 = f_of(fst)
        cre, cim = tab_line(b, tc, 1, OT)
        cre2, cim2 = tab_line(b, tc, 2, OT)
        g0 = b.mul(F0 + F1, F0 - F1)
        g1h = b.mul(F0, F1)
        mid = {k: S("TDA.%s" % k) for k in fst}
        for k, f in mid.items():
            b.bound[name(f)] = STATE_BOUNDS[k]
        finish_f(b, g0, 2 * g1h, cre, cim, mid)
        M0, M1 = f_of(mid)
        finish_f(b, M0, M1, cre2, cim2, fnew)
    T.segment("TDA", seg_tda, fin)
    tn1, tn2, tfm = (T.fixed(k, 2) for k in ("n1", "n2", "fm"))

    def seg_tnorm(b):
        F0, F1 = f_of(fst)
        b.mul(F0, F0, out="n1")
        b.mul(F1, F1, out="n2")
        b.mul(F0, F1, out="fm")
    T.segment("NORM", seg_tnorm)
    T.allocate_temps()
    return M, F, T


# ---------------------------------------------------------------------------------------------------------------
# The lane-group kernel's G1 scalar multiplication (csrc/quad/quad_g1.hpp): MultConst with per-element scalars
# ---------------------------------------------------------------------------------------------------------------
# `res.PowBig(c.C, constant)` on level 1 (bgn.go:258) for mid-size batches: sixteen lanes per element, fixed signed
# 4-bit windows over a per-element table of 1*B .. 8*B.  Jacobian coordinates with Z^2 carried along (X, Y, Z, ZZ);
# the table entries are Jacobian too (X, Y, Z, ZZ, ZZZ = Z^3) — with four quads the full addition takes as many
# rounds as the mixed one, so no inversion is spent on making them affine.  Every lane group of a wave runs the same
# segment sequence (a window = four doublings, then one addition of the entry its own digit selects, its stores to
# the state suppressed where the digit is zero or the accumulator still the identity).  The exceptional cases of the
# formulas — a doubling of a point of order two, an addition of equal or opposite points — make Z zero and keep it
# zero to the end, where the kernel tests it once and hands such an element to the exact lane kernel.
G1_STATE_BOUNDS = {"X": 19, "Y": 19, "Z": 2, "ZZ": 2}
G1_ENTRY_BOUNDS = {"tx": 19, "ty": 19, "tz": 2, "tzz": 2, "tzzz": 2}


def build_quad_g1_programs(w=QUAD_W):
    G = Program(w, w, reads_first=True)
    st = {k: G.fixed(k, b) for k, b in G1_STATE_BOUNDS.items()}
    T = {k: G.fixed(k, b) for k, b in G1_ENTRY_BOUNDS.items()}
    zzz = G.fixed("zzz", 2)                                   # Z^3 of the state, for the table's entries
    inplace = {k + "'": k for k in G1_STATE_BOUNDS}

    def declare_new(b):
        for k, v in G1_STATE_BOUNDS.items():
            b.bound[k + "'"] = v

    def seg_gdbl(b):
        """Jacobian doubling, a = 1 (y^2 = x^3 + x): the point half of pairing.hpp miller_double, Z^4 made here."""
        declare_new(b)
        X, Y, Z, ZZ = st["X"], st["Y"], st["Z"], st["ZZ"]
        XX = b.mul(X, X)
        YY = b.mul(Y, Y)
        Z3 = b.mul(2 * Y, Z, out="Z'")
        Wq = b.mul(ZZ, ZZ)
        M = 3 * XX + Wq
        M2 = b.mul(M, M)
        XYY = b.mul(X, YY)
        Y4 = b.mul(YY, YY)
        b.mul(Z3, Z3, out="ZZ'")
        b.lin(M2 - 8 * XYY, out="X'")                                    # M^2 - 2S, S = 4 X YY
        b.mul(M, 12 * XYY - M2, E=-8 * Y4, out="Y'")                     # M (S - X3) - 8 YY^2
    G.segment("GDBL", seg_gdbl, inplace)

    def seg_gadd(b):
        """(X, Y, Z, ZZ) + (tx, ty, tz, tzz, tzzz), both Jacobian: U1 = X tzz, U2 = tx ZZ, S1 = Y tzzz, S2 = ty Z^3,
        H = U2 - U1, r = S2 - S1, Z3 = Z tz H, X3 = r^2 - H^3 - 2 U1 H^2, Y3 = r (U1 H^2 - X3) - S1 H^3."""
        declare_new(b)
        X, Y, Z, ZZ = st["X"], st["Y"], st["Z"], st["ZZ"]
        U1 = b.mul(X, T["tzz"])
        U2 = b.mul(T["tx"], ZZ)
        ZZZ = b.mul(Z, ZZ)
        S1 = b.mul(Y, T["tzzz"])
        S2 = b.mul(T["ty"], ZZZ)
        ZtZ = b.mul(Z, T["tz"])
        H = U2 - U1
        r = S2 - S1
        Z3 = b.mul(ZtZ, H, out="Z'")
        HH = b.mul(H, H)
        S1H = b.mul(S1, H)
        HHH = b.mul(H, HH)
        UHH = b.mul(U1, HH)
        rr2 = b.mul(r, r)
        SH3 = b.mul(S1H, HH)
        b.lin(rr2 - HHH - 2 * UHH, out="X'")
        b.mul(r, 3 * UHH + HHH - rr2, E=-SH3, out="Y'")
        b.mul(Z3, Z3, out="ZZ'")
    G.segment("GADD", seg_gadd, inplace)

    def seg_gzzz(b):
        b.mul(st["Z"], st["ZZ"], out="zzz")
    G.segment("GZZZ", seg_gzzz)

    def seg_gadm(b):
        """(X, Y, Z, ZZ) + (tx, ty) affine — the entries of a key's fixed-base window tables (k_g1_fixed_quad): GADD with
        tz = tzz = tzzz = 1 leaves U1 = X, S1 = Y, Z3 = Z H; twelve products and X' in four rounds instead of five."""
        declare_new(b)
        X, Y, Z, ZZ = st["X"], st["Y"], st["Z"], st["ZZ"]
        U2 = b.mul(T["tx"], ZZ)
        ZZZ = b.mul(Z, ZZ)
        S2 = b.mul(T["ty"], ZZZ)
        H = U2 - X
        r = S2 - Y
        Z3 = b.mul(Z, H, out="Z'")
        HH = b.mul(H, H)
        YH = b.mul(Y, H)
        HHH = b.mul(H, HH)
        XHH = b.mul(X, HH)
        rr2 = b.mul(r, r)
        YH3 = b.mul(YH, HH)
        b.lin(rr2 - HHH - 2 * XHH, out="X'")
        b.mul(r, 3 * XHH + HHH - rr2, E=-YH3, out="Y'")
        b.mul(Z3, Z3, out="ZZ'")
    G.segment("GADM", seg_gadm, inplace, tries=QUAD_TRIES)
    G.allocate_temps()
    # ---- the last launch: affine coordinates from the parked (X, Y) and R / Z of the inversion kernel, then the
    # division by R (plain residues, as the ladder kernel of ops.hpp writes them)
    A = Program(w, w, reads_first=True)
    X, Y = A.fixed("X", 19), A.fixed("Y", 19)
    zi = A.fixed("zi", 4)
    raw1 = A.fixed("raw1", 1)
    A.fixed("out0", 2)
    A.fixed("out1", 2)

    def seg_aff(b):
        zi2 = b.mul(zi, zi)
        zi3 = b.mul(zi2, zi)
        xm = b.mul(X, zi2)
        ym = b.mul(Y, zi3)
        b.mul(xm, raw1, out="out0")
        b.mul(ym, raw1, out="out1")
    A.segment("AFF", seg_aff)
    A.allocate_temps()
    return G, A


QUAD_G1_SLOTS = ("X", "Y", "Z", "ZZ", "tx", "ty", "tz", "tzz", "tzzz", "zzz")
QUAD_AFF_SLOTS = ("X", "Y", "zi", "raw1", "out0", "out1")


def emit_quad_g1(path, verbose=True):
    G, A = build_quad_g1_programs()
    emit(G, path, prefix="QUADG", round_headers=True, slot_names=QUAD_G1_SLOTS)
    emit(A, path, prefix="QUADA", round_headers=True, slot_names=QUAD_AFF_SLOTS, append=True)
    if verbose:
        for P in (G, A):
            print(summary(P))
            print("slots:", P.nslots, "->", path)
    return G, A


# ---------------------------------------------------------------------------------------------------------------
# Emission
# ---------------------------------------------------------------------------------------------------------------
def emit(P, path, prefix="COOP", w=None, round_headers=False, slot_names=None, append=False):
    """Writes the tables as a C include.  prefix "COOP": the wave-cooperative kernel (one micro-op per wave);
    "QUAD": the lane-group kernel (csrc/quad/: one micro-op per quad of lanes, w = 4), which also gets one header
    word per round with what is uniform over its micro-ops (term counts, whether every operand is plain)."""
    w = w or P.w
    cap = prefix.capitalize()
    seg_index = {}
    rows = []
    rnd = 0
    for name, rounds in P.segments:
        seg_index[name] = (rnd, len(rounds))
        for us in rounds:
            row = list(us) + [None] * (w - len(us))
            rows.append(row)
            rnd += 1
    lines = []
    what = "wave-cooperative pairing" if prefix == "COOP" else "lane-group kernels (csrc/quad/)"
    lines.append("// GENERATED by tools/coop/gen_prog.py — do not edit.  Micro-op tables of the %s." % what)
    lines.append("// %d segments, %d rounds of %d micro-ops, %d LDS value slots." % (len(P.segments), rnd, w, P.nslots))
    lines.append("#define %s_W %d" % (prefix, w))
    lines.append("#define %s_NSLOTS %d" % (prefix, P.nslots))
    lines.append("#define %s_MAX_TERMS %d" % (prefix, MAX_TERMS))
    for gname in slot_names or (
            "ax", "ay", "bx", "by", "one", "raw1", "zero", "out0", "out1", "X@0", "Y@0", "Z@0", "ZZ@0", "W@0", "v0@0", "v1@0", "v2@0",
            "n1", "n2", "fm", "acc@0", "acc@1", "h0", "h1", "r0@0", "r1@0", "r0@1", "r1@1",
            "ta1@0", "tb1@0", "ta2@0", "tb2@0", "ta1@1", "tb1@1", "ta2@1", "tb2@1", "v0@1", "v1@1", "v2@1"):
        lines.append("#define %s_SLOT_%s %d" % (prefix, gname.replace("@", "_").upper(), P.phys[gname]))
    lines.append("enum %sSeg {" % cap)
    for i, (name, _) in enumerate(P.segments):
        lines.append("  %s_SEG_%s = %d," % (prefix, name, i))
    lines.append("  %s_NSEG = %d" % (prefix, len(P.segments)))
    lines.append("};")
    lines.append("static __device__ const unsigned int k%sSegFirst[%s_NSEG] = {%s};" %
                 (cap, prefix, ", ".join(str(seg_index[n][0]) for n, _ in P.segments)))
    lines.append("static __device__ const unsigned int k%sSegRounds[%s_NSEG] = {%s};" %
                 (cap, prefix, ", ".join(str(seg_index[n][1]) for n, _ in P.segments)))
    lines.append("// eight dwords per micro-op (scalar loads have no byte form): kind | A plain<<8 | B plain<<9 | dst<<16 | "
                 "nb<<24, ne | KA<<8 | KB<<16 | KE<<24,")
    lines.append("// (\"plain\": the operand is one stored value with coefficient 1 and no multiple of p, used as it is)")
    lines.append("// then slot indices and signed coefficients of A, B, E, four bytes each; kind: 0 nop, 1 mul, 2 lin")
    lines.append("alignas(32) static __device__ const unsigned int k%sProg[%d][8] = {" % (cap, rnd * w))

    def terms(form):
        items = sorted(form.items(), key=lambda kv: P.phys[kv[0]]) if form else []
        idx = [P.phys[k] for k, _ in items] + [0] * (MAX_TERMS - len(items))
        cf = [c for _, c in items] + [0] * (MAX_TERMS - len(items))
        assert all(-128 <= c <= 127 for c in cf)
        return len(items), idx, cf

    plain = lambda f, K: int(f is not None and len(f) == 1 and list(f.values())[0] == 1 and K == 0)
    headers = []
    for r, row in enumerate(rows):
        na_max = nb_max = ne_max = 0
        all_plain_a = all_plain_b = 1
        any_mul = any_ka = any_kb = any_ke = 0
        for u in row:
            if u is None:
                lines.append("  {0, 0, 0, 0, 0, 0, 0, 0},")
                all_plain_a = all_plain_b = 0      # an idle quad multiplies zero by zero through the general path
                continue
            na, ia, ca = terms(u.A)
            nb, ib, cb = terms(u.B)
            ne, ie, ce = terms(u.E)
            assert max(u.KA, u.KB, u.KE) <= 255
            pk = lambda v: sum((x & 0xFF) << (8 * i) for i, x in enumerate(v))
            if u.kind == "mul":
                any_mul = 1
                na_max, nb_max = max(na_max, na), max(nb_max, nb)
                all_plain_a &= plain(u.A, u.KA)
                all_plain_b &= plain(u.B, u.KB)
            else:
                all_plain_a = all_plain_b = 0
            ne_max = max(ne_max, ne)
            any_ka |= int(u.KA != 0)
            any_kb |= int(u.KB != 0)
            any_ke |= int(u.KE != 0)
            w0 = (1 if u.kind == "mul" else 2) | plain(u.A, u.KA) << 8 | plain(u.B, u.KB) << 9 | P.phys[u.dst] << 16 | nb << 24
            words = [w0, pk([ne, u.KA, u.KB, u.KE]),
                     pk(ia), pk(ca), pk(ib), pk(cb), pk(ie), pk(ce)]
            lines.append("  {%s},   // r%d %s" % (", ".join("0x%08xu" % x for x in words), r, u.dst))
        headers.append(na_max | nb_max << 4 | ne_max << 8 | all_plain_a << 12 | all_plain_b << 13 | any_mul << 14 |
                       any_ka << 15 | any_kb << 16 | any_ke << 17)
    lines.append("};")
    if round_headers:
        lines.append("// one word per round: max terms of A | of B << 4 | of E << 8 | every A plain << 12 | every B plain << 13 | "
                     "any product << 14 |")
        lines.append("// a multiple of p in any A << 15 | in any B << 16 | in any E << 17")
        lines.append("static __device__ const unsigned int k%sRound[%d] = {%s};" % (cap, rnd, ", ".join("0x%05xu" % h for h in headers)))
    with open(path, "a" if append else "w") as f:
        f.write("\n".join(lines) + "\n")
    return seg_index


QUAD_MILLER_SLOTS = ("ax", "ay", "bx", "by", "X", "Y", "Z", "ZZ", "W", "v0", "v1", "v2", "n1", "n2", "fm", "one", "pz")
QUAD_FINAL_SLOTS = ("n1", "n2", "fm", "inv", "raw1", "h0", "h1", "r0", "r1", "out0", "out1")
QUAD_TABLE_SLOTS = ("ax", "ay", "ta1", "tb1", "ta2", "tb2", "v0", "v1", "v2", "n1", "n2", "fm")


def emit_quad(path, verbose=True):
    """Both programs of the lane-group kernel into one include: QUADM (Miller loop + norms), QUADF (the rest of the
    final exponentiation)."""
    M, F, T = build_quad_programs()
    emit(M, path, prefix="QUADM", round_headers=True, slot_names=QUAD_MILLER_SLOTS)
    emit(F, path, prefix="QUADF", round_headers=True, slot_names=QUAD_FINAL_SLOTS, append=True)
    emit(T, path, prefix="QUADT", round_headers=True, slot_names=QUAD_TABLE_SLOTS, append=True)
    if verbose:
        for P in (M, F, T):
            print(summary(P))
            print("slots:", P.nslots, "->", path)
    return M, F, T


def summary(P):
    out = []
    for name, rounds in P.segments:
        nm = sum(1 for us in rounds for u in us if u.kind == "mul")
        out.append("%-6s %2d rounds, %2d products" % (name, len(rounds), nm))
    return "\n".join(out)


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    P = build_program()
    path = os.path.join(root, "bgn_amd", "csrc", "coop", "coop_prog.inc")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    emit(P, path)
    print(summary(P))
    print("slots:", P.nslots, "->", path)
    # the lane-group kernel (csrc/quad/): the same step programs scheduled for four quads of lanes per pairing
    qpath = os.path.join(root, "bgn_amd", "csrc", "quad", "quad_prog.inc")
    os.makedirs(os.path.dirname(qpath), exist_ok=True)
    emit_quad(qpath)
    emit_quad_g1(os.path.join(root, "bgn_amd", "csrc", "quad", "quad_g1_prog.inc"))
