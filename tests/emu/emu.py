"""Python driver of the host emulation harness (tests/emu/emu.cpp).

TEST INFRASTRUCTURE: runs the device lane programs of bgn_amd/csrc on the CPU
so CPU-only tests can check kernel logic against the oracle."""
from __future__ import annotations

import ctypes as C
import os
import struct
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libbgn_emu.so")
_CSRC = os.path.join(_HERE, "..", "..", "bgn_amd", "csrc")
LIMB = 29                      # bgn_amd/csrc/consts.hpp LIMB_BITS
KP_MAX, MAX_NAF, MAX_EXP_LIMBS, MASK = 32, 2112, 80, (1 << LIMB) - 1


def build() -> str:
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    srcs = [os.path.join(_HERE, "emu.cpp")] + [os.path.join(_CSRC, f) for f in
                                                ("fpmont.hpp", "pairing.hpp", "ops.hpp", "codec.hpp", "consts.hpp", "bsgs.hpp", "fixedpair.hpp", "polyops.hpp", "kernels.hpp", "fpinv.hpp", "imad.hpp", "barrett.hpp")]
    if not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["g++", "-O1", "-g", "-rdynamic", "-std=c++17", "-fPIC", "-shared", "-I" + _HERE, "-I" + _CSRC,
                               "-include", os.path.join(_HERE, "agpr.hpp"), "-include", os.path.join(_HERE, "gmem.hpp"), "-include", os.path.join(_HERE, "imad.hpp"),
                               os.path.join(_HERE, "emu.cpp"),
                               "-o", _SO])
    return _SO


def limbs(v: int, nl: int):
    return [(v >> (LIMB * j)) & MASK for j in range(nl)]


def nl_for(p: int) -> int:
    need = (p.bit_length() + 9 + LIMB - 1) // LIMB
    for nl in (3, 10, 19, 36, 37, 72):
        if nl >= need:
            return nl
    raise ValueError("field too large")


def wnaf(n: int, w: int = 3):
    """Width-w NAF, little-endian digits (hostbig.hpp BigU::wnaf)."""
    d, full, half = [], 1 << w, 1 << (w - 1)
    while n:
        if n & 1:
            z = n & (full - 1)
            if z >= half:
                z -= full
            d.append(z)
            n -= z
        else:
            d.append(0)
        n >>= 1
    return d


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


class Emu:
    def __init__(self, p: int, n: int, l: int):
        self.lib = C.CDLL(build())
        self.p, self.n, self.l = p, n, l
        self.nl = nl = nl_for(p)
        self.L = (p.bit_length() + 7) // 8
        R = 1 << (LIMB * nl)
        img = limbs(p, nl) + limbs(R % p, nl) + limbs(R * R % p, nl)
        for K in range(1, KP_MAX + 1):
            img += limbs(K * p, nl)
        img += [(-pow(p, -1, 1 << LIMB)) % (1 << LIMB), 0, 0, 0]
        self.params = (C.c_uint32 * len(img))(*img)
        d = naf(n)
        pm2 = p - 2
        buf = struct.pack("<iiQii", len(d), pm2.bit_length(), l, l.bit_length(), 0)
        buf += bytes((x & 0xFF) for x in d) + bytes(MAX_NAF - len(d))
        buf += struct.pack("<%dI" % MAX_EXP_LIMBS, *limbs(pm2, MAX_EXP_LIMBS))
        self._base = buf
        self.consts = None
        self.set_window(5)

    def set_window(self, w: int):
        """Install the width-w NAF of n for the windowed Miller loop (pairing.hpp miller_loop_w)."""
        wd = wnaf(self.n, w)
        buf = self._base + struct.pack("<ii", len(wd), w) + bytes((x & 0xFF) for x in wd) + bytes(MAX_NAF - len(wd))
        assert len(buf) == self.lib.emu_consts_size() or True
        self.lib.emu_consts_size.restype = C.c_size_t
        assert len(buf) == self.lib.emu_consts_size(), (len(buf), self.lib.emu_consts_size())
        self.consts = C.create_string_buffer(buf, len(buf))

    @classmethod
    def from_fixture(cls, fx):
        return cls(int(fx["p"], 16), int(fx["n"], 16), fx["l"])

    def fp_products(self, a: int, b: int, c: int, d: int):
        """(a*b/R, a*a/R, (a*b + c*d)/R) by fp_mul, fp_sqr, fp_mul2 on the raw values a..d < 2^(LIMB*nl) (tight limbs,
        not necessarily below p); results as integers (lazy: below 2p when the operands' bounds allow)."""
        nl = self.nl
        arr = lambda v: (C.c_uint32 * nl)(*limbs(v, nl))
        out = (C.c_uint32 * (3 * nl))()
        assert self.lib.emu_fp_products(nl, self.params, arr(a), arr(b), arr(c), arr(d), out) == 0
        val = lambda k: sum(int(out[k * nl + j]) << (LIMB * j) for j in range(nl))
        return val(0), val(1), val(2)

    # wire <-> internal
    def decode(self, wire: bytes):
        out = (C.c_uint32 * (2 * self.nl))()
        inf = C.c_uint8()
        assert self.lib.emu_decode(self.nl, self.params, wire, self.L, out, C.byref(inf)) == 0
        return out, inf.value

    def encode(self, plain, inf=0) -> bytes:
        w = C.create_string_buffer(2 * self.L)
        assert self.lib.emu_encode(self.nl, plain, self.L, inf, w) == 0
        return w.raw

    def fp_inv(self, a_mont: int) -> int:
        """Montgomery-form inverse of a Montgomery-form residue (may be lazy, < 4p)."""
        A = (C.c_uint32 * self.nl)(*limbs(a_mont, self.nl))
        out = (C.c_uint32 * self.nl)()
        assert self.lib.emu_fp_inv(self.nl, self.params, self.p.bit_length(), A, out) == 0
        return sum(int(v) << (LIMB * j) for j, v in enumerate(out))

    def pairing(self, a: bytes, b: bytes) -> bytes:
        A, ia = self.decode(a)
        B, ib = self.decode(b)
        out = (C.c_uint32 * (2 * self.nl))()
        assert self.lib.emu_pairing(self.nl, self.params, self.consts, A, B, out) == 0
        if ia or ib:
            return (1).to_bytes(self.L, "big") + bytes(self.L)
        return self.encode(out)

    def pairing_w3(self, a: bytes, b: bytes) -> bytes:
        """The windowed Miller loop (pairing.hpp miller_loop_w, width set by set_window) + final exponentiation."""
        A, _ = self.decode(a)
        B, _ = self.decode(b)
        out = (C.c_uint32 * (2 * self.nl))()
        assert self.lib.emu_pairing_w3(self.nl, self.params, self.consts, A, B, out) == 0
        return self.encode(out)

    def fixed_table(self, P_wire: bytes, ts: int = 1, te: int = 0, tab=None):
        """Line table of e(P, .) in column `te` of a table with limb stride `ts` (fixedpair.hpp); several
        points can share one table (MultPoly's per-coefficient tables) by passing `tab` back in."""
        d = naf(self.n)
        steps = (len(d) - 1) + sum(1 for i in range(1, len(d) - 1) if d[i])
        Pm, _ = self.decode(P_wire)
        if tab is None:
            tab = (C.c_uint32 * (3 * self.nl * steps * ts))()
        assert self.lib.emu_fixed_build(self.nl, self.params, self.consts, Pm, tab, C.c_size_t(ts), C.c_size_t(te)) == 0
        return tab

    def fixed_normalize(self, tab):
        """Divide every line of a key table (ts = 1) by its c, in place (fixedpair.hpp fixed_normalize_lane)."""
        d = naf(self.n)
        steps = (len(d) - 1) + sum(1 for i in range(1, len(d) - 1) if d[i])
        assert self.lib.emu_fixed_normalize(self.nl, self.params, self.consts, tab, C.c_size_t(steps)) == 0
        return tab

    def pairing_fixed(self, tab, c_wire: bytes, ts: int = 1, te: int = 0, normalized: bool = False) -> bytes:
        Cm, inf = self.decode(c_wire)
        if inf:
            return (1).to_bytes(self.L, "big") + bytes(self.L)
        out = (C.c_uint32 * (2 * self.nl))()
        assert self.lib.emu_pairing_fixed(self.nl, self.params, self.consts, tab, C.c_size_t(ts), C.c_size_t(te),
                                          1 if normalized else 0, Cm, out) == 0
        return self.encode(out)

    def pairing_fixed_multi(self, tab, ts: int, Qp: int, q: int, t_wires, v_wires, s: int) -> bytes:
        """Output coefficient s of the product of the polynomials T (d points, their line tables in columns i*Qp + q of
        `tab`) and V (d points): prod_{i+j=s} e(T_i, V_j) by ONE Miller loop (fixedpair.hpp miller_loop_fixed_multi).
        Identity points (None in the lists) contribute the factor 1."""
        d = len(t_wires)
        assert len(v_wires) == d and 0 <= s <= 2 * d - 2
        sv = d * Qp
        v = (C.c_uint32 * (2 * self.nl * sv))()
        vinf = (C.c_uint8 * sv)()
        tinf = (C.c_uint8 * sv)()
        some = next(w for w in v_wires if w is not None)
        for j, w in enumerate(v_wires):
            m, _ = self.decode(w if w is not None else some)
            for l in range(self.nl):
                v[l * sv + j * Qp + q] = m[l]
                v[(self.nl + l) * sv + j * Qp + q] = m[self.nl + l]
            vinf[j * Qp + q] = 1 if w is None else 0
        for i, w in enumerate(t_wires):
            tinf[i * Qp + q] = 1 if w is None else 0
        i0 = max(0, s - (d - 1))
        i1 = min(s, d - 1)
        out = (C.c_uint32 * (2 * self.nl))()
        assert self.lib.emu_pairing_fixed_multi(self.nl, self.params, self.consts, tab, C.c_size_t(ts), v, C.c_size_t(sv), vinf, tinf,
                                                C.c_size_t(q), C.c_size_t(Qp), C.c_size_t(i0), C.c_size_t(s - i0), i1 - i0 + 1, out) == 0
        return self.encode(out)

    def g1_mul(self, base: bytes, k: int, klen: int = None, window: int = 0) -> bytes:
        B, ib = self.decode(base)
        klen = klen or max(1, (k.bit_length() + 7) // 8)
        kb = k.to_bytes(klen, "big")
        out = (C.c_uint32 * (2 * self.nl))()
        oinf = C.c_uint8()
        assert self.lib.emu_g1_mul(self.nl, self.params, self.consts, B, ib, kb, C.c_size_t(klen), int(window), out,
                                   C.byref(oinf)) == 0
        return self.encode(out, oinf.value)

    def decode_plain(self, wire: bytes):
        """wire -> plain limbs (k_decode with plain = 1): no Montgomery conversion."""
        x, y = int.from_bytes(wire[:self.L], "big"), int.from_bytes(wire[self.L:], "big")
        return limbs(x, self.nl) + limbs(y, self.nl), int(x == 0 and y == 0)

    def g1_add(self, a_list, b_list, subtract=False, plain=False):
        """One lane processing the whole list as a single batched-inversion run; plain = the representation
        EAdd / ESub use (plain residues in and out, ops.hpp g1_add_run<PLAIN>)."""
        n = len(a_list)
        dec = self.decode_plain if plain else self.decode
        A = (C.c_uint32 * (2 * self.nl * n))()
        B = (C.c_uint32 * (2 * self.nl * n))()
        ai = (C.c_uint8 * n)()
        bi = (C.c_uint8 * n)()
        for j in range(n):
            x, i = dec(a_list[j])
            A[2 * self.nl * j:2 * self.nl * (j + 1)] = list(x)
            ai[j] = i
            x, i = dec(b_list[j])
            B[2 * self.nl * j:2 * self.nl * (j + 1)] = list(x)
            bi[j] = i
        out = (C.c_uint32 * (2 * self.nl * n))()
        oi = (C.c_uint8 * n)()
        assert self.lib.emu_g1_add(self.nl, self.params, self.consts, A, ai, B, bi, n, 1 if subtract else 0,
                                   1 if plain else 0, out, oi) == 0
        res = []
        for j in range(n):
            pl = (C.c_uint32 * (2 * self.nl))(*out[2 * self.nl * j:2 * self.nl * (j + 1)])
            res.append(self.encode(pl, oi[j]))
        return res

    def make_table(self, entries_wire):
        """entries_wire[(w << wbits) + d] = wire bytes of d*2^(wbits*w)*B (None: identity) -> device table layout."""
        tab = (C.c_uint32 * (2 * self.nl * len(entries_wire)))()
        for i, w in enumerate(entries_wire):
            if w is None:
                continue
            x, _ = self.decode(w)
            tab[2 * self.nl * i:2 * self.nl * (i + 1)] = list(x)
        return tab

    def build_table(self, wbits: int, windows: int, pow_wire, sbits: int = 0):
        """pow_wire[i] = wire bytes of 2^i * B, i < windows*sbits -> full window table, built the way the engine does
        (sbits = wbits + 1: signed windows, the top power of a window at index 0)."""
        sbits = sbits or wbits
        assert len(pow_wire) == windows * sbits
        pw = self.make_table(pow_wire)
        tab = (C.c_uint32 * (2 * self.nl * (windows << wbits)))()
        assert self.lib.emu_tab_build(self.nl, self.params, self.consts, wbits, sbits, windows, pw, tab) == 0
        return tab

    TALLY_KINDS = ("mad", "row", "flush", "pass", "final", "reduce", "select", "cmp", "agpr", "lds", "gmem")

    def tally_reset(self):
        self.lib.emu_tally_reset()

    def tally(self) -> dict:
        """Dynamic counts of the field primitives executed by this thread since tally_reset (fpmont.hpp BGN_TALLY)."""
        out = (C.c_ulonglong * 16)()
        self.lib.emu_tally_read(out)
        return {k: int(out[i]) for i, k in enumerate(self.TALLY_KINDS)}

    def window_digit(self, k: int, klen: int, wbits: int, sbits: int, window: int):
        """(digit, table index) of one window of the signed recoding (ops.hpp scalar_window_digit)."""
        idx = C.c_uint()
        self.lib.emu_window_digit.restype = C.c_longlong
        d = self.lib.emu_window_digit(k.to_bytes(klen, "big"), C.c_size_t(klen), wbits, sbits, window, C.byref(idx))
        return int(d), int(idx.value)

    def scalar_windows(self, klen: int, wbits: int, sbits: int) -> int:
        return int(self.lib.emu_scalar_windows(C.c_size_t(klen), wbits, sbits))

    def g1_fixed(self, tabP, tabQ, wbits: int, x: int, xlen: int, r=None, rlen: int = 0, sbits_q: int = 0) -> bytes:
        out = (C.c_uint32 * (2 * self.nl))()
        oinf = C.c_uint8()
        xb = x.to_bytes(xlen, "big")
        rb = r.to_bytes(rlen, "big") if r is not None else None
        assert self.lib.emu_g1_fixed(self.nl, self.params, self.consts, tabP, tabQ, wbits, sbits_q or wbits, xb, C.c_size_t(xlen), rb,
                                     C.c_size_t(rlen), out, C.byref(oinf)) == 0
        return self.encode(out, oinf.value)

    def g1_fixed_chains(self, tabP, tabQ, wbits_p: int, wbits_q: int, sbits_q: int, xs, xlen: int, rs, rlen: int):
        """P^x * Q^r for a list of elements on the chain kernels (four accumulation chains, then the chain sums):
        the launch sequence of engine.cpp fixed_base_product above the lane groups' range."""
        n = len(xs)
        xb = b"".join(int(v).to_bytes(xlen, "big") for v in xs)
        rb = b"".join(int(v).to_bytes(rlen, "big") for v in rs)
        out = (C.c_uint32 * (2 * self.nl * n))()
        oinf = (C.c_uint8 * n)()
        assert self.lib.emu_g1_fixed_chains(self.nl, self.params, self.consts, tabP, tabQ, wbits_p, wbits_q, sbits_q, xb,
                                            C.c_size_t(xlen), rb, C.c_size_t(rlen), C.c_size_t(n), out, oinf) == 0
        res = []
        for e in range(n):
            one = (C.c_uint32 * (2 * self.nl))(*out[2 * self.nl * e:2 * self.nl * (e + 1)])
            res.append(self.encode(one, oinf[e]))
        return res

    def gt_mul(self, a: bytes, b: bytes, conj_b=False, plain_a=False) -> bytes:
        if plain_a:
            pl, _ = self.decode_plain(a)
            A = (C.c_uint32 * (2 * self.nl))(*pl)
        else:
            A, _ = self.decode(a)
        B, _ = self.decode(b)
        out = (C.c_uint32 * (2 * self.nl))()
        assert self.lib.emu_gt_mul(self.nl, self.params, A, B, 1 if conj_b else 0, 1 if plain_a else 0, out) == 0
        return self.encode(out)

    def barrett_mu(self):
        """BarrettParams<NL> of barrett.hpp: mu = floor(B^(2 NL) / p), NL + 2 limbs."""
        return (C.c_uint32 * (self.nl + 2))(*limbs((1 << (2 * LIMB * self.nl)) // self.p, self.nl + 2))

    def gt_mul_plain(self, a: bytes, b: bytes, conj_b=False) -> bytes:
        """barrett.hpp fp2_mul_plain on the plain residues of two wire elements (the fused level-2 Add / Sub)."""
        A, _ = self.decode_plain(a)
        B, _ = self.decode_plain(b)
        A = (C.c_uint32 * (2 * self.nl))(*A)
        B = (C.c_uint32 * (2 * self.nl))(*B)
        out = (C.c_uint32 * (2 * self.nl))()
        assert self.lib.emu_fp2_mul_plain(self.nl, self.params, self.barrett_mu(), A, B, 1 if conj_b else 0, out) == 0
        return self.encode(out)

    def barrett(self, t: int) -> int:
        nl = self.nl
        T = (C.c_uint32 * (2 * nl))(*limbs(t, 2 * nl))
        out = (C.c_uint32 * nl)()
        assert self.lib.emu_barrett(nl, self.params, self.barrett_mu(), T, out) == 0
        return sum(int(out[j]) << (LIMB * j) for j in range(nl))

    def gt_table(self, g_wire: bytes, wbits: int, windows: int):
        g, _ = self.decode(g_wire)
        tab = (C.c_uint32 * (2 * self.nl * (windows << wbits)))()
        assert self.lib.emu_gt_tab_build(self.nl, self.params, wbits, windows, g, tab) == 0
        return tab

    def gt_fixed(self, tab, wbits: int, k: int, klen: int, r_wire: bytes = None) -> bytes:
        """g^k from the window table (times the GT element r_wire when given)."""
        out = (C.c_uint32 * (2 * self.nl))()
        R = None
        if r_wire is not None:
            pl = [int.from_bytes(r_wire[:self.L], "big"), int.from_bytes(r_wire[self.L:], "big")]
            R = (C.c_uint32 * (2 * self.nl))(*(limbs(pl[0], self.nl) + limbs(pl[1], self.nl)))
        assert self.lib.emu_gt_fixed(self.nl, self.params, tab, wbits, k.to_bytes(klen, "big"), C.c_size_t(klen), R, out) == 0
        return self.encode(out)

    def bsgs(self, g_wire: bytes, msg_space: int, xs_wire, S=None):
        """Build the table for generator g and search every x (GT wire bytes).  Returns (m list, status list)."""
        import math
        B = int(math.ceil(math.sqrt(float(msg_space))))
        Mmax = B * B + B + 2
        if S is None:
            S = 2
            while S < Mmax + 1 and S < (1 << 26):
                S <<= 1
        G = (Mmax + S) // (2 * S) + 1
        g, _ = self.decode(g_wire)
        gS = self.gt_pow(g_wire, 2 * S, 8)            # giant steps are 2S apart (bsgs.hpp)
        L = self.L
        conj = gS[:L] + ((self.p - int.from_bytes(gS[L:], "big")) % self.p).to_bytes(L, "big")
        gi, _ = self.decode(conj)
        n = len(xs_wire)
        xs = (C.c_uint32 * (2 * self.nl * n))()
        for j, w in enumerate(xs_wire):
            x, _ = self.decode(w)
            xs[2 * self.nl * j:2 * self.nl * (j + 1)] = list(x)
        m = (C.c_longlong * n)()
        st = (C.c_uint8 * n)()
        assert self.lib.emu_bsgs(self.nl, self.params, g, gi, C.c_ulonglong(S), C.c_ulonglong(G), C.c_ulonglong(Mmax),
                                 xs, n, m, st) == 0
        return list(m), list(st)

    def poly_lin(self, level: int, c_wire, scalars, dp: int):
        """polyops.hpp on one polynomial: dp > 0 convolution with the dp scalars (d + dp outputs), dp == 0 dot
        product with d scalars (one output).  Returns wire bytes per output."""
        d = len(c_wire)
        cm = (C.c_uint32 * (2 * self.nl * d))()
        cinf = (C.c_uint8 * d)()
        for j, w in enumerate(c_wire):
            x, inf = self.decode(w)
            cm[2 * self.nl * j:2 * self.nl * (j + 1)] = list(x)
            cinf[j] = inf if level == 1 else 0
        klen = max(1, max((int(v).bit_length() + 7) // 8 for v in scalars))
        kb = b"".join(int(v).to_bytes(klen, "big") for v in scalars)
        nout = d + dp if dp else 1
        out = (C.c_uint32 * (2 * self.nl * nout))()
        oinf = (C.c_uint8 * nout)()
        assert self.lib.emu_poly_lin(self.nl, self.params, self.consts, level, cm, cinf, d, dp, kb, C.c_size_t(klen),
                                     out, oinf) == 0
        res = []
        for s in range(nout):
            pl = (C.c_uint32 * (2 * self.nl))(*out[2 * self.nl * s:2 * self.nl * (s + 1)])
            res.append(self.encode(pl, oinf[s]) if level == 1 else self.encode(pl))
        return res

    def poly_acc(self, E_wire, d1: int, d2: int):
        n = d1 * d2
        E = (C.c_uint32 * (2 * self.nl * n))()
        for j, w in enumerate(E_wire):
            x, _ = self.decode(w)
            E[2 * self.nl * j:2 * self.nl * (j + 1)] = list(x)
        out = (C.c_uint32 * (2 * self.nl * (d1 + d2)))()
        assert self.lib.emu_poly_acc(self.nl, self.params, E, d1, d2, out) == 0
        res = []
        for s in range(d1 + d2):
            pl = (C.c_uint32 * (2 * self.nl))(*out[2 * self.nl * s:2 * self.nl * (s + 1)])
            res.append(self.encode(pl))
        return res

    def gt_pow_norm1(self, a: bytes, k: int, klen: int = None) -> bytes:
        """a^k for a of norm 1 by the Lucas-type ladder (ops.hpp gt_pow_norm1_lane)."""
        A, _ = self.decode(a)
        klen = klen or max(1, (k.bit_length() + 7) // 8)
        out = (C.c_uint32 * (2 * self.nl))()
        assert self.lib.emu_gt_pow_norm1(self.nl, self.params, self.p.bit_length() + 1, A, k.to_bytes(klen, "big"),
                                         C.c_size_t(klen), out) == 0
        return self.encode(out)

    def gt_pow(self, a: bytes, k: int, klen: int = None) -> bytes:
        A, _ = self.decode(a)
        klen = klen or max(1, (k.bit_length() + 7) // 8)
        out = (C.c_uint32 * (2 * self.nl))()
        assert self.lib.emu_gt_pow(self.nl, self.params, A, k.to_bytes(klen, "big"), C.c_size_t(klen), out) == 0
        return self.encode(out)
