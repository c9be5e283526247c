"""Synthetic workloads for bench.py and the scale tests."""
from __future__ import annotations

import numpy as np


def _r_shape(n: int):
    """Byte length of the blinding exponents and the mask of their top byte: r uniform below 2^(bits(n) - 2) < n
    (keygen sets the top two bits of both prime factors, bgn.go:153,160)."""
    nbits = n.bit_length()
    r_len = (nbits + 7) // 8
    top_bits = max(1, (nbits - 2) - 8 * (r_len - 1))
    return r_len, (1 << min(top_bits, 8)) - 1


def config2_ciphertexts(pk, count: int, seed: int, device, digits: bool = False):
    """SURVEY.md 8(d) Config 2 on the GPU: `count` level-1 ciphertexts P^m * Q^r of seeded random messages
    (40-bit m; with digits=True base-3 digits in {-1, 0, 1} as EncryptPoly makes them, poly.go:11-29: a negative
    digit is Sub(zero, Enc(|c|))) and full-length random r.  Returns (xs, rs, cts): the big-endian scalar arrays
    and the wire bytes, all uint8 CUDA tensors."""
    import torch
    eng = pk.engine
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    r_len, top_mask = _r_shape(pk.N)
    if digits:
        d = torch.randint(-1, 2, (count,), generator=g)
        xs = d.abs().to(torch.uint8).reshape(count, 1)
    else:
        xs = torch.randint(0, 256, (count, 5), dtype=torch.uint8, generator=g)          # 40-bit plaintexts
    rs = torch.randint(0, 256, (count, r_len), dtype=torch.uint8, generator=g)
    rs[:, 0] &= top_mask
    xs, rs = xs.to(device), rs.to(device)
    cts = torch.empty(count * eng.elem_bytes, dtype=torch.uint8, device=device)
    eng.encrypt_dev(xs, xs.shape[1], rs, r_len, cts, count)
    if digits:
        neg = torch.empty_like(cts)
        eng._lib.bgn_neg_batch_dev(eng._h, count, 1, cts.data_ptr(), neg.data_ptr(), eng._stream())
        sel = (d < 0).to(device)
        v = cts.view(count, eng.elem_bytes)
        v[sel] = neg.view(count, eng.elem_bytes)[sel]
    torch.cuda.synchronize()
    return xs, rs, cts


def permuted_copy(cts, elem_bytes: int, seed: int):
    """Config 3's second operand: a fixed (seeded) permutation of the first."""
    import torch
    count = cts.numel() // elem_bytes
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    perm = torch.randperm(count, generator=g).to(cts.device)
    return cts.view(count, elem_bytes)[perm].reshape(-1).contiguous()


def decrypt_mix(pk, fx, cts, xs, device, neg_every: int = 16, oor_every: int = 4096):
    """SURVEY.md 8(d) Config 4's inputs from Config 2's ciphertexts: every 16th negated (the retry of
    bgn.go:235-242 finds it), every 4096th replaced by the encryption of a message beyond B*B + B + 2
    (gsbs.go:105: no discrete log in either direction).  Returns (wire bytes, expected m, expected status)."""
    import numpy as np
    import torch
    eng = pk.engine
    EB = eng.elem_bytes
    n = xs.shape[0]
    neg = torch.empty_like(cts)
    eng._lib.bgn_neg_batch_dev(eng._h, n, 1, cts.data_ptr(), neg.data_ptr(), eng._stream())
    mixed = cts.view(n, EB).clone()
    mixed[::neg_every] = neg.view(n, EB)[::neg_every]
    del neg
    want = torch.zeros(n, dtype=torch.int64)
    xb = xs.cpu().numpy().astype(np.int64)
    for j in range(xb.shape[1]):
        want = want * 256 + torch.from_numpy(xb[:, j])
    want[::neg_every] = -want[::neg_every]
    status = torch.zeros(n, dtype=torch.uint8)
    idx = torch.arange(7, n, oor_every)
    if len(idx):
        T = int(fx["msg_space"])
        big = [2 * T + 12345 + 3 * int(i) for i in idx]                      # > B*B + B + 2 for every T >= 4
        xl = (max(big).bit_length() + 7) // 8
        xo = torch.from_numpy(np.frombuffer(b"".join(v.to_bytes(xl, "big") for v in big), dtype=np.uint8).copy())
        xo = xo.reshape(len(big), xl).to(device)
        co = torch.empty(len(big) * EB, dtype=torch.uint8, device=device)
        eng.encrypt_dev(xo, xl, None, 0, co, len(big))
        mixed[idx.to(device)] = co.view(len(big), EB)
        want[idx] = 0
        status[idx] = 1
    torch.cuda.synchronize()
    return mixed.reshape(-1).contiguous(), want, status


def _signed_digits(n: int, w: int):
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


def miller_schedule(n: int, window: int = 5):
    """(#doubling steps, #addition steps, #extra F_p^2 products by f_d, #set-up products) of the Miller loop in
    pairing.hpp: width-5 NAF by default (miller_loop_w), width 3 / 4, or the plain NAF with window = 2."""
    d = _signed_digits(n, window)
    dbl = len(d) - 1
    add = sum(1 for i, x in enumerate(d[:-1]) if x and i != 0)
    big = sum(1 for x in d[:-1] if abs(x) > 1)
    if window >= 3:
        npts = (1 << (window - 2)) - 1                       # odd multiples 3A .. (2^(w-1) - 1) A
        pre = STEP_DBL[0] + npts * STEP_ADD[0] + (npts - 1) * STEP_MULF[0]   # doubling step, addition steps, products by f_2
        pre += INVERSION_PRODUCTS + 7 * npts + 4 * npts      # shared inversion, peel + coordinates, canonical f_d
        if window >= 4:
            pre += INVERSION_PRODUCTS + 4 + 4                # 2A made affine, canonical f_2
    else:
        pre = 0
    return dbl, add, big, pre


# The Miller steps of pairing.hpp since round 5, as (reductions, of which squarings, of which sums of two products
# with one reduction): a reduction is what every product, squaring or sum ends with (NL^2 multiply-adds), a sum of
# two products carries a second multiplication (NL^2 more) under the same reduction.
STEP_DBL = (15, 3, 4)       # ZZ, YY, M^2 squared; M, Y3, F0, F1 sums; Y*Z, Z3*ZZ, cim, ZZ*xB, M*t, Sn, g0, F0*F1 plain
STEP_ADD = (14, 3, 4)       # ZZ, HH, rr^2 squared; Y3, cre, F0, F1 sums
STEP_MULF = (2, 0, 2)       # f * f_d: two sums
TABLE_DBL = (5, 0, 2)       # walk over a normalised line table: a*xC, g0, F0*F1 plain; F0, F1 sums
TABLE_ADD = (3, 0, 2)
TABLE_DBL_C = (6, 0, 2)     # ... over a table that keeps its c (MultPoly's per-coefficient tables): + c*yC
TABLE_ADD_C = (4, 0, 2)
BUILD_DBL = (10, 3, 2)      # fixed_build_double: ZZ, YY, M^2 squared; M, Y3 sums; YZ, a, b, c, Sn plain
BUILD_ADD = (11, 3, 2)      # fixed_build_add: ZZ, HH, rr^2 squared; b, Y3 sums


# One F_p inversion by division steps (fpinv.hpp) costs about as much as 55 field products (measured).
INVERSION_PRODUCTS = 55


LIMB_BITS = 29          # bgn_amd/csrc/consts.hpp


def limbs_for(p: int) -> int:
    """Limb count the engine instantiates for the field of p (engine.cpp pick_table)."""
    need = (p.bit_length() + 9 + LIMB_BITS - 1) // LIMB_BITS
    return next(x for x in (3, 10, 19, 36, 37, 72) if x >= need)


def flush_instructions(nl: int, plain_product: bool = False) -> int:
    """The mid-product carry passes at radix 2^29 (fpmont.hpp fp_flush: add, mask, shift per accumulator): a product
    runs in NI = ceil(NL / kRowsPerFlush) intervals of at most 19 rows (21 for a sum of two products) with a flush
    between them — none up to 19 limbs, one at 36 / 37, three at 72; counted with the multiply-adds of a product (same
    issue cost).  A plain product with ONE flush (36 / 37 limbs) carries out only the accumulators that live longer
    than 31 rows (fpmont.hpp MidFlush, BGN_TRIM_PARTIAL: nine at 36 limbs, eleven at 37)."""
    ni = -(-nl // 19)
    if ni <= 1:
        return 0
    if plain_product and ni == 2:
        done = 2 * (-(-(nl // 2) // 2))
        lo, last = max(0, 31 - done), 2 * nl - 33 - done
        return 3 * (min(last + 1, nl - 1) - lo)
    return 3 * nl * (ni - 1)


def product_mads(nl: int) -> int:
    """One Montgomery product: NL^2 multiply-adds of a*b, NL^2 of the reduction, its flush."""
    return 2 * nl * nl + flush_instructions(nl, plain_product=True)


def sop_mads(nl: int) -> int:
    """a*b + c*d with one reduction (fp_mul2): three times NL^2 and one full flush."""
    return 3 * nl * nl + flush_instructions(nl)


def square_mads(nl: int, segments: int = 5) -> int:
    """Multiply-adds of one Montgomery squaring by the segmented square of fpmont.hpp: row i of segment
    [lo, hi) multiplies a_i by the limbs j >= lo (doubled beyond hi), plus the nl reduction MADs per row."""
    if nl < 8 or segments <= 1:
        return 2 * nl * nl
    q = ((nl // segments + 1) // 2) * 2
    bounds = [k * q for k in range(segments)] + [nl]
    prod = sum((hi - lo) * (nl - lo) for lo, hi in zip(bounds[:-1], bounds[1:]))
    return prod + nl * nl + flush_instructions(nl)


def algorithmic_mads_per_pairing(fx, run: int = 16, window: int = 5, segments: int = 5) -> int:
    """32x32->64 multiply-adds one pairing executes in this formulation: general field products at 2*NL^2
    (schoolbook product + Montgomery reduction rows; + the 3*NL instructions of the mid-product flush at 36 / 37
    limbs), squarings at the segmented square's count.
    The F_p inversion of the final exponentiation is shared by `run` pairings per lane."""
    return int(mads_from_counts(*pairing_counts(fx, run, window), limbs_for(int(fx["p"], 16)), segments))


def pairing_counts(fx, run: int = 16, window: int = 5):
    """(reductions, squarings, sums of two products) of one pairing of k_pairing<NL, 0>: the windowed Miller loop of
    pairing.hpp with round 5's step programs (STEP_*), norms (both passes), the shared inversion's peel,
    conj(f)^2 / N, ^l and the conversions out of Montgomery form."""
    n, l = int(fx["n"], 16), int(fx["l"])
    dbl, add, threes, pre = miller_schedule(n, window)
    lb = l.bit_length()
    lpow = (lb - 1) * 2 + (bin(l).count("1") - 1) * 3     # F_p^2 squarings / products of ^l (no field squarings)
    npts = ((1 << (window - 2)) - 1) if window >= 3 else 0
    red = pre + dbl * STEP_DBL[0] + add * STEP_ADD[0] + threes * STEP_MULF[0] + 2 * 2 + 3 + 5 + lpow + 2
    # field squarings: in the steps, in the set-up of the windowed loop (its steps + z^2 of every affine
    # conversion), 4 + 2 in the norms / conj(f)^2
    squares = dbl * STEP_DBL[1] + add * STEP_ADD[1] + 6
    sops = dbl * STEP_DBL[2] + add * STEP_ADD[2] + threes * STEP_MULF[2]
    if window >= 3:
        squares += STEP_DBL[1] + npts * STEP_ADD[1] + (npts + 1)
        sops += STEP_DBL[2] + npts * STEP_ADD[2] + (npts - 1) * STEP_MULF[2]
    return red + INVERSION_PRODUCTS / run, float(squares), float(sops)


# ---- field products per unit of the other operations -----------------------------------------
# Every *_counts function returns (field products, how many of them are squarings); mads_from_counts prices them in
# 32x32->64 multiply-adds — a general product at 2*NL^2, a squaring at the segmented square's count — which is what
# bench.py holds against the measured v_mad_u64_u32 issue peaks (roofline_valu).  An F_p inversion by division steps
# is priced as INVERSION_PRODUCTS general products (its multiply-adds are of the same instruction, fpinv.hpp).
def mads_from_counts(products: float, squares: float, sops: float = 0.0, nl: int = 36, segments: int = 5) -> float:
    """products = everything that ends in a reduction (plain products, squarings, sums of two products); squares and
    sops say how many of them are which."""
    return ((products - squares - sops) * product_mads(nl) + squares * square_mads(nl, segments) + sops * sop_mads(nl))


def l2_add_mads(nl: int) -> int:
    """Multiply-adds of one level-2 Add / Sub in the fused kernel (csrc/barrett.hpp): two double-width sums of two
    products (4 NL^2) and, for each, the columns NL .. 2NL+1 of A * mu (the quotient estimate; A and mu have NL + 2
    limbs) and the low NL columns of q * (B^NL - p)."""
    quot = (nl + 2) ** 2 - nl * (nl + 1) // 2 - 1
    rem = nl * (nl + 1) // 2
    return 2 * (2 * nl * nl + quot + rem)


def multconst_counts(level: int, scalar_bits: int):
    """(reductions, squarings, sums of two products) of one MultConst with a per-element scalar (bgn.go:253-291).
    Level 1, k_g1_mul (ops.hpp g1_scalarmul_win_lane): a per-element table of the multiples 1..15 of the base (14 mixed
    Jacobian additions of 12 reductions, 3 of them squarings; one shared inversion over their Z: 14 prefix products,
    the inversion, 2 peels and zi^2, x, zi^3, y per entry), then per 4-bit window four doublings (9 reductions, 6
    squarings each) and one mixed addition (executed whenever a lane of the wave has a non-zero digit), and the
    affine conversion of the result (inversion + 4, one squaring).  Scalars below 128 bits: the same with 2-bit
    windows over the multiples 1..3.
    Level 2, k_gt_pow on bases of norm 1 (ops.hpp gt_pow_norm1_lane; every level-2 ciphertext is one, checked with two
    squarings and a conversion): per bit one product and one squaring on the real parts, then one inversion and two
    products for the imaginary part and two conversions out.  (The general square-and-multiply in F_p^2 it replaces
    since round 6 took 5 reductions per bit.)"""
    if level == 1:
        if scalar_bits < 24:                         # the binary ladder: a doubling and (in some lane of the wave) an addition per bit
            return float(scalar_bits * 21 + INVERSION_PRODUCTS + 4), float(scalar_bits * 9 + 1), 0.0
        wb = 4 if scalar_bits >= 128 else 2          # engine.cpp g1_mul_launch: 2-bit windows for 3 .. 15 scalar bytes
        ent = (1 << wb) - 2                          # table entries beside the base itself
        windows = -(-scalar_bits // wb)
        table = ent * 12 + ent + INVERSION_PRODUCTS + ent * 2 + ent * 4
        red = table + windows * (wb * 9 + 12) + INVERSION_PRODUCTS + 4
        sq = ent * 3 + ent + windows * (wb * 6 + 3) + 1
        return float(red), float(sq), 0.0
    return float(2 * scalar_bits + 3 + INVERSION_PRODUCTS + 2 + 2), float(scalar_bits + 2), 0.0


def _run_for(count: int) -> int:
    """Elements per lane of the batched-inversion kernels (engine.cpp run_for: the ceiling of count / 65536, no cap)."""
    return max(1, -(-count // 65536))


def eadd_counts(count: int):
    """Affine addition with Montgomery's trick over a lane's run: 7 products (one of them lambda^2) + one inversion
    per run."""
    return 7 + INVERSION_PRODUCTS / _run_for(count), 1.0, 0.0


def eadd_products(count: int) -> float:
    return eadd_counts(count)[0]


def encrypt_counts(x_bits: int, r_bits: int, wbits_p: int = 16, wbits_q: int = 20, signed_q: bool = True, chains: int = 4):
    """Fixed-base product P^m * Q^r: one affine addition per window and accumulation chain slot (four chains advanced
    together: the windows round up to a multiple of four; runs of 64) and three more to sum the chains; one squaring
    (lambda^2) per addition.  Q's windows are signed by default since round 5 (wbits_q + 1 scalar bits each over the
    2^wbits_q-entry table: 49 windows for 1024 bits instead of 52)."""
    wr = (r_bits // (wbits_q + 1) + 1) if signed_q else -(-r_bits // wbits_q)
    windows = -(-x_bits // wbits_p) + wr
    adds = -(-windows // chains) * chains + (chains - 1)
    return adds * (7 + INVERSION_PRODUCTS / 64), float(adds), 0.0


def encrypt_products(x_bits: int, r_bits: int, wbits_p: int = 16, wbits_q: int = 20, signed_q: bool = True) -> float:
    return encrypt_counts(x_bits, r_bits, wbits_p, wbits_q, signed_q)[0]


def _naf_counts(n: int):
    d = _signed_digits(n, 2)
    return len(d) - 1, sum(1 for i, x in enumerate(d[:-1]) if x and i != 0)


def decrypt_counts(fx, baby_steps: int, level: int = 1):
    """Decrypt (bgn.go:218-250): level 1 lifts with the Miller loop over the normalised line table of q1*P along
    the NAF of q2 = n/q1 (5 / 3 reductions per doubling / addition step, two of them sums of two products — f*l —,
    none of them a field squaring: f^2 is two general products) and the final exponentiation (F0^2, F1^2 of the norm are squarings); both levels raise to q1
    on the norm-1 ladder (per bit one squaring A_j^2 and one product, one inversion for the imaginary part) and
    walk G giant steps at 1.1 products each (bsgs.hpp)."""
    import math
    n, q1, l, T = int(fx["n"], 16), int(fx["q1"], 16), int(fx["l"]), int(fx["msg_space"])
    B = math.isqrt(T - 1) + 1 if T > 1 else 1
    mmax = B * B + B + 2
    S = max(1, int(baby_steps))
    G = (mmax + S) // (2 * S) + 1
    prods = 2 * q1.bit_length() + INVERSION_PRODUCTS + 1.1 * G + 40
    squares = float(q1.bit_length())
    sops = 0.0
    if level == 1:
        dbl, add = _naf_counts(n // q1)
        lb = l.bit_length()
        prods += dbl * TABLE_DBL[0] + add * TABLE_ADD[0] + 4 + 5 + INVERSION_PRODUCTS / 16 + (lb - 1) * 2 + (bin(l).count("1") - 1) * 3 + 2
        squares += 4
        sops += dbl * TABLE_DBL[2] + add * TABLE_ADD[2]
    return prods, squares, sops


def decrypt_products(fx, baby_steps: int, level: int = 1) -> float:
    return decrypt_counts(fx, baby_steps, level)[0]


def multpoly_counts_per_pair(fx, d: int, levels: int = None, multi: bool = True):
    """MultPoly of two d-coefficient polynomials (d a power of two >= 4) per coefficient pair, the way the engine runs
    2^14 of them (engine.cpp poly_plan_levels picks two Karatsuba levels for 16 x 16 there: 9 leaves of 4 x 4): per leaf
    of k x k coefficients k table builds (10 / 11 reductions per doubling / addition step, three of them squarings, two
    of them sums) and, since round 5, 2k - 1 multi-pairing lanes — per lane one f^2 per doubling step (2 reductions) and
    one final exponentiation, per term e(a_i, b_j) the line value and f*l (4 reductions a step, two of them sums of two
    products).  multi = False: the one-lane-per-pair walk (6 / 4 reductions per step and a final exponentiation per pair).
    levels: Karatsuba levels (default: down to leaves of 4 x 4)."""
    n, l = int(fx["n"], 16), int(fx["l"])
    dbl, add = _naf_counts(n)
    lb = l.bit_length()
    fe = 4 + 5 + INVERSION_PRODUCTS + (lb - 1) * 2 + (bin(l).count("1") - 1) * 3 + 2
    if levels is None:
        levels = 0
        while (d >> levels) > 4:
            levels += 1
    k = d >> levels
    leaves = 3 ** levels
    build = (dbl * BUILD_DBL[0] + add * BUILD_ADD[0], dbl * BUILD_DBL[1] + add * BUILD_ADD[1], dbl * BUILD_DBL[2] + add * BUILD_ADD[2])
    if multi:
        lane = (dbl * 2 + fe, 4.0, 0.0)
        term = ((dbl + add) * 4, 0.0, (dbl + add) * 2)
        leaf = tuple(k * build[t] + (2 * k - 1) * lane[t] + k * k * term[t] for t in range(3))
    else:
        fe16 = fe - INVERSION_PRODUCTS + INVERSION_PRODUCTS / 16
        ev = (dbl * TABLE_DBL_C[0] + add * TABLE_ADD_C[0] + fe16, 4.0, dbl * TABLE_DBL_C[2] + add * TABLE_ADD_C[2])
        leaf = tuple(k * build[t] + k * k * ev[t] for t in range(3))
    per = leaves / (d * d)
    return per * leaf[0], per * leaf[1], per * leaf[2]


def multpoly_products_per_pair(fx, d: int) -> float:
    return multpoly_counts_per_pair(fx, d)[0]
