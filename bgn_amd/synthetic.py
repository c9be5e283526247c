"""Synthetic workloads for bench.py and the scale tests."""
from __future__ import annotations

import numpy as np


def l1_ciphertext_pairs(pk, fx, count: int, seed: int, device):
    """Two arrays of `count` level-1 ciphertexts (wire bytes, uint8 CUDA tensors).

    Until the engine's Encrypt kernel is used here, operands are drawn
    (seeded, with replacement) from the key fixture's pool of valid
    ciphertexts.  The pairing kernel's control flow and instruction stream do
    not depend on operand values, so throughput is unaffected by the draw."""
    import torch
    pool = [bytes.fromhex(e["ct"]) for e in fx["encrypt"] if int(e["ct"], 16) != 0]
    pool_t = torch.from_numpy(np.frombuffer(b"".join(pool), dtype=np.uint8).reshape(len(pool), -1).copy()).to(device)
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    ia = torch.randint(0, len(pool), (count,), generator=g).to(device)
    ib = torch.randint(0, len(pool), (count,), generator=g).to(device)
    a = pool_t[ia].reshape(-1).contiguous()
    b = pool_t[ib].reshape(-1).contiguous()
    return a, b


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
        pre = 18 + npts * 17 + (npts - 1) * 3                # doubling step, addition steps, products by f_2
        pre += INVERSION_PRODUCTS + 7 * npts + 4 * npts      # shared inversion, peel + coordinates, canonical f_d
        if window >= 4:
            pre += INVERSION_PRODUCTS + 4 + 4                # 2A made affine, canonical f_2
    else:
        pre = 0
    return dbl, add, big, pre


# One F_p inversion by division steps (fpinv.hpp) costs about as much as 55 field products (measured).
INVERSION_PRODUCTS = 55


def square_mads(nl: int, segments: int = 5) -> int:
    """Multiply-adds of one Montgomery squaring by the segmented square of fp28.hpp: row i of segment
    [lo, hi) multiplies a_i by the limbs j >= lo (doubled beyond hi), plus the nl reduction MADs per row."""
    if nl < 8 or segments <= 1:
        return 2 * nl * nl
    q = ((nl // segments + 1) // 2) * 2
    bounds = [k * q for k in range(segments)] + [nl]
    prod = sum((hi - lo) * (nl - lo) for lo, hi in zip(bounds[:-1], bounds[1:]))
    return prod + nl * nl


def algorithmic_mads_per_pairing(fx, run: int = 16, window: int = 5, segments: int = 5) -> int:
    """32x32->64 multiply-adds one pairing executes in this formulation: general field products at 2*NL^2
    (schoolbook product + Montgomery reduction rows), squarings at the segmented square's count.
    The F_p inversion of the final exponentiation is shared by `run` pairings per lane."""
    p, n, l = int(fx["p"], 16), int(fx["n"], 16), int(fx["l"])
    nl = 38 if p.bit_length() > 600 else (19 if p.bit_length() > 300 else (10 if p.bit_length() > 100 else 3))
    dbl, add, threes, pre = miller_schedule(n, window)
    lb = l.bit_length()
    lpow = (lb - 1) * 2 + (bin(l).count("1") - 1) * 3     # F_p^2 squarings / products of ^l (no field squarings)
    # Miller + norms (both passes) + peel + conj(f)^2/N + ^l + from_mont
    per_pairing = pre + dbl * 18 + add * 17 + threes * 3 + 2 * 2 + 3 + 5 + lpow + 2
    # of which field squarings: 6 per doubling step, 3 per addition step (pairing.hpp), 9 in the set-up of the
    # windowed loop, 4 + 2 in the norms / conj(f)^2
    squares = dbl * 6 + add * 3 + ((6 + 3 * ((1 << (window - 2)) - 1) + 3) if window >= 3 else 0) + 6
    products = per_pairing + INVERSION_PRODUCTS / run
    return int((products - squares) * 2 * nl * nl + squares * square_mads(nl, segments))
