"""Test-side stand-in for the reference's plaintext.go — the fixed-point polynomial encoding that stays on the
Go/CPU side of the boundary (BASELINE.json north_star) and therefore is NOT part of the bgn_amd package.  The
reference's own tests (poly_test.go:68-189, cmd/main.go:24-72) are written against this layer; to run their literal
values through the HIP engine the tests need it.  Built on the oracle's restatement of the encoders
(oracle/bgn_ref.py: plaintext.go:104-317).  TEST INFRASTRUCTURE."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List

import bgn_ref as R


@dataclass
class PolyPlaintext:
    """plaintext.go:14-19 (the Pk back-pointer replaced by the two bases it is read for)."""
    Coefficients: List[int]
    Degree: int
    ScaleFactor: int
    PolyBase: int
    FPScaleBase: int

    def PolyEval(self) -> float:
        """plaintext.go:320-335: Horner from the top coefficient, then the division by FPScaleBase^ScaleFactor."""
        acc = 0.0
        for i in range(self.Degree - 1, -1, -1):
            acc = acc * float(self.PolyBase) + float(self.Coefficients[i])
        if self.ScaleFactor != 0:
            acc = acc / float(self.FPScaleBase ** self.ScaleFactor)
        return acc


class Encoding:
    """PolyEncodingParams of a PublicKey (bgn.go:43-48) with the constructors of plaintext.go."""

    def __init__(self, PolyBase: int, FPScaleBase: int, FPPrecision: float):
        self.PolyBase, self.FPScaleBase, self.FPPrecision = int(PolyBase), int(FPScaleBase), float(FPPrecision)

    def _mint(self, m: float):
        """The integer to encode and its scale factor (plaintext.go:40-62 / :77-100)."""
        mf = float(m)
        if math.remainder(mf, 1.0) != 0.0:
            numerator, scale = R.rationalize(mf - math.floor(mf), self.FPScaleBase, self.FPPrecision)
            return int(mf) * int(math.pow(float(self.FPScaleBase), float(scale))) + numerator, scale
        return int(mf), 0

    def NewPolyPlaintext(self, m: float) -> PolyPlaintext:
        """plaintext.go:66-101: balanced base-b digits."""
        if m < 0:
            raise ValueError("negative encodings not implemented")       # plaintext.go:72-74 (panic)
        mi, scale = self._mint(m)
        c = R.balancedEncode(mi, self.PolyBase)
        return PolyPlaintext(c, len(c), scale, self.PolyBase, self.FPScaleBase)

    def NewUnbalancedPlaintext(self, m: float) -> PolyPlaintext:
        """plaintext.go:34-63: unbalanced base-b digits (what MultConstPoly encodes its constant with)."""
        mi, scale = self._mint(m)
        c = R.unbalancedEncode(mi, self.PolyBase)
        return PolyPlaintext(c, len(c), scale, self.PolyBase, self.FPScaleBase)

    def decrypted(self, coeffs, scale: int) -> PolyPlaintext:
        """What DecryptPoly returns (poly.go:32-42): the decrypted coefficients with the ciphertext's scale."""
        c = [int(v) for v in coeffs]
        return PolyPlaintext(c, len(c), int(scale), self.PolyBase, self.FPScaleBase)


def f1(x: float) -> str:
    """fmt.Sprintf("%.1f\\n", x) — the comparison every reference poly test makes."""
    return "%.1f\n" % x
