"""CPU oracle A — pure-Python big-integer restatement of the BGN hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (bgn_amd/) may import
this module; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may, and there only as the checker.

PARITY UNPINNED: the reference (sachaservan/bgn) delegates every arithmetic
operation on this path to github.com/Nik-U/pbc v0.0.0-20181205041846-3e516ca0c5d6
(go.mod:5) -> libpbc 0.5.14 -> GMP, none of which is under /root/reference,
and its tests hold no golden vectors (every test draws a fresh random key,
bgn_test.go:17, gadgets_test.go:10, poly_test.go:69).  There is no Go toolchain
and no libpbc in the build container, so this oracle cannot be checked against
reference outputs.  What it restates instead:

  * the scheme logic of the reference, line by line (citations below), and
  * the *published* Type-A1 construction of PBC 0.5.14 (a1_param.c, curve.c,
    fieldquadratic.c): curve y^2 = x^3 + x over F_p with p = l*n - 1 = 3 mod 4,
    G1 = E(F_p)[n], GT in F_p^2 = F_p[i]/(i^2+1), distortion map
    phi(x, y) = (-x, i*y), reduced Tate pairing
        e(P, Q) = f_{n,P}(phi(Q)) ^ ((p^2 - 1) / n),
    wire bytes = fixed-length big-endian x||y (G1) / re||im (GT).

Every output on the path is a canonical representative (affine (x, y) mod p,
a + b*i with 0 <= a, b < p, an integer plaintext), so any correct
implementation of these conventions produces the same bytes.  The Miller loop
here is deliberately the slowest, most transparent formulation: affine
coordinates, full divisors (numerator lines AND denominator vertical lines),
and the final exponent applied as one big F_p^2 power.

Only pure Python and the standard library are used.
"""
from __future__ import annotations

import math
import random
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

Point = Optional[Tuple[int, int]]          # None = point at infinity (group identity)
Fp2 = Tuple[int, int]                      # (re, im), i^2 = -1


# --------------------------------------------------------------------------
# number theory helpers (key generation only; bgn.go:151-168, pbc a1 param gen)
# --------------------------------------------------------------------------
_SMALL_PRIMES = [2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97]


def is_probable_prime(n: int, rounds: int = 40, rng: Optional[random.Random] = None) -> bool:
    if n < 2:
        return False
    for sp in _SMALL_PRIMES:
        if n % sp == 0:
            return n == sp
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    rng = rng or random.Random(n & 0xFFFFFFFF)
    for _ in range(rounds):
        a = rng.randrange(2, n - 1)
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def rand_prime(bits: int, rng: random.Random) -> int:
    """Like Go's crypto/rand.Prime (bgn.go:153,160): top TWO bits set, odd, so a
    product of two such primes has exactly 2*bits bits."""
    while True:
        c = rng.getrandbits(bits) | (3 << (bits - 2)) | 1
        if is_probable_prime(c, rng=rng):
            return c


def a1_params_from_n(n: int) -> Tuple[int, int]:
    """PBC pbc_param_init_a1_gen: smallest l = 0 mod 4 with p = l*n - 1 prime
    (bgn.go:93; 'p + 1 = l*n' bgn.go:107-109).  Returns (p, l)."""
    l = 4
    while True:
        p = l * n - 1
        if is_probable_prime(p):
            return p, l
        l += 4


# --------------------------------------------------------------------------
# F_p^2 = F_p[i]/(i^2+1)
# --------------------------------------------------------------------------
def f2_mul(a: Fp2, b: Fp2, p: int) -> Fp2:
    return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)


def f2_inv(a: Fp2, p: int) -> Fp2:
    nrm = pow((a[0] * a[0] + a[1] * a[1]) % p, -1, p)
    return (a[0] * nrm % p, (-a[1]) * nrm % p)


def f2_pow(a: Fp2, e: int, p: int) -> Fp2:
    assert e >= 0
    r: Fp2 = (1, 0)
    for bit in bin(e)[2:] if e else "":
        r = f2_mul(r, r, p)
        if bit == "1":
            r = f2_mul(r, a, p)
    return r


F2_ONE: Fp2 = (1, 0)


# --------------------------------------------------------------------------
# E: y^2 = x^3 + x over F_p, affine, None = O
# --------------------------------------------------------------------------
def on_curve(P: Point, p: int) -> bool:
    if P is None:
        return True
    x, y = P
    return (y * y - (x * x * x + x)) % p == 0


def pt_neg(P: Point, p: int) -> Point:
    if P is None:
        return None
    return (P[0], (-P[1]) % p)


def pt_add(P: Point, Q: Point, p: int) -> Point:
    """Group law (PBC writes it multiplicatively: Element.Mul on G1)."""
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % p == 0:
            return None
        lam = (3 * x1 * x1 + 1) * pow(2 * y1, -1, p) % p
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
    x3 = (lam * lam - x1 - x2) % p
    y3 = (lam * (x1 - x3) - y1) % p
    return (x3, y3)


def pt_sub(P: Point, Q: Point, p: int) -> Point:
    return pt_add(P, pt_neg(Q, p), p)


def pt_mul(P: Point, k: int, p: int) -> Point:
    """P^k in PBC's multiplicative notation (Element.PowBig / MulBig on G1),
    k any non-negative integer (exponents >= n occur: gadgets_test.go:37-39)."""
    assert k >= 0
    R: Point = None
    for bit in bin(k)[2:] if k else "":
        R = pt_add(R, R, p)
        if bit == "1":
            R = pt_add(R, P, p)
    return R


# --------------------------------------------------------------------------
# Type-A1 reduced Tate pairing, full-divisor affine Miller loop
# --------------------------------------------------------------------------
def _line_eval(V: Tuple[int, int], W: Tuple[int, int], Qx: int, Qy: int, p: int) -> Tuple[Fp2, Point]:
    """Value at phi(Q) = (-Qx, i*Qy) of the line through V and W (tangent if
    V == W), and the point V+W.  Returns (value in F_p^2, V+W)."""
    xv, yv = V
    xw, yw = W
    if xv == xw and (yv + yw) % p == 0:
        # vertical line x - xv; V + W = O
        return (((-Qx) - xv) % p, 0), None
    if V == W:
        lam = (3 * xv * xv + 1) * pow(2 * yv, -1, p) % p
    else:
        lam = (yw - yv) * pow(xw - xv, -1, p) % p
    # l(x, y) = (y - yv) - lam (x - xv) at x = -Qx, y = i*Qy
    re = (-yv - lam * ((-Qx) - xv)) % p
    im = Qy % p
    x3 = (lam * lam - xv - xw) % p
    y3 = (lam * (xv - x3) - yv) % p
    return (re, im), (x3, y3)


def _vertical_eval(V: Point, Qx: int, p: int) -> Fp2:
    if V is None:
        return F2_ONE
    return (((-Qx) - V[0]) % p, 0)


def miller(P: Point, Q: Point, n: int, p: int) -> Fp2:
    """f_{n,P}(phi(Q)) with div(f) = n(P) - n(O); bits of n MSB -> LSB."""
    assert P is not None and Q is not None
    Qx, Qy = Q
    num: Fp2 = F2_ONE
    den: Fp2 = F2_ONE
    V: Point = P
    for bit in bin(n)[3:]:
        l, V2 = _line_eval(V, V, Qx, Qy, p)
        num = f2_mul(f2_mul(num, num, p), l, p)
        den = f2_mul(f2_mul(den, den, p), _vertical_eval(V2, Qx, p), p)
        V = V2
        if bit == "1":
            l, V2 = _line_eval(V, P, Qx, Qy, p)
            num = f2_mul(num, l, p)
            den = f2_mul(den, _vertical_eval(V2, Qx, p), p)
            V = V2
    assert V is None, "P does not have order dividing n"
    return f2_mul(num, f2_inv(den, p), p)


def pairing(P: Point, Q: Point, n: int, p: int) -> Fp2:
    """e(P, Q) = f_{n,P}(phi(Q))^((p^2-1)/n).  Identity in either slot maps to
    1 in GT (pbc pairing_apply; reached from MultPoly padding poly.go:134)."""
    if P is None or Q is None:
        return F2_ONE
    f = miller(P, Q, n, p)
    assert (p * p - 1) % n == 0
    return f2_pow(f, (p * p - 1) // n, p)


# --------------------------------------------------------------------------
# wire format (PBC element_to_bytes: fixed-length big-endian, x||y / re||im)
# --------------------------------------------------------------------------
def fp_len(p: int) -> int:
    return (p.bit_length() + 7) // 8


def elem_to_bytes(e, p: int) -> bytes:
    """G1 point or GT element -> 2L bytes.  The identity of G1 has no wire
    encoding in PBC; the engine's ABI carries a separate flag and all-zero bytes."""
    L = fp_len(p)
    if e is None:
        return bytes(2 * L)
    return int(e[0]).to_bytes(L, "big") + int(e[1]).to_bytes(L, "big")


def elem_from_bytes(b: bytes, p: int) -> Tuple[int, int]:
    L = fp_len(p)
    assert len(b) == 2 * L
    return (int.from_bytes(b[:L], "big"), int.from_bytes(b[L:], "big"))


# --------------------------------------------------------------------------
# scheme: bgn.go
# --------------------------------------------------------------------------
@dataclass
class Ciphertext:
    """ciphertext.go:12-15"""
    C: object            # Point (L1) or Fp2 (L2)
    L2: bool = False


@dataclass
class PublicKey:
    """bgn.go:28-41 (fields on the hot path only)."""
    p: int
    n: int
    l: int
    P: Tuple[int, int]
    Q: Tuple[int, int]
    MsgSpace: int
    Deterministic: bool = True
    PolyBase: int = 3
    # gsbs.go:12-15 keeps these as package globals; here they belong to the key
    tableG1: Dict[object, int] = field(default_factory=dict, repr=False)
    tableGT: Dict[object, int] = field(default_factory=dict, repr=False)
    tablesComputed: bool = False

    @property
    def N(self) -> int:
        return self.n

    # ---- encryption: bgn.go:325-353 ----
    def EncryptDeterministic(self, x: int) -> Ciphertext:
        return Ciphertext(pt_mul(self.P, x % self.n, self.p), False)

    def EncryptWithRandomness(self, x: int, r: int) -> Ciphertext:
        # Negative x: the reference passes it straight to PBC's pow whose
        # behaviour for negative exponents is not defined in-tree
        # (cmd/main.go:81); this build defines P^(x mod n).
        G = pt_mul(self.P, x % self.n if x < 0 else x, self.p)       # bgn.go:344
        H = pt_mul(self.Q, r, self.p)                                # bgn.go:346
        return Ciphertext(pt_add(G, H, self.p), False)               # bgn.go:350

    def encryptZero(self) -> Ciphertext:
        return self.EncryptDeterministic(0)                          # bgn.go:562-564

    # ---- level lift / multiplication: bgn.go:294-321 ----
    def e(self, A: Point, B: Point) -> Fp2:
        return pairing(A, B, self.n, self.p)

    def makeL2(self, ct: Ciphertext) -> Ciphertext:
        assert not ct.L2
        return Ciphertext(self.e(ct.C, self.P), True)                # bgn.go:318

    def Mult(self, ct1: Ciphertext, ct2: Ciphertext, r: Optional[int] = None) -> Ciphertext:
        assert not ct1.L2 and not ct2.L2
        res = self.e(ct1.C, ct2.C)                                   # bgn.go:300
        if not self.Deterministic:
            assert r is not None, "randomness is an input of the engine"
            pair = f2_pow(self.e(self.Q, self.Q), r, self.p)         # bgn.go:306-309
            res = f2_mul(res, pair, self.p)                          # bgn.go:310
        return Ciphertext(res, True)

    # ---- add / sub / neg: bgn.go:375-497 ----
    def _align(self, a: Ciphertext, b: Ciphertext):
        if a.L2 and not b.L2:
            b = self.makeL2(b)                                       # bgn.go:447-449
        if not a.L2 and b.L2:
            a = self.makeL2(a)                                       # bgn.go:451-453
        return a, b

    def Add(self, a: Ciphertext, b: Ciphertext, r: Optional[int] = None) -> Ciphertext:
        ct1, ct2 = self._align(a, b)
        if ct1.L2:
            res = f2_mul(ct1.C, ct2.C, self.p)                       # bgn.go:460
            if not self.Deterministic:
                res = f2_mul(res, f2_pow(self.e(self.Q, self.Q), r, self.p), self.p)   # :469-474
            return Ciphertext(res, True)
        res = pt_add(ct1.C, ct2.C, self.p)                           # bgn.go:482
        if not self.Deterministic:
            res = pt_add(res, pt_mul(self.Q, r, self.p), self.p)     # bgn.go:492-495
        return Ciphertext(res, False)

    def Sub(self, a: Ciphertext, b: Ciphertext, r: Optional[int] = None) -> Ciphertext:
        ct1, ct2 = self._align(a, b)
        if ct1.L2:
            res = f2_mul(ct1.C, f2_inv(ct2.C, self.p), self.p)       # bgn.go:397
            if not self.Deterministic:
                res = f2_mul(res, f2_pow(self.e(self.Q, self.Q), r, self.p), self.p)   # :406-410
            return Ciphertext(res, True)
        res = pt_sub(ct1.C, ct2.C, self.p)                           # bgn.go:419
        if not self.Deterministic:
            res = pt_add(res, pt_mul(self.Q, r, self.p), self.p)     # bgn.go:428-431
        return Ciphertext(res, False)

    def Neg(self, c: Ciphertext) -> Ciphertext:
        return self.Sub(self.encryptZero(), c)                       # bgn.go:436-438

    # ---- multiply by a plaintext constant: bgn.go:253-291 ----
    def MultConst(self, c: Ciphertext, k: int, r: Optional[int] = None) -> Ciphertext:
        assert k >= 0
        if not c.L2:
            res = pt_mul(c.C, k, self.p)                             # bgn.go:258
            if not self.Deterministic:
                res = pt_add(res, pt_mul(self.Q, r, self.p), self.p) # bgn.go:265-268
            return Ciphertext(res, False)
        res = f2_pow(c.C, k, self.p)                                 # bgn.go:277
        if not self.Deterministic:
            res = f2_mul(res, f2_pow(self.e(self.Q, self.Q), r, self.p), self.p)       # :283-287
        return Ciphertext(res, True)

    # ---- BSGS tables: gsbs.go:17-51 ----
    def PrecomputeTables(self, genG1: Point, genGT: Fp2) -> None:
        bound = int(math.ceil(math.sqrt(float(self.MsgSpace)))) + 1  # gsbs.go:44
        self.tableG1, self.tableGT = {}, {}
        aux = genGT
        for j in range(bound + 1):                                   # gsbs.go:33-36
            self.tableGT[aux] = j
            aux = f2_mul(aux, genGT, self.p)
        aux2 = genG1
        for j in range(bound + 1):                                   # gsbs.go:22-25
            self.tableG1[aux2] = j
            aux2 = pt_add(aux2, genG1, self.p)
        self.tablesComputed = True

    def SetupDecryption(self, sk: "SecretKey") -> None:
        genG1 = pt_mul(self.P, sk.Key, self.p)                       # bgn.go:196-197
        genGT = f2_pow(self.e(self.P, self.P), sk.Key, self.p)       # bgn.go:198-199
        self.PrecomputeTables(genG1, genGT)

    def getDL(self, csk, gsk, l2: bool) -> Optional[int]:
        """gsbs.go:54-106.  Returns None where the reference returns the error
        'cannot find discrete log; out of bounds'."""
        if not self.tablesComputed:
            raise RuntimeError("DL tables not computed!")            # gsbs.go:56-58 (panic)
        bound = int(math.ceil(math.sqrt(float(self.MsgSpace))))      # gsbs.go:60
        if l2:
            gamma_inv = f2_inv(f2_pow(gsk, bound, self.p), self.p)   # gsbs.go:71-72
            aux = csk
            for i in range(bound + 1):                               # gsbs.go:77
                v = self.tableGT.get(aux)
                if v is not None:
                    return i * bound + v + 1                         # gsbs.go:98
                aux = f2_mul(aux, gamma_inv, self.p)                 # gsbs.go:102
            return None
        gamma = pt_mul(gsk, bound, self.p)
        aux = csk
        for i in range(bound + 1):
            v = self.tableG1.get(aux)
            if v is not None:
                return i * bound + v + 1
            aux = pt_sub(aux, gamma, self.p)
        return None

    def recoverMessage(self, gsk, csk, l2: bool) -> Optional[int]:
        """bgn.go:357-372"""
        if (csk == F2_ONE) if l2 else (csk is None):
            return 0
        return self.getDL(csk, gsk, l2)

    # ---- poly layer (coefficient loops): poly.go ----
    def MultPoly(self, c1: Sequence[Ciphertext], c2: Sequence[Ciphertext]) -> List[Ciphertext]:
        """poly.go:123-156: result[i+k] = prod e(c1[i], c2[k]); degree = d1+d2
        slots, the last one stays makeL2(encryptZero()) = 1 in GT."""
        deg = len(c1) + len(c2)
        result = [self.makeL2(self.encryptZero()) for _ in range(deg)]   # poly.go:130-137
        for i in range(len(c1)):
            for k in range(len(c2)):
                coeff = self.Mult(c1[i], c2[k])                          # poly.go:146
                result[i + k] = self.Add(result[i + k], coeff)           # poly.go:148
        return result

    def AddPolyAligned(self, c1: Sequence[Ciphertext], c2: Sequence[Ciphertext]) -> List[Ciphertext]:
        """poly.go:188-204 (after scale alignment, which is host control flow)."""
        deg = max(len(c1), len(c2))
        out: List[Ciphertext] = []
        for i in range(deg):
            if i >= len(c2):
                out.append(c1[i])
            elif i >= len(c1):
                out.append(c2[i])
            else:
                out.append(self.Add(c1[i], c2[i]))
        return out

    def EvalPoly(self, coeffs: Sequence[Ciphertext]) -> Ciphertext:
        """poly.go:58-68: Horner in the poly base."""
        acc = self.EncryptDeterministic(0)
        for c in reversed(coeffs):
            acc = self.MultConst(acc, self.PolyBase)
            acc = self.Add(acc, c)
        return acc

    def NegPoly(self, coeffs: Sequence[Ciphertext]) -> List[Ciphertext]:
        """poly.go:45-55"""
        return [self.Sub(self.encryptZero(), c) for c in coeffs]

    def MultConstPoly(self, coeffs: Sequence[Ciphertext], l2: bool, digits: Sequence[int]) -> List[Ciphertext]:
        """poly.go:71-120 after the sign split and NewUnbalancedPlaintext (`digits` = poly.Coefficients):
        result[i+k] = Add(result[i+k], MultConst(ct[i], poly[k])), every slot starting from zero
        (makeL2(zero) for a level-2 operand, poly.go:85-93).  Deterministic mode only."""
        degree = len(coeffs) + len(digits)
        zero = self.encryptZero()
        if l2:
            zero = self.makeL2(zero)
        result = [zero for _ in range(degree)]
        for i in range(len(coeffs) - 1, -1, -1):
            for k in range(len(digits) - 1, -1, -1):
                coeff = self.MultConst(coeffs[i], digits[k])              # poly.go:106
                result[i + k] = self.Add(result[i + k], coeff)            # poly.go:107
        return result

    def EncryptPolyCoeffs(self, coeffs: Sequence[int], rs: Sequence[int]) -> List[Ciphertext]:
        """poly.go:11-29: negative digits become Sub(zero, Enc(|c|))."""
        out = []
        for c, r in zip(coeffs, rs):
            if c < 0:
                out.append(self.Sub(self.encryptZero(), self.EncryptWithRandomness(-c, r)))
            else:
                out.append(self.EncryptWithRandomness(c, r))
        return out


# --------------------------------------------------------------------------
# proofs (gadgets.go)
# --------------------------------------------------------------------------
@dataclass
class ProofOfPlaintextKnowledge:
    """gadgets.go:10-14"""
    Ct: Ciphertext
    Nonce: Ciphertext
    DL: Optional[int]


@dataclass
class DecryptionProof:
    """gadgets.go:18-21"""
    Value: int
    Randomness: int


def proof_hash(proof: ProofOfPlaintextKnowledge, p: int) -> int:
    """hash, gadgets.go:80-96: sha256 over Ct.C.Bytes() || Nonce.C.Bytes(), as a big-endian integer."""
    import hashlib
    return int.from_bytes(hashlib.sha256(elem_to_bytes(proof.Ct.C, p) + elem_to_bytes(proof.Nonce.C, p)).digest(), "big")


def NewProofOfPlaintextKnowledge(pk: "PublicKey", sk: "SecretKey", v: int, z: int, nonce1: int) -> ProofOfPlaintextKnowledge:
    """gadgets.go:32-54 with the nonce as an argument (the reference draws it, :33)."""
    ct = pk.EncryptWithRandomness(v, z)
    nonce = pk.EncryptWithRandomness(nonce1, 0)
    proof = ProofOfPlaintextKnowledge(ct, nonce, None)
    nonce2 = proof_hash(proof, pk.p)
    DL = nonce1 + nonce2 * v
    DL += sk.R * z * nonce2 * (pk.n // sk.Key)
    proof.DL = DL % pk.n
    return proof


def CheckDecryptionProof(pk: "PublicKey", ct: Ciphertext, proof: DecryptionProof) -> bool:
    """gadgets.go:57-61"""
    return ct.C == pk.EncryptWithRandomness(proof.Value, proof.Randomness).C


def CheckProofOfPlaintextKnoewledge(pk: "PublicKey", ct: Ciphertext, proof: ProofOfPlaintextKnowledge) -> bool:
    """gadgets.go:65-77"""
    nonce2 = proof_hash(proof, pk.p)
    res = pt_add(pt_mul(ct.C, nonce2, pk.p), proof.Nonce.C, pk.p)
    return pt_mul(pk.P, proof.DL, pk.p) == res


# --------------------------------------------------------------------------
# plaintext encoding (plaintext.go) — CPU side of the boundary; restated so the
# poly-layer tests can drive MultConstPoly / alignPolyCiphertexts like poly_test.go
# --------------------------------------------------------------------------
DEGREE_BOUND = 128                                   # plaintext.go:11


def _degree_tables(base: int):
    """computeEncodingTable, plaintext.go:104-124"""
    deg = [base ** i for i in range(DEGREE_BOUND)]
    sums, acc = [], 0
    for v in deg:
        acc += v
        sums.append(acc)
    return deg, sums


def _degree(target: int, deg, sums, bound: int, balanced: bool) -> int:
    """degree, plaintext.go:127-151"""
    if target == 1:
        return 0
    if balanced:
        for i in range(1, bound + 1):
            if sums[i] >= target:
                return i
    else:
        for i in range(1, bound + 1):
            if deg[i] > target:
                return i - 1
    return -1


def unbalancedEncode(target: int, base: int) -> List[int]:
    """plaintext.go:164-212.  Returns Coefficients (len = Degree)."""
    deg, sums = _degree_tables(base)
    if target == 0:
        return [0]
    if target < 0:
        raise ValueError("Negative encoding not supported")
    coefficients = [0] * DEGREE_BOUND
    bound = len(sums)
    last = DEGREE_BOUND
    while True:
        index = _degree(target, deg, sums, last - 1 if last == DEGREE_BOUND else last, False)
        last = index + 1
        if bound == len(sums):
            bound = index + 1
        value = deg[index]
        if 2 * value <= target:
            value = 2 * value
            coefficients[index] = 2
        else:
            coefficients[index] = 1
        if value == target:
            return coefficients[:bound + 1]
        target -= value


def balancedEncode(target: int, base: int) -> List[int]:
    """plaintext.go:214-268"""
    deg, sums = _degree_tables(base)
    if target == 0:
        return [0]
    negative = target < 0
    if negative:
        target = -target
    coefficients = [0] * DEGREE_BOUND
    bound = len(sums)
    last = DEGREE_BOUND - 1
    next_negative = False
    while True:
        index = _degree(target, deg, sums, last, True)
        last = index
        if bound == len(sums):
            bound = index
        coefficients[index] = -1 if next_negative else 1
        if deg[index] == target:
            if negative:
                for i in range(bound + 1):
                    coefficients[i] *= -1
            return coefficients[:bound + 1]
        if deg[index] > target:
            next_negative = not next_negative
            target = deg[index] - target
        else:
            target -= deg[index]


def rationalize(x: float, base: int, precision: float) -> Tuple[int, int]:
    """plaintext.go:271-317, float64 for float64."""
    factor = math.floor(x)
    x = 1.0 + math.remainder(x, 1.0)
    if abs(x) > 1.0:
        x += 1.0
    if x >= 0.0:
        x -= float(int(x))
    elif x <= -0.0:
        x += float(int(x))
    num = 1.0
    pw = 1.0
    qmin, qmax = x - precision, x + precision
    while True:
        denom = math.pow(float(base), pw)
        rat = num / denom
        if qmin <= rat <= qmax:
            while int(num) % base == 0:
                num = num / float(base)
                pw -= 1
            denom = math.pow(float(base), pw)
            return int(factor * denom + num), int(pw)
        if num + 1 >= denom:
            num = 1.0
            pw += 1
        num += 1


def NewUnbalancedPlaintext(m, base: int, fp_scale_base: int, fp_precision: float) -> Tuple[List[int], int]:
    """plaintext.go:34-63 -> (Coefficients, ScaleFactor)"""
    mf = float(m)
    if math.remainder(mf, 1.0) != 0.0:
        numerator, scale = rationalize(mf - math.floor(mf), fp_scale_base, fp_precision)
        m_int = int(mf)                                  # big.Float.Int truncates toward zero
        m_int = m_int * int(math.pow(float(fp_scale_base), float(scale))) + numerator
        return unbalancedEncode(m_int, base), scale
    return unbalancedEncode(int(mf), base), 0


def poly_eval_plain(coeffs: Sequence[int], base: int) -> int:
    """PolyEval (plaintext.go:320-339) on integer coefficients, without the scale division."""
    acc = 0
    for c in reversed(coeffs):
        acc = acc * base + c
    return acc


@dataclass
class SecretKey:
    """bgn.go:58-62: Key = q1."""
    Key: int
    R: int = 0
    PolyBase: int = 3

    def decrypt(self, ct: Ciphertext, pk: PublicKey, failed: bool = False) -> Optional[int]:
        """bgn.go:218-250.  None = error."""
        if ct.L2:
            gsk = f2_pow(pk.e(pk.P, pk.P), self.Key, pk.p)           # bgn.go:227-228
            csk = f2_pow(ct.C, self.Key, pk.p)                       # bgn.go:223
        else:
            gsk = pt_mul(pk.P, self.Key, pk.p)                       # bgn.go:222
            csk = pt_mul(ct.C, self.Key, pk.p)                       # bgn.go:223
        pt = pk.recoverMessage(gsk, csk, ct.L2)
        if pt is None and not failed:
            dec = self.decrypt(pk.Neg(ct), pk, True)                 # bgn.go:235-242
            if dec is None:
                return None
            return -dec
        return pt

    def Decrypt(self, ct: Ciphertext, pk: PublicKey) -> Optional[int]:
        return self.decrypt(ct, pk, False)

    def DecryptFailSafe(self, ct: Ciphertext, pk: PublicKey) -> int:
        v = self.decrypt(ct, pk, False)                              # bgn.go:210-216
        return 0 if v is None else v


# --------------------------------------------------------------------------
# key generation (CPU-side; bgn.go:65-138,170-192).  Seeded so fixtures are
# reproducible; the reference uses crypto/rand.
# --------------------------------------------------------------------------
def random_point_order_n(p: int, n: int, l: int, rng: random.Random) -> Tuple[int, int]:
    """PBC curve_random + cofactor clearing, then findGenerator's test
    (bgn.go:170-192) generalised: the point must have order exactly n."""
    assert p % 4 == 3
    while True:
        x = rng.randrange(p)
        rhs = (x * x * x + x) % p
        y = pow(rhs, (p + 1) // 4, p)
        if y * y % p != rhs:
            continue
        if rng.getrandbits(1):
            y = (-y) % p
        Pt = pt_mul((x, y), l, p)
        if Pt is None:
            continue
        return Pt


def NewKeyGen(keyBits: int, msgSpace: int, polyBase: int = 3, deterministic: bool = True,
              seed: int = 1) -> Tuple[PublicKey, SecretKey]:
    """bgn.go:65-138 with a seeded RNG."""
    if keyBits < 16:
        raise ValueError("key bits must be >= 16 bits in length")       # bgn.go:67-69 (panic)
    if keyBits % 2:
        raise ValueError("key bits must be divisible by 2")             # bgn.go:71-73 (panic)
    rng = random.Random(seed)
    while True:
        q1 = rand_prime(keyBits // 2, rng)
        q2 = rand_prime(keyBits // 2, rng)
        if q1 != q2:
            break
    if q1 < msgSpace or q2 < msgSpace:
        raise ValueError("Message space is greater than the group order!")  # bgn.go:87-89
    n = q1 * q2
    p, l = a1_params_from_n(n)
    while True:
        P = random_point_order_n(p, n, l, rng)
        # findGenerator: P^q1 != O, P^n == O (bgn.go:181-188); also P^q2 != O
        if pt_mul(P, q1, p) is None or pt_mul(P, q2, p) is None:
            continue
        assert pt_mul(P, n, p) is None
        break
    P = pt_mul(P, (4 * l) % n, p)                                        # bgn.go:113
    R = rng.randrange(n)                                                 # bgn.go:117
    Q = pt_mul(pt_mul(P, R, p), q2, p)                                   # bgn.go:118-119
    assert P is not None and Q is not None
    pk = PublicKey(p=p, n=n, l=l, P=P, Q=Q, MsgSpace=msgSpace, Deterministic=deterministic, PolyBase=polyBase)
    sk = SecretKey(Key=q1, R=R, PolyBase=polyBase)
    return pk, sk
