// bgn_amd.hpp — header-only C++ mirror of the reference's Go API for the hot path, over the C ABI
// (bgn_amd.h).  The reference is compiled Go and its toolchain is absent from the build image, so this
// is the compiled-language host side: same type and method names as sachaservan/bgn (bgn.go,
// ciphertext.go), same argument meaning and error behaviour.  Single-element methods are the count-1
// case of the batch entry points.  Scalars are unsigned big-endian byte strings (`Scalar`), with
// helpers for 64-bit values; randomness is always supplied by the caller.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "bgn_amd.h"

namespace bgn_amd {

using Bytes = std::vector<uint8_t>;
using Scalar = Bytes;   // unsigned, big-endian

inline Scalar scalar_u64(uint64_t v) {
  Scalar s(8);
  for (int i = 0; i < 8; ++i) s[i] = (uint8_t)(v >> (8 * (7 - i)));
  return s;
}

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& where) : std::runtime_error(where + ": " + bgn_last_error()), code(c) {}
};
// errors.New("cannot find discrete log; out of bounds"), gsbs.go:105
struct DecryptError : std::runtime_error {
  DecryptError() : std::runtime_error("cannot find discrete log; out of bounds") {}
};
inline void check(int rc, const char* where) {
  if (rc != BGN_OK) throw Error(rc, where);
}

// ciphertext.go:12-15 — C holds Element.Bytes(); the G1 identity is 2L zero bytes.
struct Ciphertext {
  Bytes C;
  bool L2 = false;
  Ciphertext Copy() const { return *this; }
};

// ciphertext.go:26-31
struct PolyCiphertext {
  std::vector<Ciphertext> Coefficients;
  int Degree = 0;
  int ScaleFactor = 0;
  bool L2 = false;
  PolyCiphertext Copy() const { return *this; }
};

// unbalancedEncode (plaintext.go:164-212) for a machine integer: greedy base-b digits, least significant
// first, with the reference's one zero coefficient above the top digit; 0 encodes as [0].  Plaintext
// encoding stays on the CPU side of the boundary.
inline std::vector<uint64_t> UnbalancedEncode(uint64_t m, uint64_t base) {
  std::vector<uint64_t> d;
  if (m == 0) return {0};
  while (m) {
    d.push_back(m % base);
    m /= base;
  }
  d.push_back(0);
  return d;
}

class SecretKey;

// bgn.go:28-41 (hot-path members)
class PublicKey {
 public:
  Bytes N;
  uint64_t MsgSpace;
  bool Deterministic;
  uint64_t PolyBase = 3;      // PolyEncodingParams, bgn.go:19-25
  uint64_t FPScaleBase = 3;

  PublicKey(const Bytes& p, const Bytes& n, uint64_t l, const Bytes& P, const Bytes& Q, uint64_t msgSpace,
            bool deterministic = true, int device = 0)
      : N(n), MsgSpace(msgSpace), Deterministic(deterministic) {
    check(bgn_ctx_create(&h_, p.data(), p.size(), n.data(), n.size(), l, P.data(), Q.data(), deterministic ? 1 : 0,
                         device),
          "bgn_ctx_create");
    E_ = 2 * bgn_fp_bytes(h_);
  }
  ~PublicKey() { bgn_ctx_destroy(h_); }
  PublicKey(const PublicKey&) = delete;
  PublicKey& operator=(const PublicKey&) = delete;

  size_t ElementBytes() const { return E_; }
  bgn_ctx* handle() const { return h_; }

  // ---- batch forms ----
  std::vector<Ciphertext> EncryptBatch(const std::vector<Scalar>& x, const std::vector<Scalar>* r) const {
    size_t xl = 0, rl = 0;
    Bytes xb = pack(x, xl), rb, out(x.size() * E_);
    if (r) rb = pack(*r, rl);
    check(bgn_encrypt_batch(h_, x.size(), xb.data(), xl, r ? rb.data() : nullptr, rl, out.data()), "bgn_encrypt_batch");
    return split(out, false);
  }
  std::vector<Ciphertext> AddBatch(const std::vector<Ciphertext>& a, const std::vector<Ciphertext>& b,
                                   const std::vector<Scalar>* r = nullptr, bool subtract = false) const {
    const int level = a.empty() || !a[0].L2 ? 1 : 2;
    size_t rl = 0;
    Bytes A = join(a), B = join(b), rb, out(a.size() * E_);
    if (r) rb = pack(*r, rl);
    check((subtract ? bgn_sub_batch : bgn_add_batch)(h_, a.size(), level, A.data(), B.data(), r ? rb.data() : nullptr, rl,
                                                     out.data()),
          subtract ? "bgn_sub_batch" : "bgn_add_batch");
    return split(out, level == 2);
  }
  std::vector<Ciphertext> MultBatch(const std::vector<Ciphertext>& a, const std::vector<Ciphertext>& b,
                                    const std::vector<Scalar>* r = nullptr) const {
    size_t rl = 0;
    Bytes A = join(a), B = join(b), rb, out(a.size() * E_);
    if (r) rb = pack(*r, rl);
    check(bgn_mult_batch(h_, a.size(), A.data(), B.data(), r ? rb.data() : nullptr, rl, out.data()), "bgn_mult_batch");
    return split(out, true);
  }

  // ---- bgn.go:325-353 ----
  Ciphertext EncryptWithRandomness(const Scalar& x, const Scalar& r) const {
    std::vector<Scalar> rs{r};
    return EncryptBatch({x}, &rs)[0];
  }
  Ciphertext EncryptDeterministic(const Scalar& x) const { return EncryptBatch({x}, nullptr)[0]; }
  Ciphertext encryptZero() const { return EncryptDeterministic(scalar_u64(0)); }   // bgn.go:562-564

  // ---- bgn.go:294-321 ----
  Ciphertext Mult(const Ciphertext& a, const Ciphertext& b, const Scalar* r = nullptr) const {
    if (r) {
      std::vector<Scalar> rs{*r};
      return MultBatch({a}, {b}, &rs)[0];
    }
    return MultBatch({a}, {b})[0];
  }
  Ciphertext makeL2(const Ciphertext& c) const {
    Ciphertext out{Bytes(E_), true};
    check(bgn_make_l2_batch(h_, 1, c.C.data(), out.C.data()), "bgn_make_l2_batch");
    return out;
  }

  // ---- bgn.go:375-497 ----
  Ciphertext Add(Ciphertext a, Ciphertext b, const Scalar* r = nullptr) const {
    align(a, b);
    return addsub(a, b, r, false);
  }
  Ciphertext Sub(Ciphertext a, Ciphertext b, const Scalar* r = nullptr) const {
    align(a, b);
    return addsub(a, b, r, true);
  }
  Ciphertext Neg(const Ciphertext& c) const { return Sub(encryptZero(), c); }     // bgn.go:436-438

  // ---- bgn.go:253-291 ----
  Ciphertext MultConst(const Ciphertext& c, const Scalar& k, const Scalar* r = nullptr) const {
    Ciphertext out{Bytes(E_), c.L2};
    check(bgn_multconst_batch(h_, 1, c.L2 ? 2 : 1, c.C.data(), k.data(), k.size(), r ? r->data() : nullptr,
                              r ? r->size() : 0, out.C.data()),
          "bgn_multconst_batch");
    return out;
  }

  // ---- bgn.go:195-201 ----
  void SetupDecryption(const SecretKey& sk) const;

  // ---- poly.go (coefficient vectors; digits are already-encoded plaintext coefficients) ----
  // EncryptPoly, poly.go:11-29: a negative digit is Sub(zero, Enc(|c|)); r[i] is the randomness of digit i.
  PolyCiphertext EncryptPoly(const std::vector<int64_t>& digits, const std::vector<Scalar>& r, int scale = 0) const {
    PolyCiphertext out;
    for (size_t i = 0; i < digits.size(); ++i) {
      const int64_t c = digits[i];
      Ciphertext e = EncryptWithRandomness(scalar_u64((uint64_t)(c < 0 ? -c : c)), r[i]);
      out.Coefficients.push_back(c < 0 ? Sub(encryptZero(), e) : e);
    }
    out.Degree = (int)digits.size();
    out.ScaleFactor = scale;
    return out;
  }
  // MultPoly, poly.go:123-156: one engine call for the d1*d2 pairings and the accumulation.
  PolyCiphertext MultPoly(const PolyCiphertext& a, const PolyCiphertext& b) const {
    Bytes A = join(a.Coefficients), B = join(b.Coefficients), out((size_t)(a.Degree + b.Degree) * E_);
    check(bgn_poly_mult_batch(h_, 1, (size_t)a.Degree, (size_t)b.Degree, A.data(), B.data(), out.data()),
          "bgn_poly_mult_batch");
    return PolyCiphertext{split(out, true), a.Degree + b.Degree, a.ScaleFactor + b.ScaleFactor, true};
  }
  // NegPoly, poly.go:45-55
  PolyCiphertext NegPoly(const PolyCiphertext& ct) const {
    PolyCiphertext out = ct;
    for (auto& c : out.Coefficients) c = Sub(ct.L2 ? makeL2(encryptZero()) : encryptZero(), c);
    return out;
  }
  // MultConstPoly, poly.go:71-120, on the encoded constant (digits, scale, sign): one engine call.
  PolyCiphertext MultConstPoly(const PolyCiphertext& ct, const std::vector<uint64_t>& digits, int scale = 0,
                               bool negative = false) const {
    std::vector<Scalar> ks;
    for (uint64_t d : digits) ks.push_back(scalar_u64(d));
    size_t kl = 0;
    Bytes A = join(ct.Coefficients), kb = pack(ks, kl), out((size_t)(ct.Degree + (int)digits.size()) * E_);
    check(bgn_poly_multconst_batch(h_, 1, (size_t)ct.Degree, digits.size(), ct.L2 ? 2 : 1, A.data(), kb.data(), kl, 0,
                                   out.data()),
          "bgn_poly_multconst_batch");
    PolyCiphertext prod{split(out, ct.L2), ct.Degree + (int)digits.size(), ct.ScaleFactor + scale, ct.L2};
    return negative ? NegPoly(prod) : prod;                                          // poly.go:115-119
  }
  PolyCiphertext MultConstPoly(const PolyCiphertext& ct, int64_t constant) const {
    return MultConstPoly(ct, UnbalancedEncode((uint64_t)(constant < 0 ? -constant : constant), PolyBase), 0,
                         constant < 0);
  }
  // MakePolyL2, poly.go:159-163: MultPoly(EncryptPoly(1), ct); r blinds the encryption of 1.
  PolyCiphertext MakePolyL2(const PolyCiphertext& ct, const Scalar& r) const {
    return MultPoly(EncryptPoly({1}, {r}), ct);
  }
  // AddPoly, poly.go:171-207 with alignPolyCiphertexts, poly.go:209-226.  r is used if a level-1 operand has
  // to be lifted (MakePolyL2).
  PolyCiphertext AddPoly(PolyCiphertext a, PolyCiphertext b, const Scalar& r = scalar_u64(1)) const {
    if (a.L2 && !b.L2) b = MakePolyL2(b, r);
    if (!a.L2 && b.L2) a = MakePolyL2(a, r);
    if (a.ScaleFactor < b.ScaleFactor) std::swap(a, b);
    if (a.ScaleFactor > b.ScaleFactor) {
      uint64_t f = 1;
      for (int i = 0; i < a.ScaleFactor - b.ScaleFactor; ++i) f *= FPScaleBase;
      b = MultConstPoly(b, UnbalancedEncode(f, PolyBase));
      b.ScaleFactor = a.ScaleFactor;
    }
    const int deg = a.Degree > b.Degree ? a.Degree : b.Degree, common = a.Degree < b.Degree ? a.Degree : b.Degree;
    std::vector<Ciphertext> ca(a.Coefficients.begin(), a.Coefficients.begin() + common),
        cb(b.Coefficients.begin(), b.Coefficients.begin() + common);
    PolyCiphertext out{AddBatch(ca, cb), deg, a.ScaleFactor, a.L2};
    const PolyCiphertext& longer = a.Degree > b.Degree ? a : b;
    out.Coefficients.insert(out.Coefficients.end(), longer.Coefficients.begin() + common, longer.Coefficients.end());
    return out;
  }
  PolyCiphertext SubPoly(const PolyCiphertext& a, const PolyCiphertext& b) const { return AddPoly(a, NegPoly(b)); }
  // EvalPoly, poly.go:58-68
  Ciphertext EvalPoly(const PolyCiphertext& ct) const {
    Bytes A = join(ct.Coefficients);
    Ciphertext out{Bytes(E_), ct.L2};
    check(bgn_poly_eval_batch(h_, 1, (size_t)ct.Degree, ct.L2 ? 2 : 1, A.data(), PolyBase, out.C.data()),
          "bgn_poly_eval_batch");
    return out;
  }

 private:
  friend class SecretKey;
  bgn_ctx* h_ = nullptr;
  size_t E_ = 0;

  void align(Ciphertext& a, Ciphertext& b) const {                                  // bgn.go:447-453
    if (a.L2 && !b.L2) b = makeL2(b);
    if (!a.L2 && b.L2) a = makeL2(a);
  }
  Ciphertext addsub(const Ciphertext& a, const Ciphertext& b, const Scalar* r, bool subtract) const {
    if (r) {
      std::vector<Scalar> rs{*r};
      return AddBatch({a}, {b}, &rs, subtract)[0];
    }
    return AddBatch({a}, {b}, nullptr, subtract)[0];
  }
  static Bytes pack(const std::vector<Scalar>& v, size_t& len) {
    len = 1;
    for (const auto& s : v) len = s.size() > len ? s.size() : len;
    Bytes out(v.size() * len, 0);
    for (size_t i = 0; i < v.size(); ++i)
      for (size_t j = 0; j < v[i].size(); ++j) out[i * len + (len - v[i].size()) + j] = v[i][j];
    return out;
  }
  Bytes join(const std::vector<Ciphertext>& v) const {
    Bytes out;
    out.reserve(v.size() * E_);
    for (const auto& c : v) out.insert(out.end(), c.C.begin(), c.C.end());
    return out;
  }
  std::vector<Ciphertext> split(const Bytes& b, bool l2) const {
    std::vector<Ciphertext> out(b.size() / E_);
    for (size_t i = 0; i < out.size(); ++i) {
      out[i].C.assign(b.begin() + i * E_, b.begin() + (i + 1) * E_);
      out[i].L2 = l2;
    }
    return out;
  }
};

// bgn.go:58-62
class SecretKey {
 public:
  Bytes Key;   // q1, big-endian
  explicit SecretKey(Bytes key) : Key(std::move(key)) {}

  // bgn.go:205-250 — returns (values, status) for a batch of one level
  std::pair<std::vector<int64_t>, std::vector<uint8_t>> DecryptBatch(const std::vector<Ciphertext>& cts,
                                                                      const PublicKey& pk) const {
    std::vector<int64_t> m(cts.size());
    std::vector<uint8_t> st(cts.size());
    if (cts.empty()) return {m, st};
    Bytes A = pk.join(cts);
    check(bgn_decrypt_batch(pk.h_, cts.size(), cts[0].L2 ? 2 : 1, A.data(), m.data(), st.data()), "bgn_decrypt_batch");
    return {m, st};
  }
  int64_t Decrypt(const Ciphertext& ct, const PublicKey& pk) const {
    auto r = DecryptBatch({ct}, pk);
    if (r.second[0] != BGN_DL_OK) throw DecryptError();
    return r.first[0];
  }
  int64_t DecryptFailSafe(const Ciphertext& ct, const PublicKey& pk) const {          // bgn.go:210-216
    auto r = DecryptBatch({ct}, pk);
    return r.second[0] == BGN_DL_OK ? r.first[0] : 0;
  }
  // DecryptPoly, poly.go:32-42 (per-coefficient errors are ignored there: a miss decrypts to 0)
  std::vector<int64_t> DecryptPoly(const PolyCiphertext& ct, const PublicKey& pk) const {
    return DecryptBatch(ct.Coefficients, pk).first;
  }
};

inline void PublicKey::SetupDecryption(const SecretKey& sk) const {
  check(bgn_ctx_set_secret(h_, sk.Key.data(), sk.Key.size()), "bgn_ctx_set_secret");
  check(bgn_ctx_setup_decryption(h_, MsgSpace), "bgn_ctx_setup_decryption");
}

}  // namespace bgn_amd
