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

class SecretKey;

// bgn.go:28-41 (hot-path members)
class PublicKey {
 public:
  Bytes N;
  uint64_t MsgSpace;
  bool Deterministic;

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
};

inline void PublicKey::SetupDecryption(const SecretKey& sk) const {
  check(bgn_ctx_set_secret(h_, sk.Key.data(), sk.Key.size()), "bgn_ctx_set_secret");
  check(bgn_ctx_setup_decryption(h_, MsgSpace), "bgn_ctx_setup_decryption");
}

}  // namespace bgn_amd
