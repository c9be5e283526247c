// bgn_amd.hpp — header-only C++ mirror of the reference's Go API for the hot path, over the C ABI
// (bgn_amd.h).  The reference is compiled Go and its toolchain is absent from the build image, so this
// is the compiled-language host side: same type and method names as sachaservan/bgn (bgn.go,
// ciphertext.go), same argument meaning and error behaviour.  Single-element methods are the count-1
// case of the batch entry points.  Scalars are unsigned big-endian byte strings (`Scalar`), with
// helpers for 64-bit values; randomness is always supplied by the caller.
#pragma once
#include <cstdint>
#include <cstring>
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

// ---- Go encoding/gob streams of the two ciphertext envelopes (ciphertext.go:17-20, :33-38) --------------
// Format notes and the reader's rules (type ids and field order come from the definitions in the stream) are
// in bgn_amd/gob.py; this is the same codec for compiled callers.
namespace gob {
struct Error : std::runtime_error {
  explicit Error(const std::string& m) : std::runtime_error(m) {}
};
inline void put_uint(Bytes& o, uint64_t v) {
  if (v < 128) {
    o.push_back((uint8_t)v);
    return;
  }
  int n = 0;
  for (uint64_t t = v; t; t >>= 8) ++n;
  o.push_back((uint8_t)(256 - n));
  for (int i = n - 1; i >= 0; --i) o.push_back((uint8_t)(v >> (8 * i)));
}
inline void put_int(Bytes& o, int64_t v) { put_uint(o, v < 0 ? (((uint64_t)~v) << 1) | 1 : ((uint64_t)v) << 1); }
inline void put_bytes(Bytes& o, const uint8_t* p, size_t n) {
  put_uint(o, n);
  o.insert(o.end(), p, p + n);
}
inline void put_str(Bytes& o, const char* s) { put_bytes(o, (const uint8_t*)s, strlen(s)); }
inline void put_message(Bytes& o, const Bytes& body) {
  put_uint(o, body.size());
  o.insert(o.end(), body.begin(), body.end());
}
// wireType{StructT: &structType{CommonType{name, id}, fields}} / wireType{SliceT: &sliceType{CommonType, elem}}
inline Bytes struct_def(int64_t id, const char* name, const std::vector<std::pair<const char*, int64_t>>& fields) {
  Bytes b;
  put_int(b, -id);
  b.push_back(3); b.push_back(1);
  b.push_back(1); put_str(b, name); b.push_back(1); put_int(b, id); b.push_back(0);
  if (!fields.empty()) {
    b.push_back(1);
    put_uint(b, fields.size());
    for (auto& f : fields) {
      b.push_back(1); put_str(b, f.first); b.push_back(1); put_int(b, f.second); b.push_back(0);
    }
  }
  b.push_back(0); b.push_back(0);
  return b;
}
inline Bytes slice_def(int64_t id, const char* name, int64_t elem) {
  Bytes b;
  put_int(b, -id);
  b.push_back(2); b.push_back(1);
  b.push_back(1); put_str(b, name); b.push_back(1); put_int(b, id); b.push_back(0);
  b.push_back(1); put_int(b, elem);
  b.push_back(0); b.push_back(0);
  return b;
}

struct Reader {
  const uint8_t* p;
  const uint8_t* e;
  const uint8_t* take(size_t n) {
    if ((size_t)(e - p) < n) throw Error("truncated gob stream");
    const uint8_t* r = p;
    p += n;
    return r;
  }
  uint64_t uint_() {
    uint8_t b = *take(1);
    if (b < 128) return b;
    size_t n = 256 - b;
    if (n > 8) throw Error("unsigned integer wider than 64 bits");
    const uint8_t* q = take(n);
    uint64_t v = 0;
    for (size_t i = 0; i < n; ++i) v = (v << 8) | q[i];
    return v;
  }
  int64_t int_() {
    uint64_t u = uint_();
    return (u & 1) ? (int64_t)~(u >> 1) : (int64_t)(u >> 1);
  }
  Bytes bytes_() {
    size_t n = (size_t)uint_();
    const uint8_t* q = take(n);
    return Bytes(q, q + n);
  }
  bool done() const { return p >= e; }
};
// A user type read from the stream: a struct (field names and type ids) or a slice (element type id).
struct Type {
  bool is_struct = false;
  std::string name;
  std::vector<std::pair<std::string, int64_t>> fields;
  int64_t elem = 0;
};
inline std::pair<std::string, int64_t> read_common(Reader& r) {   // CommonType and fieldType: {Name string; Id typeId}
  std::string name;
  int64_t id = 0;
  int field = -1;
  for (;;) {
    uint64_t d = r.uint_();
    if (!d) return {name, id};
    field += (int)d;
    if (field == 0) {
      Bytes b = r.bytes_();
      name.assign(b.begin(), b.end());
    } else if (field == 1) {
      id = r.int_();
    } else {
      throw Error("unknown CommonType field");
    }
  }
}
inline Type read_wire_type(Reader& r) {
  Type t;
  bool have = false;
  int field = -1;
  for (;;) {
    uint64_t d = r.uint_();
    if (!d) break;
    field += (int)d;
    if (field != 1 && field != 2) throw Error("gob type kind not used by the ciphertext envelopes");
    t.is_struct = field == 2;
    have = true;
    int f = -1;
    for (;;) {
      uint64_t dd = r.uint_();
      if (!dd) break;
      f += (int)dd;
      if (f == 0) {
        t.name = read_common(r).first;
      } else if (f == 1 && t.is_struct) {
        uint64_t n = r.uint_();
        for (uint64_t i = 0; i < n; ++i) t.fields.push_back(read_common(r));
      } else if (f == 1) {
        t.elem = r.int_();
      } else {
        throw Error("unknown type field");
      }
    }
  }
  if (!have) throw Error("empty type definition");
  return t;
}
// Decoded top-level struct of the envelopes: whichever of these fields the stream carries.
struct Envelope {
  Bytes CBytes;
  std::vector<Bytes> CoeffBytes;
  int64_t Degree = 0, ScaleFactor = 0;
  bool L2 = false;
};
inline Envelope decode(const Bytes& data) {
  if (data.empty()) throw Error("no data provided");   // bgn.go:503-505
  Reader r{data.data(), data.data() + data.size()};
  std::vector<std::pair<int64_t, Type>> types;
  auto find = [&](int64_t id) -> const Type* {
    for (auto& t : types)
      if (t.first == id) return &t.second;
    return nullptr;
  };
  while (!r.done()) {
    size_t n = (size_t)r.uint_();
    const uint8_t* q = r.take(n);
    Reader m{q, q + n};
    int64_t id = m.int_();
    if (id < 0) {
      types.push_back({-id, read_wire_type(m)});
      continue;
    }
    const Type* t = find(id);
    if (!t || !t->is_struct) throw Error("top-level value is not a struct");
    Envelope out;
    int field = -1;
    for (;;) {
      uint64_t d = m.uint_();
      if (!d) return out;
      field += (int)d;
      if (field >= (int)t->fields.size()) throw Error("field number out of range");
      const std::string& fn = t->fields[field].first;
      const int64_t ft = t->fields[field].second;
      if (ft == 1) {
        bool v = m.uint_() != 0;
        if (fn == "L2") out.L2 = v;
      } else if (ft == 2) {
        int64_t v = m.int_();
        if (fn == "Degree") out.Degree = v;
        if (fn == "ScaleFactor") out.ScaleFactor = v;
      } else if (ft == 3) {
        (void)m.uint_();
      } else if (ft == 5 || ft == 6) {
        Bytes v = m.bytes_();
        if (fn == "CBytes") out.CBytes = v;
      } else {
        const Type* st = find(ft);
        if (!st || st->is_struct || st->elem != 5) throw Error("unsupported field type");
        uint64_t cnt = m.uint_();
        std::vector<Bytes> v;
        for (uint64_t i = 0; i < cnt; ++i) v.push_back(m.bytes_());
        if (fn == "CoeffBytes") out.CoeffBytes = v;
      }
    }
  }
  throw Error("gob stream holds no value");
}
// Ciphertext.Bytes(), ciphertext.go:76-92
inline Bytes marshal_ciphertext(const Bytes& c, bool l2) {
  Bytes out;
  put_message(out, struct_def(65, "ciphertextWrapper", {{"CBytes", 5}, {"L2", 1}}));
  Bytes v;
  put_int(v, 65);
  if (!c.empty()) { v.push_back(1); put_bytes(v, c.data(), c.size()); }
  if (l2) { v.push_back(c.empty() ? 2 : 1); v.push_back(1); }
  v.push_back(0);
  put_message(out, v);
  return out;
}
// PolyCiphertext.Bytes(), ciphertext.go:94-116
inline Bytes marshal_poly_ciphertext(const std::vector<Bytes>& coeffs, int64_t degree, int64_t scale, bool l2) {
  Bytes out;
  put_message(out, struct_def(65, "polyCiphertextWrapper", {{"CoeffBytes", 66}, {"Degree", 2}, {"ScaleFactor", 2}, {"L2", 1}}));
  put_message(out, slice_def(66, "[][]uint8", 5));
  Bytes v;
  put_int(v, 65);
  int last = -1;
  auto delta = [&](int field) {
    put_uint(v, (uint64_t)(field - last));
    last = field;
  };
  if (!coeffs.empty()) {
    delta(0);
    put_uint(v, coeffs.size());
    for (auto& c : coeffs) put_bytes(v, c.data(), c.size());
  }
  if (degree) { delta(1); put_int(v, degree); }
  if (scale) { delta(2); put_int(v, scale); }
  if (l2) { delta(3); v.push_back(1); }
  v.push_back(0);
  put_message(out, v);
  return out;
}
}  // namespace gob

// sha256 (FIPS 180-4) for the challenge of gadgets.go:80-96
namespace sha256 {
inline Bytes digest(const Bytes& msg) {
  static const uint32_t K[64] = {
      0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
      0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
      0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
      0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
      0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
      0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
      0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  Bytes m = msg;
  const uint64_t bits = (uint64_t)msg.size() * 8;
  m.push_back(0x80);
  while (m.size() % 64 != 56) m.push_back(0);
  for (int i = 7; i >= 0; --i) m.push_back((uint8_t)(bits >> (8 * i)));
  auto rotr = [](uint32_t x, int n) { return (x >> n) | (x << (32 - n)); };
  for (size_t off = 0; off < m.size(); off += 64) {
    uint32_t w[64];
    for (int i = 0; i < 16; ++i)
      w[i] = ((uint32_t)m[off + 4 * i] << 24) | ((uint32_t)m[off + 4 * i + 1] << 16) | ((uint32_t)m[off + 4 * i + 2] << 8) | m[off + 4 * i + 3];
    for (int i = 16; i < 64; ++i) {
      const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
      const uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; ++i) {
      const uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g);
      const uint32_t t1 = hh + S1 + ch + K[i] + w[i];
      const uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), mj = (a & b) ^ (a & c) ^ (b & c);
      const uint32_t t2 = S0 + mj;
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
  Bytes out(32);
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 4; ++j) out[4 * i + j] = (uint8_t)(h[i] >> (24 - 8 * j));
  return out;
}
}  // namespace sha256

// ciphertext.go:12-15 — C holds Element.Bytes(); the G1 identity is 2L zero bytes.
struct Ciphertext {
  Bytes C;
  bool L2 = false;
  Ciphertext Copy() const { return *this; }
  Bytes Bytes_() const { return gob::marshal_ciphertext(C, L2); }   // Ciphertext.Bytes(), ciphertext.go:76-92
};

// gadgets.go:10-21
struct ProofOfPlaintextKnowledge {
  Ciphertext Ct, Nonce;
  Scalar DL;
};
struct DecryptionProof {
  Scalar Value, Randomness;
};

// ciphertext.go:26-31
struct PolyCiphertext {
  std::vector<Ciphertext> Coefficients;
  int Degree = 0;
  int ScaleFactor = 0;
  bool L2 = false;
  PolyCiphertext Copy() const { return *this; }
  Bytes Bytes_() const {                                              // PolyCiphertext.Bytes(), ciphertext.go:94-116
    std::vector<Bytes> cb;
    for (auto& c : Coefficients) cb.push_back(c.C);
    return gob::marshal_poly_ciphertext(cb, Degree, ScaleFactor, L2);
  }
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
class PublicKey;

// An array of wire elements (or scalars, plaintexts, status bytes) that stays on the key's device between calls — the
// C++ twin of go/bgn_amd.go's DeviceArray over bgn_dev_alloc / _free / _upload / _download (include/bgn_amd.h), for
// chains like MultPoly -> AddPoly -> Decrypt (poly.go:123-207 -> bgn.go:205) without a host round trip per step.
// Owned by the caller (not counted in the context's memory, not subject to its budget); movable, not copyable.
class DeviceArray {
 public:
  size_t Count = 0, Width = 0;   // elements, bytes per element
  bool L2 = false;
  DeviceArray() = default;
  DeviceArray(bgn_ctx* h, size_t count, size_t width, bool l2) : Count(count), Width(width), L2(l2), h_(h) {
    if (count * width == 0) throw Error(BGN_E_ARG, "empty device array");
    p_ = bgn_dev_alloc(h, count * width);
    if (!p_) throw Error(BGN_E_NOMEM, "bgn_dev_alloc");
  }
  ~DeviceArray() { reset(); }
  DeviceArray(const DeviceArray&) = delete;
  DeviceArray& operator=(const DeviceArray&) = delete;
  DeviceArray(DeviceArray&& o) noexcept { *this = std::move(o); }
  DeviceArray& operator=(DeviceArray&& o) noexcept {
    if (this != &o) {
      reset();
      Count = o.Count; Width = o.Width; L2 = o.L2; h_ = o.h_; p_ = o.p_;
      o.p_ = nullptr;
    }
    return *this;
  }
  uint8_t* data() const { return static_cast<uint8_t*>(p_); }
  void UploadBytes(const Bytes& b) const {
    if (b.size() != Count * Width) throw Error(BGN_E_ARG, "upload size does not match the device array");
    check(bgn_dev_upload(h_, p_, b.data(), b.size()), "bgn_dev_upload");
  }
  // after every call issued on this key before it (the copies run on the null stream, as the `_dev` calls below)
  Bytes DownloadBytes() const {
    Bytes b(Count * Width);
    check(bgn_dev_download(h_, b.data(), p_, b.size()), "bgn_dev_download");
    return b;
  }

 private:
  void reset() {
    if (p_) bgn_dev_free(h_, p_);
    p_ = nullptr;
  }
  bgn_ctx* h_ = nullptr;
  void* p_ = nullptr;
};

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

  // ---- the same over arrays that stay on the device (bgn_*_batch_dev on the null stream) ----
  DeviceArray Upload(const std::vector<Ciphertext>& cts) const {
    DeviceArray d(h_, cts.size(), E_, !cts.empty() && cts[0].L2);
    d.UploadBytes(join(cts));
    return d;
  }
  std::vector<Ciphertext> Download(const DeviceArray& d) const { return split(d.DownloadBytes(), d.L2); }
  DeviceArray EncryptBatchDev(const std::vector<Scalar>& x, const std::vector<Scalar>* r) const {
    size_t xl = 0, rl = 0;
    DeviceArray dx = scalars_dev(x, xl), dr, out(h_, x.size(), E_, false);
    if (r) dr = scalars_dev(*r, rl);
    check(bgn_encrypt_batch_dev(h_, x.size(), dx.data(), xl, r ? dr.data() : nullptr, rl, out.data(), nullptr),
          "bgn_encrypt_batch_dev");
    return out;
  }
  // operands of one level (lift with MakeL2BatchDev first where the reference would, bgn.go:444-452); the result may
  // be written over an operand with the *Into forms (accumulating, poly.go:171-207)
  DeviceArray AddBatchDev(const DeviceArray& a, const DeviceArray& b, const std::vector<Scalar>* r = nullptr,
                          bool subtract = false) const {
    DeviceArray out(h_, a.Count, E_, a.L2);
    AddBatchDevInto(out, a, b, r, subtract);
    return out;
  }
  void AddBatchDevInto(const DeviceArray& out, const DeviceArray& a, const DeviceArray& b,
                       const std::vector<Scalar>* r = nullptr, bool subtract = false) const {
    if (a.Count != b.Count || a.L2 != b.L2 || out.Count != a.Count) throw Error(BGN_E_ARG, "operand arrays differ in count or level");
    size_t rl = 0;
    DeviceArray dr;
    if (r) dr = scalars_dev(*r, rl);
    check((subtract ? bgn_sub_batch_dev : bgn_add_batch_dev)(h_, a.Count, a.L2 ? 2 : 1, a.data(), b.data(),
                                                             r ? dr.data() : nullptr, rl, out.data(), nullptr),
          subtract ? "bgn_sub_batch_dev" : "bgn_add_batch_dev");
  }
  DeviceArray NegBatchDev(const DeviceArray& a) const {
    DeviceArray out(h_, a.Count, E_, a.L2);
    check(bgn_neg_batch_dev(h_, a.Count, a.L2 ? 2 : 1, a.data(), out.data(), nullptr), "bgn_neg_batch_dev");
    return out;
  }
  DeviceArray MultBatchDev(const DeviceArray& a, const DeviceArray& b, const std::vector<Scalar>* r = nullptr) const {
    if (a.Count != b.Count || a.L2 || b.L2) throw Error(BGN_E_ARG, "Mult takes level-1 arrays of one length");
    size_t rl = 0;
    DeviceArray dr, out(h_, a.Count, E_, true);
    if (r) dr = scalars_dev(*r, rl);
    check(bgn_mult_batch_dev(h_, a.Count, a.data(), b.data(), r ? dr.data() : nullptr, rl, out.data(), nullptr),
          "bgn_mult_batch_dev");
    return out;
  }
  DeviceArray MakeL2BatchDev(const DeviceArray& a) const {
    DeviceArray out(h_, a.Count, E_, true);
    check(bgn_make_l2_batch_dev(h_, a.Count, a.data(), out.data(), nullptr), "bgn_make_l2_batch_dev");
    return out;
  }
  DeviceArray MultConstBatchDev(const DeviceArray& a, const std::vector<Scalar>& k, const std::vector<Scalar>* r = nullptr) const {
    if (k.size() != a.Count) throw Error(BGN_E_ARG, "one scalar per element");
    size_t kl = 0, rl = 0;
    DeviceArray dk = scalars_dev(k, kl), dr, out(h_, a.Count, E_, a.L2);
    if (r) dr = scalars_dev(*r, rl);
    check(bgn_multconst_batch_dev(h_, a.Count, a.L2 ? 2 : 1, a.data(), dk.data(), kl, r ? dr.data() : nullptr, rl,
                                  out.data(), nullptr),
          "bgn_multconst_batch_dev");
    return out;
  }
  // npoly products of d1 x d2 level-1 coefficient vectors (poly.go:123-156), a and b polynomial after polynomial:
  // npoly * (d1 + d2) level-2 coefficients
  DeviceArray MultPolyBatchDev(size_t npoly, size_t d1, size_t d2, const DeviceArray& a, const DeviceArray& b) const {
    if (a.Count != npoly * d1 || b.Count != npoly * d2 || a.L2 || b.L2) throw Error(BGN_E_ARG, "coefficient arrays do not match npoly*d1 / npoly*d2");
    DeviceArray out(h_, npoly * (d1 + d2), E_, true);
    check(bgn_poly_mult_batch_dev(h_, npoly, d1, d2, a.data(), b.data(), out.data(), nullptr), "bgn_poly_mult_batch_dev");
    return out;
  }
  std::vector<uint8_t> ValidateBatchDev(const DeviceArray& a) const {
    DeviceArray ok(h_, a.Count, 1, false);
    check(bgn_validate_batch_dev(h_, a.Count, a.L2 ? 2 : 1, a.data(), ok.data(), nullptr), "bgn_validate_batch_dev");
    return ok.DownloadBytes();
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

  // Device memory of this key's tables and workspace (the reference keeps the tables of every key in package
  // globals, gsbs.go:12-15): a cap for the tables built from now on (0: none), and what the context holds.
  void SetMemoryBudget(uint64_t bytes) const { check(bgn_ctx_set_memory_budget(h_, bytes), "bgn_ctx_set_memory_budget"); }
  uint64_t MemoryBytes() const { return bgn_ctx_memory_bytes(h_); }
  // per-context options (include/bgn_amd.h "Options"): kernel crossovers, table shapes, the combiner's limits
  void SetOption(const char* name, int64_t value) const { check(bgn_ctx_set_option(h_, name, value), "bgn_ctx_set_option"); }
  int64_t GetOption(const char* name) const {
    int64_t v = 0;
    check(bgn_ctx_get_option(h_, name, &v), "bgn_ctx_get_option");
    return v;
  }
  // crossovers between the kernel families re-derived from timed probes on this device (after SetupDecryption, so
  // that Decrypt's probes walk the production table)
  void Calibrate() const { check(bgn_ctx_calibrate(h_, nullptr), "bgn_ctx_calibrate"); }

  // ---- wire envelopes: bgn.go:501-560 ----
  Ciphertext NewCiphertextFromBytes(const Bytes& data) const {
    gob::Envelope w = gob::decode(data);
    return Ciphertext{element(w.CBytes), w.L2};
  }
  PolyCiphertext NewPolyCiphertextFromBytes(const Bytes& data) const {
    gob::Envelope w = gob::decode(data);
    PolyCiphertext out;
    for (auto& b : w.CoeffBytes) out.Coefficients.push_back(Ciphertext{element(b), w.L2});
    out.Degree = (int)w.Degree;
    out.ScaleFactor = (int)w.ScaleFactor;
    out.L2 = w.L2;
    return out;
  }

  // ---- untrusted inputs: 1 per element that is a valid encoding (bgn_validate_batch) ----
  std::vector<uint8_t> Validate(const std::vector<Ciphertext>& cts) const {
    std::vector<uint8_t> ok(cts.size());
    if (cts.empty()) return ok;
    Bytes A = join(cts);
    check(bgn_validate_batch(h_, cts.size(), cts[0].L2 ? 2 : 1, A.data(), ok.data()), "bgn_validate_batch");
    return ok;
  }

  // ---- gadgets.go:57-77 (verification; proof generation needs big-integer arithmetic and stays with the caller) ----
  std::vector<uint8_t> CheckDecryptionProofBatch(const std::vector<Ciphertext>& cts,
                                                 const std::vector<DecryptionProof>& proofs) const {
    std::vector<uint8_t> ok(cts.size());
    if (cts.empty()) return ok;
    std::vector<Scalar> v, r;
    for (auto& pr : proofs) {
      v.push_back(pr.Value);
      r.push_back(pr.Randomness);
    }
    size_t vl = 0, rl = 0;
    Bytes A = join(cts), vb = pack(v, vl), rb = pack(r, rl);
    check(bgn_check_decryption_proof_batch(h_, cts.size(), A.data(), vb.data(), vl, rb.data(), rl, ok.data()),
          "bgn_check_decryption_proof_batch");
    return ok;
  }
  bool CheckDecryptionProof(const Ciphertext& ct, const DecryptionProof& proof) const {
    return CheckDecryptionProofBatch({ct}, {proof})[0] != 0;
  }
  // hash(), gadgets.go:80-96
  static Scalar ProofHash(const ProofOfPlaintextKnowledge& proof) {
    Bytes m = proof.Ct.C;
    m.insert(m.end(), proof.Nonce.C.begin(), proof.Nonce.C.end());
    return sha256::digest(m);
  }
  std::vector<uint8_t> CheckProofOfPlaintextKnoewledgeBatch(const std::vector<Ciphertext>& cts,
                                                            const std::vector<ProofOfPlaintextKnowledge>& proofs) const {
    std::vector<uint8_t> ok(cts.size());
    if (cts.empty()) return ok;
    std::vector<Scalar> c, dl;
    std::vector<Ciphertext> nonces;
    for (auto& pr : proofs) {
      c.push_back(ProofHash(pr));
      dl.push_back(pr.DL);
      nonces.push_back(pr.Nonce);
    }
    size_t cl = 0, dll = 0;
    Bytes A = join(cts), N = join(nonces), cb = pack(c, cl), db = pack(dl, dll);
    check(bgn_check_plaintext_knowledge_batch(h_, cts.size(), A.data(), N.data(), cb.data(), cl, db.data(), dll, ok.data()),
          "bgn_check_plaintext_knowledge_batch");
    return ok;
  }
  bool CheckProofOfPlaintextKnoewledge(const Ciphertext& ct, const ProofOfPlaintextKnowledge& proof) const {
    return CheckProofOfPlaintextKnoewledgeBatch({ct}, {proof})[0] != 0;       // name as in the reference
  }

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
  // r (one value per output coefficient, a.Degree + b.Degree of them) blinds the result for keys with
  // Deterministic == false: the reference's per-step blinding (bgn.go:302-311, :466-474) multiplies every output
  // coefficient by e(Q,Q) to a sum of fresh random exponents, i.e. by one uniformly random power.
  PolyCiphertext MultPoly(const PolyCiphertext& a, const PolyCiphertext& b, const std::vector<Scalar>* r = nullptr) const {
    const size_t deg = (size_t)(a.Degree + b.Degree);
    Bytes A = join(a.Coefficients), B = join(b.Coefficients), out(deg * E_);
    check(bgn_poly_mult_batch(h_, 1, (size_t)a.Degree, (size_t)b.Degree, A.data(), B.data(), out.data()),
          "bgn_poly_mult_batch");
    if (r) {
      size_t rl = 0;
      Bytes rb = pack(*r, rl), one(deg * E_, 0), blinded(deg * E_);
      for (size_t i = 0; i < deg; ++i) one[i * E_ + E_ / 2 - 1] = 1;           // the GT identity: re = 1, im = 0
      check(bgn_add_batch(h_, deg, 2, out.data(), one.data(), rb.data(), rl, blinded.data()), "bgn_add_batch");
      out = blinded;
    }
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

  Bytes element(const Bytes& b) const {          // gob drops empty slices: the all-zero identity
    if (b.empty()) return Bytes(E_, 0);
    if (b.size() != E_) throw gob::Error("element of unexpected length");
    return b;
  }
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
  DeviceArray scalars_dev(const std::vector<Scalar>& v, size_t& len) const {
    Bytes b = pack(v, len);
    DeviceArray d(h_, v.size(), len, false);
    d.UploadBytes(b);
    return d;
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
  // the same for an array on the device: plaintexts and statuses come back to the host (8 + 1 bytes per element)
  std::pair<std::vector<int64_t>, std::vector<uint8_t>> DecryptBatchDev(const DeviceArray& cts, const PublicKey& pk) const {
    DeviceArray dm(pk.h_, cts.Count, 8, false), ds(pk.h_, cts.Count, 1, false);
    check(bgn_decrypt_batch_dev(pk.h_, cts.Count, cts.L2 ? 2 : 1, cts.data(), reinterpret_cast<int64_t*>(dm.data()), ds.data(),
                                nullptr),
          "bgn_decrypt_batch_dev");
    Bytes mb = dm.DownloadBytes();
    std::vector<int64_t> m(cts.Count);
    std::memcpy(m.data(), mb.data(), mb.size());
    return {m, ds.DownloadBytes()};
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
