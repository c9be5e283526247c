// hostbig.hpp — minimal unsigned big integer for one-off parameter setup on the
// host (Montgomery constants, NAF of n).  Not on the hot path; the engine has
// no GMP dependency.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <vector>
#include <algorithm>

namespace bgn {

struct BigU {
  std::vector<uint32_t> w;   // little-endian base 2^32, no trailing zero words

  BigU() {}
  explicit BigU(uint64_t v) {
    while (v) {
      w.push_back((uint32_t)v);
      v >>= 32;
    }
  }
  static BigU from_be(const uint8_t* b, size_t len) {
    BigU r;
    r.w.assign((len + 3) / 4, 0);
    for (size_t i = 0; i < len; ++i) {
      const size_t le = len - 1 - i;
      r.w[le / 4] |= (uint32_t)b[i] << (8 * (le % 4));
    }
    r.trim();
    return r;
  }
  void trim() {
    while (!w.empty() && w.back() == 0) w.pop_back();
  }
  bool is_zero() const { return w.empty(); }
  // overwrite the words before releasing them (secret keys)
  void wipe() {
    volatile uint32_t* v = w.data();
    for (size_t i = 0; i < w.size(); ++i) v[i] = 0;
    w.clear();
  }
  int bits() const {
    if (w.empty()) return 0;
    uint32_t t = w.back();
    int b = 0;
    while (t) {
      ++b;
      t >>= 1;
    }
    return (int)(w.size() - 1) * 32 + b;
  }
  bool bit(int i) const {
    const size_t k = (size_t)i / 32;
    if (k >= w.size()) return false;
    return (w[k] >> (i % 32)) & 1u;
  }
  static int cmp(const BigU& a, const BigU& b) {
    if (a.w.size() != b.w.size()) return a.w.size() < b.w.size() ? -1 : 1;
    for (size_t i = a.w.size(); i-- > 0;) {
      if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1;
    }
    return 0;
  }
  void add(const BigU& b) {
    const size_t n = std::max(w.size(), b.w.size());
    w.resize(n, 0);
    uint64_t c = 0;
    for (size_t i = 0; i < n; ++i) {
      c += (uint64_t)w[i] + (i < b.w.size() ? b.w[i] : 0);
      w[i] = (uint32_t)c;
      c >>= 32;
    }
    if (c) w.push_back((uint32_t)c);
  }
  void add_small(uint32_t v) { add(BigU((uint64_t)v)); }
  // this -= b, requires this >= b
  void sub(const BigU& b) {
    int64_t c = 0;
    for (size_t i = 0; i < w.size(); ++i) {
      int64_t d = (int64_t)w[i] - (i < b.w.size() ? (int64_t)b.w[i] : 0) + c;
      if (d < 0) {
        d += ((int64_t)1 << 32);
        c = -1;
      } else {
        c = 0;
      }
      w[i] = (uint32_t)d;
    }
    trim();
  }
  void shl1() {
    uint32_t c = 0;
    for (size_t i = 0; i < w.size(); ++i) {
      const uint32_t n = w[i] >> 31;
      w[i] = (w[i] << 1) | c;
      c = n;
    }
    if (c) w.push_back(c);
  }
  void shr1() {
    uint32_t c = 0;
    for (size_t i = w.size(); i-- > 0;) {
      const uint32_t n = w[i] & 1u;
      w[i] = (w[i] >> 1) | (c << 31);
      c = n;
    }
    trim();
  }
  static BigU mul_u64(const BigU& a, uint64_t m) {
    BigU r;
    if (a.w.empty() || m == 0) return r;
    r.w.assign(a.w.size() + 2, 0);
    const uint32_t m0 = (uint32_t)m, m1 = (uint32_t)(m >> 32);
    uint64_t c = 0;
    for (size_t i = 0; i < a.w.size(); ++i) {
      c += (uint64_t)a.w[i] * m0 + r.w[i];
      r.w[i] = (uint32_t)c;
      c >>= 32;
    }
    r.w[a.w.size()] = (uint32_t)c;
    c = 0;
    for (size_t i = 0; i < a.w.size(); ++i) {
      c += (uint64_t)a.w[i] * m1 + r.w[i + 1];
      r.w[i + 1] = (uint32_t)c;
      c >>= 32;
    }
    r.w[a.w.size() + 1] += (uint32_t)c;
    r.trim();
    return r;
  }
  // q = a / b, r = a mod b by shift-and-subtract (setup only: a few thousand word passes); b != 0
  static void divmod(const BigU& a, const BigU& b, BigU& q, BigU& r) {
    q = BigU();
    r = BigU();
    const int nb = a.bits();
    q.w.assign((size_t)(nb + 31) / 32, 0);
    for (int i = nb - 1; i >= 0; --i) {
      r.shl1();
      if (a.bit(i)) {
        if (r.w.empty()) r.w.push_back(0);
        r.w[0] |= 1u;
      }
      if (cmp(r, b) >= 0) {
        r.sub(b);
        q.w[(size_t)i / 32] |= 1u << (i % 32);
      }
    }
    q.trim();
  }
  // limbs of `limb_bits` bits, little-endian, zero padded to nl
  void to_limbs(uint32_t* out, int nl, int limb_bits) const {
    for (int j = 0; j < nl; ++j) {
      uint32_t v = 0;
      for (int b = 0; b < limb_bits; ++b)
        if (bit(limb_bits * j + b)) v |= 1u << b;
      out[j] = v;
    }
  }
  // Width-w non-adjacent form (w = 3: digits 0, +-1, +-3), little-endian; the top digit is positive.
  std::vector<signed char> wnaf(int w) const {
    std::vector<signed char> d;
    BigU n = *this;
    const int full = 1 << w, half = 1 << (w - 1);
    while (!n.is_zero()) {
      if (n.w[0] & 1u) {
        int z = (int)(n.w[0] & (uint32_t)(full - 1));
        if (z >= half) z -= full;
        d.push_back((signed char)z);
        if (z > 0)
          n.sub(BigU((uint64_t)z));
        else
          n.add_small((uint32_t)(-z));
      } else {
        d.push_back(0);
      }
      n.shr1();
    }
    return d;
  }
  // Non-adjacent form, little-endian digits in {-1,0,1}
  std::vector<signed char> naf() const {
    std::vector<signed char> d;
    BigU n = *this;
    while (!n.is_zero()) {
      if (n.w[0] & 1u) {
        const int z = 2 - (int)(n.w[0] & 3u);   // +1 or -1
        d.push_back((signed char)z);
        if (z > 0)
          n.sub(BigU((uint64_t)1));
        else
          n.add_small(1);
      } else {
        d.push_back(0);
      }
      n.shr1();
    }
    return d;
  }
};

}  // namespace bgn
