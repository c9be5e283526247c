// C++ host-mirror test: the truth table of the reference's CLI (cmd/main.go:79-104) through
// include/bgn_amd.hpp.  Usage: truth_table <p_hex> <n_hex> <l> <P_hex> <Q_hex> <q1_hex> <msgspace>
// Exit code 0 = all checks passed, 3 = no GPU (context creation failed with BGN_E_HIP), 1 = mismatch.
#include <cstdio>
#include <cstdlib>
#include <string>

#include "bgn_amd.hpp"

using namespace bgn_amd;

static Bytes unhex(const std::string& s) {
  std::string t = s.size() % 2 ? "0" + s : s;
  Bytes b(t.size() / 2);
  for (size_t i = 0; i < b.size(); ++i) b[i] = (uint8_t)strtoul(t.substr(2 * i, 2).c_str(), nullptr, 16);
  return b;
}

int main(int argc, char** argv) {
  if (argc != 8) return 2;
  try {
    PublicKey pk(unhex(argv[1]), unhex(argv[2]), strtoull(argv[3], nullptr, 10), unhex(argv[4]), unhex(argv[5]),
                 strtoull(argv[7], nullptr, 10), true, 0);
    SecretKey sk(unhex(argv[6]));
    pk.SetupDecryption(sk);
    Ciphertext zero = pk.EncryptWithRandomness(scalar_u64(0), scalar_u64(12345));
    Ciphertext one = pk.EncryptWithRandomness(scalar_u64(1), scalar_u64(67890));
    auto D = [&](const Ciphertext& c) { return sk.DecryptFailSafe(c, pk); };
    int bad = 0;
    auto expect = [&](const char* what, int64_t got, int64_t want) {
      if (got != want) {
        printf("MISMATCH %s: got %lld want %lld\n", what, (long long)got, (long long)want);
        bad++;
      }
    };
    expect("0 + 0", D(pk.Add(zero, zero)), 0);
    expect("0 + 1", D(pk.Add(zero, one)), 1);
    expect("1 + 1", D(pk.Add(one, one)), 2);
    expect("0 * 1", D(pk.Mult(zero, one)), 0);
    expect("1 * 1", D(pk.Mult(one, one)), 1);
    expect("0 - 1", D(pk.Add(zero, pk.Neg(one))), -1);
    expect("1 - 1", D(pk.Add(one, pk.Neg(one))), 0);
    expect("1 * (-1)", D(pk.Mult(one, pk.Neg(one))), -1);
    expect("(-1) * (-1)", D(pk.Mult(pk.Neg(one), pk.Neg(one))), 1);
    expect("1*1 + 1 (mixed level)", D(pk.Add(pk.Mult(one, one), one)), 2);
    expect("3 * Enc(1) via MultConst", D(pk.MultConst(one, scalar_u64(3))), 3);
    // poly layer (poly_test.go:92-189): digits of 9 and -4 in balanced base 3
    auto ev = [&](const PolyCiphertext& c) {
      int64_t acc = 0;
      std::vector<int64_t> d = sk.DecryptPoly(c, pk);
      for (size_t i = d.size(); i-- > 0;) acc = acc * 3 + d[i];
      return acc;
    };
    std::vector<Scalar> rs{scalar_u64(11), scalar_u64(22), scalar_u64(33)};
    PolyCiphertext a = pk.EncryptPoly({0, 0, 1}, rs), b = pk.EncryptPoly({-1, -1}, rs);
    expect("poly 9 + (-4)", ev(pk.AddPoly(a, b)), 5);
    expect("poly 9 - (-4)", ev(pk.SubPoly(a, b)), 13);
    expect("poly 9 * (-4)", ev(pk.MultPoly(a, b)), -36);
    expect("poly 9 * const 6", ev(pk.MultConstPoly(a, 6)), 54);
    expect("poly (-4) * const -5", ev(pk.MultConstPoly(b, -5)), 20);
    expect("poly (9 * -4) * const 2 (level 2)", ev(pk.MultConstPoly(pk.MultPoly(a, b), 2)), -72);
    expect("poly 9*(-4) + 9 (mixed level)", ev(pk.AddPoly(pk.MultPoly(a, b), a, scalar_u64(44))), -27);
    expect("EvalPoly(9)", D(pk.EvalPoly(a)), 9);
    PolyCiphertext b2 = pk.EncryptPoly({-1, -1}, rs, 2);
    PolyCiphertext s2 = pk.AddPoly(a, b2);
    expect("poly 9 + (-4 / 3^2) at scale 2", ev(s2), 77);
    expect("scale factor after alignment", s2.ScaleFactor, 2);
    bool threw = false;
    try {
      sk.Decrypt(pk.EncryptWithRandomness(scalar_u64(5000), scalar_u64(1)), pk);
    } catch (const DecryptError&) {
      threw = true;
    }
    if (!threw) {
      printf("MISMATCH: out-of-range Decrypt did not raise\n");
      bad++;
    }
    printf(bad ? "FAILED\n" : "truth table ok\n");
    return bad ? 1 : 0;
  } catch (const Error& e) {
    printf("engine error %d: %s\n", e.code, e.what());
    return e.code == BGN_E_HIP ? 3 : 1;
  }
}
