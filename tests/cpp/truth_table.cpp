// C++ host-mirror test: the truth table of the reference's CLI (cmd/main.go:79-104) through
// include/bgn_amd.hpp.  Usage: truth_table <p_hex> <n_hex> <l> <P_hex> <Q_hex> <q1_hex> <msgspace>
// Exit code 0 = all checks passed, 3 = no GPU (context creation failed with BGN_E_HIP), 1 = mismatch.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "bgn_amd.hpp"

using namespace bgn_amd;

static Bytes unhex(const std::string& s) {
  std::string t = s.size() % 2 ? "0" + s : s;
  Bytes b(t.size() / 2);
  for (size_t i = 0; i < b.size(); ++i) b[i] = (uint8_t)strtoul(t.substr(2 * i, 2).c_str(), nullptr, 16);
  return b;
}

// Host-side format code, checked before any GPU is touched: the gob streams must equal the ones of
// bgn_amd/gob.py byte for byte (both follow encoding/gob's documented layout), and decode back.
static int host_checks() {
  int bad = 0;
  const uint8_t ct_stream[] = {0x31, 0xff, 0x81, 0x03, 0x01, 0x01, 0x11, 0x63, 0x69, 0x70, 0x68, 0x65, 0x72, 0x74, 0x65, 0x78, 0x74, 0x57, 0x72, 0x61, 0x70, 0x70, 0x65, 0x72, 0x01, 0xff, 0x82, 0x00, 0x01, 0x02, 0x01, 0x06, 0x43, 0x42, 0x79, 0x74, 0x65, 0x73, 0x01, 0x0a, 0x00, 0x01, 0x02, 0x4c, 0x32, 0x01, 0x02, 0x00, 0x00, 0x00, 0x0a, 0xff, 0x82, 0x01, 0x03, 0x01, 0x02, 0x03, 0x01, 0x01, 0x00};
  const uint8_t poly_stream[] = {0x55, 0xff, 0x81, 0x03, 0x01, 0x01, 0x15, 0x70, 0x6f, 0x6c, 0x79, 0x43, 0x69, 0x70, 0x68, 0x65, 0x72, 0x74, 0x65, 0x78, 0x74, 0x57, 0x72, 0x61, 0x70, 0x70, 0x65, 0x72, 0x01, 0xff, 0x82, 0x00, 0x01, 0x04, 0x01, 0x0a, 0x43, 0x6f, 0x65, 0x66, 0x66, 0x42, 0x79, 0x74, 0x65, 0x73, 0x01, 0xff, 0x84, 0x00, 0x01, 0x06, 0x44, 0x65, 0x67, 0x72, 0x65, 0x65, 0x01, 0x04, 0x00, 0x01, 0x0b, 0x53, 0x63, 0x61, 0x6c, 0x65, 0x46, 0x61, 0x63, 0x74, 0x6f, 0x72, 0x01, 0x04, 0x00, 0x01, 0x02, 0x4c, 0x32, 0x01, 0x02, 0x00, 0x00, 0x00, 0x17, 0xff, 0x83, 0x02, 0x01, 0x01, 0x09, 0x5b, 0x5d, 0x5b, 0x5d, 0x75, 0x69, 0x6e, 0x74, 0x38, 0x01, 0xff, 0x84, 0x00, 0x01, 0x0a, 0x00, 0x00, 0x0e, 0xff, 0x82, 0x01, 0x02, 0x02, 0x01, 0x02, 0x01, 0x03, 0x01, 0x04, 0x01, 0x02, 0x00};
  const uint8_t zero_stream[] = {0x31, 0xff, 0x81, 0x03, 0x01, 0x01, 0x11, 0x63, 0x69, 0x70, 0x68, 0x65, 0x72, 0x74, 0x65, 0x78, 0x74, 0x57, 0x72, 0x61, 0x70, 0x70, 0x65, 0x72, 0x01, 0xff, 0x82, 0x00, 0x01, 0x02, 0x01, 0x06, 0x43, 0x42, 0x79, 0x74, 0x65, 0x73, 0x01, 0x0a, 0x00, 0x01, 0x02, 0x4c, 0x32, 0x01, 0x02, 0x00, 0x00, 0x00, 0x03, 0xff, 0x82, 0x00};
  Bytes a = gob::marshal_ciphertext(Bytes{1, 2, 3}, true);
  if (a != Bytes(ct_stream, ct_stream + sizeof ct_stream)) { printf("MISMATCH gob ciphertext stream\n"); bad++; }
  Bytes b = gob::marshal_poly_ciphertext({Bytes{1, 2}, Bytes{3}}, 2, 1, false);
  if (b != Bytes(poly_stream, poly_stream + sizeof poly_stream)) { printf("MISMATCH gob poly stream\n"); bad++; }
  Bytes z = gob::marshal_ciphertext(Bytes{}, false);
  if (z != Bytes(zero_stream, zero_stream + sizeof zero_stream)) { printf("MISMATCH gob zero-value stream\n"); bad++; }
  gob::Envelope e = gob::decode(a);
  if (e.CBytes != Bytes{1, 2, 3} || !e.L2) { printf("MISMATCH gob decode\n"); bad++; }
  e = gob::decode(b);
  if (e.CoeffBytes.size() != 2 || e.CoeffBytes[1] != Bytes{3} || e.Degree != 2 || e.ScaleFactor != 1 || e.L2) {
    printf("MISMATCH gob poly decode\n");
    bad++;
  }
  bool threw = false;
  try {
    gob::decode(Bytes{});
  } catch (const gob::Error&) {
    threw = true;
  }
  if (!threw) { printf("MISMATCH empty gob input accepted\n"); bad++; }
  // the documented integer encodings of encoding/gob
  Bytes t;
  gob::put_uint(t, 256);
  gob::put_int(t, -129);
  const uint8_t want[] = {0xfe, 0x01, 0x00, 0xfe, 0x01, 0x01};
  if (t.size() != 6 || memcmp(t.data(), want, 6) != 0) { printf("MISMATCH gob integer encodings\n"); bad++; }
  // sha256("abc"), FIPS 180-4 appendix B.1
  const uint8_t abc[] = {0xba, 0x78, 0x16, 0xbf, 0x8f, 0x01, 0xcf, 0xea, 0x41, 0x41, 0x40, 0xde, 0x5d, 0xae, 0x22, 0x23,
                         0xb0, 0x03, 0x61, 0xa3, 0x96, 0x17, 0x7a, 0x9c, 0xb4, 0x10, 0xff, 0x61, 0xf2, 0x00, 0x15, 0xad};
  Bytes dg = sha256::digest(Bytes{'a', 'b', 'c'});
  if (memcmp(dg.data(), abc, 32) != 0) { printf("MISMATCH sha256\n"); bad++; }
  // the shard split of the multi-GPU forms (no GPU needed): contiguous, ordered, sizes differ by at most one
  for (size_t total : {(size_t)0, (size_t)1, (size_t)7, (size_t)1048579}) {
    for (int world : {1, 2, 3, 8}) {
      size_t prev = 0;
      for (int r = 0; r < world; ++r) {
        size_t lo = 99, hi = 99;
        bgn_shard_range(total, world, r, &lo, &hi);
        if (lo != prev || hi < lo || hi - lo > total / world + 1) { printf("MISMATCH bgn_shard_range\n"); bad++; }
        prev = hi;
      }
      if (prev != total) { printf("MISMATCH bgn_shard_range end\n"); bad++; }
    }
  }
  std::vector<uint64_t> d100 = UnbalancedEncode(100, 3);
  if (d100 != std::vector<uint64_t>{1, 0, 2, 0, 1, 0}) { printf("MISMATCH UnbalancedEncode\n"); bad++; }
  return bad;
}

int main(int argc, char** argv) {
  if (argc != 8) return 2;
  if (host_checks()) return 1;
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
    // wire envelopes (bgn_test.go:37-85): Bytes() -> New...FromBytes preserves the element
    {
      Ciphertext back = pk.NewCiphertextFromBytes(one.Bytes_());
      if (back.C != one.C || back.L2) { printf("MISMATCH gob round trip (L1)\n"); bad++; }
      Ciphertext l2 = pk.Mult(one, one);
      back = pk.NewCiphertextFromBytes(l2.Bytes_());
      if (back.C != l2.C || !back.L2) { printf("MISMATCH gob round trip (L2)\n"); bad++; }
      PolyCiphertext pb = pk.NewPolyCiphertextFromBytes(b2.Bytes_());
      if (pb.Degree != b2.Degree || pb.ScaleFactor != 2 || pb.L2 || pb.Coefficients.size() != b2.Coefficients.size() ||
          pb.Coefficients[1].C != b2.Coefficients[1].C) { printf("MISMATCH gob poly round trip\n"); bad++; }
    }
    // proofs (gadgets_test.go:9-71): decryption proofs incl. the aggregate whose exponents add up
    {
      Ciphertext c1 = pk.EncryptWithRandomness(scalar_u64(1000003), scalar_u64(777)),
                 c2 = pk.EncryptWithRandomness(scalar_u64(2000003), scalar_u64(999));
      expect("decryption proof valid", pk.CheckDecryptionProof(c1, {scalar_u64(1000003), scalar_u64(777)}), 1);
      expect("decryption proof wrong randomness", pk.CheckDecryptionProof(c1, {scalar_u64(1000003), scalar_u64(778)}), 0);
      expect("decryption proof wrong value", pk.CheckDecryptionProof(c1, {scalar_u64(1000004), scalar_u64(777)}), 0);
      expect("decryption proof aggregate", pk.CheckDecryptionProof(pk.Add(c1, c2), {scalar_u64(3000006), scalar_u64(1776)}), 1);
      // proof of plaintext knowledge with v = 0, z = 0: Ct = O, DL = nonce1 whatever the challenge
      ProofOfPlaintextKnowledge pr{pk.EncryptDeterministic(scalar_u64(0)), pk.EncryptDeterministic(scalar_u64(424242)),
                                   scalar_u64(424242)};
      expect("plaintext-knowledge proof valid", pk.CheckProofOfPlaintextKnoewledge(pr.Ct, pr), 1);
      pr.DL = scalar_u64(424243);
      expect("plaintext-knowledge proof wrong DL", pk.CheckProofOfPlaintextKnoewledge(pr.Ct, pr), 0);
      std::vector<uint8_t> ok = pk.Validate({c1, c2, zero});
      Ciphertext broken = c1;
      broken.C.back() ^= 1;
      expect("validate good", ok[0] + ok[1] + ok[2], 3);
      expect("validate off-curve", pk.Validate({broken})[0], 0);
    }
    // chains on arrays that stay on the device (DeviceArray: the C++ twin of go/bgn_amd.go's, over bgn_dev_* and the
    // `_dev` entry points): Encrypt -> Mult -> Add -> Add in place -> Neg -> makeL2 -> MultConst -> Decrypt with one
    // upload of scalars per step and one download of plaintexts at the end; MultPoly on device coefficient arrays
    {
      std::vector<Scalar> xs, rs3, ks;
      for (uint64_t i = 1; i <= 6; ++i) {
        xs.push_back(scalar_u64(i));
        rs3.push_back(scalar_u64(100 + i));
        ks.push_back(scalar_u64(i % 4 + 1));
      }
      DeviceArray c = pk.EncryptBatchDev(xs, &rs3);
      std::vector<Ciphertext> host = pk.EncryptBatch(xs, &rs3), back = pk.Download(c);
      for (size_t i = 0; i < host.size(); ++i)
        if (back[i].C != host[i].C || back[i].L2) { printf("MISMATCH EncryptBatchDev element %zu\n", i); bad++; }
      DeviceArray prod = pk.MultBatchDev(c, c);                       // x^2
      DeviceArray sum = pk.AddBatchDev(prod, prod);                   // 2 x^2
      pk.AddBatchDevInto(sum, sum, prod);                             // 3 x^2, accumulated in place
      DeviceArray tot = pk.AddBatchDev(sum, pk.MakeL2BatchDev(pk.NegBatchDev(c)));   // 3 x^2 - x
      DeviceArray mc = pk.MultConstBatchDev(tot, ks);
      auto dec = sk.DecryptBatchDev(mc, pk);
      for (uint64_t i = 1; i <= 6; ++i)
        expect("device chain (3x^2 - x) * k", dec.second[i - 1] ? -1 : dec.first[i - 1], (int64_t)((3 * i * i - i) * (i % 4 + 1)));
      std::vector<uint8_t> ok = pk.ValidateBatchDev(c);
      expect("ValidateBatchDev", ok[0] + ok[1] + ok[2] + ok[3] + ok[4] + ok[5], 6);
      // two products of 2 x 2 coefficient vectors: elements 0..3 times elements 2..5
      std::vector<Ciphertext> pa(host.begin(), host.begin() + 4), pb(host.begin() + 2, host.begin() + 6);
      DeviceArray da = pk.Upload(pa), db = pk.Upload(pb);
      std::vector<Ciphertext> pd = pk.Download(pk.MultPolyBatchDev(2, 2, 2, da, db));
      Bytes A, B, ref(2 * 4 * pk.ElementBytes());
      for (auto& x : pa) A.insert(A.end(), x.C.begin(), x.C.end());
      for (auto& x : pb) B.insert(B.end(), x.C.begin(), x.C.end());
      int rc = bgn_poly_mult_batch(pk.handle(), 2, 2, 2, A.data(), B.data(), ref.data());
      if (rc != BGN_OK) throw Error(rc, "bgn_poly_mult_batch");
      for (size_t i = 0; i < pd.size(); ++i)
        if (!pd[i].L2 || memcmp(pd[i].C.data(), ref.data() + i * pk.ElementBytes(), pk.ElementBytes()) != 0) {
          printf("MISMATCH MultPolyBatchDev coefficient %zu\n", i);
          bad++;
        }
    }
    // one key on several devices from one process through the C ABI (bgn_mctx_*): the device list names GPU 0
    // three times, which gives real multi-context sharding on a one-GPU box; ragged shards, MultPoly by polynomial
    {
      const Bytes pB = unhex(argv[1]), nB = unhex(argv[2]), PB = unhex(argv[4]), QB = unhex(argv[5]), q1 = unhex(argv[6]);
      const int devs[3] = {0, 0, 0};
      bgn_mctx* m = nullptr;
      int rc = bgn_mctx_create(&m, pB.data(), pB.size(), nB.data(), nB.size(), strtoull(argv[3], nullptr, 10), PB.data(),
                               QB.data(), 1, devs, 3);
      if (rc != BGN_OK) throw Error(rc, "bgn_mctx_create");
      expect("mctx devices", bgn_mctx_device_count(m), 3);
      rc = bgn_mctx_set_secret(m, q1.data(), q1.size());
      if (rc == BGN_OK) rc = bgn_mctx_setup_decryption(m, strtoull(argv[7], nullptr, 10));
      if (rc != BGN_OK) throw Error(rc, "bgn_mctx_setup_decryption");
      const size_t cnt = 7, E = pk.ElementBytes();
      Bytes xs(cnt), rs2(cnt), cts(cnt * E), prod(cnt * E), sum(cnt * E), stat(cnt);
      std::vector<int64_t> ms(cnt);
      for (size_t i = 0; i < cnt; ++i) {
        xs[i] = (uint8_t)(i + 1);
        rs2[i] = (uint8_t)(40 + i);
      }
      rc = bgn_mencrypt_batch(m, cnt, xs.data(), 1, rs2.data(), 1, cts.data());
      if (rc == BGN_OK) rc = bgn_mmult_batch(m, cnt, cts.data(), cts.data(), nullptr, 0, prod.data());
      if (rc == BGN_OK) rc = bgn_madd_batch(m, cnt, 2, prod.data(), prod.data(), nullptr, 0, sum.data());
      if (rc == BGN_OK) rc = bgn_mdecrypt_batch(m, cnt, 2, sum.data(), ms.data(), stat.data());
      if (rc != BGN_OK) throw Error(rc, "bgn_m*_batch");
      for (size_t i = 0; i < cnt; ++i) expect("mctx Dec(x*x + x*x)", stat[i] ? -1 : ms[i], 2 * (int64_t)(i + 1) * (int64_t)(i + 1));
      // MultPoly: 3 polynomials of 2 x 2 coefficients, sharded by polynomial; against the single context
      Bytes pout(3 * 4 * E), pref(3 * 4 * E);
      rc = bgn_mpoly_mult_batch(m, 3, 2, 2, cts.data(), cts.data() + E, pout.data());
      if (rc == BGN_OK) rc = bgn_poly_mult_batch(pk.handle(), 3, 2, 2, cts.data(), cts.data() + E, pref.data());
      if (rc != BGN_OK) throw Error(rc, "bgn_mpoly_mult_batch");
      if (pout != pref) { printf("MISMATCH mctx MultPoly\n"); bad++; }
      bgn_mctx_destroy(m);
    }
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
