"""gadgets.go — decryption proofs and proofs of plaintext knowledge.
CPU: the oracle's restatement against the reference's own test pins (gadgets_test.go:9-105).  GPU: the
engine's batched verification against the oracle's verdicts on the same proofs."""
import random

import pytest

import bgn_ref as R


def fresh_key(bits=128, seed=77):
    return R.NewKeyGen(bits, 1021, 3, True, seed)


def make_cases(opk, osk, rng, count):
    """(ciphertext, proof, expected verdict) triples shaped like gadgets_test.go:73-105."""
    n = opk.n
    out = []
    for i in range(count):
        v, r, r2 = rng.randrange(n), rng.randrange(n), rng.randrange(n)
        ct = opk.EncryptWithRandomness(v, r)
        kind = i % 3
        if kind == 0:
            proof = R.NewProofOfPlaintextKnowledge(opk, osk, v, r, rng.randrange(n))       # valid
        elif kind == 1:
            proof = R.NewProofOfPlaintextKnowledge(opk, osk, v, r2, rng.randrange(n))      # wrong randomness
        else:
            proof = R.NewProofOfPlaintextKnowledge(opk, osk, r2, r, rng.randrange(n))      # wrong value
        out.append((ct, proof, kind == 0))
    return out


def test_oracle_decryption_proofs():
    opk, osk = fresh_key()
    rng = random.Random(1)
    n = opk.n
    v, r, r2 = rng.randrange(n), rng.randrange(n), rng.randrange(n)
    ct = opk.EncryptWithRandomness(v, r)
    assert R.CheckDecryptionProof(opk, ct, R.DecryptionProof(v, r))                         # gadgets_test.go:9-22
    assert not R.CheckDecryptionProof(opk, ct, R.DecryptionProof(v, r2))                    # :48-71
    assert not R.CheckDecryptionProof(opk, ct, R.DecryptionProof(r2, r))
    v2, rr2 = rng.randrange(n), rng.randrange(n)
    ct3 = opk.Add(ct, opk.EncryptWithRandomness(v2, rr2))
    assert R.CheckDecryptionProof(opk, ct3, R.DecryptionProof(v + v2, r + rr2))             # aggregate, :24-46


def test_oracle_proofs_of_plaintext_knowledge():
    opk, osk = fresh_key()
    for ct, proof, want in make_cases(opk, osk, random.Random(2), 9):
        assert R.CheckProofOfPlaintextKnoewledge(opk, ct, proof) == want


@pytest.mark.gpu
@pytest.mark.parametrize("bits", [128, 512])
def test_gpu_batched_proof_verification(bits):
    import bgn_amd
    opk, osk = fresh_key(bits, seed=5 + bits)
    p = opk.p
    pk = bgn_amd.PublicKey(p, opk.n, opk.l, R.elem_to_bytes(opk.P, p), R.elem_to_bytes(opk.Q, p), 1021)
    sk = bgn_amd.SecretKey(osk.Key, osk.R)
    W = lambda c: bgn_amd.Ciphertext(R.elem_to_bytes(c.C, p), False)
    rng = random.Random(3)
    cases = make_cases(opk, osk, rng, 12 if bits == 128 else 6)
    cts = [W(ct) for ct, _, _ in cases]
    proofs = [bgn_amd.ProofOfPlaintextKnowledge(W(pr.Ct), W(pr.Nonce), pr.DL) for _, pr, _ in cases]
    assert pk.CheckProofOfPlaintextKnoewledgeBatch(cts, proofs) == [w for _, _, w in cases]
    # the mirror's own prover agrees with the oracle's on the same nonce, and verifies
    v, z, nonce1 = rng.randrange(opk.n), rng.randrange(opk.n), rng.randrange(opk.n)
    mine = pk.NewProofOfPlaintextKnowledge(sk, v, z, nonce1)
    ref = R.NewProofOfPlaintextKnowledge(opk, osk, v, z, nonce1)
    assert (mine.Ct.C, mine.Nonce.C, mine.DL) == (R.elem_to_bytes(ref.Ct.C, p), R.elem_to_bytes(ref.Nonce.C, p), ref.DL)
    assert pk.CheckProofOfPlaintextKnoewledge(mine.Ct, mine)
    # decryption proofs, including the aggregate whose value and randomness exceed n (gadgets_test.go:24-46)
    n = opk.n
    vs = [rng.randrange(n) for _ in range(4)]
    rs = [rng.randrange(n) for _ in range(4)]
    enc = pk.EncryptBatch(vs, rs)
    agg = pk.Add(enc[0], enc[1])
    checks = [(enc[0], vs[0], rs[0], True), (enc[1], vs[1], rs[2], False), (enc[2], vs[3], rs[2], False),
              (agg, vs[0] + vs[1], rs[0] + rs[1], True), (enc[3], vs[3], rs[3], True)]
    got = pk.CheckDecryptionProofBatch([c for c, _, _, _ in checks], [bgn_amd.NewDecryptionProof(v, r) for _, v, r, _ in checks])
    assert got == [w for _, _, _, w in checks]
    assert pk.CheckDecryptionProof(enc[0], bgn_amd.NewDecryptionProof(vs[0], rs[0]))
