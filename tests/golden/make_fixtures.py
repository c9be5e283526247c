#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ from the
pure-Python oracle (oracle/bgn_ref.py).

The reference (Go + cgo -> PBC) cannot be run in the build container (no Go
toolchain, no libpbc) and its tests hold no known-answer vectors, so these
fixtures are produced by the oracle's restatement; see the "PARITY UNPINNED"
note in oracle/bgn_ref.py and DESIGN.md.  Fixtures are data only: seeded keys,
inputs, explicit randomness, expected wire bytes / plaintexts.

Usage:  python tests/golden/make_fixtures.py            (rewrites all *.json)
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import bgn_ref as R  # noqa: E402


def hx(b: bytes) -> str:
    return b.hex()


def wire(pk, e) -> str:
    return hx(R.elem_to_bytes(e, pk.p))


def make(name, key_bits, msg_space, seed, n_enc, n_pair, poly_d):
    rng = random.Random(seed * 7919 + 13)
    pk, sk = R.NewKeyGen(key_bits, msg_space, seed=seed)
    pk.SetupDecryption(sk)
    B = int(__import__("math").ceil(__import__("math").sqrt(float(msg_space))))
    fx = {
        "name": name,
        "key_bits": key_bits,
        "p": hex(pk.p), "n": hex(pk.n), "l": pk.l,
        "P": wire(pk, pk.P), "Q": wire(pk, pk.Q),
        "q1": hex(sk.Key), "msg_space": msg_space, "poly_base": pk.PolyBase,
        "fp_bytes": R.fp_len(pk.p),
    }
    # --- Encrypt: (x, r) -> ciphertext; includes x = 0 with r = 0 (identity), r = 0, x >= n
    enc = []
    xs = [0, 0, 1, 5, msg_space - 1, rng.randrange(pk.n), pk.n + 3]
    rs = [0, rng.randrange(pk.n), 0, rng.randrange(pk.n), rng.randrange(pk.n), rng.randrange(pk.n), rng.randrange(pk.n)]
    while len(xs) < n_enc:
        xs.append(rng.randrange(min(msg_space, 1 << 40)))
        rs.append(rng.randrange(pk.n))
    cts = []
    for x, r in zip(xs, rs):
        ct = pk.EncryptWithRandomness(x, r)
        cts.append(ct)
        enc.append({"x": hex(x), "r": hex(r), "ct": wire(pk, ct.C)})
    fx["encrypt"] = enc
    # --- L1 Add / Sub / Neg on consecutive pairs (deterministic mode)
    l1 = []
    for i in range(len(cts)):
        a, b = cts[i], cts[(i * 3 + 1) % len(cts)]
        l1.append({"a": i, "b": (i * 3 + 1) % len(cts), "add": wire(pk, pk.Add(a, b).C), "sub": wire(pk, pk.Sub(a, b).C),
                   "neg": wire(pk, pk.Neg(a).C)})
    # doubling case a + a and cancellation a - a
    l1.append({"a": 3, "b": 3, "add": wire(pk, pk.Add(cts[3], cts[3]).C), "sub": wire(pk, pk.Sub(cts[3], cts[3]).C),
               "neg": wire(pk, pk.Neg(cts[3]).C)})
    fx["l1"] = l1
    # --- Mult (pairing), incl. identity operands and a == b
    pairs = [(0, 3), (3, 0), (3, 3), (2, 4)]
    while len(pairs) < n_pair:
        pairs.append((rng.randrange(len(cts)), rng.randrange(len(cts))))
    mult = []
    l2 = []
    for (i, k) in pairs:
        c = pk.Mult(cts[i], cts[k])
        l2.append(c)
        mult.append({"a": i, "b": k, "out": wire(pk, c.C)})
    fx["mult"] = mult
    fx["make_l2"] = [{"a": i, "out": wire(pk, pk.makeL2(cts[i]).C)} for i in range(min(4, len(cts)))]
    # --- L2 Add / Sub / Neg
    l2v = []
    for i in range(len(l2)):
        a, b = l2[i], l2[(i + 1) % len(l2)]
        l2v.append({"a": i, "b": (i + 1) % len(l2), "add": wire(pk, pk.Add(a, b).C), "sub": wire(pk, pk.Sub(a, b).C),
                    "neg": wire(pk, pk.Neg(a).C)})
    fx["l2"] = l2v
    # --- MultConst L1 and L2
    ks = [0, 1, 2, pk.PolyBase, rng.randrange(1 << 20), rng.randrange(pk.n)]
    fx["multconst_l1"] = [{"a": (j + 2) % len(cts), "k": hex(k), "out": wire(pk, pk.MultConst(cts[(j + 2) % len(cts)], k).C)}
                          for j, k in enumerate(ks)]
    fx["multconst_l2"] = [{"a": j % len(l2), "k": hex(k), "out": wire(pk, pk.MultConst(l2[j % len(l2)], k).C)}
                          for j, k in enumerate(ks)]
    # --- Decrypt: small messages (positive, zero, negative, out of range)
    dec = []
    maxm = B * B + B + 2
    msgs = [0, 1, 2, B, B + 1, msg_space - 1, maxm, maxm + 1, -1, -5, -(msg_space - 1), -maxm, -(maxm + 1)]
    if msg_space > (1 << 20):
        # keep the Python BSGS affordable: only small |m| for big message spaces
        msgs = [0, 1, 2, 77, -1, -5, 1000, -1000]
    for m in msgs:
        r = rng.randrange(pk.n)
        ct = pk.EncryptWithRandomness(m % pk.n, r)
        got = sk.Decrypt(ct, pk) if msg_space <= (1 << 20) else m
        dec.append({"m": m, "r": hex(r), "ct": wire(pk, ct.C), "level": 1,
                    "expect": got if got is not None else None})
    # level-2 decrypts: products of small messages
    for (m1, m2) in [(0, 5), (1, 1), (3, 7), (-2, 9), (-3, -4)]:
        c1 = pk.EncryptWithRandomness(m1 % pk.n, rng.randrange(pk.n))
        c2 = pk.EncryptWithRandomness(m2 % pk.n, rng.randrange(pk.n))
        c = pk.Mult(c1, c2)
        got = sk.Decrypt(c, pk) if msg_space <= (1 << 20) else m1 * m2
        dec.append({"m": m1 * m2, "ct": wire(pk, c.C), "level": 2, "expect": got})
    fx["decrypt"] = dec
    # --- MultPoly: two coefficient vectors with digits in {-1,0,1}
    d1, d2 = poly_d
    ca = [rng.choice([-1, 0, 1]) for _ in range(d1)]
    cb = [rng.choice([-1, 0, 1]) for _ in range(d2)]
    ea = pk.EncryptPolyCoeffs(ca, [rng.randrange(pk.n) for _ in range(d1)])
    eb = pk.EncryptPolyCoeffs(cb, [rng.randrange(pk.n) for _ in range(d2)])
    prod = pk.MultPoly(ea, eb)
    fx["poly"] = {"d1": d1, "d2": d2, "ca": ca, "cb": cb,
                  "a": [wire(pk, c.C) for c in ea], "b": [wire(pk, c.C) for c in eb],
                  "out": [wire(pk, c.C) for c in prod]}
    return fx


CONFIGS = [
    # name, key bits, message space, seed, #encryptions, #pairings, poly degrees
    ("toy64", 64, 1021, 11, 10, 8, (3, 4)),
    ("k256", 256, 1021, 12, 8, 6, (2, 3)),
    ("k512", 512, 1021, 1, 8, 6, (2, 3)),          # bgn_test.go:8-13 constants
    ("k1024", 1024, 1 << 40, 1, 8, 6, (2, 2)),     # BASELINE.json configs[1..4]
    # a second 1024-bit key (l = 6336: p has 1037 bits, the top of the range A1 parameters reach at this size) for
    # the tests that keep two keys resident on one GPU
    ("k1024b", 1024, 1 << 40, 4242, 8, 5, (2, 2)),
    # a 2048-bit key (bgn.go:65-73 accepts any even key size): the 72-limb instantiation
    ("k2048", 2048, 1021, 7, 8, 5, (2, 2)),
]

if __name__ == "__main__":
    only = set(sys.argv[1:])
    for cfg in CONFIGS:
        if only and cfg[0] not in only:
            continue
        fx = make(*cfg)
        path = os.path.join(HERE, cfg[0] + ".json")
        with open(path, "w") as f:
            json.dump(fx, f, indent=1)
        print("wrote", path, os.path.getsize(path), "bytes")
