"""GPU: a 2048-bit key end to end.  The reference accepts any even key size of at least 16 bits (bgn.go:65-73); the
engine serves fields of up to 2079 bits with a 72-limb instantiation of the lane kernels — a functional one: 128-thread
workgroups, the long-lived slots in per-lane arrays instead of accumulation registers — and the lane-group pairing
kernel (18 limbs per lane).  Golden vectors of tests/golden/k2048.json (made by the Python oracle) and the C oracle."""
import random

import pytest

import bgn_amd
from conftest import load_fixture

pytestmark = pytest.mark.gpu


def H(hexes):
    return b"".join(bytes.fromhex(h) for h in hexes)


@pytest.fixture(scope="module")
def key():
    fx = load_fixture("k2048")
    pk = bgn_amd.PublicKey(int(fx["p"], 16), int(fx["n"], 16), fx["l"], bytes.fromhex(fx["P"]), bytes.fromhex(fx["Q"]),
                           fx["msg_space"], True, fx["poly_base"])
    pk.engine.set_memory_budget(64 << 30)
    return fx, pk, bgn_amd.SecretKey(int(fx["q1"], 16))


def test_field_size(key):
    fx, pk, _ = key
    assert int(fx["p"], 16).bit_length() > 2048 and pk.engine.elem_bytes == 2 * fx["fp_bytes"]


def test_encrypt_add_multconst_golden(key):
    fx, pk, _ = key
    eng = pk.engine
    xs = [int(e["x"], 16) for e in fx["encrypt"]]
    rs = [int(e["r"], 16) for e in fx["encrypt"]]
    cts = eng.encrypt(xs, rs)
    for row, e in zip(cts, fx["encrypt"]):
        assert bytes(row).hex() == e["ct"]
    ct = [e["ct"] for e in fx["encrypt"]]
    a, b = H([ct[v["a"]] for v in fx["l1"]]), H([ct[v["b"]] for v in fx["l1"]])
    for got, v in zip(eng.add(1, a, b), fx["l1"]):
        assert bytes(got).hex() == v["add"]
    for got, v in zip(eng.sub(1, a, b), fx["l1"]):
        assert bytes(got).hex() == v["sub"]
    ks = [int(v["k"], 16) for v in fx["multconst_l1"]]
    out = eng.multconst(1, H([ct[v["a"]] for v in fx["multconst_l1"]]), ks)
    for got, v in zip(out, fx["multconst_l1"]):
        assert bytes(got).hex() == v["out"]


@pytest.mark.parametrize("kernel", ["quad", "lane"])
def test_mult_and_make_l2_golden(key, kernel, engopts):
    fx, pk, _ = key
    eng = engopts.register(pk.engine)
    big = "100000000"
    engopts.set("quad_min", "0")
    for v in ("quad_max", "quad_max_l2", "quad_max_dec", "quad_max_pow"):
        engopts.set(v, big if kernel == "quad" else "0")
    ct = [e["ct"] for e in fx["encrypt"]]
    out = eng.mult(H([ct[v["a"]] for v in fx["mult"]]), H([ct[v["b"]] for v in fx["mult"]]))
    assert ("quad" in eng.last_kernel_name()) == (kernel == "quad")
    for row, v in zip(out, fx["mult"]):
        assert bytes(row).hex() == v["out"]
    out = eng.make_l2(H([ct[v["a"]] for v in fx["make_l2"]]))
    for row, v in zip(out, fx["make_l2"]):
        assert bytes(row).hex() == v["out"]


def test_decrypt_golden_and_round_trip(key):
    fx, pk, sk = key
    pk.SetupDecryption(sk)
    eng = pk.engine
    for lvl in (1, 2):
        rows = [d for d in fx["decrypt"] if d["level"] == lvl]
        m, st = eng.decrypt(lvl, H([d["ct"] for d in rows]))
        for got, s, d in zip(m.tolist(), st.tolist(), rows):
            if d["expect"] is None:
                assert s == 1
            else:
                assert s == 0 and got == d["expect"]
    rng = random.Random(5)
    n, T = int(fx["n"], 16), fx["msg_space"]
    ms = [rng.randrange(T) for _ in range(9)]
    cts = eng.encrypt(ms, [rng.randrange(n) for _ in ms])
    m, st = eng.decrypt(1, cts.tobytes())
    assert m.tolist() == ms and not st.any()


def test_random_pairs_vs_c_oracle(key):
    import oracle_c
    fx, pk, _ = key
    eng = pk.engine
    o = oracle_c.Oracle.from_fixture(fx)
    rng = random.Random(2048)
    n = int(fx["n"], 16)
    xs = [rng.randrange(fx["msg_space"]) for _ in range(10)]
    rs = [rng.randrange(n) for _ in range(10)]
    cts = eng.encrypt(xs, rs)
    assert cts.tobytes() == o.encrypt(xs, rs)
    a, b = cts[:5].tobytes(), cts[5:].tobytes()
    assert eng.mult(a, b).tobytes() == o.mult(a, b)
    assert eng.add(1, a, b).tobytes() == o.add(1, a, b)
