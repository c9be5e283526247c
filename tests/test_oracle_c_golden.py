"""CPU: the C oracle (oracle/bgn_oracle.c: 64-bit limbs, projective Miller loop)
against the golden vectors produced by the independent pure-Python oracle
(affine, full-divisor, big-int).  Two restatements with different formulations
agreeing byte for byte is what stands in for the un-runnable PBC reference."""
import subprocess
import os

import pytest

from conftest import ROOT, KEYS, load_fixture


@pytest.fixture(scope="session")
def oc():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    import oracle_c
    assert oracle_c.available()
    return oracle_c


def J(hexes):
    return b"".join(bytes.fromhex(h) for h in hexes)


@pytest.mark.parametrize("name", KEYS + ["k1024b", "k2048"])       # + the 37-limb 1024-bit key and the 2048-bit key
def test_c_oracle_matches_golden(oc, name):
    fx = load_fixture(name)
    o = oc.Oracle.from_fixture(fx)
    cts = [e["ct"] for e in fx["encrypt"]]
    assert o.encrypt([int(e["x"], 16) for e in fx["encrypt"]], [int(e["r"], 16) for e in fx["encrypt"]]) == J(cts)
    assert o.encrypt([5], None) == o.encrypt([5], [0])          # EncryptDeterministic == r = 0
    assert o.mult(J(cts[v["a"]] for v in fx["mult"]), J(cts[v["b"]] for v in fx["mult"])) == J(v["out"] for v in fx["mult"])
    assert o.mult(J(cts[v["a"]] for v in fx["make_l2"])) == J(v["out"] for v in fx["make_l2"])
    a, b = J(cts[v["a"]] for v in fx["l1"]), J(cts[v["b"]] for v in fx["l1"])
    assert o.add(1, a, b) == J(v["add"] for v in fx["l1"])
    assert o.add(1, a, b, True) == J(v["sub"] for v in fx["l1"])
    l2 = [v["out"] for v in fx["mult"]]
    a, b = J(l2[v["a"]] for v in fx["l2"]), J(l2[v["b"]] for v in fx["l2"])
    assert o.add(2, a, b) == J(v["add"] for v in fx["l2"])
    assert o.add(2, a, b, True) == J(v["sub"] for v in fx["l2"])
    for lvl, key, src in [(1, "multconst_l1", cts), (2, "multconst_l2", l2)]:
        assert o.multconst(lvl, J(src[v["a"]] for v in fx[key]), [int(v["k"], 16) for v in fx[key]]) == J(v["out"] for v in fx[key])
    po = fx["poly"]
    assert o.poly_mult(1, po["d1"], po["d2"], J(po["a"]), J(po["b"])) == J(po["out"])


@pytest.mark.parametrize("name", ["toy64", "k256", "k512"])
def test_c_oracle_decrypt(oc, name):
    """gsbs.go semantics: range [1, B*B+B+2], zero short-cut, negative retry, out-of-bounds error."""
    fx = load_fixture(name)
    o = oc.Oracle.from_fixture(fx)
    o.setup_decryption(int(fx["q1"], 16), fx["msg_space"])
    for d in fx["decrypt"]:
        m, st = o.decrypt(d["level"], bytes.fromhex(d["ct"]))
        if d["expect"] is None:
            assert st[0] == 1
        else:
            assert st[0] == 0 and m[0] == d["expect"]
