"""CPU: the micro-op tables of the wave-cooperative pairing (tools/coop/gen_prog.py) executed on Python integers
and on a lane-level model of the kernel's arithmetic (tests/coop_model.py), against the golden Mult vectors and the
oracle.  The HIP kernel (bgn_amd/csrc/coop/) interprets exactly these tables; tests/test_gpu_coop.py compares it
with the same vectors on the GPU."""
import os
import random
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_fixture, oracle_key

import coop_model as cm


def _points(fx, v):
    import bgn_ref as R
    p = int(fx["p"], 16)
    cts = [bytes.fromhex(e["ct"]) for e in fx["encrypt"]]
    A = R.elem_from_bytes(cts[v["a"]], p)
    B = R.elem_from_bytes(cts[v["b"]], p)
    return A, B


def _nonident(fx):
    return [v for v in fx["mult"] if int(fx["encrypt"][v["a"]]["ct"], 16) and int(fx["encrypt"][v["b"]]["ct"], 16)]


@pytest.mark.parametrize("key", ["toy64", "k256", "k512"])
def test_program_on_integers_matches_golden_mult(key):
    import bgn_ref as R
    fx = load_fixture(key)
    p, n, l = int(fx["p"], 16), int(fx["n"], 16), fx["l"]
    for v in _nonident(fx)[: (3 if key != "k512" else 1)]:
        A, B = _points(fx, v)
        m = cm.ValueMachine(p, cm.nl_for(p))
        re, im = m.pairing(A[0], A[1], B[0], B[1], n, l)
        assert R.elem_to_bytes((re, im), p).hex() == v["out"]


@pytest.mark.parametrize("key", ["toy64", "k256"])
def test_lane_model_matches_golden_mult(key):
    import bgn_ref as R
    fx = load_fixture(key)
    p, n, l = int(fx["p"], 16), int(fx["n"], 16), fx["l"]
    v = _nonident(fx)[0]
    A, B = _points(fx, v)
    m = cm.LaneMachine(p, cm.nl_for(p))
    re, im = m.pairing(A[0], A[1], B[0], B[1], n, l)
    assert R.elem_to_bytes((re, im), p).hex() == v["out"]


@pytest.mark.parametrize("key", ["toy64", "k256", "k512", "k1024"])
def test_lane_product_is_the_montgomery_product(key):
    """One product on lanes with signed, once-normalised limbs equals (A*B + Q*p)/R on the integers."""
    fx = load_fixture(key)
    p = int(fx["p"], 16)
    nl = cm.nl_for(p)
    m = cm.LaneMachine(p, nl)
    rng = random.Random(11)
    for _ in range(6):
        A, B = rng.randrange(0, 19 * p), rng.randrange(0, 22 * p)
        # a representation with signed limbs: value = sum of two forms, one subtracted
        X, Y = rng.randrange(0, 3 * p), rng.randrange(0, 5 * p)
        a = m.normalize64(cm.to_lanes(A + X, nl).astype(np.int64) - cm.to_lanes(X, nl).astype(np.int64))
        b = m.normalize64(cm.to_lanes(B + Y, nl).astype(np.int64) - cm.to_lanes(Y, nl).astype(np.int64))
        assert cm.from_lanes(a) == A and cm.from_lanes(b) == B
        t = m.normalize64(m.mul(a, b))
        Q = (A * B * m.pinvR) % m.R
        assert cm.from_lanes(t) == (A * B + Q * p) // m.R
        c = m.canonical(t)
        assert cm.from_lanes(c.astype(np.uint32)) == ((A * B + Q * p) // m.R) % p if (A * B + Q * p) // m.R < 2 * p else True


def test_schedule_properties():
    """No slot is read in the round that writes it; state ping-pongs; every round has at most W micro-ops."""
    P = cm.program()
    for name, rounds in P.segments:
        written = {}
        for r, us in enumerate(rounds):
            assert 1 <= len(us) <= cm.gen_prog.W
            for u in us:
                for s in u.reads():
                    assert written.get(s, -1) < r, (name, s)
            for u in us:
                assert u.dst not in written
                written[u.dst] = r
        # physical slots: a slot written in round r is not read by another micro-op of round r either
        for r, us in enumerate(rounds):
            wr = {P.phys[u.dst] for u in us}
            for u in us:
                assert not ({P.phys[s] for s in u.reads()} & (wr - {P.phys[u.dst]})), (name, r)
                assert P.phys[u.dst] not in {P.phys[s] for s in u.reads()}, (name, r, u.dst)
    seg = dict(P.segments)
    assert len(seg["DBL0"]) == 3 and len(seg["DD0"]) == 6 and len(seg["DAP0"]) == 6


def test_generated_table_is_current():
    """bgn_amd/csrc/coop/coop_prog.inc is the generator's output (regenerate with tools/coop/gen_prog.py)."""
    path = os.path.join(ROOT, "bgn_amd", "csrc", "coop", "coop_prog.inc")
    have = open(path).read()
    tmp = path + ".check"
    try:
        cm.gen_prog.emit(cm.program(), tmp)
        assert open(tmp).read() == have
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


@pytest.mark.parametrize("key", ["toy64", "k256", "k512"])
def test_table_program_on_integers_matches_golden_make_l2(key):
    """The TD / TDA segments (Miller loop over the key's normalised line table, fixedpair.hpp) on Python integers:
    e(P, C) from P's table — built here from the doubling / addition formulas with plain modular arithmetic — equals
    the golden makeL2 output (bgn.go:316-321), for every non-identity vector."""
    import bgn_ref as R
    fx = load_fixture(key)
    p, n, l = int(fx["p"], 16), int(fx["n"], 16), fx["l"]
    Pp = R.elem_from_bytes(bytes.fromhex(fx["P"]), p)
    table = cm.line_table(Pp[0], Pp[1], n, p)
    done = 0
    for v in fx["make_l2"]:
        Cc = R.elem_from_bytes(bytes.fromhex(fx["encrypt"][v["a"]]["ct"]), p)
        if Cc[0] == 0 and Cc[1] == 0:
            continue
        m = cm.ValueMachine(p, cm.nl_for(p))
        re, im = m.pairing_table(Cc[0], Cc[1], table, n, l)
        assert R.elem_to_bytes((re, im), p).hex() == v["out"]
        done += 1
        if key == "k512":
            break
    assert done
