"""CPU: the device lane programs (bgn_amd/csrc/{fp28,pairing,ops,codec}.hpp) compiled
for the host by the emulation harness (tests/emu) and checked against the golden
vectors / oracle.  This is a test of kernel *logic* (slot programs, exception
paths, bounds); it is not a product path — the product runs the same headers
through hipcc on the GPU (tests -m gpu)."""
import os
import random
import sys

import pytest

import bgn_ref as R
from conftest import ROOT, load_fixture

sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
import emu  # noqa: E402

NAMES = ["toy64", "k256"]


@pytest.fixture(scope="module", params=NAMES)
def ctx(request):
    fx = load_fixture(request.param)
    return fx, emu.Emu.from_fixture(fx)


def test_emu_pairing(ctx):
    fx, E = ctx
    cts = [bytes.fromhex(e["ct"]) for e in fx["encrypt"]]
    for v in fx["mult"][:4]:
        assert E.pairing(cts[v["a"]], cts[v["b"]]).hex() == v["out"]


def test_emu_scalar_mult_exceptional_cases(ctx):
    """acc == +-base inside the ladder: k = n, n+-1, n+2 for P; multiples of q1 (+-1, +2) for Q of order q1."""
    fx, E = ctx
    p, n, q1 = int(fx["p"], 16), int(fx["n"], 16), int(fx["q1"], 16)
    Pw, Qw = bytes.fromhex(fx["P"]), bytes.fromhex(fx["Q"])
    Pp, Qp = R.elem_from_bytes(Pw, p), R.elem_from_bytes(Qw, p)
    rng = random.Random(1)
    L = (n.bit_length() + 7) // 8 + 1
    for k in [0, 1, 2, 3, 4, 5, 7, n - 1, n, n + 1, n + 2, 2 * n + 5, rng.randrange(n)]:
        assert E.g1_mul(Pw, k, L) == R.elem_to_bytes(R.pt_mul(Pp, k, p), p), k
    for k in [q1 - 1, q1, q1 + 1, q1 + 2, 2 * q1, 2 * q1 + 2, 3 * q1 + 2]:
        assert E.g1_mul(Qw, k, L) == R.elem_to_bytes(R.pt_mul(Qp, k, p), p), k
    assert E.g1_mul(bytes(2 * E.L), 5, 2) == bytes(2 * E.L)      # identity base


def test_emu_g1_add_run(ctx):
    """The whole l1 vector list as ONE lane's batched-inversion run (identities, doubling, cancellation inside)."""
    fx, E = ctx
    cts = [bytes.fromhex(e["ct"]) for e in fx["encrypt"]]
    a = [cts[v["a"]] for v in fx["l1"]]
    b = [cts[v["b"]] for v in fx["l1"]]
    assert [x.hex() for x in E.g1_add(a, b)] == [v["add"] for v in fx["l1"]]
    assert [x.hex() for x in E.g1_add(a, b, True)] == [v["sub"] for v in fx["l1"]]


def test_emu_gt_ops(ctx):
    fx, E = ctx
    n = int(fx["n"], 16)
    l2 = [bytes.fromhex(v["out"]) for v in fx["mult"]]
    for v in fx["l2"]:
        assert E.gt_mul(l2[v["a"]], l2[v["b"]]).hex() == v["add"]
        assert E.gt_mul(l2[v["a"]], l2[v["b"]], True).hex() == v["sub"]
    for v in fx["multconst_l2"]:
        assert E.gt_pow(l2[v["a"]], int(v["k"], 16), (n.bit_length() + 7) // 8).hex() == v["out"]
