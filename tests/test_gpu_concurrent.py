"""GPU: the reference's own call shape — many host threads, each calling a single-element method on one key
(poly.go:139-153 runs one goroutine per coefficient pair around pk.Mult; poly.go:97-109 around pk.MultConst;
bgn_test.go:97-140 benchmark one op per call).  The host-buffer entry points merge such calls into one launch per
kind of call (csrc/combiner.hpp); every caller must get exactly the bytes a lone call returns — here checked
against the C oracle — and per-context options must let two keys run different kernels side by side."""
import os
import random
import threading

import numpy as np
import pytest

from conftest import engine_key, load_fixture

pytestmark = pytest.mark.gpu


def _inputs(fx, eng, n, seed):
    rng = random.Random(seed)
    nn, T = int(fx["n"], 16), fx["msg_space"]
    xs = [rng.randrange(T) for _ in range(2 * n)]
    cts = eng.encrypt(xs, [rng.randrange(nn) for _ in xs])
    return xs, cts[:n].copy(), cts[n:].copy()


@pytest.mark.parametrize("name", ["k256", "k1024"])
def test_64_threads_of_mixed_single_element_calls_equal_the_c_oracle(name):
    import oracle_c
    fx = load_fixture(name)
    pk, sk = engine_key(fx)
    pk.SetupDecryption(sk)
    eng = pk.engine
    o = oracle_c.Oracle.from_fixture(fx)
    nthreads, per = 64, 3
    n = nthreads * per
    xs, A, B = _inputs(fx, eng, n, 5)
    E = eng.elem_bytes
    ks = [random.Random(9).randrange(1, 1 << 40) for _ in range(n)]
    before = eng.combiner_stats()
    got, errs = {}, []
    start = threading.Barrier(nthreads)

    def worker(t):
        try:
            start.wait()
            for j in range(per):
                i = t * per + j
                a, b = A[i].tobytes(), B[i].tobytes()
                op = (t + j) % 4
                if op == 0:
                    got[i] = ("mult", eng.mult(a, b).tobytes())
                elif op == 1:
                    got[i] = ("add", eng.add(1, a, b).tobytes())
                elif op == 2:
                    m, st = eng.decrypt(1, a)
                    got[i] = ("decrypt", (int(m[0]), int(st[0])))
                else:
                    got[i] = ("multconst", eng.multconst(1, a, [ks[i]]).tobytes())
        except Exception as e:                                    # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs[:3]
    assert len(got) == n
    for i in range(n):
        op, val = got[i]
        a, b = A[i].tobytes(), B[i].tobytes()
        if op == "mult":
            assert val == o.mult(a, b), i
        elif op == "add":
            assert val == o.add(1, a, b), i
        elif op == "decrypt":
            assert val == (xs[i], 0), i
        else:
            assert val == o.multconst(1, a, [ks[i]]), i
    after = eng.combiner_stats()
    calls = after["calls"] - before["calls"]
    groups = after["groups"] - before["groups"]
    assert calls == n
    assert groups < calls, "no two concurrent calls were ever merged: %r -> %r" % (before, after)
    assert after["max_group"] > 1


def test_combined_calls_return_the_bytes_of_lone_calls_and_of_the_uncombined_path():
    """Counts above one, blinded and unblinded requests of the same operation (different kinds: never merged into one
    batch), level 2, Sub / Neg / makeL2 / Encrypt: the same bytes with the combiner, without it (combine = 0) and from
    one batch call."""
    fx = load_fixture("k512")
    pk, sk = engine_key(fx)
    pk.SetupDecryption(sk)
    eng = pk.engine
    n = 48
    xs, A, B = _inputs(fx, eng, n, 11)
    nn = int(fx["n"], 16)
    rng = random.Random(2)
    rs = [rng.randrange(nn) for _ in range(n)]
    L2 = eng.make_l2(A.tobytes())
    want = {
        "mult": eng.mult(A.tobytes(), B.tobytes()), "mult_r": eng.mult(A.tobytes(), B.tobytes(), rs),
        "sub": eng.sub(1, A.tobytes(), B.tobytes()), "neg": eng.neg(1, A.tobytes()), "l2": L2,
        "add2": eng.add(2, L2.tobytes(), L2.tobytes()), "enc": eng.encrypt(xs[:n], rs),
        "mc2": eng.multconst(2, L2.tobytes(), [7 + i for i in range(n)]),
    }
    for combine in (1, 0):
        with eng.options(combine=combine):
            got = {k: [None] * n for k in want}
            errs = []

            def worker(lo, hi):
                try:
                    a, b = A[lo:hi].tobytes(), B[lo:hi].tobytes()
                    l2 = L2[lo:hi].tobytes()
                    res = {"mult": eng.mult(a, b), "mult_r": eng.mult(a, b, rs[lo:hi]), "sub": eng.sub(1, a, b),
                           "neg": eng.neg(1, a), "l2": eng.make_l2(a), "add2": eng.add(2, l2, l2),
                           "enc": eng.encrypt(xs[lo:hi], rs[lo:hi]), "mc2": eng.multconst(2, l2, [7 + i for i in range(lo, hi)])}
                    for k, v in res.items():
                        for i in range(lo, hi):
                            got[k][i] = v[i - lo].tobytes()
                except Exception as e:                            # noqa: BLE001
                    errs.append(repr(e))

            cuts = [0, 1, 2, 5, 6, 13, 20, 21, 33, 40, 47, 48]     # requests of 1 .. 12 elements
            th = [threading.Thread(target=worker, args=(lo, hi)) for lo, hi in zip(cuts, cuts[1:])]
            for t in th:
                t.start()
            for t in th:
                t.join()
            assert not errs, errs[:3]
            for k, w in want.items():
                for i in range(n):
                    assert got[k][i] == w[i].tobytes(), (combine, k, i)


def test_a_failing_call_does_not_poison_its_neighbours():
    """A request whose arguments are refused (level 3) fails alone, before it reaches the combiner; concurrent valid
    requests complete."""
    import bgn_amd
    fx = load_fixture("k256")
    pk, _ = engine_key(fx)
    eng = pk.engine
    _, A, B = _inputs(fx, eng, 8, 3)
    want = eng.add(1, A.tobytes(), B.tobytes())
    res, bad = [None] * 8, []

    def ok(i):
        res[i] = eng.add(1, A[i].tobytes(), B[i].tobytes()).tobytes()

    def broken():
        try:
            eng.add(3, A[0].tobytes(), B[0].tobytes())
        except bgn_amd.BgnError as e:
            bad.append(e.code)

    th = [threading.Thread(target=ok, args=(i,)) for i in range(8)] + [threading.Thread(target=broken) for _ in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert bad == [-1] * 4
    assert res == [w.tobytes() for w in want]


def test_two_contexts_run_different_forced_kernels_concurrently():
    """Options are per context: two keys in one process, one forced onto the lane-group kernel and one onto the
    cooperative kernel, multiply at the same time from two threads; each reports its own kernel and both give the
    golden bytes.  (With environment switches this was impossible: one process, one setting.)"""
    fa, fb = load_fixture("k256"), load_fixture("k512")
    (pka, _), (pkb, _) = engine_key(fa), engine_key(fb)
    ea, eb = pka.engine, pkb.engine
    ea.force_kernel("quad")
    eb.force_kernel("coop")
    try:
        out, names, errs = {}, {}, []

        def run(tag, eng, fx):
            try:
                cts = [bytes.fromhex(e["ct"]) for e in fx["encrypt"]]
                a = b"".join(cts[v["a"]] for v in fx["mult"])
                b = b"".join(cts[v["b"]] for v in fx["mult"])
                for _ in range(5):
                    out[tag] = eng.mult(a, b)
                    names.setdefault(tag, set()).add(eng.last_kernel_name())
            except Exception as e:                                # noqa: BLE001
                errs.append(repr(e))

        th = [threading.Thread(target=run, args=("a", ea, fa)), threading.Thread(target=run, args=("b", eb, fb))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        assert all("quad" in n for n in names["a"]) and all("coop" in n for n in names["b"]), names
        for tag, fx in (("a", fa), ("b", fb)):
            for row, v in zip(out[tag], fx["mult"]):
                assert bytes(row).hex() == v["out"]
        assert ea.get_option("quad_max") != eb.get_option("quad_max")
    finally:
        ea.force_kernel(None)
        eb.force_kernel(None)


def test_calibration_yields_ordered_crossovers_and_leaves_results_alone():
    """bgn_ctx_calibrate times two sizes of the cooperative and the lane-group kernel and one round of the lane kernel
    per operation and puts the crossovers where the fitted lines meet; dispatch afterwards still returns the golden
    bytes, and explicit options keep precedence."""
    fx = load_fixture("k512")
    pk, sk = engine_key(fx)
    pk.SetupDecryption(sk)
    eng = pk.engine
    xo = eng.calibrate()
    for mode in range(4):
        assert 64 <= xo["coop"][mode] <= 4096, xo
        assert xo["coop"][mode] <= xo["quad"][mode] <= 65536, xo
    cts = [bytes.fromhex(e["ct"]) for e in fx["encrypt"]]
    a = b"".join(cts[v["a"]] for v in fx["mult"])
    b = b"".join(cts[v["b"]] for v in fx["mult"])
    for row, v in zip(eng.mult(a, b), fx["mult"]):
        assert bytes(row).hex() == v["out"]
    assert "coop" in eng.last_kernel_name()
    with eng.options(coop_max=0, quad_max=0):
        eng.mult(a, b)
        assert "coop" not in eng.last_kernel_name() and "quad" not in eng.last_kernel_name()
    # the options of the context are what they were before the probes
    for k in ("coop_max", "quad_max", "quad_min", "coop_max_l2", "quad_max_l2", "coop_max_dec", "quad_max_dec"):
        assert eng.get_option(k) == -1
    n = xo["coop"][0] + 64                      # just above the calibrated cooperative crossover: the lane groups
    rng = np.random.default_rng(1)
    idx = rng.integers(0, len(cts), 2 * n)
    aa = b"".join(cts[i] for i in idx[:n])
    bb = b"".join(cts[i] for i in idx[n:])
    with eng.options(combine=0):
        eng.mult(aa, bb)
    assert "quad" in eng.last_kernel_name(), (xo, eng.last_kernel_name())


def test_native_caller_pool_through_the_c_abi():
    """tools/concurrent_callers.cpp — std::thread callers of the C ABI with no interpreter lock between them, what a
    Go host's goroutines look like to the library — for a second per operation at the 256-bit key: every caller's last
    result equals the batch call's, with the combiner on and off (the tool exits non-zero otherwise)."""
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, CC_SECONDS="1", CC_THREADS="1,48", CC_OPS="mult,add_l1,decrypt_l1,multconst_l1_k40,decrypt_l2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "concurrent_callers.py"), "k256"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    rows = [l.split(",") for l in r.stdout.strip().splitlines()[1:]]
    assert len(rows) == 5 * 4 and all(row[-1] == "True" for row in rows), r.stdout
    merged = [row for row in rows if row[3] == "48" and row[4] == "1"]
    assert all(float(row[7]) > 3 * float(solo[7]) for row in merged for solo in rows
               if solo[1] == row[1] and solo[3] == "1" and solo[4] == "1"), r.stdout      # 48 callers get well beyond one caller's rate
