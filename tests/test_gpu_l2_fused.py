"""GPU: level-2 Add / Sub in one wire-to-wire launch (k_gt_mul_wire: barrett.hpp's F_p^2 product of plain residues
between the dword-stream codec's decode and encode) — bgn.go:455-475 (Add, level-2 branch), :392-412 (Sub).
Against the golden vectors, the C oracle, the four-launch Montgomery route it replaces, and Python integers."""
import random

import numpy as np
import pytest

import bgn_amd.synthetic as syn
from conftest import KEYS, engine_key, load_fixture

pytestmark = pytest.mark.gpu


def H(hexes):
    return b"".join(bytes.fromhex(h) for h in hexes)


@pytest.mark.parametrize("name", KEYS)
def test_l2_add_sub_golden_through_the_fused_kernel(name):
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    eng = pk.engine
    l2 = [v["out"] for v in fx["mult"]]
    a, b = H([l2[v["a"]] for v in fx["l2"]]), H([l2[v["b"]] for v in fx["l2"]])
    fused = syn.limbs_for(int(fx["p"], 16)) <= 40
    for fn, key in [(eng.add, "add"), (eng.sub, "sub")]:
        got = fn(2, a, b)
        assert eng.last_kernel_name() == ("k_gt_mul_wire" if fused else "k_gt_mul")
        for row, v in zip(got, fx["l2"]):
            assert bytes(row).hex() == v[key], f"{name}: L2 {key}({v['a']},{v['b']})"


@pytest.mark.parametrize("name,count", [("toy64", 70001), ("k256", 3000), ("k1024", 777), ("k1024b", 300)])
def test_l2_add_large_ragged_vs_c_oracle(name, count):
    """Several workgroups and a ragged tail; operands from a pool of products (GT elements) plus the GT identity
    (1, 0): Add and Sub byte for byte against the C oracle and against the four-launch route."""
    import oracle_c
    fx = load_fixture(name)
    o = oracle_c.Oracle.from_fixture(fx)
    pk, _ = engine_key(fx)
    eng = pk.engine
    EB = eng.elem_bytes
    rng = random.Random(43)
    cts = [bytes.fromhex(e["ct"]) for e in fx["encrypt"]]
    nz = [c for c in cts if any(c)][:6]
    pool = [o.mult(nz[i], nz[(i * 2 + 1) % len(nz)]) for i in range(len(nz))]
    pool.append((1).to_bytes(EB // 2, "big") + bytes(EB // 2))          # the identity of GT
    P = np.frombuffer(b"".join(pool), dtype=np.uint8).reshape(len(pool), EB)
    ia = np.array([rng.randrange(len(pool)) for _ in range(count)])
    ib = np.array([rng.randrange(len(pool)) for _ in range(count)])
    a, b = P[ia].reshape(-1).tobytes(), P[ib].reshape(-1).tobytes()
    want = {(i, k): (o.add(2, pool[i], pool[k]), o.add(2, pool[i], pool[k], True))
            for i in range(len(pool)) for k in range(len(pool))}
    wa = b"".join(want[(int(i), int(k))][0] for i, k in zip(ia, ib))
    wsub = b"".join(want[(int(i), int(k))][1] for i, k in zip(ia, ib))
    got_a, got_s = eng.add(2, a, b).tobytes(), eng.sub(2, a, b).tobytes()
    assert eng.last_kernel_name() == "k_gt_mul_wire"
    assert got_a == wa
    assert got_s == wsub
    eng.set_option("l2_fused", 0)
    try:
        assert eng.add(2, a, b).tobytes() == wa and eng.last_kernel_name() == "k_gt_mul"
        assert eng.sub(2, a, b).tobytes() == wsub
    finally:
        eng.set_option("l2_fused", 1)


@pytest.mark.parametrize("name", ["toy64", "k256", "k1024"])
def test_l2_fused_on_any_residues_vs_python_integers(name):
    """The kernel computes in F_p^2 whatever the operands are (it does not need norm 1): random residues, the edge
    values 0, 1, p - 1, and residues at or above p (reduced first), against Python integers."""
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    eng = pk.engine
    p = int(fx["p"], 16)
    Lb = eng.elem_bytes // 2
    rng = random.Random(11)
    edge = [0, 1, p - 1, p - 2, 2]
    top = min(1 << (8 * Lb), 1 << (29 * syn.limbs_for(p)))
    rows = [(a0, a1, b0, b1) for a0 in edge for a1 in edge[:3] for b0 in edge[:3] for b1 in edge[:4]]
    rows += [tuple(rng.randrange(p) for _ in range(4)) for _ in range(300)]
    big = [p, p + 1, top - 1, top - 2, 2 * p + 5 if 2 * p + 5 < top else p + 5]
    rows += [tuple(rng.choice(big) if rng.random() < 0.5 else rng.randrange(p) for _ in range(4)) for _ in range(80)]
    enc = lambda x, y: x.to_bytes(Lb, "big") + y.to_bytes(Lb, "big")
    a = b"".join(enc(r[0], r[1]) for r in rows)
    b = b"".join(enc(r[2], r[3]) for r in rows)
    add = eng.add(2, a, b)
    sub = eng.sub(2, a, b)
    assert eng.last_kernel_name() == "k_gt_mul_wire"
    for row_a, row_s, (a0, a1, b0, b1) in zip(add, sub, rows):
        assert bytes(row_a) == enc((a0 * b0 - a1 * b1) % p, (a0 * b1 + a1 * b0) % p), (hex(a0), hex(a1), hex(b0), hex(b1))
        assert bytes(row_s) == enc((a0 * b0 + a1 * b1) % p, (a1 * b0 - a0 * b1) % p)


@pytest.mark.parametrize("name", ["k256", "k1024"])
def test_l2_fused_on_misaligned_device_buffers(name):
    """Operand and result arrays that start 1, 2 and 3 bytes into a dword (callers pass sub-ranges of buffers): operand
    slices are staged at their own misalignment and decoded there; a misaligned result array is served by the
    four-launch route."""
    import torch
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    eng = pk.engine
    EB = eng.elem_bytes
    l2 = [bytes.fromhex(v["out"]) for v in fx["mult"]]
    n = 301
    a = b"".join(l2[i % len(l2)] for i in range(n))
    b = b"".join(l2[(3 * i + 1) % len(l2)] for i in range(n))
    want = eng.add(2, a, b).tobytes()
    dev = torch.device("cuda", 0)
    for ma, mb, mo in [(1, 0, 0), (0, 2, 0), (3, 1, 0), (2, 3, 0), (0, 0, 3), (3, 1, 2), (2, 2, 2)]:
        ta = torch.zeros(n * EB + 8, dtype=torch.uint8, device=dev)
        tb = torch.zeros(n * EB + 8, dtype=torch.uint8, device=dev)
        to = torch.full((n * EB + 8,), 0xEE, dtype=torch.uint8, device=dev)
        ta[ma:ma + n * EB] = torch.frombuffer(bytearray(a), dtype=torch.uint8).to(dev)
        tb[mb:mb + n * EB] = torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
        eng.add_dev(2, ta[ma:ma + n * EB], tb[mb:mb + n * EB], to[mo:mo + n * EB], n)
        torch.cuda.synchronize()
        # (a result array that does not start on a dword takes the four-launch route: the fused kernel's encoder
        # writes whole aligned words; operand arrays may start anywhere)
        assert eng.last_kernel_name() == ("k_gt_mul_wire" if mo == 0 else "k_gt_mul")
        host = to.cpu().numpy().tobytes()
        assert host[mo:mo + n * EB] == want, (ma, mb, mo)
        assert host[:mo] == b"\xee" * mo and host[mo + n * EB:] == b"\xee" * (8 - mo), "bytes outside the result written"


def test_fused_kernel_holds_everything_in_registers():
    """The fused kernel's point is two workgroups per CU with nothing in scratch memory: what the code object says
    (hipFuncGetAttributes through bgn_last_kernel_resources).  A change that makes the register allocator spill
    shows here before it shows on a clock (round 6: a third 36-limb constant in scalar registers cost 230 spilled
    vector registers and a quarter of the rate)."""
    fx = load_fixture("k1024")
    pk, _ = engine_key(fx)
    eng = pk.engine
    l2 = [bytes.fromhex(v["out"]) for v in fx["mult"]]
    eng.add(2, l2[0], l2[1])
    assert eng.last_kernel_name() == "k_gt_mul_wire"
    res = eng.last_kernel_resources()
    assert res["scratch_bytes_per_lane"] == 0, res
    assert res["vgprs"] <= 256 and res["lds_bytes_per_workgroup"] <= 80 * 1024, res


def test_code_object_resources_of_the_lane_kernels():
    """What the code objects of the kernels behind the bench line say (hipFuncGetAttributes through
    bgn_last_kernel_resources, by the name bgn_last_kernel_name reports): the 512-register kernels hold at most the
    known few hundred bytes of scratch, the one-launch codec kernels none.  Written to
    gpurun_out/r06_kernel_resources.json (tracked copy: profiles/)."""
    import json
    import os
    import random
    from conftest import ROOT
    fx = load_fixture("k1024")
    pk, _ = engine_key(fx)
    eng = pk.engine
    rng = random.Random(5)
    cts = [bytes.fromhex(e["ct"]) for e in fx["encrypt"]]
    l2 = [bytes.fromhex(v["out"]) for v in fx["mult"]]
    n = 66000                                                    # past every lane-group crossover: the lane kernels
    a1 = b"".join(cts[i % len(cts)] for i in range(n))
    a2 = b"".join(l2[i % len(l2)] for i in range(n))
    ks = [rng.randrange(1 << 40) for _ in range(n)]
    order = int(fx["n"], 16)
    rs = [rng.randrange(order) for _ in range(n)]                # full-length randomness: the chain kernel
    calls = {
        "mult": lambda: eng.mult(a1, a1),
        "add_l1": lambda: eng.add(1, a1, a1),
        "add_l2": lambda: eng.add(2, a2, a2),
        "neg_l1": lambda: eng.neg(1, a1),
        "multconst_l1": lambda: eng.multconst(1, a1, ks),
        "multconst_l2": lambda: eng.multconst(2, a2, ks),
        "encrypt": lambda: eng.encrypt(ks, rs),
    }
    want = {"mult": "k_pairing<36, 0>", "add_l1": "k_g1_add_wire", "add_l2": "k_gt_mul_wire", "neg_l1": "k_neg_wire",
            "multconst_l1": "k_g1_mul", "multconst_l2": "k_gt_pow", "encrypt": "k_g1_fixed_chain"}
    out = {}
    for op, fn in calls.items():
        fn()
        name = eng.last_kernel_name()
        assert name == want[op], (op, name)
        out[op] = dict(eng.last_kernel_resources(), kernel=name)
    for op in ("add_l2", "neg_l1"):
        assert out[op]["scratch_bytes_per_lane"] == 0, out[op]
    for op, r in out.items():
        assert r["scratch_bytes_per_lane"] <= 1024 and r["lds_bytes_per_workgroup"] <= 160 * 1024, (op, r)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06_kernel_resources.json"), "w") as f:
        json.dump(out, f, indent=1)
