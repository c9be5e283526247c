#!/usr/bin/env python3
"""bench.py — EMult (Type-A1 Tate pairing) throughput on MI355X.

Metric (BASELINE.json): EMult pairings/sec at 1024-bit params, batch = 2^20 per
GPU.  One "step" = one pass of pk.Mult over a batch of 2^20 pairs of level-1
ciphertexts: PBC wire bytes resident in HBM -> wire bytes resident in HBM
(decode, Miller loop + final exponentiation, encode), then — when more than
one GPU takes part — the RCCL all-gather of the result arrays named by the
north star.  Batches shard by contiguous ranges, one process per GPU, no
collective on the data path other than that gather (scaling: weak, 2^20 per GPU).

Usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--batch-log2 B]
For N > 1 launch through torch.distributed.run (RANK/LOCAL_RANK/WORLD_SIZE).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# profiles/ubench_valu_rates_r01.txt: v_mad_u64_u32, 4 waves/SIMD: 454.75 G wave-instr/s
VALU_MAD_PEAK = 454.75e9 * 64  # lane-MADs per second, whole chip


def cpu_baseline(fx, a_host, b_host, gpu_out_host, seconds=12.0):
    """The oracle timed on this box's host cores on a bounded sample of the same
    workload: the first pairs of the very batch the GPU just processed.  Its
    outputs are compared byte for byte with the GPU's.  Test-infrastructure code
    used as the reported baseline and checker only; never on the product path."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import oracle_c
        if oracle_c.available():
            return oracle_c.bench_pairings(fx, a_host, b_host, gpu_out_host, seconds)
    except ImportError:
        pass
    import bgn_ref as R
    from conftest import oracle_key
    opk, _ = oracle_key(fx)
    EB = 2 * R.fp_len(opk.p)
    npairs = len(a_host) // EB
    n, ok, t0 = 0, True, time.time()
    while time.time() - t0 < seconds and n < npairs:
        A = R.elem_from_bytes(a_host[n * EB:(n + 1) * EB], opk.p)
        B = R.elem_from_bytes(b_host[n * EB:(n + 1) * EB], opk.p)
        ok &= R.elem_to_bytes(opk.e(A, B), opk.p) == gpu_out_host[n * EB:(n + 1) * EB]
        n += 1
    dt = time.time() - t0
    return {"value": n / dt, "unit": "pairings/s", "cores": 1, "kind": "port",
            "sample": f"first {n} pairs of the GPU batch, pure-Python big-int oracle (oracle/bgn_ref.py), "
                      f"single thread, {dt:.1f} s", "matches_gpu_bit_exact": bool(ok)}


def config0_metrics(no_cpu: bool):
    """BASELINE configs[0]: 512-bit params, 128 ciphertexts, pk.Add and pk.Mult over 128 independent pairs — the
    shape of BenchmarkAdd / BenchmarkMult (bgn_test.go:97-140), which the reference runs on one goroutine.  Host
    buffers in and out (the size at which the boundary copies matter), next to the single-threaded C oracle on the
    same 128 pairs."""
    import numpy as np
    from conftest import load_fixture
    import bgn_amd
    fx = load_fixture("k512")
    pk = bgn_amd.PublicKey(int(fx["p"], 16), int(fx["n"], 16), fx["l"], bytes.fromhex(fx["P"]), bytes.fromhex(fx["Q"]),
                           fx["msg_space"], True, fx["poly_base"])
    eng = pk.engine
    rng = np.random.default_rng(7)
    n, nn = 128, int(fx["n"], 16)
    xs = [int(v) for v in rng.integers(0, 1021, 2 * n)]
    rs = [int.from_bytes(rng.bytes(60), "big") % nn for _ in range(2 * n)]
    cts = eng.encrypt(xs, rs)
    a, b = cts[:n].tobytes(), cts[n:].tobytes()
    out = {}
    for name, fn in (("eadd", lambda: eng.add(1, a, b)), ("emult", lambda: eng.mult(a, b))):
        fn()
        t0 = time.perf_counter()
        res = fn()
        dt = time.perf_counter() - t0
        out[name] = {"value": n / dt, "unit": "ops/s", "wall_ms_for_128": dt * 1e3, "result": res.tobytes()}
    if not no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        try:
            import oracle_c
            if oracle_c.available():
                orc = oracle_c.Oracle.from_fixture(fx)
                for name, fn in (("eadd", lambda: orc.add(1, a, b)), ("emult", lambda: orc.mult(a, b))):
                    t0 = time.perf_counter()
                    ref = fn()
                    dt = time.perf_counter() - t0
                    out[name]["cpu_single_thread_ops_per_s"] = n / dt
                    out[name]["matches_cpu_bit_exact"] = bool(ref == out[name]["result"])
        except (ImportError, AttributeError):
            pass
    for name in out:
        del out[name]["result"]
    out["workload"] = "configs[0]: 512-bit params, 128-ciphertext EAdd + EMult, host buffers through the C ABI"
    return out


def secondary_metrics(pk, fx, dev, dec_log2s):
    """BASELINE configs[1] (Encrypt) and configs[3] (BSGS Decrypt, T = 2^40, batch 2^16), reported next to the
    headline value.  Inputs resident in HBM; one warm-up pass then one timed pass each."""
    import numpy as np
    import torch
    import bgn_amd
    eng = pk.engine
    EB = eng.elem_bytes
    out = {}
    # --- Encrypt: configs[1]: 2^20 messages m uniform in [0, 2^40), r uniform below 2^1022
    n_enc = 1 << 20
    g = torch.Generator(device="cpu")
    g.manual_seed(4242)
    xs = torch.randint(0, 256, (n_enc, 5), dtype=torch.uint8, generator=g).to(dev)          # 40-bit plaintexts
    rs = torch.randint(0, 256, (n_enc, 128), dtype=torch.uint8, generator=g)
    rs[:, 0] &= 0x3F                                                                          # r < 2^1022 < n
    rs = rs.to(dev)
    cts = torch.empty(n_enc * EB, dtype=torch.uint8, device=dev)
    for it in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.encrypt_dev(xs, 5, rs, 128, cts, n_enc)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    out["encrypt"] = {"value": n_enc / dt, "unit": "encrypts/s", "batch": n_enc,
                      "workload": "configs[1]: batch=2^20 Encrypt P^m * Q^r, 40-bit m, 1022-bit r, fixed-base: one entry of the "
                                  "16-bit window tables of P and Q (HBM, 1.3 GB each) per window, affine additions over four "
                                  "accumulation chains per element (one launch adds four windows), one inversion per run of 64",
                      "kernel": eng.last_kernel_name(), "kernel_ms_per_step": eng.last_kernel_ms(),
                      "algorithmic_bytes_per_unit": 5 + 128 + EB}
    # --- EAdd (level 1): pairs of those ciphertexts
    n_add = n_enc // 2
    a1, b1 = cts[: n_add * EB], cts[n_add * EB: 2 * n_add * EB]
    o1 = torch.empty(n_add * EB, dtype=torch.uint8, device=dev)
    for it in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.add_dev(1, a1, b1, o1, n_add)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    out["eadd_l1"] = {"value": n_add / dt, "unit": "adds/s", "batch": n_add,
                      "workload": "pk.Add on level-1 ciphertexts (affine G1 addition, batched inversion)",
                      "algorithmic_bytes_per_unit": 3 * EB, "achieved_GBps": 3 * EB * n_add / dt / 1e9}
    # --- MultPoly: configs[4] shape (16x16 coefficient polynomials), 2^12 polynomials = 2^20 coefficient pairs
    npoly, d1, d2 = 1 << 12, 16, 16
    pa = cts[: npoly * d1 * EB]
    pb = cts[npoly * d1 * EB: npoly * (d1 + d2) * EB]
    po = torch.empty(npoly * (d1 + d2) * EB, dtype=torch.uint8, device=dev)
    for it in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.poly_mult_dev(npoly, d1, d2, pa, pb, po)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    out["multpoly"] = {"value": npoly * d1 * d2 / dt, "unit": "coefficient pairs/s", "polys": npoly, "d1": d1, "d2": d2,
                       "workload": "configs[4] shape on one GPU: MultPoly of 16x16-coefficient ciphertext polynomials "
                                   "(d1*d2 pairings + segmented GT accumulation), sharded by polynomial across GPUs"}
    # --- Decrypt: the first 2^k of those ciphertexts, every 16th negated; k = configs[3]'s 2^16 and the metric's 2^20
    t0 = time.perf_counter()
    pk.SetupDecryption(bgn_amd.SecretKey(int(fx["q1"], 16)))
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t0
    neg = torch.empty_like(cts)
    eng._lib.bgn_neg_batch_dev(eng._h, n_enc, 1, cts.data_ptr(), neg.data_ptr(), eng._stream())
    mixed = cts.view(n_enc, EB).clone()
    mixed[::16] = neg.view(n_enc, EB)[::16]
    del neg
    want_all = torch.zeros(n_enc, dtype=torch.int64)
    xb = xs.cpu().numpy().astype(np.int64)
    for j in range(5):
        want_all = want_all * 256 + torch.from_numpy(xb[:, j])
    want_all[::16] = -want_all[::16]
    for k in sorted(set(dec_log2s)):
        n_dec = 1 << k
        sel = mixed[:n_dec].reshape(-1).contiguous()
        m = torch.empty(n_dec, dtype=torch.int64, device=dev)
        st = torch.empty(n_dec, dtype=torch.uint8, device=dev)
        for it in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.decrypt_dev(1, sel, m, st, n_dec)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        k_ms = eng.last_kernel_ms()
        ok = bool((m.cpu() == want_all[:n_dec]).all().item()) and not bool(st.any().item())
        out["decrypt" if k == 16 else "decrypt_2^%d" % k] = {
            "value": n_dec / dt, "unit": "decrypts/s", "batch": n_dec, "level": 1,
            "workload": "configs[3]: T=2^40 BSGS Decrypt, batch=2^%d, m uniform in [0,2^40), 1/16 negative; Miller loop "
                        "over the secret order's line table + final exponentiation + ^sk, then giant steps 2S apart "
                        "on an HBM-resident baby table (%d entries)" % (k, int(eng._lib.bgn_ctx_bsgs_baby_steps(eng._h))),
            "search_kernel_ms": k_ms, "kernel": eng.last_kernel_name(), "table_setup_s": t_setup,
            "plaintexts_recovered_exactly": ok, "algorithmic_bytes_per_unit": EB + 16}
    # --- Decrypt of level-2 ciphertexts (configs[3] asks for both levels): products of 20-bit messages
    n2 = 1 << 16
    xs2 = torch.randint(0, 256, (2 * n2, 3), dtype=torch.uint8, generator=g)
    xs2[:, 0] &= 0x0F                                                                         # 20-bit plaintexts
    xs2 = xs2.to(dev)
    c2 = torch.empty(2 * n2 * EB, dtype=torch.uint8, device=dev)
    eng.encrypt_dev(xs2, 3, rs[: 2 * n2], 128, c2, 2 * n2)
    l2 = torch.empty(n2 * EB, dtype=torch.uint8, device=dev)
    eng.mult_dev(c2[: n2 * EB], c2[n2 * EB:], l2, n2)
    m = torch.empty(n2, dtype=torch.int64, device=dev)
    st = torch.empty(n2, dtype=torch.uint8, device=dev)
    for it in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.decrypt_dev(2, l2, m, st, n2)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    xv = xs2.cpu().numpy().astype(np.int64)
    val = (xv[:, 0] * 65536 + xv[:, 1] * 256 + xv[:, 2])
    ok = bool((m.cpu().numpy() == val[:n2] * val[n2:]).all()) and not bool(st.any().item())
    out["decrypt_l2"] = {"value": n2 / dt, "unit": "decrypts/s", "batch": n2, "level": 2,
                         "workload": "configs[3], level 2: Decrypt of 2^16 products of two 20-bit messages (outputs of Mult): "
                                     "^sk by the norm-1 ladder, then the same giant-step walk",
                         "plaintexts_recovered_exactly": ok}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-log2", type=int, default=20)
    ap.add_argument("--key", default="k1024")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary Encrypt / Decrypt measurements")
    ap.add_argument("--decrypt-log2", type=int, nargs="+", default=[16, 20],
                    help="batch sizes (log2, at most 20) of the secondary Decrypt measurement")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force_dist = os.environ.get("BGN_BENCH_FORCE_DIST") == "1"     # exercise the RCCL path with one rank
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from conftest import load_fixture
    import bgn_amd
    import bgn_amd.synthetic
    from bgn_amd.sharding import shard_range

    fx = load_fixture(args.key)
    pk = bgn_amd.PublicKey(int(fx["p"], 16), int(fx["n"], 16), fx["l"], bytes.fromhex(fx["P"]),
                           bytes.fromhex(fx["Q"]), fx["msg_space"], True, fx["poly_base"], device=local_rank)
    eng = pk.engine
    EB = eng.elem_bytes
    per_gpu = 1 << args.batch_log2
    total = per_gpu * world
    lo, hi = shard_range(total, world, rank)           # contiguous slice of the global batch
    count = hi - lo

    # Synthetic level-1 ciphertexts, resident in HBM before the timed region.
    a, b = bgn_amd.synthetic.l1_ciphertext_pairs(pk, fx, count, seed=1000 + rank, device=dev)
    out = torch.empty(count * EB, dtype=torch.uint8, device=dev)
    use_dist = world > 1 or force_dist
    gathered = torch.empty(total * EB, dtype=torch.uint8, device=dev) if use_dist else None

    def step():
        eng.mult_dev(a, b, out, count)
        if use_dist:
            dist.all_gather_into_tensor(gathered, out)

    kernel_ms = []
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kernel_ms.append(eng.last_kernel_ms())          # HIP events on the kernel's own stream
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kernel_name = eng.last_kernel_name()
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        assert bool((gathered[rank * count * EB:(rank + 1) * count * EB] == out).all().item()), "gather mismatch"

    # prefix of this rank's batch for the CPU leg (not timed)
    nchk = min(count, 4096)
    a_h = a[: nchk * EB].cpu().numpy().tobytes()
    b_h = b[: nchk * EB].cpu().numpy().tobytes()
    o_h = out[: nchk * EB].cpu().numpy().tobytes()

    extra = None
    if not args.no_extra and world == 1 and args.key == "k1024":
        extra = secondary_metrics(pk, fx, dev, [min(k, 20) for k in args.decrypt_log2])
        if rank == 0:
            extra["config0_512bit_128"] = config0_metrics(args.no_cpu_baseline)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = total * args.steps / dt
        k_ms = sum(kernel_ms) / len(kernel_ms)
        alg_bytes = 3 * EB                               # two G1 operands in, one GT element out
        achieved = alg_bytes * count / (k_ms * 1e-3) / 1e9
        mads = bgn_amd.synthetic.algorithmic_mads_per_pairing(
            fx, run=max(1, min(16, count // 65536)), window={"0": 2, "3": 3, "4": 4}.get(os.environ.get("BGN_MILLER_WINDOW", ""), 5))
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "r01e_pmc_summary.json")
        if os.path.exists(pmc) and args.batch_log2 == 20 and args.key == "k1024":
            with open(pmc) as f:
                traffic = json.load(f)["hbm_bytes_per_launch"]     # separate rocprofv3 --pmc passes of this command
            traffic_src = "profiles/r01e_pmc_summary.json (FETCH_SIZE + WRITE_SIZE, KB * 1024, per launch of 2^20 pairings)"
        line = {
            "metric": "EMult pairings/sec at 1024-bit, batch=2^%d per GPU" % args.batch_log2,
            "value": value, "unit": "pairings/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 (28-bit limbs, 64-bit accumulators)", "data": "synthetic",
            "config": {"workload": "configs[2]: 1024-bit params, batch=2^%d EMult (Tate pairing G1xG1->GT, "
                                   "Miller+final-exp) per MI355X" % args.batch_log2,
                       "key": fx["name"], "fp_bits": int(fx["p"], 16).bit_length(), "limbs28": 38,
                       "batch_per_gpu": per_gpu, "global_batch": total,
                       "parallelism": "batch-sharded x%d + RCCL all-gather of results" % world if world > 1 else "single GPU",
                       },
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernel_name, "kernel_ms": k_ms,
                         "algorithmic_bytes_per_pairing": alg_bytes},
            "roofline_valu": {"bound": "v_mad_u64_u32 issue", "mads_per_pairing": mads,
                              "achieved": mads * count / (k_ms * 1e-3), "peak": VALU_MAD_PEAK,
                              "unit": "lane-MAD/s", "frac": mads * count / (k_ms * 1e-3) / VALU_MAD_PEAK},
        }
        if extra:
            line["extra"] = extra
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(fx, a_h, b_h, o_h)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
