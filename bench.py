#!/usr/bin/env python3
"""bench.py — EMult (Type-A1 Tate pairing) and BSGS Decrypt throughput on MI355X.

Metric (BASELINE.json): EMult pairings/sec + BSGS decrypts/sec at 1024-bit, batch = 2^20 per GPU.
One "step" = one pass of pk.Mult over a batch of 2^20 pairs of level-1 ciphertexts — SURVEY.md 8(d) Config 3:
the outputs of Config 2 (Encrypt of 2^20 random 40-bit messages with full-length randomness, produced on the GPU
before the timed region) paired with a fixed permutation of themselves — PBC wire bytes resident in HBM -> wire
bytes resident in HBM (decode, Miller loop + final exponentiation, encode), then — when more than one GPU takes
part — the RCCL all-gather of the result arrays named by the north star.  Batches shard by contiguous ranges, one
process per GPU, no collective on the data path other than that gather (scaling: weak, 2^20 per GPU).

Usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--workload emult|multpoly] [--batch-log2 B]
  --gpus N with WORLD_SIZE unset: this process spawns N fresh rank processes of itself (before anything touches
  the GPU) with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one per GPU, RCCL over xGMI between them.
  Under an external launcher (python -m torch.distributed.run ... bench.py --gpus N) the ranks come from the
  environment and --gpus must agree with WORLD_SIZE.
  --workload multpoly: BASELINE configs[4] — 2^14 MultPoly instances of 16x16 coefficient polynomials (2^22
  coefficient pairs) per job plus one AddPoly, sharded by polynomial across the GPUs (strong scaling).
  On one GPU at the BASELINE size the run first measures the HBM-side traffic its line quotes: two child passes of
  its own timed step under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`, started before this process touches
  the GPU (live_traffic; about 13 s; --no-live-traffic quotes the committed summary under profiles/ instead).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# profiles/r02_occupancy_rates.txt (tools/ubench/occupancy_rates.hip: exactly k waves on every SIMD, wall clock):
# v_mad_u64_u32 wave-instructions per second over the chip, mean of the two forms a Montgomery row alternates
# between (VGPR multiplicand 369.5 / 508.7 G, scalar multiplicand 427.3 / 544.1 G at 1 / 4 waves per SIMD).
# Round 1's table (ubench_valu_rates_r01.txt: 378 / 455 G) did not pin its occupancy and read low at 4 waves.
VALU_MAD_PEAK_4W = 526.4e9 * 64    # lane-MADs/s at 4 waves per SIMD (the chip's ceiling)
VALU_MAD_PEAK_1W = 398.4e9 * 64    # at 1 wave per SIMD — the occupancy these 512-register kernels run at
VALU_MAD_PEAK_2W = 466.8e9 * 64    # at 2 waves per SIMD (same file: 424.1 / 509.4 G): the fused level-2 Add


def check_rc(rc):
    if rc != 0:
        raise RuntimeError("engine call failed with %d" % rc)


def cpu_budget():
    """What the box lets this process use: affinity, and the cgroup CPU quota (cpu.max) when there is one."""
    info = {"nproc": os.cpu_count()}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                info["cgroup_cpu_max"] = " ".join(txt)
                if txt[0] != "max":
                    info["cgroup_cpus"] = float(txt[0]) / float(txt[1])
            else:
                q = int(txt[0])
                if q > 0:
                    p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    info["cgroup_cpus"] = q / p
            break
        except (OSError, ValueError, IndexError):
            continue
    # what a CPU leg can actually run on: the affinity mask, capped by the cgroup's CPU quota (16 on this pool's
    # boxes, whose affinity mask shows all 256 hardware threads).  cpu_baseline.cores is THIS number and the legs
    # start that many worker threads (`threads`).
    info["cores"] = max(1, min(info.get("affinity", info["nproc"] or 1), int(-(-info.get("cgroup_cpus", 1e9) // 1))))
    return info


def cpu_baseline(fx, a_host, b_host, gpu_out_host, seconds=12.0, sample_note=""):
    """The oracle timed on this box's host cores on a bounded sample of the same workload: pairs of the very batch
    the GPU just processed, strided over it so that every position of a lane's run of sixteen pairings is in the
    sample.  Its outputs are compared byte for byte with the GPU's.
    Test-infrastructure code used as the reported baseline and checker only; never on the product path."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    res = None
    try:
        import oracle_c
        if oracle_c.available():
            res = oracle_c.bench_pairings(fx, a_host, b_host, gpu_out_host, seconds)
    except ImportError:
        pass
    if res is None:
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
        res = {"value": n / dt, "unit": "pairings/s", "cores": 1, "kind": "port",
               "sample": f"first {n} pairs of the GPU batch, pure-Python big-int oracle (oracle/bgn_ref.py), "
                         f"single thread, {dt:.1f} s", "matches_gpu_bit_exact": bool(ok)}
    res["host_cpu"] = cpu_budget()      # `cores` = min(affinity, cgroup quota) = the worker threads started
    if sample_note:
        res["sample"] = res["sample"].replace("first ", "", 1) + "; " + sample_note
    return res


def decrypt_cpu_baseline(fx, mixed, want, want_st, n_dec, EB):
    """The second half of BASELINE's metric beside its CPU figure: 32 ciphertexts of the mixed Decrypt batch the GPU
    just processed — 31 strided over it (two of them negated) and one out of range — decrypted by the C oracle on
    the host cores (oracle/oracle_c.py bench_decrypt says which route and why).  Checker / baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import oracle_c
        if not oracle_c.available():
            return None
    except ImportError:
        return None
    import torch
    stride = n_dec // 32 + 1
    idx = [i * stride for i in range(31) if i * stride < n_dec]
    if 7 < n_dec and 7 not in idx:
        idx.append(7)                                    # decrypt_mix puts an out-of-range message at 7 mod 4096
    sel = torch.tensor(idx, dtype=torch.int64)
    ct = mixed.view(-1, EB)[sel.to(mixed.device)].cpu().numpy().tobytes()
    res = oracle_c.bench_decrypt(fx, ct, want[sel].tolist(), want_st[sel].tolist(), int(fx["msg_space"]))
    res["host_cpu"] = cpu_budget()
    return res


def secondary_cpu_leg(fx, n_items, call, want, out_bytes, unit, what, **kw):
    """cpu_baseline of a secondary entry (Encrypt, EAdd, MultPoly): the C oracle on the first items of the very batch
    the GPU just processed — single thread and one thread per usable core — outputs compared byte for byte
    (oracle/oracle_c.py bench_slices).  Checker / reported baseline only; None without the built oracle."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import oracle_c
        if not oracle_c.available():
            return None
    except ImportError:
        return None
    res = oracle_c.bench_slices(fx, n_items, call, want, out_bytes, unit, what, **kw)
    res["host_cpu"] = cpu_budget()
    return res


def committed_traffic(key=None):
    """HBM-side bytes (FETCH_SIZE / WRITE_SIZE) are hardware counters: rocprofv3 collects them in separate --pmc passes
    of THIS command (tools/collect_profiles.sh), they cannot be read from inside the process.  The line therefore
    quotes the newest committed summary under profiles/ and says so: a figure of an earlier run of the same command
    on the same build family, not of this run."""
    for name in ("r06_pmc_summary.json", "r05_pmc_summary.json", "r04_pmc_summary.json", "r03_pmc_summary.json", "r02_pmc_summary.json", "r01e_pmc_summary.json"):
        pmc = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(pmc):
            continue
        with open(pmc) as f:
            d = json.load(f)
        src = "profiles/%s: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed — NOT measured in this run" % name
        if key is None:
            return d.get("hbm_bytes_per_launch"), src
        node = d.get(key)
        if node is None:
            continue
        return node, src
    return None, None


def live_traffic(timeout_s=150.0, extras=False):
    """roofline.traffic of the headline kernel measured by THIS run (verdict r05, weak 6): before this process touches
    the GPU, two child passes of this command's own timed step — `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`,
    separate passes, no tracing domain, the program itself after `--` (MI355X_MICROARCH.md, HBM section) — over
    `bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra` (with `extras`: `--pmc-extras`, which adds ONE launch
    each of the secondary kernels whose traffic the line quotes — the two Adds of 2^20, Decrypt's lifts at 2^16 and
    2^20: pmc_extras).  FETCH_SIZE is calibrated in the same pass on
    k_encode's known read volume (2 NL x 4 bytes per element of limb-major SoA at 2^20 elements), as
    tools/summarize_profiles.py does for the committed summaries.  Returns None — and the line falls back to the
    committed summary, labelled as such — when rocprofv3 is missing, this process is itself being profiled, or a pass
    fails or runs over its time; the children are started in their own process group and that group is ended on a
    timeout."""
    import csv
    import glob
    import re
    import shutil
    import signal
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "").lower():
        return None                                     # under a profiler already: no nested passes
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("BGN_BENCH_SPAWN", "BGN_BENCH_FORCE_DIST", "WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        tmp = tempfile.mkdtemp(prefix="bgn_pmc_", dir="/tmp")
    except OSError:
        return None
    t0 = time.perf_counter()
    kb, calib, sec = {}, None, {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr.lower())
            cmd = [exe, "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                   "--no-extra", "--no-live-traffic"] + (["--pmc-extras"] if extras else [])
            proc = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                    start_new_session=True)
            try:
                rc = proc.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)          # the group started above: rocprofv3 and its python child
                except OSError:
                    pass
                proc.wait()
                return None
            if rc != 0:
                return None
            rows = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    rows += [r for r in csv.DictReader(fh) if r.get("Counter_Name") == ctr]
            head = [float(r["Counter_Value"]) for r in rows if re.match(r"void bgn::k_pairing<\d+, 0>", r["Kernel_Name"])]
            if not head:
                return None
            kb[ctr] = (sum(head) / len(head), len(head))
            if extras:
                # the launches --pmc-extras adds after the timed step (pmc_extras below): ONE level-1 Add and ONE level-2
                # Add of 2^20, Decrypt of 2^16 and of 2^20 (their lifts: the two launches of k_pairing<NL, 1> on 65536
                # lanes, the shorter one the 2^16 batch)
                def avg(sel):
                    v = [float(r["Counter_Value"]) for r in sel]
                    return sum(v) / len(v) if v else None
                dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                sec.setdefault("eadd_l1", {})[ctr] = avg([r for r in rows if "k_g1_add_wire<" in r["Kernel_Name"] and int(r["Grid_Size"]) == 65536])
                sec.setdefault("eadd_l2", {})[ctr] = avg([r for r in rows if "k_gt_mul_wire<" in r["Kernel_Name"] and int(r["Grid_Size"]) == 1 << 20])
                lifts = sorted((r for r in rows if re.match(r"void bgn::k_pairing<\d+, 1>", r["Kernel_Name"]) and int(r["Grid_Size"]) == 65536), key=dur)
                if len(lifts) >= 2 and dur(lifts[-1]) > 4 * dur(lifts[0]):
                    sec.setdefault("decrypt_lift_2^16", {})[ctr] = float(lifts[0]["Counter_Value"])
                    sec.setdefault("decrypt_lift_k_pairing_1", {})[ctr] = float(lifts[-1]["Counter_Value"])
            if ctr == "FETCH_SIZE":
                enc = [r for r in rows if int(r["Grid_Size"]) == 1 << 20 and re.match(r"void bgn::k_encode<(\d+)>", r["Kernel_Name"])]
                if enc:
                    nl = int(re.match(r"void bgn::k_encode<(\d+)>", enc[0]["Kernel_Name"]).group(1))
                    seen = sum(float(r["Counter_Value"]) for r in enc) / len(enc) * 1024
                    if seen > 0:
                        calib = {"kernel": "k_encode<%d>, 2^20 elements" % nl, "known_read_bytes": 2 * nl * 4 * (1 << 20),
                                 "FETCH_SIZE_bytes": seen, "factor": 2 * nl * 4 * (1 << 20) / seen, "launches": len(enc)}
    except Exception as e:                              # a measurement aid must never take the bench line down
        print("bench.py: live PMC passes failed (%s): quoting the committed summary" % e, file=sys.stderr, flush=True)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    factor = calib["factor"] if calib else 2.0          # the guide's factor for FETCH_SIZE on gfx950 where no calibration ran
    secondary = {k: (v["FETCH_SIZE"] * factor + v["WRITE_SIZE"]) * 1024 for k, v in sec.items()
                 if v.get("FETCH_SIZE") is not None and v.get("WRITE_SIZE") is not None}
    return {"hbm_bytes_per_launch": (kb["FETCH_SIZE"][0] * factor + kb["WRITE_SIZE"][0]) * 1024, "secondary": secondary,
            "FETCH_SIZE_KB": kb["FETCH_SIZE"][0], "WRITE_SIZE_KB": kb["WRITE_SIZE"][0], "launches": kb["FETCH_SIZE"][1],
            "fetch_calibration": calib, "seconds": time.perf_counter() - t0,
            "source": "measured in this run: two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE; no tracing) of this "
                      "command's timed step%s, before the timed region; FETCH_SIZE x %.3f (%s) + WRITE_SIZE, KB -> bytes"
                      % (" and of one launch each of the secondary kernels (--pmc-extras)" if extras else "", factor, "calibrated on k_encode's known read volume in the same pass" if calib else "the guide's factor")}


def secondary_traffic(live, key, committed_field):
    """(bytes, source) of a secondary kernel's HBM traffic: from this run's own PMC passes when they covered it
    (live_traffic(extras=True)), else from the committed summary, labelled."""
    if live and key in live.get("secondary", {}):
        return live["secondary"][key], live["source"]
    node, src = committed_traffic(key)
    v = (node or {}).get(committed_field)
    return v, (src if v is not None else None)


def pmc_extras(pk, fx, dev, cts, xs, prods):
    """--pmc-extras (the child passes of live_traffic only): after the timed step, ONE launch each of the secondary
    kernels whose `traffic` the line quotes — the level-1 and the level-2 Add of 2^20, Decrypt of 2^16 and of 2^20 on the
    mixed batch of secondary_metrics — untimed, unchecked (the parent measures and checks them itself)."""
    import torch
    import bgn_amd
    import bgn_amd.synthetic as syn
    eng = pk.engine
    EB = eng.elem_bytes
    n = cts.numel() // EB
    o = torch.empty(n * EB, dtype=torch.uint8, device=dev)
    eng.add_dev(1, cts, syn.permuted_copy(cts, EB, seed=11), o, n)
    eng.add_dev(2, prods, syn.permuted_copy(prods, EB, seed=13), o, n)
    pk.SetupDecryption(bgn_amd.SecretKey(int(fx["q1"], 16)))
    mixed, _, _ = syn.decrypt_mix(pk, fx, cts, xs, dev)
    for k in (16, 20):
        m = torch.empty(1 << k, dtype=torch.int64, device=dev)
        st = torch.empty(1 << k, dtype=torch.uint8, device=dev)
        eng.decrypt_dev(1, mixed[: (EB << k)], m, st, 1 << k)
    torch.cuda.synchronize()


def config0_metrics(no_cpu: bool):
    """BASELINE configs[0]: 512-bit params, 128 ciphertexts, pk.Add and pk.Mult over 128 independent pairs — the
    shape of BenchmarkAdd / BenchmarkMult (bgn_test.go:97-140), which the reference runs on one goroutine.  Host
    buffers in and out (the size at which the boundary copies matter), next to the single-threaded C oracle on the
    same 128 pairs; and the latency of ONE Mult (count = 1), 512- and 1024-bit."""
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
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            res = fn()
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        out[name] = {"value": n / best, "unit": "ops/s", "wall_ms_for_128": best * 1e3, "result": res.tobytes()}
    E = eng.elem_bytes
    eng.mult(a[:E], b[:E])
    t0 = time.perf_counter()
    one = eng.mult(a[:E], b[:E])
    out["emult_count1_latency_ms"] = (time.perf_counter() - t0) * 1e3
    assert one.tobytes() == out["emult"]["result"][:E]
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
    for name in ("eadd", "emult"):
        del out[name]["result"]
    out["workload"] = "configs[0]: 512-bit params, 128-ciphertext EAdd + EMult, host buffers through the C ABI"
    return out


def _timed(fn, sync, reps=2):
    """Seconds of one call: reps - 1 untimed calls, then one timed call — the median of five when a call takes less than
    a quarter of a second (a single 40 ms call varies by 15 % from one sample to the next on this pool)."""
    def once():
        sync()
        t0 = time.perf_counter()
        fn()
        sync()
        return time.perf_counter() - t0

    for _ in range(max(0, reps - 1)):
        once()
    dt = once()
    if dt < 0.25:
        dt = sorted([dt] + [once() for _ in range(4)])[2]
    return dt


def op_rooflines(entry, counts, nl, n_gpus=1):
    """Every secondary entry carries the two bounds: its 32x32->64 multiply-adds — counts = (reductions, how many of
    them end a squaring, how many a sum of two products), priced by bgn_amd.synthetic.mads_from_counts — against the measured issue peaks of
    v_mad_u64_u32 (the VALU bound that applies; both occupancies), and its algorithmic bytes against the HBM peak.
    Aggregate rates over n_gpus ranks are held against n_gpus chips."""
    import bgn_amd.synthetic as syn
    products, squares, sops = counts
    mads = syn.mads_from_counts(products, squares, sops, nl)
    rate = entry["value"] * mads / n_gpus
    entry["products_per_unit"] = products           # everything that ends in a Montgomery reduction
    entry["squarings_per_unit"] = squares
    entry["sums_of_two_products_per_unit"] = sops   # reductions shared by two multiplications (fp_mul2)
    entry["roofline_valu"] = {"bound": "v_mad_u64_u32 issue", "mads_per_unit": mads, "achieved": rate, "unit": "lane-MAD/s",
                              "peak": VALU_MAD_PEAK_4W, "frac": rate / VALU_MAD_PEAK_4W,
                              "peak_at_1_wave_per_simd": VALU_MAD_PEAK_1W, "frac_at_1_wave_per_simd": rate / VALU_MAD_PEAK_1W}
    if "algorithmic_bytes_per_unit" in entry:
        gbs = entry["value"] * entry["algorithmic_bytes_per_unit"] / 1e9 / n_gpus
        entry["hbm"] = {"achieved_GBps": gbs, "peak_GBps": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS}
    return entry


def mid_batch_metrics(eng, a, b, dev):
    """Mult at the batch sizes of a MultPoly fan-out (mid-size batches: the lane-group kernel, 16 lanes per pairing): the
    first pairs of the headline's operands, device-resident, the default dispatch, best of three whole calls (wire bytes
    to wire bytes).  Called BEFORE the headline's warm-up: a mid-size request does not arrive behind a minute of full
    load, and the clocks of a chip that has just run one differ by a few per cent."""
    import torch
    EB = eng.elem_bytes
    sync = torch.cuda.synchronize
    om = torch.empty((1 << 15) * EB, dtype=torch.uint8, device=dev)
    mid = {}
    for n_mid in (1 << 12, 1 << 14, 1 << 15):
        eng.mult_dev(a[: n_mid * EB], b[: n_mid * EB], om[: n_mid * EB], n_mid)          # warm-up (workspace)
        best = min(_timed(lambda: eng.mult_dev(a[: n_mid * EB], b[: n_mid * EB], om[: n_mid * EB], n_mid), sync, reps=1)
                   for _ in range(3))
        mid[str(n_mid)] = {"ms": best * 1e3, "pairings_per_s": n_mid / best, "kernel": eng.last_kernel_name()}
    return {"unit": "ms per call of that many pairings", "sizes": mid,
            "workload": "pk.Mult on 4096 / 16384 / 32768 ciphertext pairs (wire bytes in HBM to wire bytes), kernel chosen "
                        "by the engine's batch-size dispatch; measured before the headline's warm-up steps"}


def secondary_metrics(pk, fx, dev, cts, xs, rs, dec_log2s, no_cpu=False, polys_log2=14, prods=None, live=None):
    """BASELINE configs[1] (Encrypt), EAdd on both levels, MultConst, configs[4]'s shape on one GPU and configs[3]
    (BSGS Decrypt, T = 2^40), on the Config-2 ciphertexts `cts` = Encrypt(xs, rs) the headline used and on its products
    `prods` (level-2 ciphertexts).  Inputs resident in HBM; one warm-up pass
    then one timed pass each (the median of five for calls shorter than a quarter of a second, _timed)."""
    import numpy as np
    import torch
    import bgn_amd
    import bgn_amd.synthetic as syn
    eng = pk.engine
    EB = eng.elem_bytes
    nl = syn.limbs_for(int(fx["p"], 16))
    sync = torch.cuda.synchronize
    out = {}
    n_enc = xs.shape[0]
    tmp = torch.empty_like(cts)
    dt = _timed(lambda: eng.encrypt_dev(xs, xs.shape[1], rs, rs.shape[1], tmp, n_enc), sync)
    assert bool((tmp == cts).all().item())
    del tmp
    out["encrypt"] = op_rooflines(
        {"value": n_enc / dt, "unit": "encrypts/s", "batch": n_enc,
         "workload": "configs[1]: batch=2^20 Encrypt P^m * Q^r, 40-bit m, 1022-bit r, fixed-base window tables of P "
                     "(16-bit) and Q (20-bit) in HBM, affine additions over four accumulation chains per element, one "
                     "inversion per run of 64",
         "kernel": eng.last_kernel_name(), "kernel_ms_per_step": eng.last_kernel_ms(),
         "algorithmic_bytes_per_unit": xs.shape[1] + rs.shape[1] + EB},
        syn.encrypt_counts(xs.shape[1] * 8, rs.shape[1] * 8), nl)
    if not no_cpu:
        ns = 4096                                     # the first 4096 (m, r) of the batch are enough for any host
        xh = [int.from_bytes(bytes(v), "big") for v in xs[:ns].cpu().numpy()]
        rh = [int.from_bytes(bytes(v), "big") for v in rs[:ns].cpu().numpy()]
        cb = secondary_cpu_leg(fx, ns, lambda orc, lo, hi: orc.encrypt(xh[lo:hi], rh[lo:hi]),
                               cts[: ns * EB].cpu().numpy().tobytes(), EB, "encrypts/s",
                               "(m, r) pairs (P^m * Q^r by generic scalar multiplication, as PBC's PowBig)", calibrate=4)
        if cb:
            out["encrypt"]["cpu_baseline"] = cb
    # --- EAdd (level 1): every ciphertext with its partner in a fixed permutation (the pairs of Config 3)
    n_add = n_enc
    a1, b1 = cts, syn.permuted_copy(cts, EB, seed=11)
    o1 = torch.empty(n_add * EB, dtype=torch.uint8, device=dev)
    dt = _timed(lambda: eng.add_dev(1, a1, b1, o1, n_add), sync)
    out["eadd_l1"] = op_rooflines(
        {"value": n_add / dt, "unit": "adds/s", "batch": n_add,
         "workload": "pk.Add on level-1 ciphertexts (affine G1 addition, batched inversion), wire bytes to wire bytes in "
                     "one launch",
         "kernel": eng.last_kernel_name(), "algorithmic_bytes_per_unit": 3 * EB},
        syn.eadd_counts(n_add), nl)
    # HBM-side traffic of that call (one launch of k_g1_add_wire since round 6; k_decode_plain x 2, k_g1_add, k_encode
    # before) from the committed PMC passes of this command (tools/summarize_profiles.py), next to its algorithmic bytes
    eadd_traffic, eadd_src = None, None
    if n_add == 1 << 20:
        eadd_traffic, eadd_src = secondary_traffic(live, "eadd_l1", "hbm_bytes_per_call")
    alg_add = 3 * EB * n_add
    out["eadd_l1"]["roofline"] = {"bound": "hbm", "achieved": alg_add / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": alg_add / dt / 1e9 / HBM_PEAK_GBS, "traffic": eadd_traffic, "traffic_source": eadd_src,
                                  "algorithmic_bytes_per_call": alg_add, "kernel": eng.last_kernel_name(),
                                  "kernel_ms": eng.last_kernel_ms(), "call_ms": dt * 1e3}
    if not no_cpu:
        ns = 1 << 16
        ah, bh = a1[: ns * EB].cpu().numpy().tobytes(), b1[: ns * EB].cpu().numpy().tobytes()
        cb = secondary_cpu_leg(fx, ns, lambda orc, lo, hi: orc.add(1, ah[lo * EB:hi * EB], bh[lo * EB:hi * EB]),
                               o1[: ns * EB].cpu().numpy().tobytes(), EB, "adds/s",
                               "pairs (affine G1 addition, one field inversion each, as PBC's Mul on G1)", calibrate=64)
        if cb:
            out["eadd_l1"]["cpu_baseline"] = cb
    del o1, b1
    # --- EAdd (level 2, bgn.go:455-475): every product of the headline with its partner in a fixed permutation; one
    # wire-to-wire launch (k_gt_mul_wire: Barrett F_p^2 product of plain residues, csrc/barrett.hpp)
    if prods is not None:
        n2a = prods.numel() // EB
        b2 = syn.permuted_copy(prods, EB, seed=13)
        o2 = torch.empty(n2a * EB, dtype=torch.uint8, device=dev)
        dt = _timed(lambda: eng.add_dev(2, prods, b2, o2, n2a), sync)
        k_ms, k_name = eng.last_kernel_ms(), eng.last_kernel_name()
        mads2 = syn.l2_add_mads(nl)
        alg2 = 3 * EB * n2a
        t2, src2 = secondary_traffic(live, "eadd_l2", "hbm_bytes_per_launch") if n2a == 1 << 20 else (None, None)
        rate2 = n2a / dt * mads2
        out["eadd_l2"] = {
            "value": n2a / dt, "unit": "adds/s", "batch": n2a,
            "workload": "pk.Add on level-2 ciphertexts (the headline's products; one F_p^2 product each), wire bytes to "
                        "wire bytes in one launch", "kernel": k_name, "algorithmic_bytes_per_unit": 3 * EB,
            "roofline": {"bound": "hbm", "achieved": alg2 / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg2 / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "traffic": t2, "traffic_source": src2,
                         "algorithmic_bytes_per_launch": alg2, "kernel": k_name, "kernel_ms": k_ms, "call_ms": dt * 1e3},
            "roofline_valu": {"bound": "v_mad_u64_u32 issue", "mads_per_unit": mads2, "achieved": rate2, "unit": "lane-MAD/s",
                              "peak": VALU_MAD_PEAK_4W, "frac": rate2 / VALU_MAD_PEAK_4W,
                              "peak_at_2_waves_per_simd": VALU_MAD_PEAK_2W, "frac_at_2_waves_per_simd": rate2 / VALU_MAD_PEAK_2W,
                              "note": "two workgroups per CU (LDS: the wire stage only), i.e. two waves per SIMD"}}
        if not no_cpu:
            ns = n2a                                      # the whole batch: a third of a second on sixteen cores
            ah, bh = prods[: ns * EB].cpu().numpy().tobytes(), b2[: ns * EB].cpu().numpy().tobytes()
            cb = secondary_cpu_leg(fx, ns, lambda orc, lo, hi: orc.add(2, ah[lo * EB:hi * EB], bh[lo * EB:hi * EB]),
                                   o2[: ns * EB].cpu().numpy().tobytes(), EB, "adds/s",
                                   "pairs (one F_p^2 product each, as PBC's Mul on GT)", calibrate=64)
            if cb:
                out["eadd_l2"]["cpu_baseline"] = cb
        del o2, b2
    # --- Neg (bgn.go:436-438: Sub(encryptZero(), c)): one coordinate negated, no field product — one wire-to-wire launch
    # (k_neg_wire) whose only bound is HBM: 2 * 2L bytes per element
    n_neg = n_enc
    o_neg = torch.empty(n_neg * EB, dtype=torch.uint8, device=dev)
    dt = _timed(lambda: check_rc(eng._lib.bgn_neg_batch_dev(eng._h, n_neg, 1, cts.data_ptr(), o_neg.data_ptr(), eng._stream())), sync)
    k_ms, k_name = eng.last_kernel_ms(), eng.last_kernel_name()
    alg_neg = 2 * EB * n_neg
    out["neg_l1"] = {
        "value": n_neg / dt, "unit": "negs/s", "batch": n_neg,
        "workload": "pk.Neg on level-1 ciphertexts (y -> p - y), wire bytes to wire bytes in one launch", "kernel": k_name,
        "algorithmic_bytes_per_unit": 2 * EB,
        "roofline": {"bound": "hbm", "achieved": alg_neg / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg_neg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": alg_neg, "kernel": k_name, "kernel_ms": k_ms, "call_ms": dt * 1e3}}
    if not no_cpu:
        ns = 1 << 18
        zh, ch = bytes(ns * EB), cts[: ns * EB].cpu().numpy().tobytes()
        cb = secondary_cpu_leg(fx, ns, lambda orc, lo, hi: orc.add(1, zh[lo * EB:hi * EB], ch[lo * EB:hi * EB], True),
                               o_neg[: ns * EB].cpu().numpy().tobytes(), EB, "negs/s",
                               "ciphertexts (Sub(encryptZero(), c), as the reference's Neg)", calibrate=64)
        if cb:
            out["neg_l1"]["cpu_baseline"] = cb
    del o_neg
    # --- MultConst (bgn.go:253-291, BenchmarkMultConstant bgn_test.go:112-125) with per-element scalars: 2^16
    # ciphertexts of each level, 40-bit scalars (a plaintext-sized constant) and 1024-bit ones (the width of n)
    n_mc = 1 << 16
    g = torch.Generator(device="cpu")
    g.manual_seed(4242)
    mc_src = {1: cts[: n_mc * EB], 2: (prods[: n_mc * EB] if prods is not None else None)}
    omc = torch.empty(n_mc * EB, dtype=torch.uint8, device=dev)
    for kb in (5, 128):
        ks = torch.randint(0, 256, (n_mc, kb), dtype=torch.uint8, generator=g)
        if kb == 128:
            ks[:, 0] &= 0x3F                                            # below n (1023 bits and more)
        ks = ks.to(dev)
        for level in (1, 2):
            src = mc_src[level]
            if src is None:
                continue
            call = lambda: check_rc(eng._lib.bgn_multconst_batch_dev(eng._h, n_mc, level, src.data_ptr(), ks.data_ptr(), kb,
                                                                     None, 0, omc.data_ptr(), eng._stream()))
            dt = _timed(call, sync)
            e = op_rooflines(
                {"value": n_mc / dt, "unit": "multconsts/s", "batch": n_mc, "level": level, "scalar_bits": kb * 8,
                 "workload": "pk.MultConst on 2^16 level-%d ciphertexts, one uniformly random %d-bit scalar per element "
                             "(device-resident wire bytes in and out)" % (level, kb * 8),
                 "kernel": eng.last_kernel_name(), "kernel_ms": eng.last_kernel_ms(), "call_ms": dt * 1e3,
                 "algorithmic_bytes_per_unit": 2 * EB + kb},
                syn.multconst_counts(level, kb * 8), nl)
            if not no_cpu:
                ns = 2048 if kb == 5 else 256
                ah = src[: ns * EB].cpu().numpy().tobytes()
                kh = [int.from_bytes(bytes(v), "big") for v in ks[:ns].cpu().numpy()]
                cb = secondary_cpu_leg(fx, ns, lambda orc, lo, hi, lv=level: orc.multconst(lv, ah[lo * EB:hi * EB], kh[lo:hi]),
                                       omc[: ns * EB].cpu().numpy().tobytes(), EB, "multconsts/s",
                                       "ciphertexts (generic scalar multiplication / power by a %d-bit scalar, as PBC's "
                                       "PowBig)" % (kb * 8), seconds=3.0, calibrate=2)
                if cb:
                    e["cpu_baseline"] = cb
            out["multconst_l%d_%db" % (level, kb * 8)] = e
    del omc
    # --- the decryption tables first (one-off per key, like SetupDecryption): MultPoly below is then measured on a
    # context that can also decrypt — 53 GB of tables beside its 38 GB of line tables, the state of a real service
    # (until round 6 the line tables of such a context went back to the allocator after every call: engine.cpp
    # trim_to_resident_cap)
    t0 = time.perf_counter()
    pk.SetupDecryption(bgn_amd.SecretKey(int(fx["q1"], 16)))
    sync()
    t_setup = time.perf_counter() - t0
    # --- MultPoly: configs[4] at its stated size on this one GPU — 2^14 products of 16x16 coefficient polynomials
    # (2^22 coefficient pairs, base-3 digits in {-1, 0, 1}) followed by one AddPoly of the products with each other
    if polys_log2:
        job = MultPolyJob(pk, 1 << polys_log2, dev, seed_a=2000, seed_b=3000)
        dt = _timed(job.step, sync)
        e = job.entry(fx, dt, 1)
        e["context_memory_bytes_after_the_call"] = int(eng.memory_bytes())
        e["context"] = "holds the decryption tables of T = 2^40 (bgn_ctx_setup_decryption ran first)"
        if not no_cpu:
            ah, bh = job.ca.cpu().numpy().tobytes(), job.cb.cpu().numpy().tobytes()
            d1, d2 = job.d1, job.d2
            cb = secondary_cpu_leg(
                fx, job.npoly,
                lambda orc, lo, hi: orc.poly_mult(hi - lo, d1, d2, ah[lo * d1 * EB:hi * d1 * EB], bh[lo * d2 * EB:hi * d2 * EB]),
                job.prod.cpu().numpy().tobytes(), (d1 + d2) * EB, "polynomial products/s",
                "polynomial products of 16x16 coefficients (256 pairings + the accumulation each, poly.go:123-156)",
                seconds=4.0, calibrate=1, max_per_thread=1)
            if cb:
                cb["coefficient_pairs_per_s"] = cb["value"] * d1 * d2
                cb["single_thread_coefficient_pairs_per_s"] = cb["single_thread_per_s"] * d1 * d2
                e["cpu_baseline"] = cb
        out["multpoly"] = e
        del job
    # --- Decrypt: the first 2^k of those ciphertexts, every 16th negated and every 4096th out of range
    mixed, want, want_st = syn.decrypt_mix(pk, fx, cts, xs, dev)
    dec = {}
    for k in sorted(set(dec_log2s)):
        n_dec = 1 << k
        sel = mixed[: n_dec * EB]
        m = torch.empty(n_dec, dtype=torch.int64, device=dev)
        st = torch.empty(n_dec, dtype=torch.uint8, device=dev)
        dt = _timed(lambda: eng.decrypt_dev(1, sel, m, st, n_dec), sync)
        lift_ms, walk_ms = eng.last_aux_kernel_ms(), eng.last_kernel_ms()
        ok = bool((m.cpu() == want[:n_dec]).all().item()) and bool((st.cpu() == want_st[:n_dec]).all().item())
        alg = EB + 16
        e = op_rooflines(
            {"value": n_dec / dt, "unit": "decrypts/s", "batch": n_dec, "level": 1,
             "workload": "configs[3]: T=2^40 BSGS Decrypt, batch=2^%d, m uniform in [0,2^40), 1/16 negative, 1/4096 out "
                         "of range (status NOT_FOUND); Miller loop over the secret order's line table + final "
                         "exponentiation + ^sk, then giant steps 2S apart on an HBM-resident baby table (%d entries)"
                         % (k, int(eng._lib.bgn_ctx_bsgs_baby_steps(eng._h))),
             "table_setup_s": t_setup, "plaintexts_and_statuses_exact": ok, "algorithmic_bytes_per_unit": alg},
            syn.decrypt_counts(fx, int(eng._lib.bgn_ctx_bsgs_baby_steps(eng._h))), nl)
        # the dominant kernel of Decrypt is the lift (k_pairing<NL, 1>), timed by HIP events on its stream
        traffic, traffic_src = secondary_traffic(live, "decrypt_lift_k_pairing_1" if k == 20 else "decrypt_lift_2^%d" % k,
                                                 "hbm_bytes_per_launch")
        e["roofline"] = {"bound": "hbm", "achieved": alg * n_dec / (lift_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": alg * n_dec / (lift_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src if traffic is not None else None,
                         "kernel": eng.last_aux_kernel_name(), "kernel_ms": lift_ms,
                         "walk_kernel": eng.last_kernel_name(), "walk_kernels_ms": walk_ms,
                         "algorithmic_bytes_per_decrypt": alg}
        dec[k] = e
    if dec and not no_cpu:
        top = max(dec)
        cb = decrypt_cpu_baseline(fx, mixed, want, want_st, 1 << top, EB)
        if cb:
            dec[top]["cpu_baseline"] = cb
    # --- Decrypt of level-2 ciphertexts (configs[3] asks for both levels): products of 20-bit messages
    n2 = 1 << 16
    g = torch.Generator(device="cpu")
    g.manual_seed(777)
    xs2 = torch.randint(0, 256, (2 * n2, 3), dtype=torch.uint8, generator=g)
    xs2[:, 0] &= 0x0F                                                                         # 20-bit plaintexts
    xs2 = xs2.to(dev)
    c2 = torch.empty(2 * n2 * EB, dtype=torch.uint8, device=dev)
    eng.encrypt_dev(xs2, 3, rs[: 2 * n2], rs.shape[1], c2, 2 * n2)
    l2 = torch.empty(n2 * EB, dtype=torch.uint8, device=dev)
    eng.mult_dev(c2[: n2 * EB], c2[n2 * EB:], l2, n2)
    m = torch.empty(n2, dtype=torch.int64, device=dev)
    st = torch.empty(n2, dtype=torch.uint8, device=dev)
    dt = _timed(lambda: eng.decrypt_dev(2, l2, m, st, n2), sync)
    xv = xs2.cpu().numpy().astype(np.int64)
    val = (xv[:, 0] * 65536 + xv[:, 1] * 256 + xv[:, 2])
    ok = bool((m.cpu().numpy() == val[:n2] * val[n2:]).all()) and not bool(st.any().item())
    out["decrypt_l2"] = op_rooflines(
        {"value": n2 / dt, "unit": "decrypts/s", "batch": n2, "level": 2,
         "workload": "configs[3], level 2: Decrypt of 2^16 products of two 20-bit messages (outputs of Mult): "
                     "^sk by the norm-1 ladder, then the same giant-step walk",
         "plaintexts_recovered_exactly": ok, "algorithmic_bytes_per_unit": EB + 16},
        syn.decrypt_counts(fx, int(eng._lib.bgn_ctx_bsgs_baby_steps(eng._h)), level=2), nl)
    return out, dec


def rank_command(argv):
    """Command line of one rank process (the CPU tests substitute their own children)."""
    return [sys.executable, os.path.abspath(__file__)] + list(argv)


def run_cap_seconds():
    """Wall-clock cap of one bench run (BGN_BENCH_TIMEOUT_S; 0 disables): a rank stuck in a collective — its peer
    died, a rendezvous that never completes — must end the run with a non-zero exit instead of sitting there until
    whoever launched it gives up."""
    try:
        return float(os.environ.get("BGN_BENCH_TIMEOUT_S", "2400"))
    except ValueError:
        return 2400.0


def spawn_ranks(n, argv, timeout_s=None, poll_s=0.2):
    """--gpus N without a launcher: start N fresh rank processes of this script, one per GPU, before anything in
    this process has touched the GPU (this parent never does).  All ranks are polled together: the first non-zero
    exit, or the wall-clock cap, ends the others within seconds (a failed rank leaves its peers in an RCCL barrier
    that never completes) and the run returns non-zero.  Rank 0 prints the JSON line on the shared stdout."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cap = run_cap_seconds() if timeout_s is None else timeout_s
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        procs.append(subprocess.Popen(rank_command(argv), env=env))
    t0 = time.monotonic()
    rc = 0
    try:
        while True:
            states = [p.poll() for p in procs]
            bad = [(r, st) for r, st in enumerate(states) if st not in (None, 0)]
            if bad:
                rc = bad[0][1] if 0 < bad[0][1] < 256 else 1
                print("bench.py: rank %d exited with %d; ending the other ranks" % bad[0], file=sys.stderr, flush=True)
                break
            if all(st == 0 for st in states):
                break
            if cap and time.monotonic() - t0 > cap:
                rc = 124
                print("bench.py: no result after %.0f s (BGN_BENCH_TIMEOUT_S); ending the ranks" % cap, file=sys.stderr,
                      flush=True)
                break
            time.sleep(poll_s)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                pass
    return rc


def start_watchdog():
    """Inside a rank (spawned here or by torch.distributed.run): the same cap, enforced from a daemon thread, so a
    rank hung in a collective exits non-zero by itself."""
    import threading
    cap = run_cap_seconds()
    if not cap:
        return None

    def fire():
        print("bench.py: rank %s still running after %.0f s (BGN_BENCH_TIMEOUT_S): exiting" %
              (os.environ.get("RANK", "0"), cap), file=sys.stderr, flush=True)
        os._exit(124)

    t = threading.Timer(cap, fire)
    t.daemon = True
    t.start()
    return t


_JSON_FD = None


def emit_line(line):
    data = (json.dumps(line) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, data)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["emult", "multpoly"], default="emult")
    ap.add_argument("--batch-log2", type=int, default=20, help="emult: pairs per GPU (log2)")
    ap.add_argument("--polys-log2", type=int, default=14, help="multpoly: polynomials in the whole job (log2)")
    ap.add_argument("--key", default="k1024")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary Encrypt / Decrypt measurements")
    ap.add_argument("--decrypt-log2", type=int, nargs="+", default=[16, 20],
                    help="batch sizes (log2, at most 20) of the Decrypt measurement")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and gather even with one rank")
    ap.add_argument("--pmc-extras", action="store_true", help=argparse.SUPPRESS)   # the child passes of live_traffic only
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="quote roofline.traffic from the committed PMC summary instead of measuring it in two child passes")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1 or os.environ.get("BGN_BENCH_SPAWN") == "1":     # the latter: rehearse the spawn with one rank
            sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
        world, rank, local_rank = 1, 0, 0
    else:
        world = int(os.environ["WORLD_SIZE"])
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if args.gpus != world:
            sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch N ranks for --gpus N")

    # roofline.traffic of the headline kernel, measured before this process touches the GPU (one GPU, the headline
    # workload at its BASELINE size only; the children run without it)
    live = None
    if (world == 1 and not args.force_dist and os.environ.get("BGN_BENCH_FORCE_DIST") != "1" and not args.no_live_traffic
            and args.workload == "emult" and args.key == "k1024" and args.batch_log2 == 20):
        live = live_traffic(extras=not args.no_extra)

    start_watchdog()
    # stdout carries ONE JSON line: native libraries that write to file descriptor 1 (RCCL prints a version banner
    # there when a communicator is created) are sent to stderr, the line goes to the original descriptor
    global _JSON_FD
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch
    import torch.distributed as dist

    use_dist = world > 1 or args.force_dist or os.environ.get("BGN_BENCH_FORCE_DIST") == "1"
    if use_dist:
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
    import bgn_amd.synthetic as syn
    from bgn_amd.sharding import gather_shards, shard_range

    fx = load_fixture(args.key)
    pk = bgn_amd.PublicKey(int(fx["p"], 16), int(fx["n"], 16), fx["l"], bytes.fromhex(fx["P"]),
                           bytes.fromhex(fx["Q"]), fx["msg_space"], True, fx["poly_base"], device=local_rank)
    eng = pk.engine
    EB = eng.elem_bytes
    rccl_ranks = dist.get_world_size() if use_dist else 0

    if args.workload == "multpoly":
        return bench_multpoly(args, pk, fx, dev, world, rank, use_dist, rccl_ranks)

    per_gpu = 1 << args.batch_log2
    total = per_gpu * world
    lo, hi = shard_range(total, world, rank)           # contiguous slice of the global batch
    count = hi - lo

    # Config 2 on the GPU (not timed): this rank's 2^20 ciphertexts; Config 3: paired with a fixed permutation.
    xs, rs, cts = syn.config2_ciphertexts(pk, count, seed=1000 + rank, device=dev)
    a = cts
    b = syn.permuted_copy(cts, EB, seed=5)
    out = torch.empty(count * EB, dtype=torch.uint8, device=dev)

    def timed_region(step, steps, warmup):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; MAX over ranks."""
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # ---- the headline: weak scaling, 2^20 pairs per GPU ----
    gathered = [None]
    kernel_ms = []

    def step():
        eng.mult_dev(a, b, out, count)
        kernel_ms.append(eng.last_kernel_ms())          # HIP events on the kernel's own stream
        if use_dist:
            gathered[0] = gather_shards(out, total, EB, world, rank, dist)

    mid_batch = None
    if not args.no_extra and world == 1 and not use_dist and args.key == "k1024" and args.batch_log2 == 20:
        mid_batch = mid_batch_metrics(eng, a, b, dev)
    dt = timed_region(step, args.steps, args.warmup)
    kernel_ms = kernel_ms[args.warmup:]
    kernel_name = eng.last_kernel_name()
    if use_dist:
        assert bool((gathered[0][lo * EB:hi * EB] == out).all().item()), "gather mismatch"
    gathered[0] = None

    # ---- strong scaling: the SAME global batch of 2^20 pairs over the ranks (2^17 per GPU at 8 GPUs) ----
    strong = None
    if use_dist:
        s_total = per_gpu
        s_lo, s_hi = shard_range(s_total, world, rank)
        s_cnt = s_hi - s_lo
        s_out = torch.empty(s_cnt * EB, dtype=torch.uint8, device=dev)
        s_ms = []

        def s_step():
            eng.mult_dev(a[: s_cnt * EB], b[: s_cnt * EB], s_out, s_cnt)
            s_ms.append(eng.last_kernel_ms())
            gathered[0] = gather_shards(s_out, s_total, EB, world, rank, dist)

        s_dt = timed_region(s_step, args.steps, args.warmup)
        assert bool((s_out == out[: s_cnt * EB]).all().item()), "strong-scaling shard differs from the weak-scaling result"
        strong = {"metric": "EMult pairings/sec at 1024-bit, global batch=2^%d over %d GPU(s)" % (args.batch_log2, world),
                  "value": s_total * args.steps / s_dt, "unit": "pairings/s", "scaling": "strong", "global_batch": s_total,
                  "batch_per_gpu": s_cnt, "ms_per_step": s_dt / args.steps * 1e3, "kernel": eng.last_kernel_name(),
                  "kernel_ms": sum(s_ms[args.warmup:]) / max(1, len(s_ms[args.warmup:]))}
        gathered[0] = None
        del s_out

    # sample of this rank's batch for the CPU leg (not timed): 4096 distinct pairs STRIDED over the batch (every
    # 256th pair at 2^20).  A lane of k_pairing owns a run of up to sixteen pairings that share one inversion, pair e
    # sitting at position e / 65536 of its lane's run: a prefix would check position 0 only.
    nchk = min(count, 4096)
    chk_stride = max(1, count // nchk)
    sel = torch.arange(nchk, device=dev) * chk_stride
    a_h = a.view(count, EB)[sel].cpu().numpy().tobytes()
    b_h = b.view(count, EB)[sel].cpu().numpy().tobytes()
    o_h = out.view(count, EB)[sel].cpu().numpy().tobytes()
    distinct = len({a_h[i * EB:(i + 1) * EB] + b_h[i * EB:(i + 1) * EB] for i in range(nchk)})
    run_positions = sorted({int(e) // 65536 for e in sel.tolist()}) if count > 65536 else [0]

    extra = dec = None
    full = args.key == "k1024" and args.batch_log2 == 20
    if args.pmc_extras and world == 1 and not use_dist and full:
        pmc_extras(pk, fx, dev, cts, xs, out)
    if not args.no_extra and world == 1 and not use_dist and full:
        extra, dec = secondary_metrics(pk, fx, dev, cts, xs, rs, [min(k, 20) for k in args.decrypt_log2],
                                       no_cpu=args.no_cpu_baseline, polys_log2=args.polys_log2, prods=out, live=live)
        extra["config0_512bit_128"] = config0_metrics(args.no_cpu_baseline)
        if mid_batch:
            extra["mult_mid_batch"] = mid_batch
    elif not args.no_extra and use_dist and args.key == "k1024":
        # the second half of BASELINE's metric on every GPU: Decrypt shards exactly like Mult (bgn.go:205-250 is per
        # ciphertext); plaintexts and statuses are gathered like the result arrays
        dec = {args.batch_log2: decrypt_sharded(pk, fx, dev, cts, xs, world, rank, dist, timed_region, args)}

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = total * args.steps / dt
        k_ms = sum(kernel_ms) / len(kernel_ms)
        alg_bytes = 3 * EB                               # two G1 operands in, one GT element out
        achieved = alg_bytes * count / (k_ms * 1e-3) / 1e9
        mads = syn.algorithmic_mads_per_pairing(
            fx, run=max(1, min(16, -(-count // 65536))),
            window={0: 2, 3: 3, 4: 4}.get(eng.get_option("miller_window"), 5))
        traffic, traffic_src = committed_traffic() if full else (None, None)
        if traffic is not None:
            traffic_src += " (FETCH_SIZE + WRITE_SIZE per launch of 2^20 pairings)"
        if live is not None and full:
            traffic, traffic_src = live["hbm_bytes_per_launch"], live["source"]
        mad_rate = mads * count / (k_ms * 1e-3)
        if use_dist:
            assert rccl_ranks == world == args.gpus, "RCCL world differs from --gpus"
        line = {
            "metric": "EMult pairings/sec at 1024-bit, batch=2^%d per GPU" % args.batch_log2,
            "value": value, "unit": "pairings/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 (29-bit limbs, 64-bit accumulators)", "data": "synthetic",
            "config": {"workload": "configs[2]: 1024-bit params, batch=2^%d EMult (Tate pairing G1xG1->GT, "
                                   "Miller+final-exp) per MI355X; operands = Config 2's Encrypt outputs (random 40-bit "
                                   "m, full-length r) x a fixed permutation of them" % args.batch_log2,
                       "key": fx["name"], "fp_bits": int(fx["p"], 16).bit_length(), "limbs29": syn.limbs_for(int(fx["p"], 16)),
                       "batch_per_gpu": per_gpu, "global_batch": total, "rccl_ranks": rccl_ranks,
                       "distinct_pairs_in_checked_sample": distinct, "checked_sample_stride": chk_stride,
                       "checked_run_positions": run_positions,
                       "parallelism": ("batch-sharded x%d, one process per GPU + RCCL all-gather of results" % world)
                       if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_passes": ({k: live[k] for k in ("FETCH_SIZE_KB", "WRITE_SIZE_KB", "launches", "fetch_calibration", "seconds")}
                                            if live is not None and full else None),
                         "kernel": kernel_name, "kernel_ms": k_ms,
                         "algorithmic_bytes_per_pairing": alg_bytes, "note": "per GPU (rank 0's kernel)"},
            "roofline_valu": {"bound": "v_mad_u64_u32 issue", "mads_per_pairing": mads, "achieved": mad_rate,
                              "unit": "lane-MAD/s",
                              "peak": VALU_MAD_PEAK_4W, "frac": mad_rate / VALU_MAD_PEAK_4W,
                              "peak_at_1_wave_per_simd": VALU_MAD_PEAK_1W,
                              "frac_at_1_wave_per_simd": mad_rate / VALU_MAD_PEAK_1W},
        }
        if strong:
            line["strong_2^%d" % args.batch_log2] = strong
        if dec:
            # BASELINE's metric names "EMult pairings/sec + BSGS decrypts/sec": the second headline
            top = dec.get(20) or dec[max(dec)]
            line["decrypt"] = {"metric": "BSGS decrypts/sec at 1024-bit, T=2^40, batch=2^%d%s" %
                                         (20 if 20 in dec else max(dec), " per GPU" if use_dist else ""), **top}
            if extra is not None:
                for k, e in dec.items():
                    extra["decrypt" if k == 16 else "decrypt_2^%d" % k] = e
        if extra:
            line["extra"] = extra
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(
                fx, a_h, b_h, o_h, sample_note="the sample is every %d-th pair of the batch (run positions %d..%d of "
                "the lanes' shared-inversion runs)" % (chk_stride, run_positions[0], run_positions[-1]))
        emit_line(line)
    if use_dist:
        dist.barrier()                                   # the other ranks wait for rank 0's CPU leg here
        dist.destroy_process_group()


def decrypt_sharded(pk, fx, dev, cts, xs, world, rank, dist, timed_region, args):
    """configs[3] on every rank: T = 2^40 BSGS Decrypt of this rank's batch (its Config-2 ciphertexts, every 16th
    negated, every 4096th out of range), plaintexts and statuses gathered to every rank (9 bytes per ciphertext:
    int64 + status, one RCCL all-gather).  Tables are per key and per GPU: every rank sets them up (not timed)."""
    import torch
    import bgn_amd
    import bgn_amd.synthetic as syn
    from bgn_amd.sharding import gather_shards
    eng = pk.engine
    EB = eng.elem_bytes
    t0 = time.perf_counter()
    pk.SetupDecryption(bgn_amd.SecretKey(int(fx["q1"], 16)))
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t0
    mixed, want, want_st = syn.decrypt_mix(pk, fx, cts, xs, dev)
    n = cts.numel() // EB
    total = n * world
    m = torch.empty(n, dtype=torch.int64, device=dev)
    st = torch.empty(n, dtype=torch.uint8, device=dev)
    packed = torch.empty(n, 9, dtype=torch.uint8, device=dev)
    got = [None]
    lift_ms = []

    def step():
        eng.decrypt_dev(1, mixed, m, st, n)
        lift_ms.append(eng.last_aux_kernel_ms())
        packed[:, :8] = m.view(torch.uint8).reshape(n, 8)
        packed[:, 8] = st
        got[0] = gather_shards(packed.reshape(-1), total, 9, world, rank, dist)

    dt = timed_region(step, args.steps, args.warmup)
    ok = bool((m.cpu() == want[:n]).all().item()) and bool((st.cpu() == want_st[:n]).all().item())
    mine = got[0].reshape(total, 9)[rank * n:(rank + 1) * n]
    ok = ok and bool((mine == packed).all().item())
    S = int(eng._lib.bgn_ctx_bsgs_baby_steps(eng._h))
    alg = EB + 16
    k_ms = sum(lift_ms[args.warmup:]) / max(1, len(lift_ms[args.warmup:]))
    e = op_rooflines(
        {"value": total * args.steps / dt, "unit": "decrypts/s", "n_gpus": world, "batch": n, "global_batch": total, "level": 1,
         "scaling": "weak", "ms_per_step": dt / args.steps * 1e3,
         "workload": "configs[3]: T=2^40 BSGS Decrypt, batch=2^%d per GPU, m uniform in [0,2^40), 1/16 negative, 1/4096 "
                     "out of range; sharded by ciphertext, plaintexts and statuses all-gathered (RCCL); baby table of %d "
                     "entries per GPU" % (args.batch_log2, S),
         "table_setup_s": t_setup, "plaintexts_and_statuses_exact": ok, "algorithmic_bytes_per_unit": alg},
        syn.decrypt_counts(fx, S), syn.limbs_for(int(fx["p"], 16)), n_gpus=world)
    e["roofline"] = {"bound": "hbm", "achieved": alg * n / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg * n / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "kernel": eng.last_aux_kernel_name(), "kernel_ms": k_ms, "algorithmic_bytes_per_decrypt": alg,
                     "note": "per GPU (rank 0's lift kernel)"}
    return e


class MultPolyJob:
    """BASELINE configs[4] on one rank's polynomials: `npoly` MultPoly instances of 16x16 level-1 coefficient
    polynomials (d1*d2 pairings and the GT accumulation into 31 coefficients each, poly.go:123-156) followed by one
    AddPoly of the products with each other (poly.go:171-207: product q + product q + npoly/2, coefficient-wise GT
    products).  Coefficients: Encrypt of base-3 digits in {-1, 0, 1} (plaintext.go:209-266) on the GPU, not timed."""

    def __init__(self, pk, npoly, dev, seed_a, seed_b, d1=16, d2=16):
        import torch
        import bgn_amd.synthetic as syn
        self.eng = eng = pk.engine
        self.EB = EB = eng.elem_bytes
        self.npoly, self.d1, self.d2 = npoly, d1, d2
        _, _, self.ca = syn.config2_ciphertexts(pk, npoly * d1, seed=seed_a, device=dev, digits=True)
        _, _, self.cb = syn.config2_ciphertexts(pk, npoly * d2, seed=seed_b, device=dev, digits=True)
        self.prod = torch.empty(npoly * (d1 + d2) * EB, dtype=torch.uint8, device=dev)
        self.summ = torch.empty(((npoly + 1) // 2) * (d1 + d2) * EB, dtype=torch.uint8, device=dev)
        self.mult_kernel = ""

    def step(self):
        eng, EB = self.eng, self.EB
        eng.poly_mult_dev(self.npoly, self.d1, self.d2, self.ca, self.cb, self.prod)
        self.mult_kernel = eng.last_kernel_name()
        half = self.npoly // 2
        if half:
            n = half * (self.d1 + self.d2)
            eng.add_dev(2, self.prod[: n * EB], self.prod[n * EB: 2 * n * EB], self.summ, n)

    def entry(self, fx, dt, world):
        """The measurement of one step of `dt` seconds over `world` GPUs (npoly = the whole job's polynomials)."""
        import bgn_amd.synthetic as syn
        EB, d1, d2 = self.EB, self.d1, self.d2
        nl = syn.limbs_for(int(fx["p"], 16))
        pairs = self.npoly * world * d1 * d2
        # algorithmic bytes of a step: both coefficient arrays in, the product polynomials out, then the AddPoly's two
        # operand halves in and its sums out ("780 B + amortised output" per pair, stated exactly here)
        npo = self.npoly * world
        alg = (npo * (d1 + d2) + npo * (d1 + d2) + 3 * (npo // 2) * (d1 + d2)) * EB
        e = op_rooflines(
            {"value": pairs / dt, "unit": "coefficient pairs/s", "polys": npo, "d1": d1, "d2": d2,
             "ms_per_step": dt * 1e3,
             "workload": "configs[4]: MultPoly of 2^%d pairs of 16x16-coefficient level-1 ciphertext polynomials (%d "
                         "coefficient pairs; Karatsuba over the bilinear pairing, per-coefficient line tables, segmented "
                         "GT accumulation) + one AddPoly of the products, on %d GPU(s)" %
                         (npo.bit_length() - 1, pairs, world),
             "kernel": self.mult_kernel, "algorithmic_bytes_per_unit": alg / pairs},
            syn.multpoly_counts_per_pair(fx, d1), nl, n_gpus=world)
        e["roofline"] = {"bound": "hbm", "achieved": alg / dt / 1e9 / world, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / dt / 1e9 / world / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_step": alg, "kernel": self.mult_kernel,
                         "note": "per GPU, over the whole step (table builds, walks, accumulation, AddPoly: several "
                                 "kernels, none dominant by bytes); the bound that applies is roofline_valu"}
        return e


def bench_multpoly(args, pk, fx, dev, world, rank, use_dist, rccl_ranks):
    """BASELINE configs[4] as its own workload (--workload multpoly): the 2^14 polynomials sharded by polynomial —
    rank g owns polynomials [g*N/G, (g+1)*N/G), results all-gathered (RCCL).  Strong scaling: the job is 2^14
    polynomials whatever N."""
    import torch
    import torch.distributed as dist
    from bgn_amd.sharding import gather_shards, shard_range
    EB = pk.engine.elem_bytes
    npoly = 1 << args.polys_log2
    lo, hi = shard_range(npoly, world, rank)
    mine = hi - lo
    job = MultPolyJob(pk, mine, dev, seed_a=2000 + rank, seed_b=3000 + rank)
    d1, d2, prod = job.d1, job.d2, job.prod
    gathered = None

    def step():
        nonlocal gathered
        job.step()
        if use_dist:
            gathered = gather_shards(prod, npoly, (d1 + d2) * EB, world, rank, dist)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        assert bool((gathered[lo * (d1 + d2) * EB: hi * (d1 + d2) * EB] == prod).all().item()), "gather mismatch"
    if rank == 0:
        job.npoly = npoly // world if world > 1 else npoly      # entry() prices the whole job over `world` GPUs
        e = job.entry(fx, dt / args.steps, world) if npoly % world == 0 else None
        pairs = npoly * d1 * d2
        value = pairs * args.steps / dt
        line = {"metric": "MultPoly coefficient pairs/sec at 1024-bit (configs[4])", "value": value,
                "unit": "coefficient pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "u32 (29-bit limbs, 64-bit accumulators)", "data": "synthetic",
                "config": {"workload": "configs[4]: 1024-bit poly.go MultPoly of 2^%d pairs of 16x16 coefficient "
                                       "polynomials (2^%d coefficient pairs) + one AddPoly, sharded by polynomial over "
                                       "%d GPU(s) + RCCL all-gather" % (args.polys_log2, args.polys_log2 + 8, world),
                           "key": fx["name"], "polys": npoly, "polys_per_gpu": mine, "d1": d1, "d2": d2,
                           "rccl_ranks": rccl_ranks}}
        if e:
            line["roofline"] = e["roofline"]
            line["roofline_valu"] = e["roofline_valu"]
        emit_line(line)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
