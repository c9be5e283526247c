"""GPU: the multi-GPU path on the HIP engine.

* bgn_mctx_* (one process, several devices): device lists that name cuda:0 more than once give real
  multi-context sharding on a one-GPU box — ragged shards, MultPoly by polynomial, the device-resident form
  with and without the peer-staging path — compared with the single-context engine and the oracle.
* ShardedOps (one process per rank): the code tests/test_sharding_gloo.py runs on CPU, here with the HIP engine
  at world = 1 in-process and at world = 2 with two gloo ranks sharing cuda:0.
"""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, engine_key, load_fixture

pytestmark = pytest.mark.gpu


def _pool(fx, n, step=1, off=0):
    cts = [bytes.fromhex(e["ct"]) for e in fx["encrypt"]]
    return b"".join(cts[(step * i + off) % len(cts)] for i in range(n))


def _device_lists():
    """[0, 0, 0] always: three contexts on one device are real multi-context sharding on a one-GPU box.  On a box with
    two or more GPUs the same tests also run over [0, 1] and over every visible device — the first thing a
    multi-GPU run proves: peer access enabled, hipMemcpyPeerAsync across xGMI, a root that is
    not every shard's device.  device_count() does not initialise the GPU (safe at collection time)."""
    import torch
    n = torch.cuda.device_count()
    skip = pytest.mark.skip(reason="needs more GPUs: this box shows %d" % n)
    return [pytest.param([0, 0, 0], id="same3"),
            pytest.param([0, 1], id="two", marks=[] if n >= 2 else [skip]),
            pytest.param(list(range(max(n, 3))), id="all", marks=[] if n >= 3 else [skip])]


@pytest.fixture(scope="module", params=_device_lists())
def multi(request):
    import bgn_amd
    devices = request.param
    fx = load_fixture("k256")
    me = bgn_amd.MultiEngine(int(fx["p"], 16), int(fx["n"], 16), fx["l"], bytes.fromhex(fx["P"]), bytes.fromhex(fx["Q"]),
                             True, devices=devices)
    me.set_secret(int(fx["q1"], 16))
    me.setup_decryption(fx["msg_space"])
    yield fx, me
    me.close()


@pytest.mark.parametrize("count", [1, 2, 7, 64])
def test_mctx_host_buffers_match_single_context(multi, count):
    fx, me = multi
    pk, sk = engine_key(fx)
    pk.SetupDecryption(sk)
    eng = pk.engine
    assert me.shard_ranges(count)[-1][1] == count
    a, b = _pool(fx, count), _pool(fx, count, 3, 1)
    assert me.mult(a, b).tobytes() == eng.mult(a, b).tobytes()
    assert me.add(1, a, b).tobytes() == eng.add(1, a, b).tobytes()
    assert me.sub(1, a, b).tobytes() == eng.sub(1, a, b).tobytes()
    assert me.make_l2(a).tobytes() == eng.make_l2(a).tobytes()
    xs = [(17 * i + 3) % fx["msg_space"] for i in range(count)]
    rs = [(1234567 * i + 99) % int(fx["n"], 16) for i in range(count)]
    assert me.encrypt(xs, rs).tobytes() == eng.encrypt(xs, rs).tobytes()
    ks = [5 + i for i in range(count)]
    assert me.multconst(1, a, ks).tobytes() == eng.multconst(1, a, ks).tobytes()
    m1, s1 = me.decrypt(1, a)
    m0, s0 = eng.decrypt(1, a)
    assert m1.tolist() == m0.tolist() and s1.tolist() == s0.tolist()


@pytest.mark.parametrize("npoly,d1,d2", [(1, 2, 2), (5, 3, 2), (8, 4, 4)])
def test_mctx_multpoly_shards_by_polynomial(multi, npoly, d1, d2):
    fx, me = multi
    import oracle_c
    o = oracle_c.Oracle.from_fixture(fx)
    a, b = _pool(fx, npoly * d1), _pool(fx, npoly * d2, 5, 2)
    assert me.poly_mult(npoly, d1, d2, a, b).tobytes() == o.poly_mult(npoly, d1, d2, a, b)


@pytest.mark.parametrize("staging", ["0", "1"])
def test_mctx_device_resident_forms(multi, staging):
    """Arrays resident on cuda:0, gathered into one output array on cuda:0; option mctx_force_staging = 1 sends every
    shard through the peer-copy path (scratch on the shard's device, hipMemcpyPeerAsync in and out).  With more than
    one device in the list the shards of the other devices take that path by themselves."""
    import torch
    fx, me = multi
    me.set_option("mctx_force_staging", int(staging))
    pk, _ = engine_key(fx)
    eng = pk.engine
    count, E = 11, eng.elem_bytes
    a, b = _pool(fx, count), _pool(fx, count, 3, 1)
    ta = torch.frombuffer(bytearray(a), dtype=torch.uint8).cuda()
    tb = torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()
    out = torch.zeros(count * E, dtype=torch.uint8, device="cuda")
    me.mult_dev(ta, tb, out, root=0)
    assert out.cpu().numpy().tobytes() == eng.mult(a, b).tobytes()
    m = torch.zeros(count, dtype=torch.int64, device="cuda")
    st = torch.ones(count, dtype=torch.uint8, device="cuda")
    me.decrypt_dev(1, ta, m, st, root=0)
    m0, s0 = eng.decrypt(1, a)
    assert m.cpu().tolist() == m0.tolist() and st.cpu().tolist() == s0.tolist()
    npoly, d = 5, 2
    po = torch.zeros(npoly * 2 * d * E, dtype=torch.uint8, device="cuda")
    me.poly_mult_dev(npoly, d, d, ta[: npoly * d * E], tb[: npoly * d * E], po, root=0)
    assert po.cpu().numpy().tobytes() == eng.poly_mult(npoly, d, d, a[: npoly * d * E], b[: npoly * d * E]).tobytes()


def test_mctx_device_resident_forms_wait_for_the_callers_stream(multi):
    """The ordering contract of the _dev calls (include/bgn_amd.h): operands produced asynchronously on a side
    stream of the root device — here behind tens of milliseconds of queued work — and a pending fill of the
    result array are waited for by every shard (an event on the caller's stream), staging path forced.  Without
    the wait the shards would pair the zero bytes the arrays held before."""
    import torch
    fx, me = multi
    me.set_option("mctx_force_staging", 1)
    pk, _ = engine_key(fx)
    eng = pk.engine
    count, E = 13, eng.elem_bytes
    a, b = _pool(fx, count), _pool(fx, count, 3, 1)
    src_a = torch.frombuffer(bytearray(a), dtype=torch.uint8).cuda()
    src_b = torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()
    ta, tb = torch.zeros_like(src_a), torch.zeros_like(src_b)
    out = torch.empty(count * E, dtype=torch.uint8, device="cuda")
    big = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    before = torch.cuda.current_device()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(40):                       # queued work in front of the operands
            big.fill_(7)
        ta.copy_(src_a, non_blocking=True)
        tb.copy_(src_b, non_blocking=True)
        out.fill_(255)                            # a pending fill of the result array
        me.mult_dev(ta, tb, out, root=0)          # takes torch's current stream of the root device: `side`
    assert torch.cuda.current_device() == before
    assert out.cpu().numpy().tobytes() == eng.mult(a, b).tobytes()


def test_sharded_ops_world1_on_the_engine():
    """The sharder of the multi-process form (bgn_amd/sharding.py) driving the HIP engine at world = 1."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_sharding_gloo import sharded_checks
    fx = load_fixture("toy64")
    pk, sk = engine_key(fx)
    pk.SetupDecryption(sk)
    assert sharded_checks(lambda: pk.engine, fx, 7, 1, 0, None)


def _gpu_rank(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bgn_amd
    from test_sharding_gloo import sharded_checks
    fx = load_fixture("toy64")

    def make():
        pk = bgn_amd.PublicKey(int(fx["p"], 16), int(fx["n"], 16), fx["l"], bytes.fromhex(fx["P"]),
                               bytes.fromhex(fx["Q"]), fx["msg_space"], True, fx["poly_base"], device=0)
        pk.SetupDecryption(bgn_amd.SecretKey(int(fx["q1"], 16)))
        return pk.engine

    ok = sharded_checks(make, fx, total, world, rank, dist)
    q.put((rank, ok))
    dist.destroy_process_group()


def test_sharded_ops_two_ranks_share_one_gpu():
    """Two ranks (gloo rendezvous; both on cuda:0, the box has one GPU) each run their shard on their own HIP
    engine; the gathered arrays equal the oracle's on the whole batch."""
    import torch.multiprocessing as mp
    from test_sharding_gloo import _free_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_rank, args=(r, 2, port, 9, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def test_mctx_root_is_not_a_shard_device():
    """Operands resident on the LAST visible device while the contexts live on the others: every shard fetches its
    slice by peer DMA and writes its results back the same way (asynchronously produced operands).  Two GPUs or more."""
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("needs two or more GPUs: this box shows %d" % n)
    import bgn_amd
    fx = load_fixture("k256")
    pk, _ = engine_key(fx)
    eng = pk.engine
    root = n - 1
    me = bgn_amd.MultiEngine(int(fx["p"], 16), int(fx["n"], 16), fx["l"], bytes.fromhex(fx["P"]), bytes.fromhex(fx["Q"]),
                             True, devices=list(range(n - 1)) or [0])
    try:
        count, E = 37, eng.elem_bytes
        a, b = _pool(fx, count), _pool(fx, count, 3, 1)
        dev = torch.device("cuda", root)
        src_a = torch.frombuffer(bytearray(a), dtype=torch.uint8).to(dev)
        src_b = torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
        ta, tb = torch.zeros_like(src_a), torch.zeros_like(src_b)
        out = torch.zeros(count * E, dtype=torch.uint8, device=dev)
        side = torch.cuda.Stream(device=dev)
        with torch.cuda.device(dev), torch.cuda.stream(side):
            ta.copy_(src_a, non_blocking=True)
            tb.copy_(src_b, non_blocking=True)
            me.mult_dev(ta, tb, out, root=root)
        assert out.cpu().numpy().tobytes() == eng.mult(a, b).tobytes()
    finally:
        me.close()


def test_bench_two_ranks_over_rccl():
    """bench.py --gpus 2 --steps 1: two rank processes, one per GPU, RCCL all-gather of the shards — the bench
    contract's N > 1 form, on a box that has the GPUs for it."""
    import json
    import subprocess
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("needs two or more GPUs: this box shows %d" % n)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BGN_BENCH_TIMEOUT_S="900")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--batch-log2", "16", "--no-extra", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["rccl_ranks"] == 2 and line["value"] > 0


def _bench_line(argv, spawn):
    """One run of bench.py as a child process (fresh: this test process has initialised the GPU and never re-execs).
    spawn=True goes through bench.py's own rank spawner (BGN_BENCH_SPAWN=1: a parent that never touches the GPU starts
    rank 0 with RANK / WORLD_SIZE / MASTER_* set, exactly as for --gpus N)."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BGN_BENCH_TIMEOUT_S="600")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    if spawn:
        env["BGN_BENCH_SPAWN"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "bench.py must print ONE JSON line on stdout, got %d:\n%s" % (len(lines), r.stdout[-2000:])
    return json.loads(lines[0])


def _keep(name, line):
    """Leave the line under gpurun_out/ (scratch that travels back from the GPU box) so that the one committed under
    profiles/ is the one this test produced.  Best effort: the driver's box may not want it."""
    import json
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name), "w") as f:
            json.dump(line, f, indent=1)
    except OSError:
        pass


def test_bench_one_rank_over_rccl():
    """The rank path of bench.py on the final build with ONE rank: spawned child, `nccl` (= RCCL) process group,
    all_gather_into_tensor of the result shards, the strong-scaling block, the sharded Decrypt with its gather of
    plaintexts and statuses — everything the driver's 2/4/8-GPU runs execute except a second GPU.  The rate has to
    be that of the undistributed call (the gather of one shard is a device copy)."""
    argv = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--batch-log2", "16", "--no-cpu-baseline"]
    line = _bench_line(argv + ["--force-dist"], spawn=True)
    _keep("r06_bench_dist1_line.json", line)
    assert line["n_gpus"] == 1 and line["config"]["rccl_ranks"] == 1 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 1 << 16 and line["value"] > 0
    strong = line["strong_2^16"]
    assert strong["scaling"] == "strong" and strong["batch_per_gpu"] == 1 << 16 and strong["value"] > 0
    dec = line["decrypt"]
    assert dec["plaintexts_and_statuses_exact"] is True and dec["global_batch"] == 1 << 16 and dec["value"] > 0
    assert dec["roofline"]["kernel_ms"] > 0 and line["roofline"]["kernel_ms"] > 0
    plain = _bench_line(argv + ["--no-extra"], spawn=False)
    assert plain["config"]["rccl_ranks"] == 0 and "strong_2^16" not in plain
    assert plain["roofline"]["kernel"] == line["roofline"]["kernel"]
    ok = all(abs(v / plain["value"] - 1) < 0.03 for v in (line["value"], strong["value"]))
    if not ok:                                        # two steps of 150 ms each: one noisy sample gets a second pair
        line2 = _bench_line(argv + ["--force-dist"], spawn=True)
        plain = _bench_line(argv + ["--no-extra"], spawn=False)
        line, strong = line2, line2["strong_2^16"]
    for v in (line["value"], strong["value"]):
        assert abs(v / plain["value"] - 1) < 0.03, (line["value"], strong["value"], plain["value"])


def test_bench_multpoly_one_rank_over_rccl():
    """--workload multpoly through the same rank path: polynomials sharded by polynomial (one shard here), product
    polynomials all-gathered over RCCL, strong scaling."""
    argv = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--workload", "multpoly", "--polys-log2", "8"]
    line = _bench_line(argv + ["--force-dist"], spawn=True)
    _keep("r06_bench_multpoly_dist1_line.json", line)
    assert line["n_gpus"] == 1 and line["config"]["rccl_ranks"] == 1 and line["scaling"] == "strong"
    assert line["config"]["polys"] == 256 and line["config"]["polys_per_gpu"] == 256 and line["value"] > 0
    assert line["roofline"]["kernel"] and line["roofline_valu"]["frac"] > 0
    plain = _bench_line(argv, spawn=False)
    assert plain["config"]["rccl_ranks"] == 0
    if abs(line["value"] / plain["value"] - 1) >= 0.03:      # one noisy sample gets a second pair
        line = _bench_line(argv + ["--force-dist"], spawn=True)
        plain = _bench_line(argv, spawn=False)
    assert abs(line["value"] / plain["value"] - 1) < 0.03, (line["value"], plain["value"])
