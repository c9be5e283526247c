#!/usr/bin/env python3
"""Randomised differential run on the GPU: random operations at random batch sizes (log-uniform, so the dispatch
boundaries between the cooperative, the lane-group and the one-element-per-lane kernels and the cuts into whole
rounds are crossed all the time), each computed under the default dispatch and again with a kernel family forced or a
planner switched off; every byte / plaintext / status must agree, and a sample of every batch is checked against
the C oracle (test infrastructure: this tool is not part of the product).
    python tools/soak.py [seconds] [seed] > profiles/r03_soak.txt"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import torch  # noqa: E402

from conftest import load_fixture  # noqa: E402
import bgn_amd  # noqa: E402
import bgn_amd.synthetic as syn  # noqa: E402
import oracle_c  # noqa: E402

ENGINES = []


def force(kernel, **extra):
    """Every dispatch alternative is an option of the contexts (bgn_ctx_set_option); nothing goes through the
    environment."""
    for eng in ENGINES:
        eng.reset_options()
        if kernel == "one launch":
            eng.set_option("split_rounds", 0)
        elif kernel != "default":
            eng.force_kernel(kernel)
        for k, v in extra.items():
            eng.set_option(k, v)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    dev = torch.device("cuda", 0)
    keys = {}
    NMAX = 1 << 18
    names = os.environ.get("SOAK_KEYS", "k256,k512,k1024,k1024b").split(",")
    for name in names:
        fx = load_fixture(name)
        pk = bgn_amd.PublicKey(int(fx["p"], 16), int(fx["n"], 16), fx["l"], bytes.fromhex(fx["P"]), bytes.fromhex(fx["Q"]),
                               fx["msg_space"], True, fx["poly_base"])
        pk.engine.set_memory_budget(60 << 30)
        pk.SetupDecryption(bgn_amd.SecretKey(int(fx["q1"], 16)))
        eng = pk.engine
        ENGINES.append(eng)
        EB = eng.elem_bytes
        T = fx["msg_space"]
        g = torch.Generator().manual_seed(seed * 7 + len(keys))
        nbytes = 3 if T >= (1 << 24) else 1
        vals = torch.randint(0, min(T, 1 << (8 * nbytes)) - 1, (NMAX,), generator=g, dtype=torch.int64)
        xs = torch.empty((NMAX, nbytes), dtype=torch.uint8)
        v = vals.clone()
        for j in range(nbytes - 1, -1, -1):
            xs[:, j] = (v & 0xFF).to(torch.uint8)
            v >>= 8
        r_len, top_mask = syn._r_shape(int(fx["n"], 16))
        rs = torch.randint(0, 256, (NMAX, r_len), dtype=torch.uint8, generator=g)
        rs[:, 0] &= top_mask
        xs, rs = xs.to(dev), rs.to(dev)
        cts = torch.empty(NMAX * EB, dtype=torch.uint8, device=dev)
        eng.encrypt_dev(xs, nbytes, rs, r_len, cts, NMAX)
        perm = syn.permuted_copy(cts, EB, seed + 5)
        l2 = torch.empty_like(cts)
        force("default")
        eng.make_l2_dev(cts, l2, NMAX)
        torch.cuda.synchronize()
        keys[name] = dict(fx=fx, pk=pk, eng=eng, EB=EB, vals=vals, cts=cts, perm=perm, l2=l2, xs=xs, rs=rs, nbytes=nbytes, r_len=r_len,
                          oracle=oracle_c.Oracle.from_fixture(fx))
    print("# soak: %d s, seed %d, keys %s" % (seconds, seed, ",".join(keys)), flush=True)
    t_end = time.time() + seconds
    done = {}
    elements = 0
    last_note = time.time()
    while time.time() < t_end:
        name = rng.choice(list(keys))
        K = keys[name]
        eng, EB = K["eng"], K["EB"]
        op = rng.choice(["mult", "mult", "make_l2", "decrypt_l1", "decrypt_l2", "multpoly", "add_l1", "add_l2", "encrypt",
                         "multconst_l1", "multconst_l2", "callers"])
        hi = {"k256": 18, "k512": 17.6, "k1024": 17.2, "k1024b": 16.5, "k2048": 12.5}[name]
        n = max(1, int(2 ** rng.uniform(0, hi)))
        if rng.random() < 0.25 and name != "k2048":          # around the round boundaries
            n = min(NMAX, rng.choice([65536, 131072]) + rng.randrange(-3000, 3000))
        off = rng.randrange(0, NMAX - n + 1)
        a = K["cts"][off * EB: (off + n) * EB]
        b = K["perm"][off * EB: (off + n) * EB]
        variants = ["default"]
        if name == "k2048":                                   # 72 limbs: no cooperative kernel, the lane kernels are the slow functional ones
            variants += ["lane"] if n <= 48 else []
        else:
            if n <= 20000:
                variants += ["coop"]
            variants += [rng.choice(["quad", "lane"])]
        if n > 65536:
            variants += ["one launch"]
        ref = None
        if op in ("mult", "make_l2"):
            for kv in variants:
                force(kv)
                out = torch.empty(n * EB, dtype=torch.uint8, device=dev)
                if op == "mult":
                    eng.mult_dev(a, b, out, n)
                else:
                    eng.make_l2_dev(a, out, n)
                torch.cuda.synchronize()
                if ref is None:
                    ref = out
                    k = min(n, 3)
                    sa, sb = bytes(a[: k * EB].cpu().numpy()), bytes(b[: k * EB].cpu().numpy())
                    want = K["oracle"].mult(sa, sb) if op == "mult" else K["oracle"].mult(sa)      # makeL2 = e(., P)
                    assert bytes(out[: k * EB].cpu().numpy()) == want, (name, op, n, "oracle")
                else:
                    assert torch.equal(ref, out), (name, op, n, kv)
        elif op in ("decrypt_l1", "decrypt_l2"):
            lvl = 1 if op == "decrypt_l1" else 2
            src = a if lvl == 1 else K["l2"][off * EB: (off + n) * EB]
            want = K["vals"][off: off + n]
            for kv in variants:
                force(kv)
                m = torch.empty(n, dtype=torch.int64, device=dev)
                st = torch.empty(n, dtype=torch.uint8, device=dev)
                eng.decrypt_dev(lvl, src, m, st, n)
                torch.cuda.synchronize()
                assert not bool(st.any().item()) and torch.equal(m.cpu(), want), (name, op, n, kv)
        elif op in ("multconst_l1", "multconst_l2"):
            # per-element scalars of a random length, the special ones among them (0, 1, chains of the digits 8 / -1,
            # multiples of the group order: the ladder's flagged fallback); lane groups against one element per lane
            lvl = 1 if op == "multconst_l1" else 2
            nn = int(K["fx"]["n"], 16)
            klen = rng.choice([1, 3, 5, 8, 15, 16, 32, (nn.bit_length() + 7) // 8, (nn.bit_length() + 7) // 8 + 1])
            n = min(n, 70000 if klen <= 16 else 8192 if name != "k2048" else 64)
            src = (a if lvl == 1 else K["l2"][off * EB: (off + n) * EB])[: n * EB]
            g = torch.Generator().manual_seed(rng.randrange(1 << 30))
            ks = torch.randint(0, 256, (n, klen), dtype=torch.uint8, generator=g)
            special = [0, 1, int("8" * (2 * klen), 16), (16 ** (2 * klen - 1)) - 1]
            if 8 * klen > nn.bit_length():
                special += [nn, 2 * nn, nn - 1]
            for i, v in enumerate(special[: n]):
                ks[i] = torch.tensor(list(int(v % (1 << (8 * klen))).to_bytes(klen, "big")), dtype=torch.uint8)
            ksd = ks.to(dev)
            kvs = ["default"] + (["quad", "lane"] if (name != "k2048" or n <= 48) else [])
            if lvl == 2 and name != "k2048":
                kvs += ["lane, general power"]                # the lane kernel without the norm-1 ladder (round 6)
            if lvl == 1 and 3 <= klen < 16 and name != "k2048":
                kvs += ["lane, binary ladder"]                # the lane kernel without its 2-bit windows (round 6)
            for kv in kvs:
                if kv == "lane, general power":
                    force("lane", multconst_l2_ladder=0)
                elif kv == "lane, binary ladder":
                    force("lane", g1_mul_window_short=0)
                else:
                    force(kv)
                out = torch.empty(n * EB, dtype=torch.uint8, device=dev)
                eng._lib.bgn_multconst_batch_dev(eng._h, n, lvl, src.data_ptr(), ksd.data_ptr(), klen, None, 0, out.data_ptr(), eng._stream())
                torch.cuda.synchronize()
                if ref is None:
                    ref = out
                    k = min(n, 6 if klen > 16 else 12)
                    want = K["oracle"].multconst(lvl, bytes(src[: k * EB].cpu().numpy()),
                                                 [int.from_bytes(bytes(ks[i].numpy()), "big") for i in range(k)])
                    assert bytes(out[: k * EB].cpu().numpy()) == want, (name, op, n, klen, "oracle")
                else:
                    assert torch.equal(ref, out), (name, op, n, klen, kv)
        elif op == "callers":
            # the reference's call shape: threads issuing small host-buffer calls of mixed kinds on this context at
            # once (the combiner merges them); every result equals the device-resident batch
            import threading
            force("default")
            nthreads, per = rng.choice([(8, 4), (32, 2), (64, 1)]), None
            nthreads, per = nthreads
            tot = nthreads * per
            n = tot
            off2 = rng.randrange(0, NMAX - tot + 1)
            ha = K["cts"][off2 * EB: (off2 + tot) * EB].cpu().numpy().reshape(tot, EB)
            hb = K["perm"][off2 * EB: (off2 + tot) * EB].cpu().numpy().reshape(tot, EB)
            want_m = eng.mult(ha.tobytes(), hb.tobytes())
            want_a = eng.add(1, ha.tobytes(), hb.tobytes())
            want_d = K["vals"][off2: off2 + tot].tolist()
            errs = []

            def worker(t):
                try:
                    for j in range(per):
                        i = t * per + j
                        which = (t + j) % 3
                        if which == 0:
                            assert eng.mult(ha[i].tobytes(), hb[i].tobytes()).tobytes() == want_m[i].tobytes()
                        elif which == 1:
                            assert eng.add(1, ha[i].tobytes(), hb[i].tobytes()).tobytes() == want_a[i].tobytes()
                        else:
                            m, st = eng.decrypt(1, ha[i].tobytes())
                            assert int(m[0]) == want_d[i] and int(st[0]) == 0
                except Exception as e:                        # noqa: BLE001
                    errs.append((t, repr(e)))

            th = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            assert not errs, (name, op, errs[:2])
        elif op in ("add_l1", "add_l2"):
            # Add and Sub on both levels: the one-launch wire-to-wire kernels (round 6) against the decode / decode /
            # add / encode routes they replace, and a sample against the C oracle
            lvl = 1 if op == "add_l1" else 2
            sa_dev = a if lvl == 1 else K["l2"][off * EB: (off + n) * EB]
            o2 = rng.randrange(0, NMAX - n + 1)
            sb_dev = b if lvl == 1 else K["l2"][o2 * EB: (o2 + n) * EB]
            sub = rng.random() < 0.5
            fn = eng._lib.bgn_sub_batch_dev if sub else eng._lib.bgn_add_batch_dev
            for env in ({}, {"l1_fused": 0, "l2_fused": 0}):
                force("default", **env)
                out = torch.empty(n * EB, dtype=torch.uint8, device=dev)
                assert fn(eng._h, n, lvl, sa_dev.data_ptr(), sb_dev.data_ptr(), None, 0, out.data_ptr(), eng._stream()) == 0
                torch.cuda.synchronize()
                if ref is None:
                    ref = out
                    k = min(n, 4)
                    j = rng.randrange(0, n - k + 1)
                    ha, hb = bytes(sa_dev[j * EB: (j + k) * EB].cpu().numpy()), bytes(sb_dev[j * EB: (j + k) * EB].cpu().numpy())
                    assert bytes(out[j * EB: (j + k) * EB].cpu().numpy()) == K["oracle"].add(lvl, ha, hb, sub), (name, op, n, sub)
                else:
                    assert torch.equal(ref, out), (name, op, n, sub, env)
        elif op == "encrypt":
            # the chain kernels and the lane groups (k_g1_fixed_quad): one is the default, the other forced
            alts = [{}]
            if name == "k2048":
                alts += [{"quad_max_enc": 0}] if n <= 4096 else []
            else:
                alts += [{"quad_max_enc": 0}]                         # the chain kernels at every size
                if n <= 100000:
                    alts += [{"quad_max_enc": 1 << 20}]               # the lane groups beyond their default range too
            for env in alts:
                force("default", **env)
                out = torch.empty(n * EB, dtype=torch.uint8, device=dev)
                eng.encrypt_dev(K["xs"][off: off + n], K["nbytes"], K["rs"][off: off + n], K["r_len"], out, n)
                torch.cuda.synchronize()
                assert torch.equal(out, a), (name, op, n, env)      # the same inputs gave cts at set-up (one launch of 2^18)
        else:
            d1, d2 = rng.choice([(2, 2), (3, 5), (4, 4), (8, 8), (5, 1), (6, 6), (16, 16), (7, 3)])
            npoly = max(1, min(n // (d1 * d2), NMAX // max(d1, d2) - 1, 3000))
            pa = K["cts"][: npoly * d1 * EB]
            pb = K["perm"][: npoly * d2 * EB]
            n = npoly * d1 * d2
            for kv, env in (("default", {}), ("direct", {"poly_tables": 0, "poly_karatsuba": 0}), ("tables", {"poly_tables": 1}),
                            ("levels", {"poly_levels": rng.randrange(0, 4)})):
                if kv == "direct" and n > 70000:
                    continue
                force("default", **env)
                out = torch.empty(npoly * (d1 + d2) * EB, dtype=torch.uint8, device=dev)
                eng.poly_mult_dev(npoly, d1, d2, pa, pb, out)
                torch.cuda.synchronize()
                if ref is None:
                    ref = out
                else:
                    assert torch.equal(ref, out), (name, op, npoly, d1, d2, kv)
        done[(name, op)] = done.get((name, op), 0) + 1
        elements += n
        if time.time() - last_note > 30:                     # a sign of life (a silent command is taken to be hung)
            last_note = time.time()
            print("# ... %d calls, %d elements so far" % (sum(done.values()), elements), flush=True)
    force("default")
    print("# %d calls compared, %d elements; no difference" % (sum(done.values()), elements))
    for (name, op), c in sorted(done.items()):
        print("%s,%s,%d" % (name, op, c))


if __name__ == "__main__":
    main()
