"""GPU: BASELINE.json's full sizes, checked through size-independent properties
(the oracles cannot reach 2^20 elements in seconds): homomorphic round trips whose
expected plaintexts are computed with plain integer arithmetic."""
import numpy as np
import pytest
import torch

from conftest import engine_key, load_fixture

pytestmark = pytest.mark.gpu


def _scalars(vals: torch.Tensor, nbytes: int) -> torch.Tensor:
    """int64 tensor -> big-endian fixed-length byte rows (uint8, on the same device)."""
    out = torch.empty((vals.numel(), nbytes), dtype=torch.uint8, device=vals.device)
    v = vals.clone()
    for j in range(nbytes - 1, -1, -1):
        out[:, j] = (v & 0xFF).to(torch.uint8)
        v >>= 8
    return out.contiguous()


def _rand_r(count, gen, dev):
    r = torch.randint(0, 256, (count, 128), dtype=torch.uint8, generator=gen)
    r[:, 0] &= 0x3F
    return r.to(dev)


def test_config2_config3_encrypt_mult_decrypt_2pow20():
    """configs[1]+[2]+[3] at batch 2^20, 1024-bit: Dec(Mult(Enc(a), Enc(b))) == a*b for 2^20 independent pairs
    (20-bit a, b so that a*b < T = 2^40), plus Dec(Enc(a)) == a on level 1."""
    fx = load_fixture("k1024")
    pk, sk = engine_key(fx)
    pk.SetupDecryption(sk)
    eng, dev = pk.engine, torch.device("cuda")
    n = 1 << 20
    g = torch.Generator().manual_seed(20)
    a = torch.randint(0, 1 << 20, (n,), generator=g, dtype=torch.int64)
    b = torch.randint(0, 1 << 20, (n,), generator=g, dtype=torch.int64)
    EB = eng.elem_bytes
    ca = torch.empty(n * EB, dtype=torch.uint8, device=dev)
    cb = torch.empty_like(ca)
    eng.encrypt_dev(_scalars(a.to(dev), 3), 3, _rand_r(n, g, dev), 128, ca, n)
    eng.encrypt_dev(_scalars(b.to(dev), 3), 3, _rand_r(n, g, dev), 128, cb, n)
    prod = torch.empty_like(ca)
    eng.mult_dev(ca, cb, prod, n)
    m = torch.empty(n, dtype=torch.int64, device=dev)
    st = torch.empty(n, dtype=torch.uint8, device=dev)
    eng.decrypt_dev(2, prod, m, st, n)
    torch.cuda.synchronize()
    assert not bool(st.any().item())
    assert bool((m.cpu() == a * b).all().item())
    # level-1 round trip on a 2^16 slice (configs[3] batch)
    k = 1 << 16
    eng.decrypt_dev(1, ca[: k * EB], m[:k], st[:k], k)
    torch.cuda.synchronize()
    assert not bool(st[:k].any().item()) and bool((m[:k].cpu() == a[:k]).all().item())


def test_add_is_homomorphic_at_2pow20():
    """EAdd over 2^20 pairs: Dec(Add(Enc(a), Enc(b))) == a + b, incl. Sub giving negatives on a slice."""
    fx = load_fixture("k1024")
    pk, sk = engine_key(fx)
    pk.SetupDecryption(sk)
    eng, dev = pk.engine, torch.device("cuda")
    n = 1 << 20
    g = torch.Generator().manual_seed(21)
    a = torch.randint(0, 1 << 30, (n,), generator=g, dtype=torch.int64)
    b = torch.randint(0, 1 << 30, (n,), generator=g, dtype=torch.int64)
    EB = eng.elem_bytes
    ca = torch.empty(n * EB, dtype=torch.uint8, device=dev)
    cb = torch.empty_like(ca)
    eng.encrypt_dev(_scalars(a.to(dev), 4), 4, _rand_r(n, g, dev), 128, ca, n)
    eng.encrypt_dev(_scalars(b.to(dev), 4), 4, _rand_r(n, g, dev), 128, cb, n)
    s = torch.empty_like(ca)
    eng.add_dev(1, ca, cb, s, n)
    k = 1 << 17
    m = torch.empty(k, dtype=torch.int64, device=dev)
    st = torch.empty(k, dtype=torch.uint8, device=dev)
    eng.decrypt_dev(1, s[: k * EB], m, st, k)
    torch.cuda.synchronize()
    assert not bool(st.any().item()) and bool((m.cpu() == (a + b)[:k]).all().item())
    from bgn_amd._lib import check
    d = torch.empty(k * EB, dtype=torch.uint8, device=dev)
    check(eng._lib.bgn_sub_batch_dev(eng._h, k, 1, ca.data_ptr(), cb.data_ptr(), None, 0, d.data_ptr(), eng._stream()),
          "bgn_sub_batch_dev")
    eng.decrypt_dev(1, d, m, st, k)
    torch.cuda.synchronize()
    assert not bool(st.any().item()) and bool((m.cpu() == (a - b)[:k]).all().item())


@pytest.mark.parametrize("polys_log2", [12, 14])
def test_config5_multpoly_shape_decrypts_to_convolution(polys_log2):
    """configs[4] on one GPU: 2^12 polynomial pairs of 16x16 base-3 digits in {-1,0,1} (2^20 pairings), and the
    config's stated size, 2^14 pairs = 2^22 coefficient pairs (poly.go:123-156 sixteen thousand times over);
    DecryptPoly of every product equals the integer convolution (poly_test.go:172-189), the AddPoly of the products
    with each other (poly.go:171-207) the sum of the convolutions, and — at the stated size — the first, a middle and
    the last product equal the C oracle's byte for byte (the chunks of whole table rounds and the directly paired
    remainder all take part)."""
    fx = load_fixture("k1024")
    pk, sk = engine_key(fx)
    pk.SetupDecryption(sk)
    eng, dev = pk.engine, torch.device("cuda")
    npoly, d = 1 << polys_log2, 16
    g = torch.Generator().manual_seed(22)
    ca = torch.randint(-1, 2, (npoly, d), generator=g, dtype=torch.int64)
    cb = torch.randint(-1, 2, (npoly, d), generator=g, dtype=torch.int64)
    EB = eng.elem_bytes
    n = npoly * d

    def enc(c):
        # EncryptPoly (poly.go:11-29): a negative digit is Sub(zero, Enc(|c|)) == Neg(Enc(|c|))
        mag = c.abs().reshape(-1)
        ct = torch.empty(n * EB, dtype=torch.uint8, device=dev)
        eng.encrypt_dev(_scalars(mag.to(dev), 1), 1, _rand_r(n, g, dev), 128, ct, n)
        neg = torch.empty_like(ct)
        from bgn_amd._lib import check
        check(eng._lib.bgn_neg_batch_dev(eng._h, n, 1, ct.data_ptr(), neg.data_ptr(), eng._stream()), "neg")
        sel = (c.reshape(-1) < 0).to(dev)
        out = ct.view(n, EB).clone()
        out[sel] = neg.view(n, EB)[sel]
        return out.reshape(-1).contiguous()

    ea, eb = enc(ca), enc(cb)
    out = torch.empty(npoly * 2 * d * EB, dtype=torch.uint8, device=dev)
    eng.poly_mult_dev(npoly, d, d, ea, eb, out)
    m = torch.empty(npoly * 2 * d, dtype=torch.int64, device=dev)
    st = torch.empty(npoly * 2 * d, dtype=torch.uint8, device=dev)
    eng.decrypt_dev(2, out, m, st, npoly * 2 * d)
    torch.cuda.synchronize()
    assert not bool(st.any().item())
    conv = torch.zeros((npoly, 2 * d), dtype=torch.int64)
    for i in range(d):
        for k in range(d):
            conv[:, i + k] += ca[:, i] * cb[:, k]
    assert bool((m.cpu().view(npoly, 2 * d) == conv).all().item())
    # AddPoly of product q with product q + npoly/2: coefficient-wise products in GT
    half = npoly // 2
    nh = half * 2 * d
    summ = torch.empty(nh * EB, dtype=torch.uint8, device=dev)
    eng.add_dev(2, out[: nh * EB], out[nh * EB: 2 * nh * EB], summ, nh)
    eng.decrypt_dev(2, summ, m[:nh], st[:nh], nh)
    torch.cuda.synchronize()
    assert not bool(st[:nh].any().item())
    assert bool((m[:nh].cpu().view(half, 2 * d) == conv[:half] + conv[half:]).all().item())
    if polys_log2 >= 14:
        import threading
        import oracle_c
        W = 2 * d * EB
        picks = [0, npoly // 2 + 77, npoly - 1]
        got = [None] * len(picks)

        def work(i):
            q = picks[i]
            o = oracle_c.Oracle.from_fixture(fx)
            got[i] = o.poly_mult(1, d, d, ea[q * d * EB:(q + 1) * d * EB].cpu().numpy().tobytes(),
                                 eb[q * d * EB:(q + 1) * d * EB].cpu().numpy().tobytes())

        th = [threading.Thread(target=work, args=(i,)) for i in range(len(picks))]
        [t.start() for t in th]
        [t.join() for t in th]
        for q, ref in zip(picks, got):
            assert out[q * W:(q + 1) * W].cpu().numpy().tobytes() == ref, "MultPoly product %d differs from the C oracle" % q


def test_mult_longer_than_one_piece():
    """Mult processes its arrays in pieces of 2^22 pairs (bounded workspace): a batch that spans two pieces, with
    inputs cycling through a small pool, reproduces the pool's pairings at every position."""
    import oracle_c
    fx = load_fixture("toy64")
    o = oracle_c.Oracle.from_fixture(fx)
    pk, _ = engine_key(fx)
    eng = pk.engine
    EB = eng.elem_bytes
    cts = [bytes.fromhex(e["ct"]) for e in fx["encrypt"]]
    pool = 7
    pa = b"".join(cts[i % len(cts)] for i in range(pool))
    pb = b"".join(cts[(3 * i + 1) % len(cts)] for i in range(pool))
    want = torch.frombuffer(bytearray(o.mult(pa, pb)), dtype=torch.uint8).view(pool, EB).cuda()
    count = (1 << 22) + 77
    idx = torch.arange(count, device="cuda") % pool
    a = torch.frombuffer(bytearray(pa), dtype=torch.uint8).view(pool, EB).cuda()[idx].reshape(-1).contiguous()
    b = torch.frombuffer(bytearray(pb), dtype=torch.uint8).view(pool, EB).cuda()[idx].reshape(-1).contiguous()
    out = torch.empty(count * EB, dtype=torch.uint8, device="cuda")
    eng.mult_dev(a, b, out, count)
    torch.cuda.synchronize()
    assert bool((out.view(count, EB) == want[idx]).all().item())


@pytest.mark.parametrize("name,count", [("k1024", 1 << 16), ("k1024b", 1 << 14), ("k512", 1 << 16), ("k2048", 1 << 11)])
def test_pairing_kernels_agree_on_large_random_batches(name, count, engopts):
    """Three formulations of `res.Pair` (bgn.go:300) — one pairing per lane (unsigned lazy limbs, Jacobian,
    windowed NAF), sixteen lanes per pairing and one workgroup per pairing (signed lazy limbs, generated step
    programs, plain NAF) — on the same seeded random ciphertext pairs with full-length randomness: every byte of
    every result equal.  A rare carry pattern mishandled by one of them shows here (65536 pairings x ~21 000 field
    products each); the C oracle checks a sample of the same batch."""
    import oracle_c
    from bgn_amd.synthetic import config2_ciphertexts, permuted_copy
    fx = load_fixture(name)
    pk, _ = engine_key(fx)
    eng, dev = pk.engine, torch.device("cuda")
    EB = eng.elem_bytes
    _, _, ca = config2_ciphertexts(pk, count, 4242, dev)
    cb = permuted_copy(ca, EB, 4243)
    kernels = ("quad", "lane") if name == "k2048" else ("quad", "coop", "lane")   # no cooperative kernel at 72 limbs
    outs = {}
    for kernel in kernels:
        engopts.set("quad_min", "0")
        engopts.set("quad_max", "100000000" if kernel == "quad" else "0")
        engopts.set("coop_max", "100000000" if kernel == "coop" else "0")
        n = count if kernel != "lane" or name != "k2048" else 256                 # the 72-limb lane kernel is the functional one
        out = torch.empty(n * EB, dtype=torch.uint8, device=dev)
        eng.mult_dev(ca[: n * EB], cb[: n * EB], out, n)
        torch.cuda.synchronize()
        assert (kernel in eng.last_kernel_name()) == (kernel != "lane"), eng.last_kernel_name()
        outs[kernel] = out
    for kernel in kernels[1:]:
        n = outs[kernel].numel()
        assert torch.equal(outs["quad"][:n], outs[kernel]), (name, kernel)
    k = 24 if name != "k2048" else 6
    o = oracle_c.Oracle.from_fixture(fx)
    sel = slice((count - k) * EB, count * EB)
    assert bytes(outs["quad"][sel].cpu().numpy()) == o.mult(bytes(ca[sel].cpu().numpy()), bytes(cb[sel].cpu().numpy()))


@pytest.mark.parametrize("name,count", [("k1024", 1 << 15)])
def test_table_walks_and_decrypt_agree_across_kernels_on_large_batches(name, count, engopts):
    """makeL2 (the walk over P's line table, bgn.go:316-321) and Decrypt of both levels (lift over the secret order's
    table, power by q1, BSGS; bgn.go:205-250) on 32768 random ciphertexts, every 16th negated: the lane-group kernels,
    the cooperative kernels and the lane kernels return the same bytes / plaintexts / statuses, and the plaintexts
    are the ones encrypted."""
    from bgn_amd.synthetic import config2_ciphertexts, decrypt_mix
    fx = load_fixture(name)
    pk, sk = engine_key(fx)
    pk.SetupDecryption(sk)
    eng, dev = pk.engine, torch.device("cuda")
    EB = eng.elem_bytes
    xs, _, ca = config2_ciphertexts(pk, count, 777, dev)
    assert fx["msg_space"] >= (1 << 40)                  # the synthetic plaintexts have 40 bits
    mixed, want, want_st = decrypt_mix(pk, fx, ca, xs, dev)
    big = "100000000"
    res = {}
    for kernel in ("quad", "coop", "lane"):
        engopts.set("quad_min", "0")
        for v in ("quad_max_l2", "quad_max_dec", "quad_max_pow"):
            engopts.set(v, big if kernel == "quad" else "0")
        for v in ("coop_max_l2", "coop_max_dec"):
            engopts.set(v, big if kernel == "coop" else "0")
        l2 = torch.empty(count * EB, dtype=torch.uint8, device=dev)
        eng.make_l2_dev(mixed, l2, count)
        torch.cuda.synchronize()
        assert (kernel in eng.last_kernel_name()) == (kernel != "lane"), eng.last_kernel_name()
        m1 = torch.empty(count, dtype=torch.int64, device=dev)
        s1 = torch.empty(count, dtype=torch.uint8, device=dev)
        eng.decrypt_dev(1, mixed, m1, s1, count)
        torch.cuda.synchronize()
        assert (kernel in eng.last_aux_kernel_name()) == (kernel != "lane"), eng.last_aux_kernel_name()
        m2 = torch.empty_like(m1)
        s2 = torch.empty_like(s1)
        eng.decrypt_dev(2, l2, m2, s2, count)
        torch.cuda.synchronize()
        res[kernel] = (l2, m1.cpu(), s1.cpu(), m2.cpu(), s2.cpu())
    for kernel in ("coop", "lane"):
        assert torch.equal(res["quad"][0], res[kernel][0]), (name, kernel, "makeL2 bytes")
        for i in range(1, 5):
            assert torch.equal(res["quad"][i], res[kernel][i]), (name, kernel, i)
    _, m1, s1, m2, s2 = res["quad"]
    assert torch.equal(s1, want_st) and torch.equal(s2, want_st)
    ok = want_st == 0
    assert torch.equal(m1[ok], want[ok]) and torch.equal(m2[ok], want[ok])


def test_batches_that_do_not_fill_lane_kernel_rounds_are_cut(engopts):
    """65536 + r elements: the engine runs whole rounds of the lane kernel and hands the remainder to the kernel that
    is fastest at ITS size (engine.cpp lane_rounds_head / decrypt_rounds_head).  Same bytes / plaintexts as the single
    launch (option split_rounds = 0), for Mult, makeL2 and level-1 Decrypt.  The kernel the measurement hooks report is
    the HEAD piece's (the lane kernel either way: it does nearly all the work), so the cut shows in the time: one round
    of the lane kernel + a short launch instead of two rounds."""
    import time
    from bgn_amd.synthetic import config2_ciphertexts, permuted_copy
    dev = torch.device("cuda")
    for name, count in (("k512", 65536 + 1500), ("k1024", 65536 + 700)):
        fx = load_fixture(name)
        pk, sk = engine_key(fx)
        eng = pk.engine
        EB = eng.elem_bytes
        xs, _, ca = config2_ciphertexts(pk, count, 99, dev)
        cb = permuted_copy(ca, EB, 98)
        res = {}
        for split in ("1", "0"):
            engopts.set("split_rounds", split)
            prod = torch.empty(count * EB, dtype=torch.uint8, device=dev)
            eng.mult_dev(ca, cb, prod, count)                                    # (also grows the workspace)
            torch.cuda.synchronize()
            t_mult = 1e9
            for _ in range(3):                                                   # best of three: a timing, on a shared box
                t0 = time.perf_counter()
                eng.mult_dev(ca, cb, prod, count)
                torch.cuda.synchronize()
                t_mult = min(t_mult, time.perf_counter() - t0)
            k_mult = eng.last_kernel_name()
            l2 = torch.empty_like(prod)
            eng.make_l2_dev(ca, l2, count)
            torch.cuda.synchronize()
            k_l2 = eng.last_kernel_name()
            res[split] = (prod, l2, k_mult, k_l2, t_mult)
        assert torch.equal(res["1"][0], res["0"][0]) and torch.equal(res["1"][1], res["0"][1]), name
        if name == "k512":
            # blinded Mult (Deterministic == false, bgn.go:302-311): the randomness array is cut with the batch
            r_len = (int(fx["n"], 16).bit_length() + 7) // 8
            g = torch.Generator().manual_seed(5)
            r = torch.randint(0, 256, (count, r_len), dtype=torch.uint8, generator=g)
            r[:, 0] &= 0x3F
            r = r.to(dev)
            blinded = {}
            for split in ("1", "0"):
                engopts.set("split_rounds", split)
                o = torch.empty(count * EB, dtype=torch.uint8, device=dev)
                eng.mult_dev(ca, cb, o, count, r=r, r_len=r_len)
                torch.cuda.synchronize()
                blinded[split] = o
            assert torch.equal(blinded["1"], blinded["0"]) and not torch.equal(blinded["1"], res["1"][0])
        for split in ("0", "1"):
            assert "k_pairing<" in res[split][2] and ", 0>" in res[split][2], res[split][2]
            assert "k_pairing<" in res[split][3] and ", 1>" in res[split][3], res[split][3]
        assert res["1"][4] < 0.8 * res["0"][4], (res["1"][4], res["0"][4])         # 1 round + remainder against 2 rounds
        if name == "k1024":
            pk.SetupDecryption(sk)
            want = torch.zeros(count, dtype=torch.int64)
            xb = xs.cpu().to(torch.int64)
            for j in range(xb.shape[1]):
                want = want * 256 + xb[:, j]
            for split in ("1", "0"):
                engopts.set("split_rounds", split)
                m = torch.empty(count, dtype=torch.int64, device=dev)
                st = torch.empty(count, dtype=torch.uint8, device=dev)
                eng.decrypt_dev(1, ca, m, st, count)
                torch.cuda.synchronize()
                assert not bool(st.any().item()) and torch.equal(m.cpu(), want), split
                aux = eng.last_aux_kernel_name()
                assert "k_pairing<" in aux and ", 1>" in aux, aux                # the head piece's lift
                eng.decrypt_dev(2, res["1"][1], m, st, count)                  # level 2: the power by q1 + BSGS
                torch.cuda.synchronize()
                assert not bool(st.any().item()) and torch.equal(m.cpu(), want), ("level 2", split)


def test_add_beyond_64_elements_per_lane():
    """EAdd over 2^22 + 1000 pairs (65 additions on the busiest lanes of the batched-inversion kernel: the run is the
    ceiling of count / 65536 whatever the count): the same bytes as the four 2^20 slices and the remainder added
    on their own."""
    from bgn_amd.synthetic import config2_ciphertexts, permuted_copy
    fx = load_fixture("k1024")
    pk, _ = engine_key(fx)
    eng, dev = pk.engine, torch.device("cuda")
    EB = eng.elem_bytes
    base = 1 << 20
    _, _, c0 = config2_ciphertexts(pk, base, 31, dev)
    n = 4 * base + 1000
    ca = torch.cat([c0, c0, c0, c0, c0[: 1000 * EB]])
    cb = torch.cat([permuted_copy(c0, EB, 32 + k) for k in range(4)] + [c0[2000 * EB: 3000 * EB]])
    whole = torch.empty(n * EB, dtype=torch.uint8, device=dev)
    eng.add_dev(1, ca, cb, whole, n)
    torch.cuda.synchronize()
    piece = torch.empty(base * EB, dtype=torch.uint8, device=dev)
    for k in range(5):
        m = base if k < 4 else 1000
        lo = k * base * EB
        eng.add_dev(1, ca[lo: lo + m * EB], cb[lo: lo + m * EB], piece[: m * EB], m)
        torch.cuda.synchronize()
        assert torch.equal(whole[lo: lo + m * EB], piece[: m * EB]), k


def test_short_device_arrays_are_refused_by_the_host_mirror():
    """MultPoly writes npoly * (d1 + d2) coefficients (include/bgn_amd.h): an output array of npoly * (d1 + d2 - 1)
    is refused before the call; so are short operands of Mult / Decrypt."""
    fx = load_fixture("k256")
    pk, _ = engine_key(fx)
    eng, dev = pk.engine, torch.device("cuda")
    EB = eng.elem_bytes
    a = torch.zeros(4 * 3 * EB, dtype=torch.uint8, device=dev)
    with pytest.raises(ValueError, match="out"):
        eng.poly_mult_dev(4, 3, 3, a, a, torch.empty(4 * 5 * EB, dtype=torch.uint8, device=dev))
    with pytest.raises(ValueError, match="b:"):
        eng.mult_dev(a, a[: 5 * EB], torch.empty_like(a), 12)
    with pytest.raises(ValueError, match="status"):
        eng.decrypt_dev(1, a, torch.empty(12, dtype=torch.int64, device=dev), torch.empty(11, dtype=torch.uint8, device=dev), 12)


def test_level2_add_sub_at_2pow20():
    """EAdd / ESub on level 2 at 2^20 (the fused wire-to-wire kernel): Sub(Add(x, y), y) gives x back byte for byte
    (every product is an element of GT, whose inverse is the conjugate), Add(x, x') commutes, and a slice decrypts to
    the sum and the difference of the plaintext products."""
    fx = load_fixture("k1024")
    pk, sk = engine_key(fx)
    pk.SetupDecryption(sk)
    eng, dev = pk.engine, torch.device("cuda")
    n = 1 << 20
    g = torch.Generator().manual_seed(23)
    a = torch.randint(0, 1 << 19, (n,), generator=g, dtype=torch.int64)
    b = torch.randint(0, 1 << 19, (n,), generator=g, dtype=torch.int64)
    EB = eng.elem_bytes
    ca = torch.empty(n * EB, dtype=torch.uint8, device=dev)
    cb = torch.empty_like(ca)
    eng.encrypt_dev(_scalars(a.to(dev), 4), 4, _rand_r(n, g, dev), 128, ca, n)
    eng.encrypt_dev(_scalars(b.to(dev), 4), 4, _rand_r(n, g, dev), 128, cb, n)
    x = torch.empty_like(ca)
    eng.mult_dev(ca, cb, x, n)                                   # x[i] = Enc2(a[i] * b[i])
    perm = torch.randperm(n, generator=g).to(dev)
    y = x.view(n, EB)[perm].contiguous().view(-1)                # y[i] = x[perm[i]]
    s = torch.empty_like(x)
    eng.add_dev(2, x, y, s, n)
    assert eng.last_kernel_name() == "k_gt_mul_wire"
    s2 = torch.empty_like(x)
    eng.add_dev(2, y, x, s2, n)
    assert torch.equal(s, s2)
    from bgn_amd._lib import check
    back = torch.empty_like(x)
    check(eng._lib.bgn_sub_batch_dev(eng._h, n, 2, s.data_ptr(), y.data_ptr(), None, 0, back.data_ptr(), eng._stream()),
          "bgn_sub_batch_dev")
    torch.cuda.synchronize()
    assert torch.equal(back, x)
    k = 1 << 16
    m = torch.empty(k, dtype=torch.int64, device=dev)
    st = torch.empty(k, dtype=torch.uint8, device=dev)
    eng.decrypt_dev(2, s[: k * EB], m, st, k)
    torch.cuda.synchronize()
    prod = a * b
    want = prod + prod[perm.cpu()]
    assert not bool(st.any().item()) and bool((m.cpu() == want[:k]).all().item())
    d = torch.empty(k * EB, dtype=torch.uint8, device=dev)
    check(eng._lib.bgn_sub_batch_dev(eng._h, k, 2, x.data_ptr(), y.data_ptr(), None, 0, d.data_ptr(), eng._stream()),
          "bgn_sub_batch_dev")
    eng.decrypt_dev(2, d, m, st, k)
    torch.cuda.synchronize()
    assert not bool(st.any().item()) and bool((m.cpu() == (prod - prod[perm.cpu()])[:k]).all().item())
