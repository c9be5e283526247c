#!/usr/bin/env python3
"""What the default sizes of the per-key tables buy (include/bgn_amd.h "Device memory"), 1024-bit key, T = 2^40:

  * Decrypt against the baby-step table: decrypts/s at 2^16 and 2^20 ciphertexts (level 1, the bench's mixed batch) and
    the set-up time for tables of 2^24 .. 2^31 entries (option bsgs_max_log2; 16 B per entry since round 5) — the walk is
    T / (2 S) products per ciphertext beside a lift of ~3.8 k products;
  * Encrypt against the window width of Q's table (option fixed_window_bits_q: 2^16 .. 2^22 entries per window, signed
    windows of one scalar bit more since round 5; a fresh context each, the tables are built on first use).

    python tools/decrypt_vs_table.py > profiles/r04_decrypt_vs_table.csv
    python tools/decrypt_vs_table.py encrypt > profiles/r04_encrypt_vs_window.csv
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from conftest import load_fixture  # noqa: E402
import bgn_amd  # noqa: E402
import bgn_amd.synthetic as syn  # noqa: E402


def best(fn, reps=3):
    t = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        t.append(time.perf_counter() - t0)
    return min(t)


def new_key(fx):
    return bgn_amd.PublicKey(int(fx["p"], 16), int(fx["n"], 16), fx["l"], bytes.fromhex(fx["P"]), bytes.fromhex(fx["Q"]),
                             fx["msg_space"], True, fx["poly_base"])


def decrypt_sweep(fx, dev):
    pk = new_key(fx)
    eng = pk.engine
    EB = eng.elem_bytes
    n = 1 << 20
    xs, rs, cts = syn.config2_ciphertexts(pk, n, seed=1000, device=dev)
    sk = bgn_amd.SecretKey(int(fx["q1"], 16))
    print("key,T_log2,table_log2,table_GB,setup_s,batch,ms,decrypts_per_s,giant_steps,exact")
    m = torch.empty(n, dtype=torch.int64, device=dev)
    st = torch.empty(n, dtype=torch.uint8, device=dev)
    mixed = want = want_st = None
    for lg in (31, 30, 29, 28, 27, 26, 25, 24):
        eng.set_option("bsgs_max_log2", lg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pk.SetupDecryption(sk)
        torch.cuda.synchronize()
        setup = time.perf_counter() - t0
        if mixed is None:
            mixed, want, want_st = syn.decrypt_mix(pk, fx, cts, xs, dev)
        S = int(eng._lib.bgn_ctx_bsgs_baby_steps(eng._h))
        for k in (16, 20):
            cnt = 1 << k
            eng.decrypt_dev(1, mixed[: cnt * EB], m, st, cnt)
            dt = best(lambda: eng.decrypt_dev(1, mixed[: cnt * EB], m, st, cnt), 2)
            ok = bool((m[:cnt].cpu() == want[:cnt]).all().item()) and bool((st[:cnt].cpu() == want_st[:cnt]).all().item())
            T = int(fx["msg_space"])
            print("%s,%d,%d,%.2f,%.2f,%d,%.2f,%.0f,%d,%s" % (fx["name"], T.bit_length() - 1, S.bit_length() - 1, 16.0 * S / 1e9, setup,
                                                          cnt, dt * 1e3, cnt / dt, T // (2 * S) + 1, ok), flush=True)


def encrypt_sweep(fx, dev):
    print("key,q_table_index_bits,signed_windows,q_windows,q_table_GB,first_call_s,batch,ms,encrypts_per_s")
    n = 1 << 20
    ref = None
    for wb in (22, 20, 18, 16):
        pk = new_key(fx)
        eng = pk.engine
        eng.set_option("fixed_window_bits_q", wb)

        torch.cuda.synchronize()
        t0 = time.perf_counter()
        xs, rs, cts = syn.config2_ciphertexts(pk, n, seed=1000, device=dev)
        first = time.perf_counter() - t0

        out = torch.empty_like(cts)
        dt = best(lambda: eng.encrypt_dev(xs, xs.shape[1], rs, rs.shape[1], out, n))
        assert bool((out == cts).all().item())
        if ref is None:
            ref = cts.clone()
        assert bool((ref == cts).all().item()), "window width changed the ciphertexts"
        nbits = int(fx["n"], 16).bit_length()
        signed = int(eng.get_option("fixed_signed_q")) != 0
        W = (8 * ((nbits + 7) // 8)) // (wb + 1) + 1 if signed else (nbits + wb - 1) // wb + 1     # engine.cpp fixed_table_windows
        tab = (W << wb) * 2 * syn.limbs_for(int(fx["p"], 16)) * 4
        print("%s,%d,%d,%d,%.2f,%.2f,%d,%.2f,%.0f" % (fx["name"], wb, 1 if signed else 0, W, tab / 1e9, first, n, dt * 1e3, n / dt), flush=True)
        del xs, rs, cts, out
        pk.engine.close()


if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    fx = load_fixture("k1024")
    if len(sys.argv) > 1 and sys.argv[1] == "encrypt":
        encrypt_sweep(fx, dev)
    else:
        decrypt_sweep(fx, dev)
