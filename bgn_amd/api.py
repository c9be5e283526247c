"""Host-side mirror of the reference's Go API for the batched hot path.

Names and argument meaning follow sachaservan/bgn (bgn.go, ciphertext.go,
poly.go, gsbs.go) so tests read like the reference's own; every operation is
executed by the HIP engine through the C ABI (include/bgn_amd.h).  Single
element methods are the count = 1 case of the batch entry points.

Error behaviour mirrors the reference: Decrypt raises DecryptError with the
reference's message (gsbs.go:105); misuse that panics in Go raises here.
"""
from __future__ import annotations

import ctypes as C
import math
import secrets
from dataclasses import dataclass
from typing import List, Optional, Sequence, Union

import numpy as np

from . import _lib, gob
from ._lib import check

BytesLike = Union[bytes, bytearray, memoryview, np.ndarray]


class DecryptError(ValueError):
    """errors.New("cannot find discrete log; out of bounds") — gsbs.go:105"""

    def __init__(self):
        super().__init__("cannot find discrete log; out of bounds")


def _int_bytes(v: int, length: int) -> bytes:
    return int(v).to_bytes(length, "big")


def _scalars(vals: Sequence[int], length: Optional[int] = None) -> np.ndarray:
    """Pack non-negative integers as fixed-length big-endian rows."""
    vals = [int(v) for v in vals]
    if any(v < 0 for v in vals):
        raise ValueError("scalars must be non-negative (reduce modulo n first)")
    if length is None:
        length = max(1, max(((v.bit_length() + 7) // 8 for v in vals), default=1))
    buf = b"".join(v.to_bytes(length, "big") for v in vals)
    return np.frombuffer(buf, dtype=np.uint8).reshape(len(vals), length)


def _as_u8(a: BytesLike, row: int) -> np.ndarray:
    arr = np.frombuffer(a, dtype=np.uint8) if not isinstance(a, np.ndarray) else a
    arr = np.ascontiguousarray(arr, dtype=np.uint8).reshape(-1)
    if arr.size % row:
        raise ValueError(f"buffer of {arr.size} bytes is not a multiple of the element size {row}")
    return arr.reshape(-1, row)


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Engine:
    """One bgn_ctx: a public key's pairing context on one GPU."""

    def __init__(self, p: int, n: int, l: int, P_wire: bytes, Q_wire: bytes, deterministic: bool = True,
                 device: int = 0):
        self._lib = _lib.load()
        self.p, self.n, self.l = int(p), int(n), int(l)
        pb = _int_bytes(self.p, (self.p.bit_length() + 7) // 8)
        nb = _int_bytes(self.n, (self.n.bit_length() + 7) // 8)
        self._h = C.c_void_p()
        self._keep = (pb, nb, bytes(P_wire), bytes(Q_wire))
        check(self._lib.bgn_ctx_create(C.byref(self._h), pb, len(pb), nb, len(nb), self.l, self._keep[2],
                                       self._keep[3], 1 if deterministic else 0, device), "bgn_ctx_create")
        self.L = int(self._lib.bgn_fp_bytes(self._h))
        self.elem_bytes = 2 * self.L
        self.device = device

    def close(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            self._lib.bgn_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- secret key / tables -------------------------------------------------
    def set_secret(self, q1: int) -> None:
        b = _int_bytes(q1, (int(q1).bit_length() + 7) // 8)
        check(self._lib.bgn_ctx_set_secret(self._h, b, len(b)), "bgn_ctx_set_secret")

    def setup_decryption(self, msg_space: int) -> None:
        check(self._lib.bgn_ctx_setup_decryption(self._h, int(msg_space)), "bgn_ctx_setup_decryption")

    # ---- host-buffer batch ops -------------------------------------------------
    def _out(self, count: int) -> np.ndarray:
        return np.zeros((count, self.elem_bytes), dtype=np.uint8)

    def encrypt(self, x: Sequence[int], r: Optional[Sequence[int]] = None) -> np.ndarray:
        xs = _scalars(x)
        rs = _scalars(r) if r is not None else None
        out = self._out(len(xs))
        check(self._lib.bgn_encrypt_batch(self._h, len(xs), _ptr(xs), xs.shape[1], _ptr(rs),
                                          rs.shape[1] if rs is not None else 0, _ptr(out)), "bgn_encrypt_batch")
        return out

    def _binop(self, fn, name, level, a, b, r):
        A, B = _as_u8(a, self.elem_bytes), _as_u8(b, self.elem_bytes)
        if len(A) != len(B):
            raise ValueError("operand counts differ")
        rs = _scalars(r) if r is not None else None
        out = self._out(len(A))
        check(fn(self._h, len(A), level, _ptr(A), _ptr(B), _ptr(rs), rs.shape[1] if rs is not None else 0, _ptr(out)),
              name)
        return out

    def add(self, level: int, a: BytesLike, b: BytesLike, r=None) -> np.ndarray:
        return self._binop(self._lib.bgn_add_batch, "bgn_add_batch", level, a, b, r)

    def sub(self, level: int, a: BytesLike, b: BytesLike, r=None) -> np.ndarray:
        return self._binop(self._lib.bgn_sub_batch, "bgn_sub_batch", level, a, b, r)

    def neg(self, level: int, a: BytesLike) -> np.ndarray:
        A = _as_u8(a, self.elem_bytes)
        out = self._out(len(A))
        check(self._lib.bgn_neg_batch(self._h, len(A), level, _ptr(A), _ptr(out)), "bgn_neg_batch")
        return out

    def mult(self, a: BytesLike, b: BytesLike, r=None) -> np.ndarray:
        A, B = _as_u8(a, self.elem_bytes), _as_u8(b, self.elem_bytes)
        if len(A) != len(B):
            raise ValueError("operand counts differ")
        rs = _scalars(r) if r is not None else None
        out = self._out(len(A))
        check(self._lib.bgn_mult_batch(self._h, len(A), _ptr(A), _ptr(B), _ptr(rs),
                                       rs.shape[1] if rs is not None else 0, _ptr(out)), "bgn_mult_batch")
        return out

    def make_l2(self, a: BytesLike) -> np.ndarray:
        A = _as_u8(a, self.elem_bytes)
        out = self._out(len(A))
        check(self._lib.bgn_make_l2_batch(self._h, len(A), _ptr(A), _ptr(out)), "bgn_make_l2_batch")
        return out

    def multconst(self, level: int, a: BytesLike, k: Sequence[int], r=None) -> np.ndarray:
        A = _as_u8(a, self.elem_bytes)
        ks = _scalars(k)
        if len(ks) != len(A):
            raise ValueError("operand counts differ")
        rs = _scalars(r) if r is not None else None
        out = self._out(len(A))
        check(self._lib.bgn_multconst_batch(self._h, len(A), level, _ptr(A), _ptr(ks), ks.shape[1], _ptr(rs),
                                            rs.shape[1] if rs is not None else 0, _ptr(out)), "bgn_multconst_batch")
        return out

    def decrypt(self, level: int, ct: BytesLike):
        A = _as_u8(ct, self.elem_bytes)
        m = np.zeros(len(A), dtype=np.int64)
        st = np.zeros(len(A), dtype=np.uint8)
        check(self._lib.bgn_decrypt_batch(self._h, len(A), level, _ptr(A), _ptr(m), _ptr(st)), "bgn_decrypt_batch")
        return m, st

    def poly_mult(self, npoly: int, d1: int, d2: int, a: BytesLike, b: BytesLike) -> np.ndarray:
        A, B = _as_u8(a, self.elem_bytes), _as_u8(b, self.elem_bytes)
        if len(A) != npoly * d1 or len(B) != npoly * d2:
            raise ValueError("coefficient array sizes do not match npoly*d1 / npoly*d2")
        out = self._out(npoly * (d1 + d2))
        check(self._lib.bgn_poly_mult_batch(self._h, npoly, d1, d2, _ptr(A), _ptr(B), _ptr(out)),
              "bgn_poly_mult_batch")
        return out

    def validate(self, level: int, a: BytesLike) -> np.ndarray:
        """1 per element that is a valid encoding (range; on the curve / norm 1), 0 otherwise."""
        A = _as_u8(a, self.elem_bytes)
        ok = np.zeros(len(A), dtype=np.uint8)
        check(self._lib.bgn_validate_batch(self._h, len(A), level, _ptr(A), _ptr(ok)), "bgn_validate_batch")
        return ok

    def poly_multconst(self, npoly: int, d: int, level: int, ct: BytesLike, coeffs, shared: bool = True) -> np.ndarray:
        """MultConstPoly of `npoly` ciphertext polynomials (d coefficients each) by encoded plaintext
        constants: `coeffs` is one list of dp digits (shared) or npoly lists.  Returns npoly*(d+dp) rows."""
        rows = [list(coeffs)] if shared else [list(r) for r in coeffs]
        dp = len(rows[0])
        if any(len(r) != dp for r in rows) or (not shared and len(rows) != npoly):
            raise ValueError("plaintext polynomials must have one common degree")
        k = _scalars([v for r in rows for v in r])
        a = _as_u8(ct, self.elem_bytes)
        if len(a) != npoly * d:
            raise ValueError("coefficient count mismatch")
        out = self._out(npoly * (d + dp))
        check(self._lib.bgn_poly_multconst_batch(self._h, npoly, d, dp, level, _ptr(a), _ptr(k), k.shape[1],
                                                 0 if shared else 1, _ptr(out)), "bgn_poly_multconst_batch")
        return out

    def poly_eval(self, npoly: int, d: int, level: int, ct: BytesLike, base: int) -> np.ndarray:
        a = _as_u8(ct, self.elem_bytes)
        if len(a) != npoly * d:
            raise ValueError("coefficient count mismatch")
        out = self._out(npoly)
        check(self._lib.bgn_poly_eval_batch(self._h, npoly, d, level, _ptr(a), base, _ptr(out)), "bgn_poly_eval_batch")
        return out

    # ---- device-buffer batch ops (torch uint8 CUDA tensors; asynchronous) ---------
    @staticmethod
    def _stream():
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _need(what: str, t, nbytes: int) -> None:
        """The C ABI takes bare pointers: a device array shorter than the call reads or writes would fault the GPU,
        so the host mirror refuses it here."""
        if t is None:
            return
        have = t.numel() * t.element_size()
        if have < nbytes:
            raise ValueError(f"{what}: {have} bytes on the device, the call needs {nbytes}")
        if not t.is_cuda or not t.is_contiguous():
            raise ValueError(f"{what}: a contiguous CUDA tensor is required")

    def mult_dev(self, a, b, out, count: Optional[int] = None, r=None, r_len: int = 0) -> None:
        """Device-resident Mult; r (count x r_len big-endian bytes on the device) blinds with e(Q,Q)^r."""
        count = a.numel() // self.elem_bytes if count is None else count
        for what, t in (("a", a), ("b", b), ("out", out)):
            self._need(what, t, count * self.elem_bytes)
        self._need("r", r, count * r_len)
        check(self._lib.bgn_mult_batch_dev(self._h, count, a.data_ptr(), b.data_ptr(),
                                           r.data_ptr() if r is not None else None, r_len, out.data_ptr(),
                                           self._stream()), "bgn_mult_batch_dev")

    def make_l2_dev(self, a, out, count: Optional[int] = None) -> None:
        count = a.numel() // self.elem_bytes if count is None else count
        self._need("a", a, count * self.elem_bytes)
        self._need("out", out, count * self.elem_bytes)
        check(self._lib.bgn_make_l2_batch_dev(self._h, count, a.data_ptr(), out.data_ptr(), self._stream()),
              "bgn_make_l2_batch_dev")

    def encrypt_dev(self, x, x_len: int, r, r_len: int, out, count: int) -> None:
        self._need("x", x, count * x_len)
        self._need("r", r, count * r_len)
        self._need("out", out, count * self.elem_bytes)
        check(self._lib.bgn_encrypt_batch_dev(self._h, count, x.data_ptr(), x_len, r.data_ptr() if r is not None else None,
                                              r_len, out.data_ptr(), self._stream()), "bgn_encrypt_batch_dev")

    def add_dev(self, level: int, a, b, out, count: Optional[int] = None, r=None, r_len: int = 0) -> None:
        """Device-resident Add; r blinds with Q^r (level 1) resp. e(Q,Q)^r (level 2)."""
        count = a.numel() // self.elem_bytes if count is None else count
        for what, t in (("a", a), ("b", b), ("out", out)):
            self._need(what, t, count * self.elem_bytes)
        self._need("r", r, count * r_len)
        check(self._lib.bgn_add_batch_dev(self._h, count, level, a.data_ptr(), b.data_ptr(),
                                          r.data_ptr() if r is not None else None, r_len, out.data_ptr(),
                                          self._stream()), "bgn_add_batch_dev")

    def sub_dev(self, level: int, a, b, out, count: Optional[int] = None, r=None, r_len: int = 0) -> None:
        """Device-resident Sub (bgn.go:375-433); `out` may be `a` or `b` itself, as for add_dev."""
        count = a.numel() // self.elem_bytes if count is None else count
        for what, t in (("a", a), ("b", b), ("out", out)):
            self._need(what, t, count * self.elem_bytes)
        self._need("r", r, count * r_len)
        check(self._lib.bgn_sub_batch_dev(self._h, count, level, a.data_ptr(), b.data_ptr(),
                                          r.data_ptr() if r is not None else None, r_len, out.data_ptr(),
                                          self._stream()), "bgn_sub_batch_dev")

    def neg_dev(self, level: int, a, out, count: Optional[int] = None) -> None:
        """Device-resident Neg (bgn.go:436-438); `out` may be `a` itself."""
        count = a.numel() // self.elem_bytes if count is None else count
        self._need("a", a, count * self.elem_bytes)
        self._need("out", out, count * self.elem_bytes)
        check(self._lib.bgn_neg_batch_dev(self._h, count, level, a.data_ptr(), out.data_ptr(), self._stream()),
              "bgn_neg_batch_dev")

    def multconst_dev(self, level: int, a, k, k_len: int, out, count: Optional[int] = None, r=None, r_len: int = 0) -> None:
        """Device-resident MultConst (bgn.go:253-291): k holds count x k_len big-endian scalar bytes on the device;
        `out` may be `a` itself."""
        count = a.numel() // self.elem_bytes if count is None else count
        self._need("a", a, count * self.elem_bytes)
        self._need("k", k, count * k_len)
        self._need("r", r, count * r_len)
        self._need("out", out, count * self.elem_bytes)
        check(self._lib.bgn_multconst_batch_dev(self._h, count, level, a.data_ptr(), k.data_ptr(), k_len,
                                                r.data_ptr() if r is not None else None, r_len, out.data_ptr(),
                                                self._stream()), "bgn_multconst_batch_dev")

    def validate_dev(self, level: int, a, ok, count: Optional[int] = None) -> None:
        """ok[i] = 1 per element that is a valid encoding (bgn_validate_batch_dev)."""
        count = a.numel() // self.elem_bytes if count is None else count
        self._need("a", a, count * self.elem_bytes)
        self._need("ok", ok, count)
        check(self._lib.bgn_validate_batch_dev(self._h, count, level, a.data_ptr(), ok.data_ptr(), self._stream()),
              "bgn_validate_batch_dev")

    def decrypt_dev(self, level: int, ct, m, status, count: Optional[int] = None) -> None:
        count = ct.numel() // self.elem_bytes if count is None else count
        self._need("ct", ct, count * self.elem_bytes)
        self._need("m", m, count * 8)
        self._need("status", status, count)
        check(self._lib.bgn_decrypt_batch_dev(self._h, count, level, ct.data_ptr(), m.data_ptr(), status.data_ptr(),
                                              self._stream()), "bgn_decrypt_batch_dev")

    def poly_mult_dev(self, npoly: int, d1: int, d2: int, a, b, out) -> None:
        """out: npoly x (d1 + d2) level-2 coefficients (the last one of each product is the identity)."""
        self._need("a", a, npoly * d1 * self.elem_bytes)
        self._need("b", b, npoly * d2 * self.elem_bytes)
        self._need("out", out, npoly * (d1 + d2) * self.elem_bytes)
        check(self._lib.bgn_poly_mult_batch_dev(self._h, npoly, d1, d2, a.data_ptr(), b.data_ptr(), out.data_ptr(),
                                                self._stream()), "bgn_poly_mult_batch_dev")

    def poly_multconst_dev(self, npoly: int, d: int, dp: int, level: int, ct, p, k_len: int, per_poly: bool, out) -> None:
        self._need("ct", ct, npoly * d * self.elem_bytes)
        self._need("p", p, (npoly if per_poly else 1) * dp * k_len)
        self._need("out", out, npoly * (d + dp) * self.elem_bytes)
        check(self._lib.bgn_poly_multconst_batch_dev(self._h, npoly, d, dp, level, ct.data_ptr(), p.data_ptr(), k_len,
                                                     1 if per_poly else 0, out.data_ptr(), self._stream()),
              "bgn_poly_multconst_batch_dev")

    def poly_eval_dev(self, npoly: int, d: int, level: int, ct, base: int, out) -> None:
        self._need("ct", ct, npoly * d * self.elem_bytes)
        self._need("out", out, npoly * self.elem_bytes)
        check(self._lib.bgn_poly_eval_batch_dev(self._h, npoly, d, level, ct.data_ptr(), base, out.data_ptr(),
                                                self._stream()), "bgn_poly_eval_batch_dev")

    def field_ops(self, xy: BytesLike):
        """Diagnostics: (x*y || 1/x, x^2 || y^2) for elements x||y of plain residues (bgn_field_ops_batch)."""
        A = _as_u8(xy, self.elem_bytes)
        o1, o2 = self._out(len(A)), self._out(len(A))
        check(self._lib.bgn_field_ops_batch(self._h, len(A), _ptr(A), _ptr(o1), _ptr(o2)), "bgn_field_ops_batch")
        return o1, o2

    def field_sums(self, xy: BytesLike):
        """Diagnostics: (x^2 + y^2) || (x*y + y^2), each as one sum of two products with a shared reduction
        (bgn_field_sums_batch)."""
        A = _as_u8(xy, self.elem_bytes)
        o = self._out(len(A))
        check(self._lib.bgn_field_sums_batch(self._h, len(A), _ptr(A), _ptr(o)), "bgn_field_sums_batch")
        return o

    def memory_bytes(self) -> int:
        """Device memory the context holds now (tables of the key + workspace)."""
        return int(self._lib.bgn_ctx_memory_bytes(self._h))

    def set_memory_budget(self, nbytes: int) -> None:
        """Cap on memory_bytes(): tables built afterwards are sized within it (0: none).  The reference keeps the tables
        of every key it has seen (gsbs.go:12-15); with a budget per key several keys share one GPU."""
        check(self._lib.bgn_ctx_set_memory_budget(self._h, int(nbytes)), "bgn_ctx_set_memory_budget")

    # ---- options / calibration / combiner ---------------------------------------
    def set_option(self, name: str, value: int) -> None:
        """bgn_ctx_set_option: a named knob of this context (include/bgn_amd.h; names in csrc/options.hpp)."""
        check(self._lib.bgn_ctx_set_option(self._h, name.encode(), int(value)), "bgn_ctx_set_option(%s)" % name)

    def get_option(self, name: str) -> int:
        v = C.c_int64()
        check(self._lib.bgn_ctx_get_option(self._h, name.encode(), C.byref(v)), "bgn_ctx_get_option(%s)" % name)
        return int(v.value)

    def reset_options(self) -> None:
        check(self._lib.bgn_ctx_reset_options(self._h), "bgn_ctx_reset_options")

    def options(self, **kw):
        """Context manager: set the given options, restore their previous values on exit."""
        eng = self

        class _Scope:
            def __enter__(self_):
                self_.keep = {k: eng.get_option(k) for k in kw}
                for k, v in kw.items():
                    eng.set_option(k, v)
                return eng

            def __exit__(self_, *exc):
                for k, v in self_.keep.items():
                    eng.set_option(k, v)
                return False

        return _Scope()

    def force_kernel(self, kernel: Optional[str]) -> None:
        """'coop', 'quad' or 'lane': every Mult / makeL2 / Decrypt / MultConst batch on that kernel family whatever
        its size; None: back to the dispatch by batch size."""
        names = ("coop_max", "coop_max_l2", "coop_max_dec", "quad_max", "quad_max_l2", "quad_max_dec", "quad_max_pow",
                 "quad_max_mc", "quad_min")
        if kernel is None:
            for k in names:
                self.set_option(k, -1)
            return
        if kernel not in ("coop", "quad", "lane"):
            raise ValueError(kernel)
        big = 1 << 40
        self.set_option("quad_min", 0)
        for k in names[:3]:
            self.set_option(k, big if kernel == "coop" else 0)
        for k in names[3:8]:
            self.set_option(k, big if kernel == "quad" else 0)

    def calibrate(self):
        """bgn_ctx_calibrate: crossovers re-derived from timed probes on this device.  Returns
        {'coop': [Mult, makeL2, lift, power], 'quad': [...]} in elements (-1: not calibrated)."""
        out = (C.c_int64 * 8)()
        check(self._lib.bgn_ctx_calibrate(self._h, out), "bgn_ctx_calibrate")
        return {"coop": [int(v) for v in out[:4]], "quad": [int(v) for v in out[4:]]}

    def combiner_stats(self):
        out = (C.c_uint64 * 5)()
        check(self._lib.bgn_ctx_combiner_stats(self._h, out), "bgn_ctx_combiner_stats")
        return dict(zip(("calls", "rounds", "groups", "elements", "max_group"), (int(v) for v in out)))

    def last_kernel_ms(self) -> float:
        return float(self._lib.bgn_last_kernel_ms(self._h))

    def last_kernel_resources(self) -> dict:
        """Registers, scratch and LDS of the kernel last_kernel_name names (bgn_last_kernel_resources)."""
        out = (C.c_int64 * 4)()
        check(self._lib.bgn_last_kernel_resources(self._h, out), "bgn_last_kernel_resources")
        return {"vgprs": int(out[0]), "scratch_bytes_per_lane": int(out[1]), "lds_bytes_per_workgroup": int(out[2]),
                "max_threads_per_workgroup": int(out[3])}

    def last_kernel_name(self) -> str:
        return self._lib.bgn_last_kernel_name(self._h).decode()

    def last_aux_kernel_ms(self) -> float:
        return float(self._lib.bgn_last_aux_kernel_ms(self._h))

    def last_aux_kernel_name(self) -> str:
        return self._lib.bgn_last_aux_kernel_name(self._h).decode()


class MultiEngine:
    """One bgn_mctx: a public key replicated on several GPUs of the node, driven from this process
    (include/bgn_amd.h, "several GPUs of one node").  Batches split into contiguous shards (MultPoly by
    polynomial, poly.go:139-153) and every shard runs on its own device; results land in one array."""

    def __init__(self, p: int, n: int, l: int, P_wire: bytes, Q_wire: bytes, deterministic: bool = True,
                 devices: Sequence[int] = (0,)):
        self._lib = _lib.load()
        self.p, self.n, self.l = int(p), int(n), int(l)
        pb = _int_bytes(self.p, (self.p.bit_length() + 7) // 8)
        nb = _int_bytes(self.n, (self.n.bit_length() + 7) // 8)
        self.devices = [int(d) for d in devices]
        devs = (C.c_int * len(self.devices))(*self.devices)
        self._h = C.c_void_p()
        self._keep = (pb, nb, bytes(P_wire), bytes(Q_wire))
        check(self._lib.bgn_mctx_create(C.byref(self._h), pb, len(pb), nb, len(nb), self.l, self._keep[2],
                                        self._keep[3], 1 if deterministic else 0, devs, len(self.devices)),
              "bgn_mctx_create")
        self.L = int(self._lib.bgn_fp_bytes(self._lib.bgn_mctx_ctx(self._h, 0)))
        self.elem_bytes = 2 * self.L

    def close(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            self._lib.bgn_mctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def shard_ranges(self, total: int):
        out = []
        for r in range(len(self.devices)):
            lo, hi = C.c_size_t(), C.c_size_t()
            self._lib.bgn_shard_range(total, len(self.devices), r, C.byref(lo), C.byref(hi))
            out.append((lo.value, hi.value))
        return out

    def set_secret(self, q1: int) -> None:
        b = _int_bytes(q1, (int(q1).bit_length() + 7) // 8)
        check(self._lib.bgn_mctx_set_secret(self._h, b, len(b)), "bgn_mctx_set_secret")

    def setup_decryption(self, msg_space: int) -> None:
        check(self._lib.bgn_mctx_setup_decryption(self._h, int(msg_space)), "bgn_mctx_setup_decryption")

    def set_option(self, name: str, value: int) -> None:
        check(self._lib.bgn_mctx_set_option(self._h, name.encode(), int(value)), "bgn_mctx_set_option(%s)" % name)

    def _out(self, count: int) -> np.ndarray:
        return np.zeros((count, self.elem_bytes), dtype=np.uint8)

    def encrypt(self, x: Sequence[int], r: Optional[Sequence[int]] = None) -> np.ndarray:
        xs = _scalars(x)
        rs = _scalars(r) if r is not None else None
        out = self._out(len(xs))
        check(self._lib.bgn_mencrypt_batch(self._h, len(xs), _ptr(xs), xs.shape[1], _ptr(rs),
                                           rs.shape[1] if rs is not None else 0, _ptr(out)), "bgn_mencrypt_batch")
        return out

    def _binop(self, fn, name, level, a, b):
        A, B = _as_u8(a, self.elem_bytes), _as_u8(b, self.elem_bytes)
        if len(A) != len(B):
            raise ValueError("operand counts differ")
        out = self._out(len(A))
        check(fn(self._h, len(A), level, _ptr(A), _ptr(B), None, 0, _ptr(out)), name)
        return out

    def add(self, level: int, a: BytesLike, b: BytesLike) -> np.ndarray:
        return self._binop(self._lib.bgn_madd_batch, "bgn_madd_batch", level, a, b)

    def sub(self, level: int, a: BytesLike, b: BytesLike) -> np.ndarray:
        return self._binop(self._lib.bgn_msub_batch, "bgn_msub_batch", level, a, b)

    def mult(self, a: BytesLike, b: BytesLike) -> np.ndarray:
        A, B = _as_u8(a, self.elem_bytes), _as_u8(b, self.elem_bytes)
        if len(A) != len(B):
            raise ValueError("operand counts differ")
        out = self._out(len(A))
        check(self._lib.bgn_mmult_batch(self._h, len(A), _ptr(A), _ptr(B), None, 0, _ptr(out)), "bgn_mmult_batch")
        return out

    def make_l2(self, a: BytesLike) -> np.ndarray:
        A = _as_u8(a, self.elem_bytes)
        out = self._out(len(A))
        check(self._lib.bgn_mmake_l2_batch(self._h, len(A), _ptr(A), _ptr(out)), "bgn_mmake_l2_batch")
        return out

    def multconst(self, level: int, a: BytesLike, k: Sequence[int]) -> np.ndarray:
        A = _as_u8(a, self.elem_bytes)
        ks = _scalars(k)
        if len(ks) != len(A):
            raise ValueError("operand counts differ")
        out = self._out(len(A))
        check(self._lib.bgn_mmultconst_batch(self._h, len(A), level, _ptr(A), _ptr(ks), ks.shape[1], None, 0,
                                             _ptr(out)), "bgn_mmultconst_batch")
        return out

    def decrypt(self, level: int, ct: BytesLike):
        A = _as_u8(ct, self.elem_bytes)
        m = np.zeros(len(A), dtype=np.int64)
        st = np.zeros(len(A), dtype=np.uint8)
        check(self._lib.bgn_mdecrypt_batch(self._h, len(A), level, _ptr(A), _ptr(m), _ptr(st)), "bgn_mdecrypt_batch")
        return m, st

    def poly_mult(self, npoly: int, d1: int, d2: int, a: BytesLike, b: BytesLike) -> np.ndarray:
        A, B = _as_u8(a, self.elem_bytes), _as_u8(b, self.elem_bytes)
        if len(A) != npoly * d1 or len(B) != npoly * d2:
            raise ValueError("coefficient array sizes do not match npoly*d1 / npoly*d2")
        out = self._out(npoly * (d1 + d2))
        check(self._lib.bgn_mpoly_mult_batch(self._h, npoly, d1, d2, _ptr(A), _ptr(B), _ptr(out)),
              "bgn_mpoly_mult_batch")
        return out

    # device arrays (torch uint8 CUDA tensors) resident on device `root`; every shard waits for the work queued on
    # torch's current stream of that device (where the caller produced the arrays); returns after the gather
    @staticmethod
    def _root_stream(root: int):
        import torch
        return torch.cuda.current_stream(root).cuda_stream

    def mult_dev(self, a, b, out, root: int, count: Optional[int] = None) -> None:
        count = a.numel() // self.elem_bytes if count is None else count
        for what, t in (("a", a), ("b", b), ("out", out)):
            Engine._need(what, t, count * self.elem_bytes)
        check(self._lib.bgn_mmult_batch_dev(self._h, count, a.data_ptr(), b.data_ptr(), out.data_ptr(), root,
                                            self._root_stream(root)), "bgn_mmult_batch_dev")

    def decrypt_dev(self, level: int, ct, m, status, root: int, count: Optional[int] = None) -> None:
        count = ct.numel() // self.elem_bytes if count is None else count
        Engine._need("ct", ct, count * self.elem_bytes)
        Engine._need("m", m, count * 8)
        Engine._need("status", status, count)
        check(self._lib.bgn_mdecrypt_batch_dev(self._h, count, level, ct.data_ptr(), m.data_ptr(), status.data_ptr(),
                                               root, self._root_stream(root)), "bgn_mdecrypt_batch_dev")

    def poly_mult_dev(self, npoly: int, d1: int, d2: int, a, b, out, root: int) -> None:
        Engine._need("a", a, npoly * d1 * self.elem_bytes)
        Engine._need("b", b, npoly * d2 * self.elem_bytes)
        Engine._need("out", out, npoly * (d1 + d2) * self.elem_bytes)
        check(self._lib.bgn_mpoly_mult_batch_dev(self._h, npoly, d1, d2, a.data_ptr(), b.data_ptr(), out.data_ptr(),
                                                 root, self._root_stream(root)), "bgn_mpoly_mult_batch_dev")


# ---------------------------------------------------------------------------
# Reference-shaped objects
# ---------------------------------------------------------------------------
@dataclass
class Ciphertext:
    """ciphertext.go:12-15.  C holds the element's PBC wire bytes (2L); the G1
    identity is 2L zero bytes."""
    C: bytes
    L2: bool = False

    def Copy(self) -> "Ciphertext":
        return Ciphertext(self.C, self.L2)

    def Bytes(self) -> bytes:
        """ciphertext.go:76-92: the gob envelope of {CBytes, L2}."""
        return gob.marshal_ciphertext(self.C, self.L2)


@dataclass
class PolyCiphertext:
    """ciphertext.go:26-31"""
    Coefficients: List[Ciphertext]
    Degree: int
    ScaleFactor: int
    L2: bool

    def Copy(self) -> "PolyCiphertext":
        return PolyCiphertext(self.Coefficients, self.Degree, self.ScaleFactor, self.L2)   # ciphertext.go:41-43

    def Bytes(self) -> bytes:
        """ciphertext.go:94-116"""
        return gob.marshal_poly_ciphertext([c.C for c in self.Coefficients], self.Degree, self.ScaleFactor, self.L2)


@dataclass
class ProofOfPlaintextKnowledge:
    """gadgets.go:10-14"""
    Ct: Ciphertext
    Nonce: Ciphertext
    DL: int


@dataclass
class DecryptionProof:
    """gadgets.go:18-21"""
    Value: int
    Randomness: int


def NewDecryptionProof(v: int, r: int) -> DecryptionProof:
    return DecryptionProof(int(v), int(r))                                    # gadgets.go:24-28


class PublicKey:
    """bgn.go:28-41 — hot-path methods only (keygen and plaintext encoding stay
    on the CPU side of the boundary, north_star)."""

    def __init__(self, p: int, n: int, l: int, P: bytes, Q: bytes, MsgSpace: int, Deterministic: bool = True,
                 PolyBase: int = 3, device: int = 0, FPScaleBase: int = 3, FPPrecision: float = 0.0001):
        self.N = int(n)
        self.FPScaleBase, self.FPPrecision = int(FPScaleBase), float(FPPrecision)     # PolyEncodingParams, bgn.go:43-48
        self.P, self.Q = bytes(P), bytes(Q)
        self.MsgSpace = int(MsgSpace)
        self.Deterministic = bool(Deterministic)
        self.PolyBase = int(PolyBase)
        self.engine = Engine(p, n, l, P, Q, Deterministic, device)
        self._zero = bytes(self.engine.elem_bytes)

    # -- helpers --
    def _r(self, count: int = 1):
        """Blinding randomness: None in deterministic mode (bgn.go:462,484)."""
        if self.Deterministic:
            return None
        return [secrets.randbelow(self.N) for _ in range(count)]     # newCryptoRandom, bgn.go:567-574

    def _blind(self, rows: np.ndarray, l2: bool) -> np.ndarray:
        """Results of a fused loop on a key with Deterministic == false: the reference blinds every Mult / MultConst /
        Add of the loop with fresh randomness (bgn.go:260-269, :302-311, :466-474, :488-495), so each result ends up
        multiplied by Q (resp. e(Q,Q)) to a sum of fresh exponents — one uniformly random power, applied here by
        adding the identity with blinding."""
        if self.Deterministic or len(rows) == 0:
            return rows
        E = self.engine.elem_bytes
        ident = ((1).to_bytes(E // 2, "big") + bytes(E // 2)) if l2 else bytes(E)
        return self.engine.add(2 if l2 else 1, rows.tobytes(), ident * len(rows), self._r(len(rows)))

    def _lvl(self, ct: Ciphertext) -> int:
        return 2 if ct.L2 else 1

    # -- wire envelopes: bgn.go:501-560 --
    def _elem(self, b: bytes) -> bytes:
        if len(b) == 0:
            return self._zero                        # gob drops empty slices; the identity is all zero
        if len(b) != self.engine.elem_bytes:
            raise ValueError(f"element of {len(b)} bytes, expected {self.engine.elem_bytes}")
        return bytes(b)

    def NewCiphertextFromBytes(self, data: bytes) -> Ciphertext:
        c, l2 = gob.unmarshal_ciphertext(data)       # raises "no data provided" on empty input (bgn.go:503-505)
        return Ciphertext(self._elem(c), l2)

    def NewPolyCiphertextFromBytes(self, data: bytes) -> PolyCiphertext:
        coeffs, degree, scale, l2 = gob.unmarshal_poly_ciphertext(data)
        return PolyCiphertext([Ciphertext(self._elem(c), l2) for c in coeffs], degree, scale, l2)

    # -- encryption: bgn.go:325-353 --
    def EncryptBatch(self, xs: Sequence[int], rs: Optional[Sequence[int]] = None) -> List[Ciphertext]:
        # negative plaintexts: defined as P^(x mod n) (the reference hands a negative
        # exponent to PBC, cmd/main.go:81, whose result is not defined in-tree)
        xs = [int(x) % self.N if int(x) < 0 else int(x) for x in xs]
        out = self.engine.encrypt(xs, rs)
        return [Ciphertext(bytes(row), False) for row in out]

    def EncryptWithRandomness(self, x: int, r: int) -> Ciphertext:
        return self.EncryptBatch([x], [r])[0]

    def Encrypt(self, x: int) -> Ciphertext:
        return self.EncryptWithRandomness(x, secrets.randbelow(self.N))      # bgn.go:334-337

    def EncryptDeterministic(self, x: int) -> Ciphertext:
        return self.EncryptBatch([x], None)[0]                              # bgn.go:325-331

    def encryptZero(self) -> Ciphertext:
        return self.EncryptDeterministic(0)                                 # bgn.go:562-564

    # -- level lift / Mult: bgn.go:294-321 --
    def makeL2(self, ct: Ciphertext) -> Ciphertext:
        return Ciphertext(bytes(self.engine.make_l2(ct.C)[0]), True)

    def MultBatch(self, a: Sequence[Ciphertext], b: Sequence[Ciphertext]) -> List[Ciphertext]:
        out = self.engine.mult(b"".join(c.C for c in a), b"".join(c.C for c in b), self._r(len(a)))
        return [Ciphertext(bytes(row), True) for row in out]

    def Mult(self, ct1: Ciphertext, ct2: Ciphertext) -> Ciphertext:
        return self.MultBatch([ct1], [ct2])[0]

    # -- Add / Sub / Neg: bgn.go:375-497 --
    def _align(self, a: Ciphertext, b: Ciphertext):
        if a.L2 and not b.L2:
            b = self.makeL2(b)                                              # bgn.go:447-449
        if not a.L2 and b.L2:
            a = self.makeL2(a)                                              # bgn.go:451-453
        return a, b

    def Add(self, a: Ciphertext, b: Ciphertext) -> Ciphertext:
        a, b = self._align(a, b)
        return Ciphertext(bytes(self.engine.add(self._lvl(a), a.C, b.C, self._r())[0]), a.L2)

    def Sub(self, a: Ciphertext, b: Ciphertext) -> Ciphertext:
        a, b = self._align(a, b)
        return Ciphertext(bytes(self.engine.sub(self._lvl(a), a.C, b.C, self._r())[0]), a.L2)

    def Neg(self, c: Ciphertext) -> Ciphertext:
        return self.Sub(self.encryptZero(), c)                              # bgn.go:436-438

    def AddBatch(self, a: Sequence[Ciphertext], b: Sequence[Ciphertext]) -> List[Ciphertext]:
        if not a:
            return []
        lvl = self._lvl(a[0])
        if any(self._lvl(c) != lvl for c in list(a) + list(b)):
            raise ValueError("AddBatch needs operands of one level")
        out = self.engine.add(lvl, b"".join(c.C for c in a), b"".join(c.C for c in b), self._r(len(a)))
        return [Ciphertext(bytes(row), lvl == 2) for row in out]

    # -- MultConst: bgn.go:253-291 --
    def MultConst(self, c: Ciphertext, constant: int) -> Ciphertext:
        if constant < 0:
            raise ValueError("negative constants are not defined by the reference's PowBig path")
        return Ciphertext(bytes(self.engine.multconst(self._lvl(c), c.C, [constant], self._r())[0]), c.L2)

    # -- decryption tables: bgn.go:195-201, gsbs.go:41-51 --
    def SetupDecryption(self, sk: "SecretKey") -> None:
        self.engine.set_secret(sk.Key)
        self.engine.setup_decryption(self.MsgSpace)

    # -- poly layer: poly.go --
    def EncryptPoly(self, coeffs: Sequence[int], scale: int = 0) -> PolyCiphertext:
        """poly.go:11-29 on already-encoded digits (plaintext.go stays CPU-side)."""
        enc = []
        for c in coeffs:
            if c < 0:
                enc.append(self.Sub(self.encryptZero(), self.Encrypt(-c)))     # poly.go:17-22
            else:
                enc.append(self.Encrypt(c))
        return PolyCiphertext(enc, len(enc), scale, False)

    def MultPoly(self, ct1: PolyCiphertext, ct2: PolyCiphertext) -> PolyCiphertext:
        """poly.go:123-156 — one engine call for all d1*d2 pairings and the
        segmented GT accumulation."""
        d1, d2 = ct1.Degree, ct2.Degree
        out = self.engine.poly_mult(1, d1, d2, b"".join(c.C for c in ct1.Coefficients),
                                    b"".join(c.C for c in ct2.Coefficients))
        out = self._blind(out, True)
        coeffs = [Ciphertext(bytes(row), True) for row in out]
        return PolyCiphertext(coeffs, d1 + d2, ct1.ScaleFactor + ct2.ScaleFactor, True)

    def NegPoly(self, ct: PolyCiphertext) -> PolyCiphertext:
        res = [self.Sub(self.encryptZero(), c) for c in ct.Coefficients]        # poly.go:45-55
        return PolyCiphertext(res, ct.Degree, ct.ScaleFactor, ct.L2)

    def MultConstPoly(self, ct: PolyCiphertext, constant, negative: bool = False) -> PolyCiphertext:
        """poly.go:71-120.  `constant` is the ENCODED plaintext — the reference encodes its *big.Float argument
        with NewUnbalancedPlaintext (plaintext.go:34-63), which stays on the CPU side of the boundary: an object
        with .Coefficients / .ScaleFactor, a (digits, scale) pair, or a plain integer (expanded here, sign split
        as poly.go:73-76).  `negative`: the constant was negative and |constant| was encoded (NegPoly of the
        product, poly.go:115-119).  One engine call computes the whole convolution
        result[i+k] += MultConst(ct[i], p[k])."""
        if hasattr(constant, "Coefficients"):
            digits, scale = [int(c) for c in constant.Coefficients], int(constant.ScaleFactor)
        elif isinstance(constant, tuple):
            digits, scale = list(constant[0]), int(constant[1])
        else:
            if int(constant) != constant:
                raise TypeError("encode fractional constants on the host side (plaintext.go) and pass the digits")
            negative = negative != (constant < 0)                            # poly.go:73-76
            digits, scale = self.NewUnbalancedPlaintext(abs(int(constant)))
        out = self.engine.poly_multconst(1, ct.Degree, 2 if ct.L2 else 1, b"".join(c.C for c in ct.Coefficients), digits)
        out = self._blind(out, ct.L2)                                        # bgn.go:260-269 on every step of the loop
        prod = PolyCiphertext([Ciphertext(bytes(r), ct.L2) for r in out], ct.Degree + len(digits),
                              ct.ScaleFactor + scale, ct.L2)
        return self.NegPoly(prod) if negative else prod                      # poly.go:115-119

    def MakePolyL2(self, ct: PolyCiphertext) -> PolyCiphertext:
        """poly.go:159-163: MultPoly(EncryptPoly(1), ct).  The encoding of 1 is the single digit [1]."""
        one = PolyCiphertext([self.Encrypt(1)], 1, 0, False)                 # EncryptPoly -> pk.Encrypt, poly.go:24
        return self.MultPoly(one, ct)

    def alignPolyCiphertexts(self, ct1: PolyCiphertext, ct2: PolyCiphertext):
        """poly.go:209-226: bring both operands to the larger scale factor by MultConstPoly with
        FPScaleBase^diff.  Returns the pair in the reference's order (larger scale first)."""
        if ct1.ScaleFactor > ct2.ScaleFactor:
            diff = ct1.ScaleFactor - ct2.ScaleFactor
            ct2 = self.MultConstPoly(ct2, self.FPScaleBase ** diff)
            ct2.ScaleFactor = ct1.ScaleFactor
        elif ct2.ScaleFactor > ct1.ScaleFactor:
            return self.alignPolyCiphertexts(ct2, ct1)
        return ct1, ct2

    def SubPoly(self, ct1: PolyCiphertext, ct2: PolyCiphertext) -> PolyCiphertext:
        return self.AddPoly(ct1, self.NegPoly(ct2))                          # poly.go:166-168

    def AddPoly(self, p1: PolyCiphertext, p2: PolyCiphertext) -> PolyCiphertext:
        """poly.go:171-207: lift to a common level, align the scale factors, add coefficient-wise (one batch
        call for the common prefix)."""
        if p1.L2 or p2.L2:
            if not p1.L2:
                return self.AddPoly(self.MakePolyL2(p1), p2)                 # poly.go:175-177
            if not p2.L2:
                return self.AddPoly(p1, self.MakePolyL2(p2))                 # poly.go:179-181
        p1, p2 = self.alignPolyCiphertexts(p1, p2)
        deg, common = max(p1.Degree, p2.Degree), min(p1.Degree, p2.Degree)
        res = self.AddBatch(p1.Coefficients[:common], p2.Coefficients[:common])
        res += (p1 if p1.Degree > p2.Degree else p2).Coefficients[common:deg]
        return PolyCiphertext(res, deg, p1.ScaleFactor, p1.L2)

    def EvalPoly(self, ct: PolyCiphertext) -> Ciphertext:
        """poly.go:58-68 (Horner over MultConst/Add) as one multi-scalar sum on the device."""
        out = self.engine.poly_eval(1, ct.Degree, 2 if ct.L2 else 1, b"".join(c.C for c in ct.Coefficients), self.PolyBase)
        return Ciphertext(bytes(self._blind(out, ct.L2)[0]), ct.L2)

    # -- proofs: gadgets.go --
    def _hash(self, ct_bytes: bytes, nonce_bytes: bytes) -> int:
        import hashlib
        return int.from_bytes(hashlib.sha256(bytes(ct_bytes) + bytes(nonce_bytes)).digest(), "big")   # gadgets.go:80-96

    def NewProofOfPlaintextKnowledge(self, sk: "SecretKey", v: int, z: int, nonce1: Optional[int] = None):
        """gadgets.go:32-54 (`nonce1` may be supplied to make the proof reproducible)."""
        if nonce1 is None:
            nonce1 = secrets.randbelow(self.N)
        ct, nonce = self.EncryptBatch([v, nonce1], [z, 0])
        nonce2 = self._hash(ct.C, nonce.C)
        DL = (nonce1 + nonce2 * v + sk.R * z * nonce2 * (self.N // sk.Key)) % self.N
        return ProofOfPlaintextKnowledge(ct, nonce, DL)

    def CheckDecryptionProofBatch(self, cts: Sequence[Ciphertext], proofs: Sequence["DecryptionProof"]) -> List[bool]:
        if not cts:
            return []
        v = _scalars([pr.Value for pr in proofs])
        r = _scalars([pr.Randomness for pr in proofs])
        a = _as_u8(b"".join(c.C for c in cts), self.engine.elem_bytes)
        ok = np.zeros(len(cts), dtype=np.uint8)
        check(self.engine._lib.bgn_check_decryption_proof_batch(self.engine._h, len(cts), _ptr(a), _ptr(v), v.shape[1],
                                                                _ptr(r), r.shape[1], _ptr(ok)),
              "bgn_check_decryption_proof_batch")
        return [bool(x) for x in ok]

    def CheckDecryptionProof(self, ct: Ciphertext, proof: "DecryptionProof") -> bool:
        return self.CheckDecryptionProofBatch([ct], [proof])[0]                # gadgets.go:57-61

    def CheckProofOfPlaintextKnoewledgeBatch(self, cts: Sequence[Ciphertext],
                                             proofs: Sequence["ProofOfPlaintextKnowledge"]) -> List[bool]:
        if not cts:
            return []
        E = self.engine.elem_bytes
        c = _scalars([self._hash(pr.Ct.C, pr.Nonce.C) for pr in proofs], 32)   # nonce2 := hash(proof), gadgets.go:67
        dl = _scalars([pr.DL for pr in proofs])
        a = _as_u8(b"".join(x.C for x in cts), E)
        nn = _as_u8(b"".join(pr.Nonce.C for pr in proofs), E)
        ok = np.zeros(len(cts), dtype=np.uint8)
        check(self.engine._lib.bgn_check_plaintext_knowledge_batch(self.engine._h, len(cts), _ptr(a), _ptr(nn), _ptr(c), 32,
                                                                   _ptr(dl), dl.shape[1], _ptr(ok)),
              "bgn_check_plaintext_knowledge_batch")
        return [bool(x) for x in ok]

    def CheckProofOfPlaintextKnoewledge(self, ct: Ciphertext, proof: "ProofOfPlaintextKnowledge") -> bool:
        return self.CheckProofOfPlaintextKnoewledgeBatch([ct], [proof])[0]     # gadgets.go:65-77 (name as in the reference)

    # -- the integer expansion alignPolyCiphertexts needs (poly.go:209-226 multiplies by FPScaleBase^diff) --
    def NewUnbalancedPlaintext(self, m: int):
        """Unbalanced base-b digits of an integer m >= 0 and scale factor 0 (plaintext.go:58-62).  Fractional
        values go through plaintext.go's rationalize on the host side, outside this package."""
        if int(m) != m:
            raise TypeError("integers only: the fixed-point encoding of plaintext.go stays on the host side")
        m = int(m)
        if m < 0:
            raise ValueError("Negative encoding not supported")             # plaintext.go:175-177
        return unbalanced_encode(m, self.PolyBase), 0


def unbalanced_encode(target: int, base: int) -> List[int]:
    """unbalancedEncode (plaintext.go:164-212): greedy from the top power, digit 2 when 2*b^i still fits, else 1;
    for base 3 this is the ordinary ternary expansion.  Like the reference, the result carries one zero
    coefficient above the top digit (`coefficients[:bound+1]` with bound = top index + 1).  Zero encodes as [0]."""
    if target == 0:
        return [0]
    digits: List[int] = []
    top = None
    while True:
        i = 0
        while base ** (i + 1) <= target:                                   # degree(): largest i with b^i <= target
            i += 1
        if top is None:
            top = i
            digits = [0] * (top + 2)
        v = base ** i
        if 2 * v <= target:
            v, digits[i] = 2 * v, 2
        else:
            digits[i] = 1
        if v == target:
            return digits
        target -= v


class SecretKey:
    """bgn.go:58-62"""

    def __init__(self, Key: int, R: int = 0, PolyBase: int = 3):
        self.Key, self.R, self.PolyBase = int(Key), int(R), int(PolyBase)

    def DecryptBatch(self, cts: Sequence[Ciphertext], pk: PublicKey):
        """Returns (values, status) arrays; status 1 = reference's error."""
        if not cts:
            return np.zeros(0, np.int64), np.zeros(0, np.uint8)
        lvl = 2 if cts[0].L2 else 1
        if any((2 if c.L2 else 1) != lvl for c in cts):
            raise ValueError("DecryptBatch needs ciphertexts of one level")
        return pk.engine.decrypt(lvl, b"".join(c.C for c in cts))

    def Decrypt(self, ct: Ciphertext, pk: PublicKey) -> int:
        m, st = self.DecryptBatch([ct], pk)                                  # bgn.go:205-207
        if st[0] != _lib.BGN_DL_OK:
            raise DecryptError()
        return int(m[0])

    def DecryptFailSafe(self, ct: Ciphertext, pk: PublicKey) -> int:
        m, st = self.DecryptBatch([ct], pk)                                  # bgn.go:210-216
        return int(m[0]) if st[0] == _lib.BGN_DL_OK else 0

    def DecryptPoly(self, ct: PolyCiphertext, pk: PublicKey) -> List[int]:
        m, _ = self.DecryptBatch(ct.Coefficients, pk)                        # poly.go:32-42 ignores errors
        return [int(v) for v in m]
