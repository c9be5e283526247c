"""ctypes binding of libbgn_amd.so (include/bgn_amd.h).

The product path has no CPU fallback: if the HIP library is missing this
module raises at import of the symbols, and every compute call fails with
BGN_E_HIP when no GPU is present.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BGN_AMD_LIB selects another build of the same library (A/B measurements, packaging)
LIB_PATH = os.environ.get("BGN_AMD_LIB") or os.path.join(_HERE, "lib", "libbgn_amd.so")

BGN_OK = 0
BGN_E_ARG, BGN_E_PARAM, BGN_E_HIP, BGN_E_STATE, BGN_E_POINT, BGN_E_NOMEM = -1, -2, -3, -4, -5, -6
BGN_DL_OK, BGN_DL_NOT_FOUND = 0, 1

_u8p = C.c_void_p      # raw addresses (host bytes or device pointers)
_sz = C.c_size_t
_ctx = C.c_void_p

# name -> (restype, argtypes); mirrors include/bgn_amd.h one to one
PROTOTYPES = {
    "bgn_ctx_create": (C.c_int, [C.POINTER(_ctx), _u8p, _sz, _u8p, _sz, C.c_uint64, _u8p, _u8p, C.c_int, C.c_int]),
    "bgn_ctx_destroy": (None, [_ctx]),
    "bgn_fp_bytes": (_sz, [_ctx]),
    "bgn_last_error": (C.c_char_p, []),
    "bgn_version": (C.c_char_p, []),
    "bgn_ctx_set_option": (C.c_int, [_ctx, C.c_char_p, C.c_int64]),
    "bgn_ctx_get_option": (C.c_int, [_ctx, C.c_char_p, C.POINTER(C.c_int64)]),
    "bgn_ctx_reset_options": (C.c_int, [_ctx]),
    "bgn_option_name": (C.c_char_p, [_sz]),
    "bgn_ctx_calibrate": (C.c_int, [_ctx, C.POINTER(C.c_int64)]),
    "bgn_ctx_combiner_stats": (C.c_int, [_ctx, C.POINTER(C.c_uint64)]),
    "bgn_mctx_set_option": (C.c_int, [_ctx, C.c_char_p, C.c_int64]),
    "bgn_ctx_set_secret": (C.c_int, [_ctx, _u8p, _sz]),
    "bgn_ctx_setup_decryption": (C.c_int, [_ctx, C.c_uint64]),
    "bgn_encrypt_batch": (C.c_int, [_ctx, _sz, _u8p, _sz, _u8p, _sz, _u8p]),
    "bgn_add_batch": (C.c_int, [_ctx, _sz, C.c_int, _u8p, _u8p, _u8p, _sz, _u8p]),
    "bgn_sub_batch": (C.c_int, [_ctx, _sz, C.c_int, _u8p, _u8p, _u8p, _sz, _u8p]),
    "bgn_neg_batch": (C.c_int, [_ctx, _sz, C.c_int, _u8p, _u8p]),
    "bgn_mult_batch": (C.c_int, [_ctx, _sz, _u8p, _u8p, _u8p, _sz, _u8p]),
    "bgn_make_l2_batch": (C.c_int, [_ctx, _sz, _u8p, _u8p]),
    "bgn_multconst_batch": (C.c_int, [_ctx, _sz, C.c_int, _u8p, _u8p, _sz, _u8p, _sz, _u8p]),
    "bgn_decrypt_batch": (C.c_int, [_ctx, _sz, C.c_int, _u8p, _u8p, _u8p]),
    "bgn_poly_mult_batch": (C.c_int, [_ctx, _sz, _sz, _sz, _u8p, _u8p, _u8p]),
    "bgn_poly_multconst_batch": (C.c_int, [_ctx, _sz, _sz, _sz, C.c_int, _u8p, _u8p, _sz, C.c_int, _u8p]),
    "bgn_poly_eval_batch": (C.c_int, [_ctx, _sz, _sz, C.c_int, _u8p, C.c_uint64, _u8p]),
    "bgn_validate_batch": (C.c_int, [_ctx, _sz, C.c_int, _u8p, _u8p]),
    "bgn_validate_batch_dev": (C.c_int, [_ctx, _sz, C.c_int, _u8p, _u8p, C.c_void_p]),
    "bgn_check_decryption_proof_batch": (C.c_int, [_ctx, _sz, _u8p, _u8p, _sz, _u8p, _sz, _u8p]),
    "bgn_check_plaintext_knowledge_batch": (C.c_int, [_ctx, _sz, _u8p, _u8p, _u8p, _sz, _u8p, _sz, _u8p]),
    "bgn_encrypt_batch_dev": (C.c_int, [_ctx, _sz, _u8p, _sz, _u8p, _sz, _u8p, C.c_void_p]),
    "bgn_add_batch_dev": (C.c_int, [_ctx, _sz, C.c_int, _u8p, _u8p, _u8p, _sz, _u8p, C.c_void_p]),
    "bgn_sub_batch_dev": (C.c_int, [_ctx, _sz, C.c_int, _u8p, _u8p, _u8p, _sz, _u8p, C.c_void_p]),
    "bgn_neg_batch_dev": (C.c_int, [_ctx, _sz, C.c_int, _u8p, _u8p, C.c_void_p]),
    "bgn_mult_batch_dev": (C.c_int, [_ctx, _sz, _u8p, _u8p, _u8p, _sz, _u8p, C.c_void_p]),
    "bgn_make_l2_batch_dev": (C.c_int, [_ctx, _sz, _u8p, _u8p, C.c_void_p]),
    "bgn_multconst_batch_dev": (C.c_int, [_ctx, _sz, C.c_int, _u8p, _u8p, _sz, _u8p, _sz, _u8p, C.c_void_p]),
    "bgn_decrypt_batch_dev": (C.c_int, [_ctx, _sz, C.c_int, _u8p, _u8p, _u8p, C.c_void_p]),
    "bgn_poly_mult_batch_dev": (C.c_int, [_ctx, _sz, _sz, _sz, _u8p, _u8p, _u8p, C.c_void_p]),
    "bgn_poly_multconst_batch_dev": (C.c_int, [_ctx, _sz, _sz, _sz, C.c_int, _u8p, _u8p, _sz, C.c_int, _u8p, C.c_void_p]),
    "bgn_poly_eval_batch_dev": (C.c_int, [_ctx, _sz, _sz, C.c_int, _u8p, C.c_uint64, _u8p, C.c_void_p]),
    "bgn_check_decryption_proof_batch_dev": (C.c_int, [_ctx, _sz, _u8p, _u8p, _sz, _u8p, _sz, _u8p, C.c_void_p]),
    "bgn_check_plaintext_knowledge_batch_dev": (C.c_int, [_ctx, _sz, _u8p, _u8p, _u8p, _sz, _u8p, _sz, _u8p, C.c_void_p]),
    "bgn_shard_range": (None, [_sz, C.c_int, C.c_int, C.POINTER(_sz), C.POINTER(_sz)]),
    "bgn_mctx_create": (C.c_int, [C.POINTER(_ctx), _u8p, _sz, _u8p, _sz, C.c_uint64, _u8p, _u8p, C.c_int,
                                  C.POINTER(C.c_int), C.c_int]),
    "bgn_mctx_destroy": (None, [_ctx]),
    "bgn_mctx_device_count": (C.c_int, [_ctx]),
    "bgn_mctx_ctx": (_ctx, [_ctx, C.c_int]),
    "bgn_mctx_set_secret": (C.c_int, [_ctx, _u8p, _sz]),
    "bgn_mctx_setup_decryption": (C.c_int, [_ctx, C.c_uint64]),
    "bgn_mencrypt_batch": (C.c_int, [_ctx, _sz, _u8p, _sz, _u8p, _sz, _u8p]),
    "bgn_madd_batch": (C.c_int, [_ctx, _sz, C.c_int, _u8p, _u8p, _u8p, _sz, _u8p]),
    "bgn_msub_batch": (C.c_int, [_ctx, _sz, C.c_int, _u8p, _u8p, _u8p, _sz, _u8p]),
    "bgn_mmult_batch": (C.c_int, [_ctx, _sz, _u8p, _u8p, _u8p, _sz, _u8p]),
    "bgn_mmake_l2_batch": (C.c_int, [_ctx, _sz, _u8p, _u8p]),
    "bgn_mmultconst_batch": (C.c_int, [_ctx, _sz, C.c_int, _u8p, _u8p, _sz, _u8p, _sz, _u8p]),
    "bgn_mdecrypt_batch": (C.c_int, [_ctx, _sz, C.c_int, _u8p, _u8p, _u8p]),
    "bgn_mpoly_mult_batch": (C.c_int, [_ctx, _sz, _sz, _sz, _u8p, _u8p, _u8p]),
    "bgn_ctx_memory_bytes": (C.c_uint64, [_ctx]),
    "bgn_ctx_set_memory_budget": (C.c_int, [_ctx, C.c_uint64]),
    "bgn_mmult_batch_dev": (C.c_int, [_ctx, _sz, _u8p, _u8p, _u8p, C.c_int, C.c_void_p]),
    "bgn_mdecrypt_batch_dev": (C.c_int, [_ctx, _sz, C.c_int, _u8p, _u8p, _u8p, C.c_int, C.c_void_p]),
    "bgn_mpoly_mult_batch_dev": (C.c_int, [_ctx, _sz, _sz, _sz, _u8p, _u8p, _u8p, C.c_int, C.c_void_p]),
    "bgn_field_ops_batch": (C.c_int, [_ctx, _sz, _u8p, _u8p, _u8p]),
    "bgn_field_sums_batch": (C.c_int, [_ctx, _sz, _u8p, _u8p]),
    "bgn_host_alloc": (C.c_void_p, [_sz]),
    "bgn_host_free": (None, [C.c_void_p]),
    "bgn_dev_alloc": (C.c_void_p, [_ctx, _sz]),
    "bgn_dev_free": (None, [_ctx, C.c_void_p]),
    "bgn_dev_upload": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, _sz]),
    "bgn_dev_download": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, _sz]),
    "bgn_last_kernel_ms": (C.c_double, [_ctx]),
    "bgn_last_kernel_name": (C.c_char_p, [_ctx]),
    "bgn_last_aux_kernel_ms": (C.c_double, [_ctx]),
    "bgn_last_aux_kernel_name": (C.c_char_p, [_ctx]),
    "bgn_ctx_bsgs_baby_steps": (C.c_uint64, [_ctx]),
    "bgn_last_kernel_resources": (C.c_int, [_ctx, C.POINTER(C.c_int64)]),
}

_lib = None


def load() -> C.CDLL:
    """Load the HIP library; raises OSError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C bgn_amd/csrc).  There is no CPU fallback.")
    # A process that also uses PyTorch-ROCm (device buffers, streams: every caller of this binding does) must have
    # torch's copy of the HIP runtime loaded FIRST: the library then binds to that one.  The other order leaves two HIP
    # runtimes of different versions in the process, and this library's sees no device (bgn_ctx_create: "no HIP
    # device available") — python __graft_entry__.py smoke, which builds and loads before it imports torch, hit it.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class BgnError(RuntimeError):
    def __init__(self, code: int, where: str):
        msg = load().bgn_last_error().decode("utf-8", "replace")
        super().__init__(f"{where} failed with code {code}: {msg}")
        self.code = code


def check(code: int, where: str) -> None:
    if code != BGN_OK:
        raise BgnError(code, where)
