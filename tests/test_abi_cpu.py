"""CPU: the C-ABI library loads and exports every symbol include/bgn_amd.h
declares; without a GPU compute calls fail loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT, load_fixture


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "bgn_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bgn_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from bgn_amd import _lib
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in bgn_amd.h but not exported"
    assert set(names) == set(_lib.PROTOTYPES), "ctypes prototypes out of sync with the header"
    assert b"gfx950" in lib.bgn_version()


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import bgn_amd
    fx = load_fixture("toy64")
    with pytest.raises(bgn_amd.BgnError) as ei:
        bgn_amd.PublicKey(int(fx["p"], 16), int(fx["n"], 16), fx["l"], bytes.fromhex(fx["P"]),
                          bytes.fromhex(fx["Q"]), fx["msg_space"])
    assert ei.value.code == -3   # BGN_E_HIP


def test_bad_parameters_rejected_before_touching_the_gpu():
    from bgn_amd import _lib
    lib = _lib.load()
    fx = load_fixture("toy64")
    p = int(fx["p"], 16)
    pb = (p + 2).to_bytes(fx["fp_bytes"], "big")          # p + 1 != l * n
    nb = int(fx["n"], 16).to_bytes(8, "big")
    h = ctypes.c_void_p()
    rc = lib.bgn_ctx_create(ctypes.byref(h), pb, len(pb), nb, len(nb), fx["l"], bytes.fromhex(fx["P"]),
                            bytes.fromhex(fx["Q"]), 1, 0)
    assert rc == -2 and b"l * n" in lib.bgn_last_error()


def test_product_never_references_the_oracle():
    """The oracle is test infrastructure: nothing under bgn_amd/ may import, link or call it."""
    bad = []
    for dp, _, fns in os.walk(os.path.join(ROOT, "bgn_amd")):
        if "build" in dp:
            continue
        for fn in fns:
            if fn.endswith((".py", ".hpp", ".cpp", ".hip", ".h")) or fn == "Makefile":
                txt = open(os.path.join(dp, fn), errors="replace").read()
                if re.search(r"oracle|bgn_ref|bgn_oracle|tests[/.]emu", txt):
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_host_mirror_refuses_device_arrays_shorter_than_the_call_needs():
    """The C ABI takes bare pointers (include/bgn_amd.h): the Python mirror checks the byte length of every device
    array before it hands the pointer over — a short `out` is a ValueError, not a GPU fault."""
    import pytest
    import torch
    from bgn_amd.api import Engine
    with pytest.raises(ValueError, match="the call needs 4"):
        Engine._need("out", torch.empty(3, dtype=torch.uint8), 4)
    with pytest.raises(ValueError, match="the call needs 16"):
        Engine._need("m", torch.empty(1, dtype=torch.int64), 16)
    with pytest.raises(ValueError, match="CUDA tensor"):
        Engine._need("a", torch.empty(8, dtype=torch.uint8), 8)
    Engine._need("r", None, 100)                      # optional operands


def test_options_are_named_and_the_library_reads_no_environment_per_call():
    """The options surface (bgn_ctx_set_option): every name of csrc/options.hpp is enumerable through the ABI, a
    null context or an unknown option is BGN_E_ARG — and the host sources call getenv in exactly one place, the
    parse of the environment when a context is created (options_from_environment)."""
    from bgn_amd import _lib
    lib = _lib.load()
    names = []
    while True:
        n = lib.bgn_option_name(len(names))
        if not n:
            break
        names.append(n.decode())
    assert len(names) == len(set(names)) >= 30
    for must in ("coop_max", "quad_max", "quad_min", "split_rounds", "combine", "combine_max_count", "test_bsgs_fp_bits"):
        assert must in names
    v = ctypes.c_int64()
    assert lib.bgn_ctx_set_option(None, b"coop_max", 1) == -1
    assert lib.bgn_ctx_get_option(None, b"coop_max", ctypes.byref(v)) == -1
    csrc = os.path.join(ROOT, "bgn_amd", "csrc")
    hits = {}
    for fn in os.listdir(csrc):
        if fn.endswith((".cpp", ".hpp", ".hip")):
            c = len(re.findall(r"\bgetenv\s*\(", open(os.path.join(csrc, fn)).read()))
            if c:
                hits[fn] = c
    assert hits == {"options.hpp": 1}, hits
    src = open(os.path.join(csrc, "options.hpp")).read()
    for n in names:                                    # test hooks are not reachable from the environment
        m = re.search(r'\{"%s", &Options::%s, (true|false)' % (n, n), src)
        assert m and (m.group(1) == "false") == n.startswith("test_"), n
