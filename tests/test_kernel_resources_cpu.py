"""CPU: what the built code objects say about the kernels whose design depends on it (no GPU needed: the metadata
notes of the gfx950 code object inside bgn_amd/csrc/build/kern_nl*.o).  The fused level-2 Add (k_gt_mul_wire) is
meant to hold everything in registers at two workgroups per CU — no scratch memory, at most 256 registers, at most
80 KB of LDS; a change that makes the register allocator spill shows here at build time, before it shows on a clock
(round 6: a third 36-limb constant in scalar registers cost 230 spilled vector registers and a quarter of the rate).
The GPU suite asserts the same through hipFuncGetAttributes (tests/test_gpu_l2_fused.py)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

from conftest import ROOT

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_notes(obj):
    """{mangled kernel name: {field: int}} of the gfx950 code object bundled in a host object file."""
    with tempfile.TemporaryDirectory() as td:
        local = os.path.join(td, "k.o")
        shutil.copy(obj, local)
        subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=td, stdout=subprocess.DEVNULL)
        dev = [f for f in os.listdir(td) if "amdgcn" in f]
        assert dev, "no device code object in %s" % obj
        txt = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(td, dev[0])], text=True)
    out = {}
    for block in txt.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block)
        if not name:
            continue
        f = {"agpr_count": int(block.split()[0])}
        for key in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size"):
            m = re.search(r"\.%s:\s+(\d+)" % key, block)
            if m:
                f[key] = int(m.group(1))
        out[name.group(1)] = f
    return out


@pytest.mark.parametrize("nl", [36, 37, 19])
def test_fused_level2_add_holds_everything_in_registers(nl):
    obj = os.path.join(ROOT, "bgn_amd", "csrc", "build", "kern_nl%d.o" % nl)
    if not os.path.exists(obj) or not os.path.exists(os.path.join(LLVM, "llvm-readelf")):
        pytest.skip("no built kernel objects / LLVM tools here")
    notes = kernel_notes(obj)
    mine = [v for k, v in notes.items() if "k_gt_mul_wireILi%dE" % nl in k]
    assert len(mine) == 1, [k for k in notes if "gt_mul_wire" in k]
    f = mine[0]
    assert f["private_segment_fixed_size"] == 0, f                   # nothing spilled to scratch memory
    assert f["vgpr_count"] + f["agpr_count"] <= 256, f               # two waves per SIMD
    assert f["group_segment_fixed_size"] <= 80 * 1024, f             # two workgroups per CU
