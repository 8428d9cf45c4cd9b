"""The C-ABI boundary without a GPU: the shared library loads, exports every
symbol include/blance_hip.h declares, agrees with the ctypes mirror on struct
layout, and validates problems on the host (no compute calls here)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from blance_amd import abi, hip, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "blance_hip.h")


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build_hip()                      # hipcc cross-compiles for gfx950 without a device
    return hip.load_library()


def test_exports_every_declared_symbol(lib):
    src = open(HEADER).read()
    declared = set(re.findall(r"\b(blance_[a-z_]+)\s*\(", src))
    assert declared == set(hip.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.blance_abi_version() == abi.ABI_VERSION


def test_struct_layout_matches_the_header(tmp_path):
    prog = tmp_path / "sz.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "%s"\n'
                    'int main(){printf("%%zu %%zu %%zu %%zu %%zu\\n", sizeof(blance_problem), sizeof(blance_result),'
                    ' sizeof(blance_options), offsetof(blance_problem, vertex_parent),'
                    ' offsetof(blance_result, pass_kernel_ms));return 0;}\n' % HEADER)
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-o", str(exe), str(prog)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got == [C.sizeof(abi.Problem), C.sizeof(abi.Result), C.sizeof(abi.Options),
                   abi.Problem.vertex_parent.offset, abi.Result.pass_kernel_ms.offset]


def test_validate_on_host(lib):
    fp = synth.config_flat(3, P=64, N=64)
    assert lib.blance_validate(C.byref(fp.as_struct())) == abi.OK
    assert lib.blance_result_capacity(C.byref(fp.as_struct())) == fp.result_capacity() == 64 * 3
    bad = synth.config_flat(3, P=64, N=64)
    bad.arrays["part_order"][0] = 1                       # no longer a permutation
    bad._struct = None
    assert lib.blance_validate(C.byref(bad.as_struct())) == abi.ERR_BAD_ARG
    assert b"part_order" in lib.blance_last_error()
    big = synth.config_flat(2, P=4, N=9000)
    assert lib.blance_validate(C.byref(big.as_struct())) == abi.ERR_UNSUPPORTED
    k9 = synth.config_flat(2, P=4, N=16)
    k9.arrays["state_constraints"][1] = 9
    k9._struct = None
    assert lib.blance_validate(C.byref(k9.as_struct())) == abi.ERR_UNSUPPORTED


def test_no_device_fails_loudly(lib):
    """On a box without a GPU the product refuses to run -- there is no CPU fallback."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(hip.BlanceError) as e:
        hip.Planner(device_id=0)
    assert e.value.status in (abi.ERR_NO_DEVICE, abi.ERR_DEVICE)
