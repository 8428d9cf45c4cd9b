"""Kernel logic without a GPU: the *same kernel source* as the product
(blance_amd/csrc/blance_hip.hip) compiled against the SIMT emulator of
tests/simt/hip_emu.h and driven through the same C ABI, against the oracle.
This is test infrastructure only -- the product library is hipcc/gfx950 and has
no CPU path; the GPU parity tests proper are tests/test_hip_parity.py."""
import os
import subprocess

import pytest

from blance_amd import hip, problem, synth
from helpers import build_from_case
from randgen import random_case

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_SRC = os.path.join(HERE, "simt", "emu_lib.cpp")
EMU_SO = os.path.join(HERE, "simt", "_build", "libblance_emu.so")
DEPS = [EMU_SRC, os.path.join(HERE, "simt", "hip_emu.h"),
        os.path.join(HERE, "..", "blance_amd", "csrc", "blance_hip.hip"),
        os.path.join(HERE, "..", "blance_amd", "csrc", "blance_kernels.h"),
        os.path.join(HERE, "..", "include", "blance_hip.h")]


@pytest.fixture(scope="module")
def emu_lib():
    stale = (not os.path.exists(EMU_SO)) or any(os.path.getmtime(d) > os.path.getmtime(EMU_SO) for d in DEPS)
    if stale:
        os.makedirs(os.path.dirname(EMU_SO), exist_ok=True)
        subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-shared",
                               "-ffp-contract=off", "-Wno-unknown-pragmas", "-o", EMU_SO, EMU_SRC])
    return EMU_SO


def _oracle(fp):
    from oracle import loader
    return loader.plan(fp)


def test_golden_cases_wave64(emu_lib, golden_cases):
    pl = hip.Planner(lib_path=emu_lib, force_threads=64)
    for c in golden_cases:
        fp = build_from_case(c)
        got, want = pl.plan(fp), _oracle(fp)
        assert got.digest() == want.digest(), c["source"]
        out, _ = problem.decode_result(fp, got)
        assert out == c["exp"], c["source"]
    pl.close()


def test_golden_cases_multi_wave(emu_lib, golden_cases):
    pl = hip.Planner(lib_path=emu_lib, force_threads=256)
    for c in golden_cases[::4]:
        fp = build_from_case(c)
        assert pl.plan(fp).digest() == _oracle(fp).digest(), c["source"]
    pl.close()


def test_random_instances(emu_lib):
    pl = hip.Planner(lib_path=emu_lib, force_threads=64)
    n = 0
    for seed in range(0, 120):
        try:
            fp = build_from_case(random_case(seed))
        except problem.Unsupported:
            continue
        assert pl.plan(fp).digest() == _oracle(fp).digest(), seed
        n += 1
    assert n > 80
    pl.close()


def test_several_nodes_per_thread(emu_lib):
    """NX > T exercises the NPT > 1 register tiles."""
    pl = hip.Planner(lib_path=emu_lib, force_threads=64)
    for fp in (synth.config_flat(3, P=96, N=200), synth.config_flat(2, P=150, N=100)):
        assert pl.plan(fp).digest() == _oracle(fp).digest()
    pl.close()
