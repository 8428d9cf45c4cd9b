"""The periodic form of the all-blank chain pass (blance_amd/csrc/k_period.h): two periods walked, the rest of
the periodic stretch copied, whatever lies behind it walked -- bit for bit the oracle's plan, on the emulated kernels.
(The GPU counterpart is tests/test_periodic_gpu.py::test_periodic_all_blank_pass.)"""
import os
import subprocess
import sys

import pytest

from blance_amd import hip, synth
from helpers import build_from_case
from randgen import random_regular_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(8192, 256), (16384, 512), (5000, 256), (9000, 384), (4096, 4096), (6000, 300)]


def check_shapes(pl, shapes=SHAPES):
    from oracle import loader
    for P, N in shapes:
        fp = synth.config_flat(3, P=P, N=N)
        got, want = pl.plan(fp), loader.plan(fp)
        assert (got.digest(), got.iterations) == (want.digest(), want.iterations), (P, N)


TREES = [(12, 8, 2), (8, 16, 3), (16, 8, 1), (4, 4, 2), (16, 8, 4), (2, 8, 2)]      # rack, racks per zone, replicas


def check_trees(pl, P=16384, zones=4):
    """Other trees and replica counts (periods of 96, 128 and 16 steps; class masks by lane reads and by arithmetic)."""
    from oracle import loader
    for rack, rpz, k in TREES:
        N = rack * rpz * zones
        c = synth.config_case(3, P=P, N=N)
        c["nodeHierarchy"] = synth.hierarchy_names(N, rack=rack, racks_per_zone=rpz, zones_per_dc=8)
        c["model"] = {"primary": {"priority": 0, "constraints": 1}, "replica": {"priority": 1, "constraints": k}}
        fp = synth.case_to_flat(c)
        got, want = pl.plan(fp), loader.plan(fp)
        assert (got.digest(), got.iterations) == (want.digest(), want.iterations), (rack, rpz, k)


def check_weights_and_gaps(pl, P=8192, N=512):
    """One partition weight for all (copied in units of that weight), two weights (the stretch ends where the weight
    changes; the chain kernel refuses the rest and the pass is redone in full), nodes missing from the tree (period 120)."""
    from oracle import loader
    for mode in ("w3", "mixed", "gaps"):
        c = synth.config_case(3, P=P, N=N)
        if mode == "w3":
            c["partitionWeights"] = {p: 3 for p in c["partitionsToAssign"]}
        if mode == "mixed":
            c["partitionWeights"] = {p: (3 if int(p) % 2 else 1) for p in c["partitionsToAssign"]}
        if mode == "gaps":
            gone = set(c["nodesAll"][5::17])
            c["nodesAll"] = [n for n in c["nodesAll"] if n not in gone]
            c["nodesToAdd"] = list(c["nodesAll"])
        fp = synth.case_to_flat(c)
        got, want = pl.plan(fp), loader.plan(fp)
        assert (got.digest(), got.iterations) == (want.digest(), want.iterations), mode


def check_wide_regions(make_planner, P=12288):
    """Regions of 192 and 256 leaves (k_pass_chain_blank walks the segments), and the same kernel forced on a narrow tree."""
    from oracle import loader
    for planes, (rack, rpz, k, zones) in ((True, (16, 12, 2, 2)), (True, (16, 16, 3, 2)), (False, (16, 8, 2, 4)), (False, (12, 8, 2, 4))):
        pl = make_planner(planes)
        N = rack * rpz * zones
        c = synth.config_case(3, P=P, N=N)
        c["nodeHierarchy"] = synth.hierarchy_names(N, rack=rack, racks_per_zone=rpz, zones_per_dc=8)
        c["model"] = {"primary": {"priority": 0, "constraints": 1}, "replica": {"priority": 1, "constraints": k}}
        fp = synth.case_to_flat(c)
        got, want = pl.plan(fp), loader.plan(fp)
        pl.close()
        assert (got.digest(), got.iterations) == (want.digest(), want.iterations), (planes, rack, rpz, k)


def check_random(pl, seeds):
    from oracle import loader
    for seed in seeds:
        fp = build_from_case(random_regular_case(seed))
        got, want = pl.plan(fp), loader.plan(fp)
        assert (got.digest(), got.iterations, got.n_warnings) == (want.digest(), want.iterations, want.n_warnings), seed


def test_periodic_pass_equals_the_oracle():
    from test_simt_emulated import build_emu
    pl = hip.Planner(lib_path=build_emu(), chain_min_parts=8, periodic=True)
    check_shapes(pl)
    check_trees(pl)
    check_weights_and_gaps(pl)
    pl.close()
    pl = hip.Planner(lib_path=build_emu(), chain_min_parts=1, periodic=True)      # tiny chains, odd trees: mostly the ways out
    check_random(pl, range(7000, 7080))
    pl.close()


def test_periodic_pass_on_the_lane_minimum_kernel():
    from test_simt_emulated import build_emu
    check_wide_regions(lambda planes: hip.Planner(lib_path=build_emu(), chain_min_parts=8, periodic=True, planes=planes))


def test_periodic_stretch_is_taken(capfd):
    """The path is really taken (trace line), and copies most of the steps of config 3's shape."""
    from test_simt_emulated import build_emu
    code = ("import sys; sys.path.insert(0, %r); from blance_amd import hip, synth; "
            "pl = hip.Planner(lib_path=%r, chain_min_parts=8, periodic=True); pl.plan(synth.config_flat(3, P=16384, N=512)); pl.close()"
            % (ROOT, build_emu()))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BLANCE_TRACE="1"), capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stderr.splitlines() if "periodic records" in ln]
    assert line and "in 4 of 4 regions (period 128" in line[0] and "15360 of 16384 steps copied" in line[0], out.stderr[-2000:]
    # (the sweep's last pass: its flags come back with the convergence word)
    assert "all-blank kernel (planes) did the pass" in out.stderr or "deferred verdict (all-blank kernel): stands" in out.stderr


@pytest.mark.parametrize("cut", [300, 1000, 4000])
def test_chain_behind_the_periodic_stretch_is_walked(cut):
    """BLANCE_PERIODIC_CUT ends the periodic stretch early: the third segment is walked from the copied counters."""
    from test_simt_emulated import build_emu
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_periodic_emulated as T; from blance_amd import hip; "
            "pl = hip.Planner(lib_path=%r, chain_min_parts=8, periodic=True); T.check_shapes(pl, T.SHAPES[:4]); pl.close(); print('ok')"
            % (ROOT, os.path.join(ROOT, "tests"), build_emu()))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BLANCE_PERIODIC_CUT=str(cut)), capture_output=True, text=True)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-3000:]
