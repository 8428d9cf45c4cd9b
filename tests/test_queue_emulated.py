"""k_pass_queue (flat passes with k <= 2 on one wave64, the candidates as a sorted window over the lanes,
blance_amd/csrc/k_pass_queue.h) on the SIMT emulator against the oracle: golden cases, random instances, the weighted
rebalance shape on clusters wider than the window (overflow, rebuilds), partitions without a top priority node (the
folded row), promotions (the launch stops and k_pass_tree takes over), the lean walk switched off (every step through
the general code), and a cluster whose node weights are not powers of two."""
import os
import subprocess
import sys

import pytest

from blance_amd import hip, problem, synth
from helpers import build_from_case, edge_cases
from randgen import random_case, random_flat_wide_case
from test_simt_emulated import build_emu
from test_tree_emulated import _rebalance

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emu_lib():
    return build_emu()


def _oracle(fp):
    from oracle import loader
    return loader.plan(fp)


@pytest.mark.parametrize("lean", [True, False])
def test_golden_cases_queue(emu_lib, golden_cases, lean):
    for eager in (0, 1):                         # 1: the flat bulk driver hands sub-ranges to the queue kernel
        pl = hip.Planner(lib_path=emu_lib, chain_min_parts=eager, queue="on" if lean else "general")
        for c in golden_cases:
            fp = build_from_case(c)
            got = pl.plan(fp)
            assert got.digest() == _oracle(fp).digest(), c["source"]
            out, _ = problem.decode_result(fp, got)
            assert out == c["exp"], c["source"]
        pl.close()


def test_random_instances_queue_dense(emu_lib):
    pl = hip.Planner(lib_path=emu_lib, queue="dense")
    n = 0
    for seed in range(700, 860):
        try:
            fp = build_from_case(random_case(seed))
        except problem.Unsupported:
            continue
        got, want = pl.plan(fp), _oracle(fp)
        assert (got.digest(), got.iterations, got.n_warnings) == (want.digest(), want.iterations, want.n_warnings), seed
        n += 1
    assert n > 90
    pl.close()


@pytest.mark.parametrize("lean", [True, False])
def test_random_instances_queue(emu_lib, lean):
    pl = hip.Planner(lib_path=emu_lib, queue="on" if lean else "general")
    n = 0
    for seed in range(0, 500) if lean else range(500, 700):
        try:
            fp = build_from_case(random_case(seed))
        except problem.Unsupported:
            continue
        got, want = pl.plan(fp), _oracle(fp)
        assert (got.digest(), got.iterations, got.n_warnings) == (want.digest(), want.iterations, want.n_warnings), seed
        n += 1
    assert n > (330 if lean else 120)
    pl.close()


def test_rebalance_queue(emu_lib):
    """Config 5's ingredients (Zipf partition weights, node weights 1/2/4, stickiness, a tenth of the nodes removed
    and added) on flat clusters below and above the window's 64 entries."""
    pl = hip.Planner(lib_path=emu_lib)
    _rebalance(pl, 200, 24)
    _rebalance(pl, 300, 100)
    _rebalance(pl, 300, 300, check_stays=True)
    _rebalance(pl, 900, 200)
    pl.close()
    pl = hip.Planner(lib_path=emu_lib, queue="general")
    _rebalance(pl, 200, 90)
    pl.close()
    pl = hip.Planner(lib_path=emu_lib, queue="dense")       # every moving step scores every node (bit map, bound, matrix)
    _rebalance(pl, 300, 150)
    _rebalance(pl, 200, 24)
    pl.close()


def test_rowless_runs_queue(emu_lib):
    """Half of the nodes removed: hundreds of consecutive steps have no top priority node and share the row "" of
    nodeToNodeCounts -- whole batches of them run folded (the row in LDS, part of the window keys), mixed batches
    through the re-read path."""
    pl = hip.Planner(lib_path=emu_lib)
    _rebalance(pl, 400, 40, remove_frac=0.5, add_frac=0.3)
    _rebalance(pl, 260, 130, remove_frac=0.6, add_frac=0.1)
    _rebalance(pl, 700, 90, remove_frac=0.4, add_frac=0.4)
    pl.close()


def test_config3_rebalance_reduced_queue(emu_lib):
    """bench.py's general-regime workload (a) at 16,384 x 512: config 3's plan, every tenth node leaves -- the partitions that lost
    their primary come first in the pass and share row "": folded batches, with promotions taken by the lean walk itself
    (the new primary is sometimes a node that holds a replica of the partition, plan.go:294-297); no launch stops."""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); from blance_amd import hip, synth; from oracle import loader; "
            "pl = hip.Planner(lib_path=%r); fp = synth.config_flat(3, 16384, 512); r = pl.plan(fp); assert r.digest() == loader.plan(fp).digest(); "
            "fp2 = synth.config3_rebalance_flat(fp, r); r2 = pl.plan(fp2); w = loader.plan(fp2); "
            "assert (r2.digest(), r2.iterations) == (w.digest(), w.iterations); pl.close()" % (os.path.dirname(HERE), HERE, build_emu()))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BLANCE_QUEUE_STATS="1"), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    import re
    lines = [ln for ln in out.stderr.splitlines() if "k_pass_queue:" in ln]
    assert lines, out.stderr
    m = re.search(r"(\d+) launches, (\d+) stops, (\d+) moving steps", lines[-1])
    assert m and int(m.group(2)) == 0 and int(m.group(3)) > 1000, lines[-1]


def test_statistics_say_which_paths_ran():
    """BLANCE_QUEUE_STATS: the weighted rebalance on 300 nodes takes the lean walk for most moving steps, rebuilds its
    window, and the kernel is launched (no silent fall back to k_pass_tree)."""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); from blance_amd import hip; from test_tree_emulated import _rebalance; "
            "pl = hip.Planner(lib_path=%r); _rebalance(pl, 900, 200); pl.close()" % (os.path.dirname(HERE), HERE, build_emu()))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BLANCE_QUEUE_STATS="1"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stderr.splitlines() if "k_pass_queue:" in ln]
    assert len(lines) == 2, out.stderr
    import re
    for ln in lines:
        m = re.search(r"(\d+) launches, (\d+) stops, (\d+) moving steps \((\d+) with matrix reads, (\d+) scoring every node\), (\d+) window rebuilds", ln)
        assert m, ln
        launches, stops, moved, exact, dense, rebuilds = map(int, m.groups())
        assert launches > 0 and moved > 100 and rebuilds >= launches and exact < moved


def test_odd_node_weights_queue(emu_lib):
    """Node weights that are not powers of two (3, 5, 6): the lean walk is off for the whole launch, the general code
    divides -- same plans."""
    c = synth.rebalance_case(P=300, N=80)
    import random
    rnd = random.Random(11)
    c["nodeWeights"] = {n: rnd.choice([1, 3, 5, 6, 2]) for n in c["nodesAll"]}
    fresh = {p: {"name": p, "nodesByState": {}} for p in c["partitions"]}
    opts = dict(partition_weights=c["partitionWeights"], state_stickiness=c["stateStickiness"], node_weights=c["nodeWeights"],
                node_hierarchy=None, hierarchy_rules=None)
    pl = hip.Planner(lib_path=emu_lib)
    fp1 = problem.build_problem({}, fresh, c["oldNodes"], [], c["oldNodes"], c["model"], **opts)
    r1 = pl.plan(fp1)
    assert r1.digest() == _oracle(fp1).digest()
    plan1, _ = problem.decode_result(fp1, r1)
    fp2 = problem.build_problem(plan1, plan1, c["nodesAll"], c["nodesToRemove"], c["nodesToAdd"], c["model"], **opts)
    assert pl.plan(fp2).digest() == _oracle(fp2).digest()
    pl.close()


def test_edge_shapes_queue(emu_lib):
    cases = edge_cases()
    for mode in ("on", "general", "dense"):
        pl = hip.Planner(lib_path=emu_lib, queue=mode)
        for i, (a, k) in enumerate(cases):
            fp = problem.build_problem(*a, **k)
            got, want = pl.plan(fp), _oracle(fp)
            assert (got.digest(), got.iterations, got.n_warnings) == (want.digest(), want.iterations, want.n_warnings), i
        pl.close()


def test_reduced_configs_queue(emu_lib):
    pl = hip.Planner(lib_path=emu_lib, chain_min_parts=64)
    for fp in (synth.config_flat(1), synth.config_flat(2, P=2048, N=32), synth.config_flat(2, P=600, N=300)):
        got = pl.plan(fp)
        assert got.digest() == _oracle(fp).digest()
    pl.close()
