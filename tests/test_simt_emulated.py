"""Kernel logic without a GPU: the *same kernel source* as the product
(blance_amd/csrc/blance_hip.hip) compiled against the SIMT emulator of
tests/simt/hip_emu.h and driven through the same C ABI, against the oracle.
This is test infrastructure only -- the product library is hipcc/gfx950 and has
no CPU path; the GPU parity tests proper are tests/test_hip_parity.py."""
import os
import subprocess

import pytest

from blance_amd import hip, problem, synth
from helpers import build_from_case, edge_cases
from randgen import random_case, random_regular_case

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_SRC = os.path.join(HERE, "simt", "emu_lib.cpp")
EMU_SO = os.path.join(HERE, "simt", "_build", "libblance_emu.so")
def _deps():
    """Everything the emulator library is compiled from: its own two files, the ABI header, and EVERY source under
    blance_amd/csrc (a list kept by hand missed k_queue_walk.h once)."""
    import glob
    csrc = os.path.join(HERE, "..", "blance_amd", "csrc")
    return [EMU_SRC, os.path.join(HERE, "simt", "hip_emu.h"), os.path.join(HERE, "..", "include", "blance_hip.h")] + \
        sorted(glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.hip")))


def build_emu():
    stale = (not os.path.exists(EMU_SO)) or any(os.path.getmtime(d) > os.path.getmtime(EMU_SO) for d in _deps())
    if stale:
        os.makedirs(os.path.dirname(EMU_SO), exist_ok=True)
        subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-shared",
                               "-ffp-contract=off", "-Wno-unknown-pragmas", "-o", EMU_SO, EMU_SRC])
    return EMU_SO


@pytest.fixture(scope="module")
def emu_lib():
    return build_emu()


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
    for c in golden_cases:
        fp = build_from_case(c)
        assert pl.plan(fp).digest() == _oracle(fp).digest(), c["source"]
    pl.close()


def test_golden_cases_bulk_engines(emu_lib, golden_cases):
    """Flat bulk driver (certain stays, fresh identical runs, radix sort) and region
    chains switched on for passes of any size."""
    pl = hip.Planner(lib_path=emu_lib, force_threads=64, chain_min_parts=1)
    bulk = 0
    for c in golden_cases:
        fp = build_from_case(c)
        got = pl.plan(fp)
        assert got.digest() == _oracle(fp).digest(), c["source"]
        bulk += got.struct.steps_batched > 0
    assert bulk > 40
    pl.close()


def test_random_instances(emu_lib):
    pl = hip.Planner(lib_path=emu_lib, force_threads=64)
    n = 0
    for seed in range(0, 600):
        try:
            fp = build_from_case(random_case(seed))
        except problem.Unsupported:
            continue
        assert pl.plan(fp).digest() == _oracle(fp).digest(), seed
        n += 1
    assert n > 400
    pl.close()


def test_region_chains(emu_lib):
    """Uniform rack/zone trees: the chain kernel (general and integer-key mode,
    verified-stay speculation) and its escape to the sequential pass."""
    pl = hip.Planner(lib_path=emu_lib, force_threads=64, chain_min_parts=1)
    chains = 0
    for seed in range(0, 60):                # the GPU suite walks 800 of these (tests/test_hip_parity.py)
        try:
            fp = build_from_case(random_regular_case(seed))
        except problem.Unsupported:
            continue
        got = pl.plan(fp)
        assert got.digest() == _oracle(fp).digest(), seed
        chains += got.struct.steps_batched > 0
    assert chains >= 25
    fp = synth.config_flat(3, P=160, N=200)
    got = pl.plan(fp)
    assert got.digest() == _oracle(fp).digest() and got.struct.steps_batched > 0
    pl.close()


def test_chain_rounds_of_several_waves(emu_lib):
    """k_pass_chain's stay rounds on four waves (round 6): 256 steps tested at once, a lane adding the row bumps of the earlier
    steps of the round that share its top priority node.  Small zones make top priority nodes repeat inside a round (16 and 32
    leaves per zone: every 16th / 32nd step, also inside one wave's 64: the prefix then ends at the second one), zones of 128
    leaves make them repeat across the waves only; k_stay_by_top is off so that the converged sweeps run here too; with
    partition and node weights (no packed keys) and without."""
    from blance_amd import problem
    for rack, rpz, N, P in ((4, 4, 64, 3000), (8, 4, 128, 2500), (16, 8, 256, 4096)):
        for weighted in (False, True):
            c = synth.config_case(3, P=P, N=N)
            c["nodeHierarchy"] = synth.hierarchy_names(N, rack=rack, racks_per_zone=rpz, zones_per_dc=2)
            if weighted:
                c["partitionWeights"] = {str(i): 1 + (i * 7) % 3 for i in range(P)}
                c["nodeWeights"] = {("n%04d" % i): [1, 1, 2, 4][(i * 5) % 4] for i in range(N)}
                c["stateStickiness"] = {"primary": 100, "replica": 10}
            fp = synth.case_to_flat(c)
            want = _oracle(fp)
            for kw in (dict(stay_top="off"), dict()):
                pl = hip.Planner(lib_path=emu_lib, chain_min_parts=8, **kw)
                got = pl.plan(fp)
                assert (got.digest(), got.iterations) == (want.digest(), want.iterations), (rack, rpz, N, weighted, kw)
                assert got.struct.steps_batched > 0
                # and the rebalance from it (events, general steps between the rounds)
                fp2 = synth.config3_rebalance_flat(fp, want) if not weighted else None
                if fp2 is not None:
                    assert pl.plan(fp2).digest() == _oracle(fp2).digest(), (rack, rpz, N, "rebalance", kw)
                pl.close()


def test_stays_verified_per_top_priority_node(emu_lib):
    """k_stay_by_top tried in EVERY chain pass with NumPartitions > 0 (knob "force"): passes of stays are taken by it,
    any other pass makes it raise its flag and the chain kernel redoes the pass -- same results either way; and the
    converged sweep of config 3's shape is one it takes on its own ("auto")."""
    pl = hip.Planner(lib_path=emu_lib, force_threads=64, chain_min_parts=1, stay_top="force")
    for seed in range(100, 145):
        try:
            fp = build_from_case(random_regular_case(seed))
        except problem.Unsupported:
            continue
        got, want = pl.plan(fp), _oracle(fp)
        assert (got.digest(), got.iterations, got.n_warnings) == (want.digest(), want.iterations, want.n_warnings), seed
    for fp in (synth.config_flat(3, P=3000, N=256), synth.config_flat(3, P=900, N=300)):
        assert pl.plan(fp).digest() == _oracle(fp).digest()
    pl.close()
    for mode in ("auto", "off"):
        pl = hip.Planner(lib_path=emu_lib, chain_min_parts=8, stay_top=mode)
        fp = synth.config_flat(3, P=4096, N=256)
        assert pl.plan(fp).digest() == _oracle(fp).digest()
        pl.close()


def test_random_instances_bulk_engines(emu_lib):
    pl = hip.Planner(lib_path=emu_lib, force_threads=64, chain_min_parts=1)
    n = bulk = 0
    for seed in range(600, 700):             # the GPU suite walks 1,000 of these
        try:
            fp = build_from_case(random_case(seed))
        except problem.Unsupported:
            continue
        got = pl.plan(fp)
        assert got.digest() == _oracle(fp).digest(), seed
        n += 1
        bulk += got.struct.steps_batched > 0
    assert n > 60 and bulk > 12
    pl.close()


def test_multi_wave_random(emu_lib):
    pl = hip.Planner(lib_path=emu_lib, force_threads=256, chain_min_parts=1)
    for seed in range(2000, 2060):
        try:
            fp = build_from_case(random_case(seed, max_nodes=40, max_parts=40))
        except problem.Unsupported:
            continue
        assert pl.plan(fp).digest() == _oracle(fp).digest(), seed
    pl.close()


def test_reduced_configs(emu_lib):
    """BASELINE.json shapes at a size the emulator finishes in seconds."""
    pl = hip.Planner(lib_path=emu_lib, chain_min_parts=64)
    for fp in (synth.config_flat(1), synth.config_flat(2, P=2048, N=32), synth.config_flat(3, P=1024, N=256),
               synth.config_flat(3, P=700, N=300)):
        got = pl.plan(fp)
        assert got.digest() == _oracle(fp).digest()
        assert got.struct.steps_batched > 0
    pl.close()


def test_rebalance_miniature(emu_lib):
    """Config 5's shape: weights, node weights, stickiness, remove + add nodes."""
    pl = hip.Planner(lib_path=emu_lib, chain_min_parts=8)
    for hierarchy, N in ((False, 24), (True, 64)):
        c = synth.rebalance_case(P=300, N=N, hierarchy=hierarchy)
        fresh = {p: {"name": p, "nodesByState": {}} for p in c["partitions"]}
        opts = dict(partition_weights=c["partitionWeights"], state_stickiness=c["stateStickiness"],
                    node_weights=c["nodeWeights"], node_hierarchy=c["nodeHierarchy"],
                    hierarchy_rules=c["hierarchyRules"])
        fp1 = problem.build_problem({}, fresh, c["oldNodes"], [], c["oldNodes"], c["model"], **opts)
        r1 = pl.plan(fp1)
        assert r1.digest() == _oracle(fp1).digest()
        plan1, _ = problem.decode_result(fp1, r1)
        fp2 = problem.build_problem(plan1, plan1, c["nodesAll"], c["nodesToRemove"], c["nodesToAdd"], c["model"], **opts)
        assert pl.plan(fp2).digest() == _oracle(fp2).digest()
    pl.close()


def test_sequential_pass_speculation(emu_lib):
    """k_pass_seq on a flat cluster too wide for one wave64 (300 nodes): verified stays are
    committed in batches, the other steps one by one; bit-exact with and without."""
    c = synth.rebalance_case(P=300, N=300, hierarchy=False)
    fresh = {p: {"name": p, "nodesByState": {}} for p in c["partitions"]}
    opts = dict(partition_weights=c["partitionWeights"], state_stickiness=c["stateStickiness"],
                node_weights=c["nodeWeights"], node_hierarchy=None, hierarchy_rules=None)
    fp1 = problem.build_problem({}, fresh, c["oldNodes"], [], c["oldNodes"], c["model"], **opts)
    digests = []
    for spec in (True, False):
        pl = hip.Planner(lib_path=emu_lib, seq_speculation=spec, tree="off")
        r1 = pl.plan(fp1)
        assert r1.digest() == _oracle(fp1).digest()
        plan1, _ = problem.decode_result(fp1, r1)
        fp2 = problem.build_problem(plan1, plan1, c["nodesAll"], c["nodesToRemove"], c["nodesToAdd"], c["model"], **opts)
        r2 = pl.plan(fp2)
        assert r2.digest() == _oracle(fp2).digest()
        assert (r2.struct.steps_batched > 300) == spec       # beyond the primary pass's bulk stays
        digests.append(r2.digest())
        pl.close()
    assert digests[0] == digests[1]


def test_wide_hierarchy_regions(emu_lib):
    """Regions of 320 leaves: 5 leaves per lane of the region's wave64."""
    c = synth.config_case(3, P=300, N=700)
    c["nodeHierarchy"] = synth.hierarchy_names(700, rack=16, racks_per_zone=20, zones_per_dc=2)
    fp = synth.case_to_flat(c)
    pl = hip.Planner(lib_path=emu_lib, chain_min_parts=8)
    got = pl.plan(fp)
    assert got.digest() == _oracle(fp).digest()
    assert got.struct.steps_batched > 0
    pl.close()


def test_edge_shapes(emu_lib):
    """Degenerate and unusual inputs: nothing to plan, no nodes, iteration caps, constraints the
    cluster cannot meet, 8 copies, three states, names that occur only in prevMap, rules with
    exclude level 0, two rules for one state, rules on the top priority state."""
    cases = edge_cases()
    for eager in (0, 1):
        pl = hip.Planner(lib_path=emu_lib, chain_min_parts=eager)
        for i, (a, k) in enumerate(cases):
            fp = problem.build_problem(*a, **k)
            got, want = pl.plan(fp), _oracle(fp)
            assert (got.digest(), got.iterations, got.n_warnings) == (want.digest(), want.iterations, want.n_warnings), i
        pl.close()


def test_one_context_from_several_threads(emu_lib):
    """calls on one blance_ctx are serialised by the library; goroutines
    (here: Python threads, ctypes releases the GIL during the call) may share it."""
    import threading
    pl = hip.Planner(lib_path=emu_lib, chain_min_parts=8)
    fps = [synth.config_flat(3, P=96 + 8 * i, N=64) for i in range(4)] + [synth.config_flat(2, P=200 + i, N=16) for i in range(4)]
    want = [_oracle(fp).digest() for fp in fps]
    got = [None] * len(fps)

    def work(i):
        for _ in range(3):
            got[i] = pl.plan(fps[i]).digest()
    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(fps))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert got == want
    pl.close()


def test_eight_waves(emu_lib):
    """512 threads: the second argmin stage reads every wave's slot twice in a DPP row of 16."""
    pl = hip.Planner(lib_path=emu_lib, force_threads=512)
    for fp in (synth.config_flat(3, P=40, N=300), synth.config_flat(2, P=60, N=900)):
        assert pl.plan(fp).digest() == _oracle(fp).digest()
    pl.close()


def test_several_nodes_per_thread(emu_lib):
    """NX > T exercises the NPT > 1 register tiles."""
    pl = hip.Planner(lib_path=emu_lib, force_threads=64)
    for fp in (synth.config_flat(3, P=96, N=200), synth.config_flat(2, P=150, N=100)):
        assert pl.plan(fp).digest() == _oracle(fp).digest()
    pl.close()


def test_fresh_runs_with_excluded_node(emu_lib):
    """k_fresh_excl: a fresh plan's replica pass (NumPartitions == 0) where every partition only excludes
    its primary -- skewed primaries give long waits of the excluded node at the head of the queue, few
    nodes make it come up again while it waits (the run is cut there), node weights bend the sequence."""
    import random
    model = {"primary": {"priority": 0, "constraints": 1}, "replica": {"priority": 1, "constraints": 1}}
    pl = hip.Planner(lib_path=emu_lib, chain_min_parts=1)
    bulk = 0
    for seed in range(24):
        rnd = random.Random(seed)
        n = rnd.choice([2, 3, 5, 9, 40, 70])
        nodes = ["n%02d" % i for i in range(n)]
        hot = nodes[: rnd.choice([1, 2, n])]
        parts = {}
        for i in range(rnd.choice([30, 200, 700])):
            prim = rnd.choice(hot) if rnd.random() < 0.8 else rnd.choice(nodes)
            nbs = {"primary": [prim]} if rnd.random() < 0.95 else {}
            parts[str(i)] = {"name": str(i), "nodesByState": nbs}
        kw = {}
        if seed % 3 == 0:
            kw["node_weights"] = {x: rnd.choice([1, 2, 3]) for x in nodes}
        fp = problem.build_problem({}, parts, nodes, [], nodes, model, max_iterations=rnd.choice([1, 10]), **kw)
        got, want = pl.plan(fp), _oracle(fp)
        assert (got.digest(), got.iterations, got.n_warnings) == (want.digest(), want.iterations, want.n_warnings), seed
        bulk += got.struct.steps_batched > 0
    assert bulk > 10
    pl.close()


def test_fresh_runs_two_picks(emu_lib):
    """k_fresh_excl with two picks per step: a fresh plan's pass for a state with two copies (NumPartitions == 0),
    with and without an excluded node per step; few nodes force cuts (a node would be taken twice in a step, the
    pending node comes up again)."""
    import random
    pl = hip.Planner(lib_path=emu_lib, chain_min_parts=1)
    bulk = 0
    for seed in range(18):
        rnd = random.Random(1000 + seed)
        n = rnd.choice([2, 3, 4, 7, 30, 70])
        nodes = ["n%02d" % i for i in range(n)]
        if seed % 2:
            model = {"primary": {"priority": 0, "constraints": 1}, "replica": {"priority": 1, "constraints": 2}}
        else:
            model = {"primary": {"priority": 0, "constraints": 2}}
        hot = nodes[: rnd.choice([1, 2, n])]
        parts = {}
        for i in range(rnd.choice([30, 200, 400])):
            nbs = {}
            if seed % 4 == 1:
                nbs = {"primary": [rnd.choice(hot) if rnd.random() < 0.7 else rnd.choice(nodes)]}
            parts[str(i)] = {"name": str(i), "nodesByState": nbs}
        kw = {}
        if seed % 3 == 0:
            kw["node_weights"] = {x: rnd.choice([1, 2, 3]) for x in nodes}
        fp = problem.build_problem({}, parts, nodes, [], nodes, model, max_iterations=rnd.choice([1, 10]), **kw)
        got, want = pl.plan(fp), _oracle(fp)
        assert (got.digest(), got.iterations, got.n_warnings) == (want.digest(), want.iterations, want.n_warnings), seed
        bulk += got.struct.steps_batched > 0
    assert bulk > 9
    for fp in (synth.config5_initial(400, 60), synth.config5_initial(900, 30)):
        assert pl.plan(fp).digest() == _oracle(fp).digest()
    pl.close()


def test_plan_from_nothing_opening_pass(emu_lib, monkeypatch):
    """A plan from nothing (round 6): the opening pass needs no scan and no sort -- the empty cluster's greedy plan is a round
    robin over nodesNext by id.  Shapes that bend the closed form: nodes removed up front (nodesNext is not every node),
    partition counts that are no multiple of the node count, one and two picks per step, flat and hierarchical, extra states;
    with the host's shortcuts on and off, against the oracle."""
    shapes = []
    for cfg, P, N in ((2, 3000, 37), (2, 2048, 64), (3, 2500, 96), (3, 4096, 128)):
        c = synth.config_case(cfg, P=P, N=N)
        shapes.append((cfg, P, N, "all", c))
        # (nodesToRemove needs the partitions in prevMap, plan.go:545: there, holding nothing -- NumPartitions > 0 then)
        c2 = synth.config_case(cfg, P=P, N=N)
        c2["prevMap"] = c2["partitionsToAssign"]
        c2["aliased"] = True
        c2["nodesToRemove"] = [n for i, n in enumerate(c2["nodesAll"]) if i % 7 == 3]
        c2["nodesToAdd"] = [n for n in c2["nodesAll"] if n not in set(c2["nodesToRemove"])]
        shapes.append((cfg, P, N, "removed", c2))
    for cfg, P, N, what, c in shapes:
        fp = synth.case_to_flat(c)
        want = _oracle(fp)
        for spec in ("1", "0"):
            monkeypatch.setenv("BLANCE_SPECULATE", spec)
            pl = hip.Planner(lib_path=emu_lib, chain_min_parts=64)
            got = pl.plan(fp)
            assert (got.digest(), got.iterations) == (want.digest(), want.iterations), (cfg, P, N, what, spec)
            if spec == "1":
                fewer = got.struct.host_syncs
            else:
                assert fewer < got.struct.host_syncs, (cfg, P, N, what, fewer, got.struct.host_syncs)
            pl.close()
