"""GPU parity tests proper: the hand-written HIP planner, called through the C
ABI (blance_plan), against the CPU oracle on the same inputs -- bit exact
(node ids, list kinds, warnings, sweep count)."""
import numpy as np
import pytest

from blance_amd import hip, problem, synth
from helpers import build_from_case, edge_cases
from randgen import random_case, random_regular_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def planner():
    pl = hip.Planner(device_id=0)
    yield pl
    pl.close()


@pytest.fixture(scope="module")
def eager_planner():
    """Bulk engines (region chains, flat stay / fresh runs) switched on for passes
    of any size, so small random inputs exercise them and their fallbacks."""
    pl = hip.Planner(device_id=0, chain_min_parts=1)
    yield pl
    pl.close()


def _oracle(fp):
    from oracle import loader
    return loader.plan(fp)


def _same(got, want, tag):
    assert got.iterations == want.iterations, tag
    assert got.converged == want.converged, tag
    assert got.warnings() == want.warnings(), tag
    assert got.digest() == want.digest(), tag


def test_golden_cases(planner, golden_cases):
    """The reference's own 69 end-to-end tables (plan_test.go, control_test.go)."""
    from oracle import blance_ref as R
    for c in golden_cases:
        fp = build_from_case(c)
        got = planner.plan(fp)
        out, w = problem.decode_result(fp, got)
        assert out == c["exp"], (c["suite"], c["index"], c["source"])
        assert R.count_warnings(w, c["warningCountMode"]) == c["expNumWarnings"], c["source"]
        _same(got, _oracle(fp), c["source"])


@pytest.mark.parametrize("block", range(4))
def test_random_instances(planner, block):
    n = 0
    for seed in range(block * 250, (block + 1) * 250):
        try:
            fp = build_from_case(random_case(seed))
        except problem.Unsupported:
            continue
        _same(planner.plan(fp), _oracle(fp), seed)
        n += 1
    assert n > 150


@pytest.mark.parametrize("block", range(4))
def test_random_instances_bulk_engines(eager_planner, block):
    n = bulk = 0
    for seed in range(block * 250, (block + 1) * 250):
        try:
            fp = build_from_case(random_case(seed))
        except problem.Unsupported:
            continue
        got = eager_planner.plan(fp)
        _same(got, _oracle(fp), seed)
        n += 1
        bulk += got.struct.steps_batched > 0
    assert n > 150 and bulk > 20


@pytest.mark.parametrize("block", range(4))
def test_regular_hierarchies_region_chains(eager_planner, block):
    n = bulk = 0
    for seed in range(block * 200, (block + 1) * 200):
        try:
            fp = build_from_case(random_regular_case(seed))
        except problem.Unsupported:
            continue
        got = eager_planner.plan(fp)
        _same(got, _oracle(fp), seed)
        n += 1
        bulk += got.struct.steps_batched > 0
    assert n > 150 and bulk > 40


def test_golden_cases_bulk_engines(eager_planner, golden_cases):
    for c in golden_cases:
        fp = build_from_case(c)
        _same(eager_planner.plan(fp), _oracle(fp), c["source"])


def test_sequential_engine_option(golden_cases):
    """BLANCE_ENGINE_SEQUENTIAL: every step one at a time, same answer."""
    from blance_amd import abi
    pl = hip.Planner(device_id=0, engine=abi.ENGINE_SEQUENTIAL, chain_min_parts=1)
    for c in golden_cases[::2]:
        fp = build_from_case(c)
        got = pl.plan(fp)
        _same(got, _oracle(fp), c["source"])
        assert got.struct.steps_batched == 0
    fp = synth.config_flat(3, P=3000, N=512)
    _same(pl.plan(fp), _oracle(fp), "cfg3 sequential")
    pl.close()


def test_random_larger_instances(planner):
    for seed in range(5000, 5040):
        try:
            fp = build_from_case(random_case(seed, max_nodes=300, max_parts=400))
        except problem.Unsupported:
            continue
        _same(planner.plan(fp), _oracle(fp), seed)


@pytest.mark.parametrize("threads", [64, 256, 512, 1024])
def test_workgroup_shapes(threads, golden_cases):
    """Every workgroup shape of the pass kernel gives the same answer."""
    pl = hip.Planner(device_id=0, force_threads=threads)
    for c in golden_cases[::3]:
        fp = build_from_case(c)
        _same(pl.plan(fp), _oracle(fp), c["source"])
    for seed in range(40):
        try:
            fp = build_from_case(random_case(seed, max_nodes=100, max_parts=60))
        except problem.Unsupported:
            continue
        _same(pl.plan(fp), _oracle(fp), seed)
    pl.close()


def test_config1(planner):
    fp = synth.config_flat(1)
    got = planner.plan(fp)
    _same(got, _oracle(fp), "cfg1")
    # partition i -> node i mod 4, two sweeps
    assert got.iterations == 2
    assert got.out_nodes[:64].tolist() == [i % 4 for i in range(64)]


def test_config2_full(planner):
    fp = synth.config_flat(2)
    got = planner.plan(fp)
    _same(got, _oracle(fp), "cfg2")
    prim = got.out_nodes[0::2][:65536]
    rep = got.out_nodes[1::2][:65536]
    i = np.arange(65536)
    assert np.array_equal(prim, i % 256) and np.array_equal(rep, (i % 256) ^ 1)


@pytest.mark.parametrize("P,N", [(4096, 4096), (16384, 1024), (20000, 777)])
def test_config3_reduced(planner, P, N):
    fp = synth.config_flat(3, P=P, N=N)
    _same(planner.plan(fp), _oracle(fp), ("cfg3", P, N))


def _rebalance(planner_obj, P, N, hierarchy):
    """Config 5 in miniature: plan over the old nodes, then rebalance with 10 % of the
    nodes removed and 10 % added; both calls bit-exact against the oracle."""
    c = synth.rebalance_case(P=P, N=N, hierarchy=hierarchy)
    fresh = {p: {"name": p, "nodesByState": {}} for p in c["partitions"]}
    opts = dict(partition_weights=c["partitionWeights"], state_stickiness=c["stateStickiness"],
                node_weights=c["nodeWeights"], node_hierarchy=c["nodeHierarchy"],
                hierarchy_rules=c["hierarchyRules"])
    fp1 = problem.build_problem({}, fresh, c["oldNodes"], [], c["oldNodes"], c["model"], **opts)
    r1 = planner_obj.plan(fp1)
    _same(r1, _oracle(fp1), ("initial plan", P, N))
    plan1, _ = problem.decode_result(fp1, r1)
    fp2 = problem.build_problem(plan1, plan1, c["nodesAll"], c["nodesToRemove"], c["nodesToAdd"], c["model"], **opts)
    r2 = planner_obj.plan(fp2)
    _same(r2, _oracle(fp2), ("rebalance", P, N))
    plan2, _ = problem.decode_result(fp2, r2)
    gone = set(c["nodesToRemove"])
    assert not any(n in gone for p in plan2.values() for lst in p["nodesByState"].values() for n in lst)
    return r1, r2


def test_config5_miniature_flat(planner):
    r1, r2 = _rebalance(planner, P=6000, N=96, hierarchy=False)
    assert r1.iterations >= 2 and r2.iterations >= 2


def test_config5_miniature_hierarchy(planner):
    _rebalance(planner, P=6000, N=256, hierarchy=True)


@pytest.mark.parametrize("N", [600, 1500, 4096, 5000])
def test_config5_wide_flat_cluster(planner, N):
    """Flat clusters beyond one wave64's registers: up to 4,096 names the tournament-tree pass
    (k_pass_tree), beyond that the workgroup pass (k_pass_seq) with verified-stay speculation;
    weights and stickiness.  Same answer from k_pass_seq with and without its speculation."""
    r1, r2 = _rebalance(planner, P=5000, N=N, hierarchy=False)
    assert r2.struct.steps_batched > 0
    for kw in (dict(tree="off"), dict(tree="off", seq_speculation=False)):
        plain = hip.Planner(device_id=0, **kw)
        p1, p2 = _rebalance(plain, P=5000, N=N, hierarchy=False)
        plain.close()
        assert (p1.digest(), p2.digest()) == (r1.digest(), r2.digest())


def test_tree_pass_dense_mode(golden_cases):
    """k_pass_tree with its candidate walk switched off: every general step scores all nodes."""
    pl = hip.Planner(device_id=0, tree="dense")
    for c in golden_cases:
        fp = build_from_case(c)
        _same(pl.plan(fp), _oracle(fp), c["source"])
    for seed in range(300):
        try:
            fp = build_from_case(random_case(seed))
        except problem.Unsupported:
            continue
        _same(pl.plan(fp), _oracle(fp), seed)
    _rebalance(pl, P=3000, N=700, hierarchy=False)
    pl.close()


@pytest.mark.parametrize("queue", ["on", "lean-cpp", "general"])
def test_rowless_runs_folded_row(queue):
    """Half of the nodes removed: hundreds of consecutive steps have no top priority node and share row "" of
    nodeToNodeCounts; whole batches of them run with that row folded into the window keys -- through the assembly walk
    (k_queue_walk.h, bit 25), through its C++ twin, and with the lean walk off (the launch then stops and k_pass_tree takes
    the stretch).  Promotions inside such runs are taken by the lean walk itself."""
    from test_tree_emulated import _rebalance as rebalance_case
    pl = hip.Planner(device_id=0, queue=queue)
    rebalance_case(pl, 400, 40, remove_frac=0.5, add_frac=0.3)
    rebalance_case(pl, 260, 130, remove_frac=0.6, add_frac=0.1)
    rebalance_case(pl, 700, 90, remove_frac=0.4, add_frac=0.4)
    rebalance_case(pl, 5000, 600, remove_frac=0.3, add_frac=0.0)
    pl.close()


@pytest.mark.parametrize("P,N", [(16384, 512), (65536, 1024)])
def test_config3_rebalance_reduced(planner, P, N):
    """bench.py's general-regime workload (a) at reduced size: config 3's plan, every tenth node leaves.  The partitions that
    lost their primary come first in the pass (plan.go:542-561) and have no top priority node: a run of folded batches with
    promotions (the new primary is sometimes a node that holds a replica, plan.go:294-297)."""
    fp = synth.config_flat(3, P, N)
    res = planner.plan(fp)
    _same(res, _oracle(fp), ("config3", P, N))
    fp2 = synth.config3_rebalance_flat(fp, res)
    _same(planner.plan(fp2), _oracle(fp2), ("config3 rebalance", P, N))


@pytest.mark.parametrize("P,N", [(30000, 1000), (20000, 4096), (50000, 300)])
def test_tree_pass_flat_weighted(planner, P, N):
    """Larger weighted flat instances (config 5's generator at reduced size): initial plan and rebalance."""
    fp1 = synth.config5_initial(P, N)
    r1 = planner.plan(fp1)
    _same(r1, _oracle(fp1), ("config5 initial", P, N))
    fp2 = synth.config5_rebalance(fp1, r1, P, N)
    _same(planner.plan(fp2), _oracle(fp2), ("config5 rebalance", P, N))


def test_wide_hierarchy_regions(planner):
    """Zones of 320 nodes: 5 leaves per lane of the region's wave64."""
    c = synth.config_case(3, P=6000, N=1300)
    c["nodeHierarchy"] = synth.hierarchy_names(1300, rack=16, racks_per_zone=20, zones_per_dc=2)
    fp = synth.case_to_flat(c)
    got = planner.plan(fp)
    _same(got, _oracle(fp), "zones of 320")
    assert got.struct.steps_batched > 0


def _golden_digests():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config_digests.json")) as f:
        return json.load(f)


def test_config3_full_size_digest(planner):
    """BASELINE.json's headline configuration at its full size (1,048,576 x 4,096): the result's
    SHA-256 equals the CPU oracle's (tests/tools/make_config_digests.py, 141 s on one core)."""
    want = _golden_digests()["config3"]
    got = planner.plan(synth.config_flat(3))
    assert (got.iterations, got.n_warnings) == (want["iterations"], want["warnings"])
    assert got.digest() == want["digest"]


def test_config5_full_size_digest(planner):
    """Config 5 at its full size: weighted plan over the old nodes, then the rebalance from it
    (10 sweeps each, about 70 s of device time); digests from tests/tools/make_config5_digest.py
    (the CPU oracle needs 8 minutes for each)."""
    want = _golden_digests()["config5"]
    P, N = want["partitions"], want["nodes"]
    fp1 = synth.config5_initial(P, N)
    r1 = planner.plan(fp1)
    assert (r1.iterations, r1.digest()) == (want["initial"]["iterations"], want["initial"]["digest"])
    fp2 = synth.config5_rebalance(fp1, r1, P, N)
    r2 = planner.plan(fp2)
    assert (r2.iterations, r2.digest()) == (want["rebalance"]["iterations"], want["rebalance"]["digest"])
    # and what holds at any size without an oracle (tests/properties.py): list lengths, no removed node, no node twice
    # in a partition, the weighted load of every state adds up
    import properties
    properties.plan_properties(fp1, r1)
    properties.plan_properties(fp2, r2)


def test_edge_shapes(planner, eager_planner):
    """Nothing to plan, no nodes, iteration caps, unmet constraints, 8 copies, three states, names only
    in prevMap, exclude level 0, two rules for a state, rules on the top priority state."""
    for pl in (planner, eager_planner):
        for i, (a, k) in enumerate(edge_cases()):
            fp = problem.build_problem(*a, **k)
            got, want = pl.plan(fp), _oracle(fp)
            assert (got.digest(), got.iterations, got.n_warnings) == (want.digest(), want.iterations, want.n_warnings), i


def test_resident_replan_is_deterministic(planner):
    fp = synth.config_flat(3, P=8192, N=512)
    planner.upload(fp)
    planner.plan_resident()
    a = planner.download().digest()
    planner.plan_resident()
    b = planner.download().digest()
    assert a == b == _oracle(fp).digest()


def test_unsupported_is_refused(planner):
    fp = synth.config_flat(2, P=16, N=9000)          # beyond the register-resident pass
    with pytest.raises(hip.BlanceError) as e:
        planner.plan(fp)
    assert e.value.status == -2


def test_widest_clusters_8192_nodes(planner):
    """The register-resident workgroup pass at its widest (k_pass_seq<1024, 8>: 8,192 node names): hierarchy
    states it has to walk itself (two rules for the state, or the sequential engine), and a flat weighted
    rebalance beyond k_pass_tree's 4,096 names."""
    from blance_amd import abi
    c = synth.config_case(3, P=1200, N=8192)
    fp = synth.case_to_flat(c)
    _same(planner.plan(fp), _oracle(fp), "config 3 shape at 8192 nodes")
    seq = hip.Planner(device_id=0, engine=abi.ENGINE_SEQUENTIAL)
    _same(seq.plan(fp), _oracle(fp), "sequential engine at 8192 nodes")
    seq.close()
    c["hierarchyRules"] = {"replica": [{"includeLevel": 2, "excludeLevel": 1}, {"includeLevel": 3, "excludeLevel": 2}]}
    fp2 = synth.case_to_flat(c)
    _same(planner.plan(fp2), _oracle(fp2), "two rules for the state at 8192 nodes")
    _rebalance(planner, P=2500, N=8192, hierarchy=False)


def test_region_chain_envelope_edges(planner):
    """Passes just outside the region chains' envelope fall back to the exact workgroup pass: k = 5 copies,
    regions of more than 512 leaves."""
    c = synth.config_case(3, P=2500, N=1024)
    c["model"] = {"primary": {"priority": 0, "constraints": 1}, "replica": {"priority": 1, "constraints": 5}}
    fp = synth.case_to_flat(c)
    _same(planner.plan(fp), _oracle(fp), "k = 5")
    c = synth.config_case(3, P=2500, N=1400)
    c["nodeHierarchy"] = synth.hierarchy_names(1400, rack=16, racks_per_zone=40, zones_per_dc=2)   # zones of 640 leaves
    fp = synth.case_to_flat(c)
    _same(planner.plan(fp), _oracle(fp), "regions of 640 leaves")


@pytest.mark.parametrize("which", ["planner", "eager_planner"])
def test_general_regime_b_reduced(request, which):
    """bench.py's general regime (b) in pytest: config 3's model, tree and rule {2,1} with scrambled
    NON-NUMERIC partition names (plan.go:525-528: the raw name is the key) and Zipf partition weights, 65,536 x 1,024 --
    through the string interning (config3_named_weighted_case -> problem.build_problem) AND as the flat generator bench.py
    uses; both must be the same problem and give the oracle's result (10 sweeps, not converged: weighted chain passes with
    general steps + the weighted flat primary pass on k_pass_queue in its tie regime)."""
    pl = request.getfixturevalue(which)
    P, N = 65536, 1024
    fp = synth.config3_named_weighted_flat(P, N)
    want = _oracle(fp)
    assert want.iterations == 10 and not want.converged
    got = pl.plan(fp)
    _same(got, want, "regime (b), flat generator")
    assert got.struct.steps_batched > 0
    fps = synth.case_to_flat(synth.config3_named_weighted_case(P, N))
    for n in ("part_order", "part_weight", "part_has_weight", "vertex_leaf_lo", "vertex_leaf_hi", "node_leaf_pos"):
        assert np.array_equal(fps.arrays[n], fp.arrays[n]), n
    _same(pl.plan(fps), want, "regime (b), through the interning layer")


@pytest.mark.parametrize("which", ["planner", "eager_planner"])
def test_config3_with_node_weights_reduced(request, which):
    """Config 3 with NodeWeights in {1, 1, 2, 4} on its hierarchy (plan.go:675-679: the score divided by the weight -- no
    packed keys, no plane automaton: k_pass_chain's general path with weights), 65,536 x 1,024: the fresh plan, and the
    rebalance of it after every tenth node left (events + weighted folded batches of k_pass_queue)."""
    import literal_cases as L
    pl = request.getfixturevalue(which)
    P, N = 65536, 1024
    c = synth.config_case(3, P=P, N=N)
    c["nodeWeights"] = L.node_weights_1124(N)
    fp = synth.case_to_flat(c)
    want = _oracle(fp)
    got = pl.plan(fp)
    _same(got, want, "config 3 + node weights")
    fp2 = synth.config3_rebalance_flat(fp, want)
    _same(pl.plan(fp2), _oracle(fp2), "config 3 + node weights, rebalanced")


def test_rccl_communicator_of_one_rank(planner):
    """blance_comm_unique_id / blance_comm_init_rccl on the device (librccl.so bound at run time): a communicator of one
    rank, after which plans run as before (the multi-rank exchange is covered on gloo by tests/test_dist.py)."""
    class OneRank:
        @staticmethod
        def get_rank():
            return 0

        @staticmethod
        def get_world_size():
            return 1

        @staticmethod
        def broadcast_object_list(box, src=0):
            return None
    pl = hip.Planner(device_id=0)
    assert pl.comm_init_rccl(OneRank) == 1
    fp = synth.config_flat(3, P=4096, N=512)
    _same(pl.plan(fp), _oracle(fp), "after comm_init_rccl")
    pl.close()


def test_rccl_one_rank_runs_both_collectives():
    """The RCCL transport itself on the one GPU there is: with Planner(shard_one_rank=True) a communicator of one rank takes
    the sharded branch of every chain pass, so ncclAllReduce (collective A: [flags | load-vector change]) and ncclAllGather
    (collective B: the output slice) really execute on the planner's stream -- dlsym'd signatures, enum values, in-place
    semantics, stream ordering, the event pairs of blance_comm_time_ms.  Config 3's generator at 65,536 x 4,096 (32 regions):
    the oracle's result, two collectives per chain pass (= per sweep), device time inside them > 0."""
    fp = synth.config_flat(3, P=65536, N=4096)
    want = _oracle(fp)
    pl = hip.Planner(device_id=0, shard_one_rank=True)
    assert pl.comm_init_rccl_one_rank() == 1
    for rep in range(2):
        calls0, words0 = pl.comm_stats()
        ms0 = pl.comm_time_ms()
        got = pl.plan(fp)
        _same(got, want, ("one-rank RCCL plan", rep))
        calls1, words1 = pl.comm_stats()
        assert calls1 - calls0 == 2 * got.iterations, (calls0, calls1, got.iterations)
        # A: 16 header words + (M + 1) x NX loads; B: the whole output (one slice), 1 + k words per step
        a_words = 16 + (fp.n_states + 1) * fp.n_nodes_ext
        assert words1 - words0 == got.iterations * (a_words + fp.n_parts * 3), (words1 - words0, a_words)
        assert pl.comm_time_ms() > ms0
    per_call_us = pl.comm_time_ms() * 1e3 / pl.comm_stats()[0]
    assert 0.5 < per_call_us < 50000, per_call_us
    pl.close()
    # a weighted hierarchical plan and its rebalance (events: nodes outside their partition's region) the same way
    c = synth.rebalance_case(P=20000, N=1024, hierarchy=True)
    fresh = {p: {"name": p, "nodesByState": {}} for p in c["partitions"]}
    opts = dict(partition_weights=c["partitionWeights"], state_stickiness=c["stateStickiness"],
                node_weights=c["nodeWeights"], node_hierarchy=c["nodeHierarchy"], hierarchy_rules=c["hierarchyRules"])
    fp1 = problem.build_problem({}, fresh, c["oldNodes"], [], c["oldNodes"], c["model"], **opts)
    w1 = _oracle(fp1)
    plan1, _ = problem.decode_result(fp1, w1)
    fp2 = problem.build_problem(plan1, plan1, c["nodesAll"], c["nodesToRemove"], c["nodesToAdd"], c["model"], **opts)
    pl = hip.Planner(device_id=0, shard_one_rank=True)
    pl.comm_init_rccl_one_rank()
    _same(pl.plan(fp1), w1, "weighted hierarchical plan over one-rank RCCL")
    n1 = pl.comm_stats()[0]
    _same(pl.plan(fp2), _oracle(fp2), "its rebalance over one-rank RCCL")
    assert n1 > 0 and pl.comm_stats()[0] > n1
    pl.close()


@pytest.mark.parametrize("G", [2, 4, 8])
def test_sharded_plan_contexts_on_one_gpu(G):
    """BASELINE.json config 4 on a one-GPU box: G ranks = G contexts on this device, each walking the region
    chains of its slice (region_base > 0), the load-vector change summed and the output slices gathered by
    the embedder's collectives (blance_amd.dist_util.LocalGroup stages them through the host).  Every rank's
    result at config 3's FULL size carries the oracle's digest; a weighted hierarchical plan and the
    rebalance from it (events: nodes outside their partition's region) as well."""
    from blance_amd import dist_util
    want = _golden_digests()["config3"]
    fp3 = synth.config_flat(3)
    c = synth.rebalance_case(P=20000, N=1024, hierarchy=True)             # 8 zones
    fresh = {p: {"name": p, "nodesByState": {}} for p in c["partitions"]}
    opts = dict(partition_weights=c["partitionWeights"], state_stickiness=c["stateStickiness"],
                node_weights=c["nodeWeights"], node_hierarchy=c["nodeHierarchy"], hierarchy_rules=c["hierarchyRules"])
    fp1 = problem.build_problem({}, fresh, c["oldNodes"], [], c["oldNodes"], c["model"], **opts)
    w1 = _oracle(fp1)
    plan1, _ = problem.decode_result(fp1, w1)
    fp2 = problem.build_problem(plan1, plan1, c["nodesAll"], c["nodesToRemove"], c["nodesToAdd"], c["model"], **opts)
    w2 = _oracle(fp2)
    grp, planners = dist_util.local_sharded_planners(G, lambda: hip.Planner(device_id=0))

    def work(rank, pl):
        r3 = pl.plan(fp3)
        n3 = pl.comm_stats()[0]
        r1 = pl.plan(fp1)
        n1 = pl.comm_stats()[0]
        r2 = pl.plan(fp2)
        return r3, r1, r2, n3, n1, pl.comm_stats()[0]
    res = grp.run(planners, work)
    for r3, r1, r2, n3, n1, n2 in res:
        assert (r3.iterations, r3.n_warnings, r3.digest()) == (want["iterations"], want["warnings"], want["digest"]), G
        _same(r1, w1, ("weighted hierarchical plan", G))
        _same(r2, w2, ("its rebalance", G))
        assert n3 == 2 * r3.iterations, (G, n3)                  # collectives A and B of every replica pass
        assert n1 > n3 and n2 > n1, (G, n3, n1, n2)              # the weighted plans sharded as well
    for pl in planners:
        pl.close()


def _flat_wide_k(planner_obj, k, P, N):
    """A flat state with k = 3 or 4 copies (k_pass_tree<4>): weighted fresh plan and the rebalance from it."""
    c = synth.rebalance_case(P=P, N=N, hierarchy=False)
    c["model"] = {"primary": {"priority": 0, "constraints": 1}, "replica": {"priority": 1, "constraints": k}}
    fresh = {p: {"name": p, "nodesByState": {}} for p in c["partitions"]}
    opts = dict(partition_weights=c["partitionWeights"], state_stickiness=c["stateStickiness"],
                node_weights=c["nodeWeights"], node_hierarchy=None, hierarchy_rules=None)
    fp1 = problem.build_problem({}, fresh, c["oldNodes"], [], c["oldNodes"], c["model"], **opts)
    r1 = planner_obj.plan(fp1)
    _same(r1, _oracle(fp1), ("flat k", k, "initial"))
    plan1, _ = problem.decode_result(fp1, r1)
    fp2 = problem.build_problem(plan1, plan1, c["nodesAll"], c["nodesToRemove"], c["nodesToAdd"], c["model"], **opts)
    _same(planner_obj.plan(fp2), _oracle(fp2), ("flat k", k, "rebalance"))


@pytest.mark.parametrize("k", [3, 4])
def test_tree_pass_flat_three_and_four_copies(planner, k):
    """k_pass_tree<4>: flat states with 3 and 4 copies, clusters of one and of several leaf groups."""
    _flat_wide_k(planner, k, P=4000, N=50)
    _flat_wide_k(planner, k, P=6000, N=700)
    from randgen import random_flat_wide_case
    n = 0
    for seed in range(120):
        try:
            fp = build_from_case(random_flat_wide_case(seed, k))
        except problem.Unsupported:
            continue
        _same(planner.plan(fp), _oracle(fp), ("random flat k", k, seed))
        n += 1
    assert n > 80


def test_stays_verified_per_top_priority_node():
    """k_stay_by_top on the device: forced in every chain pass with NumPartitions > 0 (random regular hierarchies: mostly
    refused, the chain kernel redoes the pass), taken on its own in config 3's converged sweep, switched off."""
    pl = hip.Planner(device_id=0, chain_min_parts=1, stay_top="force")
    n = 0
    for seed in range(300):
        try:
            fp = build_from_case(random_regular_case(seed))
        except problem.Unsupported:
            continue
        _same(pl.plan(fp), _oracle(fp), ("forced stay verification", seed))
        n += 1
    assert n > 200
    pl.close()
    for mode in ("auto", "off", "force"):
        pl = hip.Planner(device_id=0, stay_top=mode)
        for P, N in ((16384, 1024), (20000, 777), (4096, 4096)):
            fp = synth.config_flat(3, P=P, N=N)
            _same(pl.plan(fp), _oracle(fp), ("cfg3", mode, P, N))
        pl.close()


@pytest.mark.parametrize("k_rep", [1, 2, 3, 4])
def test_all_blank_pass_plane_automaton(k_rep):
    """k_pass_chain_planes (the first replica pass of a fresh plan as a scalar bit-plane automaton, hand-written
    assembly) over its envelope: zones of 40 .. 128 leaves (one and two 64-bit words per plane), racks that are
    aligned power-of-two runs (class masks by arithmetic) and racks of 5 / 7 / 12 (class masks by lane reads),
    k = 1 .. 4 picks per step, leaves without a node (names of the tree that are not in nodesAll), one partition weight
    other than 1 for every partition, a rule that excludes only the anchor itself; and the same plans with the
    kernel switched off (k_pass_chain_blank) -- all against the oracle."""
    import random
    shapes = [(16, 8, 3), (8, 16, 2), (4, 10, 3), (5, 9, 4), (7, 11, 2), (12, 10, 2), (16, 5, 3), (2, 30, 2), (32, 4, 2)]
    for planes in (True, False):
        pl = hip.Planner(device_id=0, chain_min_parts=1, planes=planes, periodic=False)     # every step walked
        for si, (rack, racks_per_zone, n_zones) in enumerate(shapes):
            if rack * racks_per_zone > 128 or racks_per_zone <= k_rep + 1:
                continue
            rnd = random.Random(1000 * k_rep + si)
            N = rack * racks_per_zone * n_zones
            c = synth.config_case(3, P=rnd.choice([3000, 9000]), N=N)
            c["nodeHierarchy"] = synth.hierarchy_names(N, rack=rack, racks_per_zone=racks_per_zone, zones_per_dc=2)
            c["model"] = {"primary": {"priority": 0, "constraints": 1}, "replica": {"priority": 1, "constraints": k_rep}}
            if si % 3 == 1:
                c["hierarchyRules"] = {"replica": [{"includeLevel": 2, "excludeLevel": 0}]}
            if si % 2 == 0:                                   # leaves of the tree that carry no node of nodesAll
                gone = set(rnd.sample(c["nodesAll"], max(1, N // 17)))
                c["nodesAll"] = [n for n in c["nodesAll"] if n not in gone]
                c["nodesToAdd"] = list(c["nodesAll"])
            if si % 4 == 3:
                c["partitionWeights"] = {p: 3 for p in c["partitionsToAssign"]}
            fp = synth.case_to_flat(c)
            _same(pl.plan(fp), _oracle(fp), ("planes" if planes else "lanes", k_rep, rack, racks_per_zone, n_zones))
        pl.close()


@pytest.mark.parametrize("rack,racks_per_zone,k_rep", [(12, 8, 2), (8, 16, 3), (16, 8, 1)])
def test_config3_shapes_at_scale(planner, rack, racks_per_zone, k_rep):
    """Config 3's generator with other trees at 131,072 partitions x 3,072 / 4,096 nodes: racks of 12 (class masks by
    lane reads, zones of 96 leaves), racks of 8 in zones of 128 (class masks by arithmetic, k = 3), k = 1 -- every
    chain of thousands of steps, all three sweeps, against the oracle."""
    N = rack * racks_per_zone * 32
    c = synth.config_case(3, P=131072, N=N)
    c["nodeHierarchy"] = synth.hierarchy_names(N, rack=rack, racks_per_zone=racks_per_zone, zones_per_dc=8)
    c["model"] = {"primary": {"priority": 0, "constraints": 1}, "replica": {"priority": 1, "constraints": k_rep}}
    fp = synth.case_to_flat(c)
    _same(planner.plan(fp), _oracle(fp), ("config 3 at scale", rack, racks_per_zone, k_rep))



@pytest.mark.gpu
def test_host_shortcuts_on_the_device(monkeypatch):
    """Round 6's shortcuts of the host on the device, where streams really run side by side: plans from nothing (config 2's and
    config 3's shapes, partition counts that are no multiple of the node count) with the shortcuts on, off and forced to fail
    -- the same map, fewer round trips with them on; k_stay_by_top's work list made a sweep ahead on the second stream and used
    by the next sweep (the converged sweep is a k_stay_by_top pass; repeated, so that an ordering left to chance would show)."""
    shapes = [(2, 50000, 97), (2, 65536, 256), (3, 65536, 1024), (3, 100000, 768), (3, 131072, 4096)]
    for cfg, P, N in shapes:
        fp = synth.config_flat(cfg, P=P, N=N)
        want = _oracle(fp)
        syncs = {}
        for spec in ("1", "0", "fail"):
            monkeypatch.setenv("BLANCE_SPECULATE", spec)
            pl = hip.Planner(device_id=0)
            for rep in range(3 if spec == "1" else 1):
                got = pl.plan(fp)
                _same(got, want, ("host shortcuts", cfg, P, N, spec, rep))
                if cfg == 3 and spec == "1":
                    assert got.struct.stay_pass_launches >= 1, (P, N)
            syncs[spec] = got.struct.host_syncs
            pl.close()
        assert syncs["1"] < syncs["0"], (cfg, P, N, syncs)
