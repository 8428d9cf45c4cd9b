"""k_pass_tree (flat passes on one wave64 with bound-ordered candidates, blance_amd/csrc/k_pass_tree.h)
on the SIMT emulator against the oracle: golden cases, random instances, the rebalance shape, edge
shapes -- with the bounded candidate walk and with every general step scoring all nodes ("dense")."""
import pytest

from blance_amd import hip, problem, synth
from helpers import build_from_case, edge_cases
from randgen import random_case, random_flat_wide_case
from test_simt_emulated import build_emu


@pytest.fixture(scope="module")
def emu_lib():
    return build_emu()


def _oracle(fp):
    from oracle import loader
    return loader.plan(fp)


@pytest.mark.parametrize("mode", ["on", "dense", "long"])
def test_golden_cases_tree(emu_lib, golden_cases, mode):
    for eager in (0, 1):                         # 1: the flat bulk driver hands sub-ranges to the tree kernel
        pl = hip.Planner(lib_path=emu_lib, tree=mode, chain_min_parts=eager)
        for c in golden_cases:
            fp = build_from_case(c)
            got = pl.plan(fp)
            assert got.digest() == _oracle(fp).digest(), c["source"]
            out, _ = problem.decode_result(fp, got)
            assert out == c["exp"], c["source"]
        pl.close()


def test_random_instances_tree(emu_lib):
    pl = hip.Planner(lib_path=emu_lib, tree="on")
    n = 0
    for seed in range(0, 500):
        try:
            fp = build_from_case(random_case(seed))
        except problem.Unsupported:
            continue
        got, want = pl.plan(fp), _oracle(fp)
        assert (got.digest(), got.iterations, got.n_warnings) == (want.digest(), want.iterations, want.n_warnings), seed
        n += 1
    assert n > 330
    pl.close()


def test_random_instances_tree_dense_and_bulk(emu_lib):
    pd = hip.Planner(lib_path=emu_lib, tree="dense")
    pb = hip.Planner(lib_path=emu_lib, tree="long", chain_min_parts=1)
    for seed in range(500, 580):             # (the GPU suite walks 300 of these in dense mode)
        try:
            fp = build_from_case(random_case(seed))
        except problem.Unsupported:
            continue
        want = _oracle(fp).digest()
        assert pd.plan(fp).digest() == want, seed
        assert pb.plan(fp).digest() == want, seed
    pd.close()
    pb.close()


def _rebalance(pl, P, N, check_stays=False, model=None, **kw):
    c = synth.rebalance_case(P=P, N=N, hierarchy=False, **kw)
    if model:
        c["model"] = model
    fresh = {p: {"name": p, "nodesByState": {}} for p in c["partitions"]}
    opts = dict(partition_weights=c["partitionWeights"], state_stickiness=c["stateStickiness"],
                node_weights=c["nodeWeights"], node_hierarchy=None, hierarchy_rules=None)
    fp1 = problem.build_problem({}, fresh, c["oldNodes"], [], c["oldNodes"], c["model"], **opts)
    r1 = pl.plan(fp1)
    assert r1.digest() == _oracle(fp1).digest()
    plan1, _ = problem.decode_result(fp1, r1)
    fp2 = problem.build_problem(plan1, plan1, c["nodesAll"], c["nodesToRemove"], c["nodesToAdd"], c["model"], **opts)
    r2 = pl.plan(fp2)
    assert r2.digest() == _oracle(fp2).digest()
    if check_stays:
        assert r2.struct.steps_batched > P       # beyond the primary pass's bulk stays
    return r2


def test_rebalance_tree(emu_lib):
    """Config 5's ingredients (Zipf partition weights, node weights 1/2/4, stickiness, a tenth of the
    nodes removed and added) on flat clusters of 1, 2 and 5 leaf groups."""
    pl = hip.Planner(lib_path=emu_lib, tree="on")
    _rebalance(pl, 200, 24)
    _rebalance(pl, 300, 100)
    _rebalance(pl, 300, 300, check_stays=True)
    pl.close()
    for mode in ("dense", "long", "dense-long"):
        pl = hip.Planner(lib_path=emu_lib, tree=mode)
        _rebalance(pl, 120, 70)
        pl.close()


def test_folded_row_tree(emu_lib):
    """Half of the nodes removed: hundreds of consecutive steps have no top priority node and share the
    row "" of nodeToNodeCounts -- k_pass_tree folds that row into its leaves for whole batches, and
    unfolds when a mixed batch comes."""
    for mode in ("on", "dense", "long"):
        pl = hip.Planner(lib_path=emu_lib, tree=mode)
        _rebalance(pl, 400, 40, remove_frac=0.5, add_frac=0.3)
        if mode == "on":
            _rebalance(pl, 260, 130, remove_frac=0.6, add_frac=0.1)
        pl.close()


def test_reduced_configs_tree(emu_lib):
    pl = hip.Planner(lib_path=emu_lib, tree="on", chain_min_parts=64)
    for fp in (synth.config_flat(1), synth.config_flat(2, P=2048, N=32), synth.config_flat(2, P=600, N=300)):
        got = pl.plan(fp)
        assert got.digest() == _oracle(fp).digest()
    pl.close()


def test_edge_shapes_tree(emu_lib):
    cases = edge_cases()
    for mode in ("on", "dense", "long"):
        pl = hip.Planner(lib_path=emu_lib, tree=mode)
        for i, (a, k) in enumerate(cases):
            fp = problem.build_problem(*a, **k)
            got, want = pl.plan(fp), _oracle(fp)
            assert (got.digest(), got.iterations, got.n_warnings) == (want.digest(), want.iterations, want.n_warnings), i
        pl.close()


@pytest.mark.parametrize("k", [3, 4])
def test_flat_three_and_four_copies_tree(emu_lib, k):
    """k_pass_tree<4> (flat state, k = 3 / 4): random instances in the bounded-walk, dense and record-decoding
    modes, and the weighted rebalance shape."""
    planners = [hip.Planner(lib_path=emu_lib, tree=mode) for mode in ("on", "dense", "long")]
    n = 0
    for seed in range(70):
        try:
            fp = build_from_case(random_flat_wide_case(seed, k))
        except problem.Unsupported:
            continue
        want = _oracle(fp)
        for pl in planners[:1] if seed % 3 else planners:
            got = pl.plan(fp)
            assert (got.digest(), got.iterations, got.n_warnings) == (want.digest(), want.iterations, want.n_warnings), seed
        n += 1
    assert n > 45
    model = {"primary": {"priority": 0, "constraints": 1}, "replica": {"priority": 1, "constraints": k}}
    _rebalance(planners[0], 160, 30, model=model)
    _rebalance(planners[0], 200, 150, model=model)
    _rebalance(planners[2], 100, 70, model=model)
    for pl in planners:
        pl.close()
