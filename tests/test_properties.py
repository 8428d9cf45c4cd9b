"""Size-independent properties of a plan (tests/properties.py) -- a check that does not pass through the oracle:
on the CPU for the oracle's and the emulated kernels' results at reduced sizes, on the MI355X for BASELINE.json
config 3 at its FULL size (1,048,576 x 4,096), where the oracle is only present as a committed digest."""
import json
import os

import pytest

import properties
from blance_amd import hip, synth

HERE = os.path.dirname(os.path.abspath(__file__))


def _golden():
    with open(os.path.join(HERE, "golden", "config3_full_size_properties.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("P,N", [(4096, 4096), (16384, 1024), (20000, 777), (5000, 300)])
def test_oracle_results_keep_the_promises(P, N):
    from oracle import loader
    fp = synth.config_flat(3, P=P, N=N)
    r = loader.plan(fp)
    props = properties.config3_properties(fp, r)
    assert props["primary_sum"] == P and props["replica_sum"] == 2 * P
    if N % 128 == 0 and P % N == 0:                 # whole zones, whole rounds: the greedy levels exactly
        assert props["primary_spread"] == 0 and props["replica_spread"] == 0
    r2 = loader.plan(synth.replan_problem(fp, r))   # the reference's own fixed point test, plan.go:32-57
    assert r2.iterations == 1 and r2.converged and properties.same_lists(r, r2, P, 2)


def test_emulated_kernels_keep_the_promises():
    from test_simt_emulated import build_emu
    pl = hip.Planner(lib_path=build_emu(), chain_min_parts=8)
    for P, N in ((2048, 256), (1500, 300)):
        fp = synth.config_flat(3, P=P, N=N)
        r = pl.plan(fp)
        properties.config3_properties(fp, r)
        r2 = pl.plan(synth.replan_problem(fp, r))
        assert r2.iterations == 1 and r2.converged and properties.same_lists(r, r2, P, 2)
    pl.close()


@pytest.mark.parametrize("hier", [False, True])
def test_weighted_rebalance_keeps_the_general_promises(hier):
    """Config 5 in miniature (Zipf partition weights, node weights, stickiness, a tenth of the nodes removed / added):
    the oracle's and the emulated kernels' plans; the GPU suite applies the same function to config 5 at full size."""
    from oracle import loader
    from test_simt_emulated import build_emu
    P, N = (1500, 128) if hier else (2000, 96)
    pl = hip.Planner(lib_path=build_emu(), chain_min_parts=8)
    fp1 = synth.config5_initial(P, N, hierarchy=hier)
    r1 = pl.plan(fp1)
    assert properties.plan_properties(fp1, r1) == properties.plan_properties(fp1, loader.plan(fp1))
    fp2 = synth.config5_rebalance(fp1, r1, P, N, hierarchy=hier)
    r2 = pl.plan(fp2)
    assert properties.plan_properties(fp2, r2) == properties.plan_properties(fp2, loader.plan(fp2))
    pl.close()


def test_committed_full_size_numbers_are_the_oracles():
    """tests/golden/config3_full_size_properties.json was written by tests/tools/full_size_properties_oracle.py from the
    oracle's full-size result -- the one whose digest tests/golden/config_digests.json holds."""
    g = _golden()
    with open(os.path.join(HERE, "golden", "config_digests.json")) as f:
        want = json.load(f)["config3"]
    assert (g["partitions"], g["nodes"]) == (1 << 20, 4096)
    assert g["digest"] == want["digest"] and g["iterations"] == want["iterations"]
    assert g["replan_iterations"] == 1 and g["replan_converged"] and g["replan_same_lists"]


@pytest.mark.gpu
def test_config3_full_size_properties_on_the_device():
    """The headline configuration at its full size: model and rule promises for every one of the 1,048,576 partitions,
    the per-node spreads the oracle's result has, and idempotence -- the plan from the result converges in its first
    sweep on the same map (its digest: the oracle's, committed)."""
    g = _golden()
    pl = hip.Planner(device_id=0)
    try:
        fp = synth.config_flat(3)
        r = pl.plan(fp)
        props = properties.config3_properties(fp, r)
        assert props == g["properties"]
        r2 = pl.plan(synth.replan_problem(fp, r))
        assert r2.iterations == 1 and r2.converged
        assert properties.same_lists(r, r2, 1 << 20, 2)
        assert r2.digest() == g["replan_digest"]
    finally:
        pl.close()


def _moves_scenario(pl, P, N):
    """begMap = config 3's plan; endMap = the plan after a tenth of the nodes left (a real rebalance: adds, deletes,
    promotions where a replica takes over a lost primary), both as flat lists; -> calc_moves arguments."""
    import numpy as np
    fp = synth.config_flat(3, P=P, N=N)
    r1 = pl.plan(fp)
    fp2 = synth.replan_problem(fp, r1)
    rm = np.zeros(N, dtype=np.uint8)
    rm[np.arange(N) % 10 == 3] = 1
    fp2.set("node_removed", rm)
    r2 = pl.plan(fp2)
    M = 2

    def csr(res):                                   # p * (M + 1) + state, the extra slot (other states) empty
        off, nodes, _ = properties.lists_of(res, P, M)
        ln = np.zeros((P, M + 1), dtype=np.int64)
        ln[:, :M] = np.diff(off).reshape(P, M)
        o = np.zeros(P * (M + 1) + 1, dtype=np.int32)
        o[1:] = np.cumsum(ln.reshape(-1))
        return o, nodes.astype(np.int32)
    return fp2, r2, M, csr(r1), csr(r2)


def _check_moves(pl, P, N):
    fp2, r2, M, (bo, bn), (eo, en) = _moves_scenario(pl, P, N)
    properties.plan_properties(fp2, r2)
    counts = []
    for favor in (False, True):
        op_off, op_node, op_state, op_kind, _ = pl.calc_moves(M, favor, bo, bn, eo, en)
        counts.append(properties.moves_round_trip(P, M, N, bo, bn, eo, en, op_off, op_node, op_state, op_kind))
    assert counts[0] == counts[1] and counts[0] > P // 10      # favorMinNodes reorders the moves, it does not change them
    return fp2, r2


def test_moves_carried_out_on_the_first_map_give_the_second_emulated():
    from oracle import loader
    from test_simt_emulated import build_emu
    pl = hip.Planner(lib_path=build_emu(), chain_min_parts=8)
    for P, N in ((2048, 256), (1500, 300)):
        fp2, r2 = _check_moves(pl, P, N)
        want = loader.plan(fp2)                     # the rebalance itself: the oracle's (flat problem made by replan_problem)
        assert (r2.digest(), r2.iterations) == (want.digest(), want.iterations)
    pl.close()


@pytest.mark.gpu
def test_moves_carried_out_on_the_first_map_give_the_second_full_size():
    """blance_calc_moves for the 1,048,576 partitions of config 3 between the plan and its rebalance after a tenth of
    the nodes left: begMap + moves == endMap, checked without an oracle; the rebalanced map (a hierarchical rebalance
    at full size: 4 sweeps) keeps the general promises and has the oracle's digest."""
    want = _golden()["rebalance"]                   # the oracle's rebalance of the same flat problem, 164 s on one core
    pl = hip.Planner(device_id=0)
    try:
        fp2, r2 = _check_moves(pl, 1 << 20, 4096)
        assert (r2.digest(), r2.iterations, bool(r2.converged), int(r2.n_warnings)) == (
            want["digest"], want["iterations"], want["converged"], want["warnings"])
        assert properties.plan_properties(fp2, r2) == want["properties"]
    finally:
        pl.close()
