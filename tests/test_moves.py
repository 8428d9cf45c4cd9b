"""CalcPartitionMoves (moves.go:41-136), the next row after the planner:
the Python oracle against the reference's own tables, and the batched HIP kernel
(through the C ABI) against the oracle."""
import json
import os
import random

import pytest

from blance_amd import hip, planner
from oracle import moves_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "moves_cases.json")) as f:
        return json.load(f)


def _matches(got, exp):
    return len(got) == len(exp) and all(m[0] == e["node"] and m[1] == e["state"] and m[2] in e["ops"]
                                        for m, e in zip(got, exp))


def test_oracle_find_state_changes(golden):
    assert len(golden["findStateChanges"]) == 9
    for row in golden["findStateChanges"]:
        got = R.find_state_changes(row["begStateIdx"], row["endStateIdx"], row["state"], row["states"],
                                   row["begNodesByState"], row["endNodesByState"])
        assert (got or None) == (row["expected"] or None), row


def test_oracle_calc_partition_moves(golden):
    assert len(golden["calcPartitionMoves"]) == 29
    for c in golden["calcPartitionMoves"]:
        assert _matches(R.calc_partition_moves(c["states"], c["before"], c["after"], c["favorMinNodes"]), c["exp"]), c


def random_pair(rng, states, nodes):
    def nbs():
        pool = nodes[:]
        rng.shuffle(pool)
        d = {}
        for s in states + (["extra"] if rng.random() < 0.2 else []):
            if rng.random() < 0.85:
                d[s] = [pool.pop() for _ in range(rng.randint(0, min(3, len(pool))))]
        if rng.random() < 0.1 and d:
            k = rng.choice(list(d))
            d[k] = d[k] + d[k][:1]                     # a duplicate inside one list
        return d
    return nbs(), nbs()


def _check_device(pl, golden):
    for favor in (False, True):
        cases = [c for c in golden["calcPartitionMoves"] if c["favorMinNodes"] == favor]
        beg = {str(i): c["before"] for i, c in enumerate(cases)}
        end = {str(i): c["after"] for i, c in enumerate(cases)}
        got = planner.CalcPartitionMovesBatch(["primary", "replica"], beg, end, favor, planner=pl)
        for i, c in enumerate(cases):
            assert _matches([(m.Node, m.State, m.Op) for m in got[str(i)]], c["exp"]), c
    rng = random.Random(7)
    for trial in range(30):
        states = ["primary", "replica", "readonly"][:rng.randint(1, 3)]
        nodes = ["n%d" % i for i in range(rng.randint(1, 9))]
        favor = rng.random() < 0.5
        beg, end = {}, {}
        for i in range(rng.randint(1, 200)):
            b, e = random_pair(rng, states, nodes)
            if rng.random() < 0.9:
                beg[str(i)] = b
            if rng.random() < 0.9:
                end[str(i)] = e
        got = planner.CalcPartitionMovesBatch(states, beg, end, favor, planner=pl)
        for name in set(beg) | set(end):
            want = R.calc_partition_moves(states, beg.get(name), end.get(name), favor)
            assert [(m.Node, m.State, m.Op) for m in got[name]] == want, (trial, name)
    one = planner.CalcPartitionMoves(["primary", "replica"], {"primary": ["a"], "replica": ["b"]},
                                     {"primary": ["b"], "replica": ["c"]}, False, planner=pl)
    assert [(m.Node, m.State, m.Op) for m in one] == R.calc_partition_moves(
        ["primary", "replica"], {"primary": ["a"], "replica": ["b"]}, {"primary": ["b"], "replica": ["c"]}, False)


def test_kernel_emulated(golden):
    from test_simt_emulated import build_emu
    pl = hip.Planner(lib_path=build_emu())
    _check_device(pl, golden)
    pl.close()


@pytest.mark.gpu
def test_kernel_gpu(golden):
    pl = hip.Planner(device_id=0)
    _check_device(pl, golden)
    pl.close()


@pytest.mark.gpu
def test_moves_of_a_planned_rebalance_gpu():
    """End to end: plan, rebalance, then the moves between the two maps for 20,000 partitions."""
    from blance_amd import problem, synth
    c = synth.rebalance_case(P=20000, N=128, hierarchy=False)
    fresh = {p: {"name": p, "nodesByState": {}} for p in c["partitions"]}
    pl = hip.Planner(device_id=0)
    fp1 = problem.build_problem({}, fresh, c["oldNodes"], [], c["oldNodes"], c["model"])
    beg, _ = problem.decode_result(fp1, pl.plan(fp1))
    fp2 = problem.build_problem(beg, beg, c["nodesAll"], c["nodesToRemove"], c["nodesToAdd"], c["model"])
    end, _ = problem.decode_result(fp2, pl.plan(fp2))
    states = ["primary", "replica"]
    b = {k: v["nodesByState"] for k, v in beg.items()}
    e = {k: v["nodesByState"] for k, v in end.items()}
    got = planner.CalcPartitionMovesBatch(states, b, e, False, planner=pl)
    moved = 0
    for name in b:
        want = R.calc_partition_moves(states, b[name], e[name], False)
        assert [(m.Node, m.State, m.Op) for m in got[name]] == want, name
        moved += bool(want)
    assert moved > 1000
    pl.close()
