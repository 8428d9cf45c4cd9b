"""The host mirror of the reference API (blance_amd/planner.py): results,
warnings text and the caller-visible mutations of plan.go:49-52 against the
literal Python oracle, on the reference's golden inputs."""
import copy

import pytest

from blance_amd import hip, planner, problem
from oracle import blance_ref as R


def _objects(d):
    return None if d is None else {k: planner.Partition(v.get("name", ""), copy.deepcopy(v.get("nodesByState")))
                                   for k, v in d.items()}


def _json(m):
    return None if m is None else {k: {"name": p.Name, "nodesByState": p.NodesByState} for k, p in m.items()}


def _check_case(c, pl):
    # the oracle, on its own copies
    prev_o = R.partition_map_from_json(copy.deepcopy(c["prevMap"]))
    assign_o = prev_o if c.get("aliased") else R.partition_map_from_json(copy.deepcopy(c["partitionsToAssign"]))
    opts_o = R.Options(c.get("modelStateConstraints"), c.get("partitionWeights"), c.get("stateStickiness"),
                       c.get("nodeWeights"), c.get("nodeHierarchy"), c.get("hierarchyRules"))
    want, want_w = R.plan_next_map_ex(prev_o, assign_o, list(c["nodesAll"]), c["nodesToRemove"], c["nodesToAdd"],
                                      c["model"], opts_o, c.get("booster"))
    # the product API
    prev = _objects(c["prevMap"])
    assign = prev if c.get("aliased") else _objects(c["partitionsToAssign"])
    model = {s: planner.PartitionModelState(v["priority"], v["constraints"]) for s, v in c["model"].items()}
    rules = c.get("hierarchyRules")
    if rules is not None:
        rules = {s: [planner.HierarchyRule(r["includeLevel"], r["excludeLevel"]) for r in rl] for s, rl in rules.items()}
    opts = planner.PlanNextMapOptions(c.get("modelStateConstraints"), c.get("partitionWeights"),
                                      c.get("stateStickiness"), c.get("nodeWeights"), c.get("nodeHierarchy"), rules)
    got, got_w = planner.PlanNextMapEx(prev, assign, list(c["nodesAll"]), c["nodesToRemove"], c["nodesToAdd"], model,
                                       opts, booster=c.get("booster"), planner=pl)
    assert _json(got) == R.partition_map_to_json(want) == c["exp"], c["source"]
    assert (got_w or {}) == (want_w or {}), c["source"]
    # caller-visible mutations (plan.go:49-52)
    assert _json(prev) == R.partition_map_to_json(prev_o), c["source"]
    assert _json(assign) == R.partition_map_to_json(assign_o), c["source"]
    # object identity: what is stored in the input maps are the objects of the last sweep that did NOT converge
    # (plan.go:49-52), the returned ones are fresh (plan.go:334-343) -- editing nextMap[p] edits prevMap[p] only where
    # the reference would
    if want is not None:
        for name in want:
            assert (got[name] is prev.get(name)) == (want[name] is prev_o.get(name)), (c["source"], name)
            assert (got[name] is assign.get(name)) == (want[name] is assign_o.get(name)), (c["source"], name)
            if name in prev_o and name in assign_o:
                assert (prev[name] is assign[name]) == (prev_o[name] is assign_o[name]), (c["source"], name)
        if any(want[n] is not prev_o.get(n) for n in want):
            n0 = next(n for n in want if want[n] is not prev_o.get(n))
            if got[n0].NodesByState and n0 in prev:
                before = copy.deepcopy(prev[n0].NodesByState)
                got[n0].NodesByState["edited-by-the-caller"] = ["x"]
                assert prev[n0].NodesByState == before, c["source"]


@pytest.mark.gpu
def test_api_mirror_on_golden_cases_gpu(golden_cases):
    pl = hip.Planner(device_id=0)
    for c in golden_cases:
        _check_case(c, pl)
    pl.close()


def test_api_mirror_on_golden_cases_emulated(golden_cases):
    from test_simt_emulated import build_emu
    pl = hip.Planner(lib_path=build_emu(), force_threads=64)
    for c in golden_cases:
        _check_case(c, pl)
    pl.close()


def test_helpers(helper_tables):
    for row in helper_tables["TestStringsToMap"]:
        assert planner.StringsToMap(row["s"]) == row["exp"]
    for row in helper_tables["TestStringsRemoveStrings"]:
        assert planner.StringsRemoveStrings(row["a"], row["b"]) == row["exp"]
    for row in helper_tables["TestStringsIntersectStrings"]:
        assert planner.StringsIntersectStrings(row["a"], row["b"]) == row["exp"]


def test_unsupported_inputs_raise():
    with pytest.raises(problem.Unsupported):
        problem.build_problem({}, {"0": {"name": "0", "nodesByState": {}}}, ["a", "a"], [], [],
                              {"primary": {"priority": 0, "constraints": 1}})


def test_callbacks_of_the_caller_are_refused(monkeypatch):
    """plan.go:580 CustomNodeSorter and plan.go:693 NodeScoreBooster are hooks in the caller's language:
    the device path refuses them before touching the library (the Go shim runs plan.go for such calls)."""
    args = ({}, {"0": {"name": "0", "nodesByState": {}}}, ["a", "b"], [], [],
            {"primary": {"priority": 0, "constraints": 1}})
    with pytest.raises(problem.Unsupported):
        planner.PlanNextMapEx(*args, booster=lambda w, s: s)
    monkeypatch.setattr(planner, "CustomNodeSorter", lambda *a: None)
    with pytest.raises(problem.Unsupported):
        planner.PlanNextMapEx(*args)
