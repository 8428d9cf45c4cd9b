"""The oracles against the reference's own golden tables:
69 end-to-end planner cases transcribed from plan_test.go / control_test.go and
the helper-function tables of plan_test.go:21-390 / misc_test.go:18-89."""
import pytest

from blance_amd import problem
from oracle import blance_ref as R
from helpers import run_c_oracle, run_ref


def test_case_count(golden_cases):
    assert len(golden_cases) == 69


def test_python_oracle_matches_reference_tables(golden_cases):
    iters = {}
    for c in golden_cases:
        out, w, info = run_ref(c)
        assert out == c["exp"], (c["suite"], c["index"], c["source"])
        assert R.count_warnings(w, c["warningCountMode"]) == c["expNumWarnings"], c["source"]
        iters[info["iterations"]] = iters.get(info["iterations"], 0) + 1
    # sweeps needed over the 69 cases
    assert iters == {1: 5, 2: 57, 3: 6, 4: 1}


def test_c_oracle_matches_reference_tables(golden_cases):
    for c in golden_cases:
        out, w, info, _, _ = run_c_oracle(c)
        assert out == c["exp"], (c["suite"], c["index"], c["source"])
        assert R.count_warnings(w, c["warningCountMode"]) == c["expNumWarnings"], c["source"]


def test_c_oracle_matches_python_oracle_on_golden_inputs(golden_cases):
    for c in golden_cases:
        out, w, info = run_ref(c)
        out2, w2, info2, _, _ = run_c_oracle(c)
        assert out2 == out and w2 == w and info2 == info, c["source"]


def test_warning_text():
    # plan.go:231-234
    c = {"prevMap": {}, "partitionsToAssign": {"0": {"name": "0", "nodesByState": {}}},
         "aliased": False, "nodesAll": ["a"], "nodesToRemove": [], "nodesToAdd": ["a"],
         "model": {"primary": {"priority": 0, "constraints": 1},
                   "replica": {"priority": 1, "constraints": 1}}}
    out, w, _, _, _ = run_c_oracle(c)
    assert w == {"0": ["could not meet constraints: 1, stateName: replica, partitionName: 0"]}
    assert out == {"0": {"name": "0", "nodesByState": {"primary": ["a"], "replica": []}}}


# ---- helper tables ---------------------------------------------------------

def test_flatten_nodes_by_state(helper_tables):
    for row in helper_tables["TestFlattenNodesByState"]:
        assert sorted(R.flatten_nodes_by_state(row["a"])) == sorted(row["exp"])


def test_remove_nodes_from_nodes_by_state(helper_tables):
    for row in helper_tables["TestRemoveNodesFromNodesByState"]:
        got = R.remove_nodes_from_nodes_by_state(row["nodesByState"], row["removeNodes"], None)
        assert got == row["exp"]


def test_state_name_sorter(helper_tables):
    for row in helper_tables["TestStateNameSorter"]:
        model = {k: {"priority": v.get("Priority", 0), "constraints": v.get("Constraints", 0)}
                 for k, v in (row["m"] or {}).items()}
        names = list(row["s"])
        # plan_test.go:118-180 sorts an explicit slice with the model's comparator
        for i in range(1, len(names)):
            j = i
            while j > 0 and R.state_name_less(model, names[j], names[j - 1]):
                names[j], names[j - 1] = names[j - 1], names[j]
                j -= 1
        assert names == row["exp"]
        if set(row["s"]) == set(model):
            assert problem.sort_state_names(model) == row["exp"]


def test_count_state_nodes(helper_tables):
    for row in helper_tables["TestCountStateNodes"]:
        pm = R.partition_map_from_json(
            {k: {"name": v.get("Name", ""), "nodesByState": v.get("NodesByState")}
             for k, v in row["m"].items()})
        assert R.count_state_nodes(pm, row.get("w")) == row["exp"]


def test_find_ancestor(helper_tables):
    for row in helper_tables["TestFindAncestor"]:
        assert R.find_ancestor("a", row["mapParents"], row["level"]) == row["exp"]


def test_find_leaves(helper_tables):
    for row in helper_tables["TestFindLeaves"]:
        assert R.find_leaves("a", row["mapChildren"]) == row["exp"]


def test_map_parents_to_map_children(helper_tables):
    for row in helper_tables["TestMapParentsToMapChildren"]:
        assert R.map_parents_to_map_children(row["in"]) == row["exp"]


def test_strings_helpers(helper_tables):
    for row in helper_tables["TestStringsToMap"]:
        assert R.strings_to_map(row["s"]) == row["exp"]
    for row in helper_tables["TestStringsRemoveStrings"]:
        assert R.strings_remove_strings(row["a"], row["b"]) == row["exp"]
    for row in helper_tables["TestStringsIntersectStrings"]:
        assert R.strings_intersect_strings(row["a"], row["b"]) == row["exp"]
