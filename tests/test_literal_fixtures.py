"""Results of the LITERAL string-keyed oracle at 4,096 partitions (tests/golden/literal_oracle_cases.json, made by
tests/tools/make_literal_fixtures.py) against the implementations that sit behind blance_amd/problem.py's interning:
the id-based C oracle (CPU suite) and the device (`-m gpu`).  This is the check that does not share the interning code
(static partition order of non-numeric names -- plan.go:525-528 --, weights, hierarchy intervals) with what it checks."""
import json
import os

import pytest

import literal_cases as L
from blance_amd import problem
from helpers import build_from_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fixture():
    with open(os.path.join(ROOT, "tests", "golden", "literal_oracle_cases.json")) as f:
        return {e["case"]: e for e in json.load(f)["cases"]}


def _check(plan, fixture, tag):
    """plan(fp) -> a FlatResult; walks the four cases (the rebalance starts from the config-3 plan the same solver made)."""
    from blance_amd import synth
    cases = [("named_weighted", L.case_named_weighted()), ("node_weights", L.case_node_weights()),
             ("config3", synth.config_case(3, P=L.P_LITERAL, N=L.N_LITERAL))]
    plan3 = None
    for name, c in cases + [("rebalance_of_config3", None)]:
        if c is None:
            c = L.case_rebalance(plan3)
        fp = build_from_case(c)
        res = plan(fp)
        out, w = problem.decode_result(fp, res)
        want = fixture[name]
        assert (res.iterations, bool(res.converged)) == (want["iterations"], want["converged"]), (tag, name)
        assert L.canonical_sha(out) == want["map_sha256"], (tag, name)
        assert L.warnings_sha(w) == want["warnings_sha256"], (tag, name)
        if name == "config3":
            plan3 = out


def test_c_oracle_equals_literal_oracle_at_4096_partitions(fixture):
    from oracle import loader
    assert len(fixture) == 4 and all(e["partitions"] == L.P_LITERAL and e["nodes"] == L.N_LITERAL for e in fixture.values())
    _check(loader.plan, fixture, "C oracle")


@pytest.mark.gpu
@pytest.mark.parametrize("eager", [False, True])
def test_device_equals_literal_oracle_at_4096_partitions(fixture, eager):
    """The device against the literal oracle's fixture: default engine selection (4,096 steps run as region chains /
    k_pass_queue already) and with the bulk engines forced on for passes of any size."""
    from blance_amd import hip
    pl = hip.Planner(device_id=0, chain_min_parts=1 if eager else 0)
    try:
        _check(pl.plan, fixture, "device, eager=%s" % eager)
    finally:
        pl.close()
