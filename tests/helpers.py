"""Shared test helpers: run one API-level case through an implementation."""
import copy

from blance_amd import problem


def build_from_case(c, max_iterations=10):
    prev = c["prevMap"]
    assign = prev if c.get("aliased") else c["partitionsToAssign"]
    return problem.build_problem(
        prev, assign, c["nodesAll"], c["nodesToRemove"], c["nodesToAdd"], c["model"],
        c.get("modelStateConstraints"), c.get("partitionWeights"), c.get("stateStickiness"),
        c.get("nodeWeights"), c.get("nodeHierarchy"), c.get("hierarchyRules"), c.get("booster"),
        max_iterations=max_iterations)


def run_ref(c):
    """Literal Python oracle on a deep copy; returns (map, warnings, info) or
    raises RuntimeError where the reference would panic."""
    from oracle import blance_ref as R
    info = {}
    out, w = R.run_case(copy.deepcopy(c), info)
    return out, (w or {}), info


def run_c_oracle(c):
    from oracle import loader
    fp = build_from_case(c)
    res = loader.plan(fp)
    out, w = problem.decode_result(fp, res)
    return out, w, {"iterations": res.iterations, "converged": res.converged}, fp, res
