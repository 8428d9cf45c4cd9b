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


def edge_cases():
    """(args, kwargs) of problem.build_problem for degenerate and unusual inputs."""
    from blance_amd import synth
    M1 = {"primary": {"priority": 0, "constraints": 1}, "replica": {"priority": 1, "constraints": 2}}
    fresh = lambda P: {str(i): {"name": str(i), "nodesByState": {}} for i in range(P)}     # noqa: E731
    hier24 = synth.hierarchy_names(24, rack=4, racks_per_zone=3, zones_per_dc=2, width=2)
    n24 = ["n%02d" % i for i in range(24)]
    prev = {str(i): {"name": str(i), "nodesByState": {"primary": ["n%d" % (i % 3)],
                                                      "replica": ["n%d" % ((i + 1) % 3), "gone"]}} for i in range(30)}
    cases = [
        (({}, {}, ["a", "b"], [], ["a", "b"], M1), {}),
        (({}, fresh(5), [], [], [], M1), {}),
        (({}, fresh(5), ["a"], [], ["a"], M1), {}),
        (({}, fresh(5), ["a", "b", "c"], [], [], M1), {"max_iterations": 0}),
        (({}, fresh(5), ["a", "b", "c"], [], [], M1), {"max_iterations": 1}),
        (({}, fresh(5), ["a", "b", "c"], [], [], {"primary": {"priority": 0, "constraints": 0},
                                                  "replica": {"priority": 1, "constraints": 0}}), {}),
        (({}, fresh(5), ["a", "b"], [], [], {"primary": {"priority": 0, "constraints": 1},
                                             "replica": {"priority": 1, "constraints": 5}}), {}),
        (({}, fresh(40), ["n%d" % i for i in range(12)], [], [], {"primary": {"priority": 0, "constraints": 1},
                                                                  "replica": {"priority": 1, "constraints": 8}}), {}),
        (({}, fresh(50), ["n%d" % i for i in range(7)], [], [], {"a": {"priority": 0, "constraints": 1},
                                                                 "b": {"priority": 1, "constraints": 2},
                                                                 "c": {"priority": 2, "constraints": 1}}), {}),
        ((prev, prev, ["n0", "n1", "n2", "n3"], ["n0"], ["n3"], M1), {}),
        (({}, fresh(64), n24, [], [], {"primary": {"priority": 0, "constraints": 1}, "replica": {"priority": 1, "constraints": 3}}),
         {"node_hierarchy": hier24, "hierarchy_rules": {"replica": [{"includeLevel": 1, "excludeLevel": 0}]}}),
        (({}, fresh(64), n24, [], [], M1),
         {"node_hierarchy": hier24, "hierarchy_rules": {"replica": [{"includeLevel": 2, "excludeLevel": 1},
                                                                    {"includeLevel": 3, "excludeLevel": 2}]}}),
        (({}, fresh(64), n24, [], [], M1),
         {"node_hierarchy": hier24, "hierarchy_rules": {"primary": [{"includeLevel": 3, "excludeLevel": 0}],
                                                        "replica": [{"includeLevel": 2, "excludeLevel": 1}]}}),
    ]
    return cases


def sharded_cases():
    """Problems whose replica pass runs as region chains over 2..8 regions (enough for plans sharded over up to
    8 ranks): config 3's shape, a small-rack tree with 8 zones, and config 5's ingredients (Zipf partition
    weights, node weights, stickiness) on a tree of 5 zones.  Returns (problems, rebalance case, its options):
    problems[-1] is the initial plan of the rebalance case."""
    from blance_amd import synth
    cases = [synth.config_flat(3, P=300, N=256), synth.config_flat(3, P=700, N=300), synth.config_flat(2, P=300, N=20)]
    c8 = synth.config_case(3, P=500, N=96)
    c8["nodeHierarchy"] = synth.hierarchy_names(96, rack=4, racks_per_zone=3, zones_per_dc=4)
    cases.append(synth.case_to_flat(c8))
    c = synth.rebalance_case(P=300, N=60, hierarchy=True)
    c["nodeHierarchy"] = synth.hierarchy_names(60, rack=3, racks_per_zone=4, zones_per_dc=2)
    fresh = {p: {"name": p, "nodesByState": {}} for p in c["partitions"]}
    opts = dict(partition_weights=c["partitionWeights"], state_stickiness=c["stateStickiness"],
                node_weights=c["nodeWeights"], node_hierarchy=c["nodeHierarchy"], hierarchy_rules=c["hierarchyRules"])
    cases.append(problem.build_problem({}, fresh, c["oldNodes"], [], c["oldNodes"], c["model"], **opts))
    return cases, c, opts
