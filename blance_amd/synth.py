"""Synthetic PlanNextMapEx() inputs of the shapes BASELINE.json names
(SURVEY.md section 8(d)), at two levels:

  *_case(...)  -> the API-level arguments (dicts of strings), for any size the
                  host interning layer can chew (used by the parity tests);
  *_flat(...)  -> the same problem built directly as the int32 SoA of
                  include/blance_hip.h with numpy (used by bench.py at 1M
                  partitions, where interning a million Go-style map entries
                  in Python would dominate).

tests/test_synth.py checks that both routes give byte-identical problems.
"""
import numpy as np

from . import abi, problem

MODEL_P1 = {"primary": {"priority": 0, "constraints": 1}}
MODEL_P1R1 = {"primary": {"priority": 0, "constraints": 1},
              "replica": {"priority": 1, "constraints": 1}}
MODEL_P1R2 = {"primary": {"priority": 0, "constraints": 1},
              "replica": {"priority": 1, "constraints": 2}}


def _node_names(n, width):
    return [("n%0" + str(width) + "d") % i for i in range(n)]


def _fresh_partitions(P):
    return {str(i): {"name": str(i), "nodesByState": {}} for i in range(P)}


def hierarchy_names(N, rack=16, racks_per_zone=8, zones_per_dc=8, width=4):
    """3-level rack/zone/DC tree of config 3: NodeHierarchy child -> parent."""
    nodes = _node_names(N, width)
    hier = {}
    n_racks = (N + rack - 1) // rack
    n_zones = (n_racks + racks_per_zone - 1) // racks_per_zone
    for i, n in enumerate(nodes):
        hier[n] = "r%03d" % (i // rack)
    for r in range(n_racks):
        hier["r%03d" % r] = "z%02d" % (r // racks_per_zone)
    for z in range(n_zones):
        hier["z%02d" % z] = "d%d" % (z // zones_per_dc)
    return hier


def config_case(cfg, P=None, N=None):
    """API-level arguments of BASELINE.json config `cfg` (1, 2 or 3); P and N
    override the named size (for reduced-scale parity tests)."""
    if cfg == 1:
        P, N = P or 64, N or 4
        nodes = ["n%d" % i for i in range(N)]
        model, hier, rules = MODEL_P1, None, None
    elif cfg == 2:
        P, N = P or 65536, N or 256
        nodes = _node_names(N, 3)
        model, hier, rules = MODEL_P1R1, None, None
    elif cfg == 3:
        P, N = P or 1048576, N or 4096
        nodes = _node_names(N, 4)
        model = MODEL_P1R2
        hier = hierarchy_names(N)
        rules = {"replica": [{"includeLevel": 2, "excludeLevel": 1}]}
    else:
        raise ValueError("config %r" % (cfg,))
    return {"prevMap": {}, "partitionsToAssign": _fresh_partitions(P), "aliased": False,
            "nodesAll": nodes, "nodesToRemove": [], "nodesToAdd": list(nodes), "model": model,
            "nodeHierarchy": hier, "hierarchyRules": rules}


def case_to_flat(c, max_iterations=10):
    prev = c["prevMap"]
    assign = prev if c.get("aliased") else c["partitionsToAssign"]
    return problem.build_problem(
        prev, assign, c["nodesAll"], c["nodesToRemove"], c["nodesToAdd"], c["model"],
        c.get("modelStateConstraints"), c.get("partitionWeights"), c.get("stateStickiness"),
        c.get("nodeWeights"), c.get("nodeHierarchy"), c.get("hierarchyRules"), c.get("booster"),
        max_iterations=max_iterations)


def config_flat(cfg, P=None, N=None, max_iterations=10, rotate=0):
    """The flat problem of config `cfg` without going through strings.  rotate = s (config 3): nodesAll starts at the s-th
    name -- node id i carries the name n[(i + s) mod N], so every per-node hierarchy table differs from the s = 0 instance;
    for s a multiple of the zone size (128) zones and racks stay aligned blocks of ids and the plan, as ids, is the same
    (bench.py: the replicas of an N-GPU run plan instances rotated by 512 rank)."""
    if cfg == 1:
        P, N = P or 64, N or 4
        prios, cons = [0], [1]
        hier = False
    elif cfg == 2:
        P, N = P or 65536, N or 256
        prios, cons = [0, 1], [1, 1]
        hier = False
    elif cfg == 3:
        P, N = P or 1048576, N or 4096
        prios, cons = [0, 1], [1, 2]
        hier = True
    else:
        raise ValueError("config %r" % (cfg,))
    M = len(prios)
    fp = abi.FlatProblem()
    z8 = lambda n: np.zeros(n, dtype=np.uint8)
    z32 = lambda n: np.zeros(n, dtype=np.int32)
    fp.set("state_priority", prios)
    fp.set("state_constraints", cons)
    fp.set("state_stickiness", z32(M))
    fp.set("state_has_stickiness", z8(M))
    fp.set("node_removed", z8(N))
    fp.set("node_added", np.ones(N, dtype=np.uint8))
    fp.set("node_weight", z32(N))
    fp.set("node_has_weight", z8(N))
    fp.set("part_order", np.arange(P, dtype=np.int32))        # names "0".."P-1": numeric order
    fp.set("part_weight", np.ones(P, dtype=np.int32))
    fp.set("part_has_weight", z8(P))
    fp.set("part_in_prev", z8(P))
    fp.set("part_prev_never_equal", z8(P))
    fp.set("assign_off", z32(P * M + 1))
    fp.set("assign_nodes", z32(0))
    fp.set("assign_kind", z8(P * M))
    fp.set("prev_off", z32(P * M + 1))
    fp.set("prev_nodes", z32(0))
    fp.set("prev_kind", z8(P * M))
    for k in ("load_state", "load_node", "load_weight"):
        fp.set(k, z32(0))
    fp.set("load_first_sweep_only", z8(0))
    VX, v_empty = 0, 0
    if hier:
        rack, rpz, zpd = 16, 8, 8
        n_racks = (N + rack - 1) // rack
        n_zones = (n_racks + rpz - 1) // rpz
        n_dcs = (n_zones + zpd - 1) // zpd
        r0, z0, d0 = N, N + n_racks, N + n_racks + n_zones
        v_empty = d0 + n_dcs
        VX = v_empty + 1
        parent = np.full(VX, v_empty, dtype=np.int32)
        nid = (np.arange(N) + int(rotate)) % N      # the name (= leaf) index of node id i
        parent[:N] = r0 + nid // rack
        parent[r0:z0] = z0 + np.arange(n_racks) // rpz
        parent[z0:d0] = d0 + np.arange(n_zones) // zpd
        lo = np.zeros(VX, dtype=np.int32)
        hi = np.zeros(VX, dtype=np.int32)
        lo[:N] = nid
        hi[:N] = nid + 1
        rk = np.arange(n_racks)
        lo[r0:z0] = rk * rack
        hi[r0:z0] = np.minimum((rk + 1) * rack, N)
        zn = np.arange(n_zones)
        lo[z0:d0] = zn * rack * rpz
        hi[z0:d0] = np.minimum((zn + 1) * rack * rpz, N)
        dc = np.arange(n_dcs)
        lo[d0:v_empty] = dc * rack * rpz * zpd
        hi[d0:v_empty] = np.minimum((dc + 1) * rack * rpz * zpd, N)
        lo[v_empty] = N
        hi[v_empty] = N + 1
        fp.set("rule_off", [0, 0, 1])
        fp.set("rule_inc", [2])
        fp.set("rule_exc", [1])
        fp.set("vertex_parent", parent)
        fp.set("vertex_leaf_lo", lo)
        fp.set("vertex_leaf_hi", hi)
        fp.set("node_leaf_pos", nid.astype(np.int32))
    else:
        fp.set("rule_off", z32(M + 1))
        fp.set("rule_inc", z32(0))
        fp.set("rule_exc", z32(0))
        fp.set("vertex_parent", z32(0))
        fp.set("vertex_leaf_lo", z32(0))
        fp.set("vertex_leaf_hi", z32(0))
        fp.set("node_leaf_pos", np.full(N, -1, dtype=np.int32))
    fp.scalars.update(n_nodes=N, n_nodes_ext=N, n_states=M, n_parts=P, n_prev=0, n_loads=0,
                      n_rules=1 if hier else 0, n_vertices=VX, max_iterations=int(max_iterations),
                      partition_weights_nil=1, nodes_to_add_nil=0,
                      hierarchy_rules_nil=0 if hier else 1, booster_kind=abi.BOOSTER_NONE,
                      top_state=0, vertex_empty=v_empty)
    return fp


def assignments(fp):
    """Metric numerator: node slots in the returned map."""
    k = np.maximum(fp.arrays["state_constraints"].astype(np.int64), 0)
    return int(fp.scalars["n_parts"]) * int(k.sum())


def _state_has_rules(fp, m):
    roff = fp.arrays["rule_off"]
    return (not fp.scalars["hierarchy_rules_nil"]) and int(roff[m + 1]) > int(roff[m])


def algorithmic_bytes_per_state(fp):
    """one state pass reads P * (N*(16 + 4*k*[rules]) + 40)
    bytes in the reference's dense formulation (0 for states without a pass)."""
    N = int(fp.scalars["n_nodes"])
    P = int(fp.scalars["n_parts"])
    cons = fp.arrays["state_constraints"]
    out = []
    for m in range(int(fp.scalars["n_states"])):
        k = int(cons[m])
        out.append(P * (N * (16 + (4 * k if _state_has_rules(fp, m) else 0)) + 40) if k > 0 else 0)
    return out


def algorithmic_bytes_per_sweep(fp):
    return sum(algorithmic_bytes_per_state(fp))


def pass_kernel_states(fp):
    """States whose pass is ONE kernel launch (k_pass_chain / k_pass_seq); single
    constraint states without hierarchy rules go through the flat bulk driver."""
    cons = fp.arrays["state_constraints"]
    return [m for m in range(int(fp.scalars["n_states"]))
            if int(cons[m]) > 0 and (_state_has_rules(fp, m) or int(cons[m]) != 1)]


def rebalance_case(P=4096, N=128, seed=5, remove_frac=0.1, add_frac=0.1, hierarchy=False):
    """BASELINE.json config 5 in miniature: Zipf-weighted partitions,
    heterogeneous node weights, stickiness, and a rebalance that removes / adds a
    tenth of the nodes.  Returns the API-level arguments WITHOUT a prevMap: the
    caller plans once over the old nodes and feeds that plan back (see
    tests/test_hip_parity.py::test_config5_miniature)."""
    import random
    rng = random.Random(seed)
    nodes = ["n%04d" % i for i in range(N)]
    order = nodes[:]
    rng.shuffle(order)
    n_rm = max(1, int(N * remove_frac))
    n_add = max(1, int(N * add_frac))
    to_add = sorted(order[:n_add])
    to_remove = sorted(order[n_add:n_add + n_rm])
    old_nodes = [n for n in nodes if n not in to_add]
    ranks = list(range(1, P + 1))
    rng.shuffle(ranks)
    weights = {str(i): max(1, min(1000, 1000 // ranks[i])) for i in range(P)}
    node_weights = {n: rng.choice([1, 1, 2, 4]) for n in nodes}
    case = {"model": MODEL_P1R2, "nodesAll": nodes, "oldNodes": old_nodes, "nodesToRemove": to_remove,
            "nodesToAdd": to_add, "partitionWeights": weights, "nodeWeights": node_weights,
            "stateStickiness": {"primary": 100, "replica": 10}, "partitions": [str(i) for i in range(P)],
            "nodeHierarchy": None, "hierarchyRules": None}
    if hierarchy:
        case["nodeHierarchy"] = hierarchy_names(N)
        case["hierarchyRules"] = {"replica": [{"includeLevel": 2, "excludeLevel": 1}]}
    return case


def _config5_opts(c):
    return dict(partition_weights=c["partitionWeights"], state_stickiness=c["stateStickiness"],
                node_weights=c["nodeWeights"], node_hierarchy=c["nodeHierarchy"], hierarchy_rules=c["hierarchyRules"])


def config5_initial(P=1048576, N=4096, hierarchy=False):
    """BASELINE.json config 5, first half: the plan over the old nodes that the rebalance
    starts from (fresh partitions, weights, stickiness).  Goes through the interning layer."""
    c = rebalance_case(P=P, N=N, hierarchy=hierarchy)
    fresh = {p: {"name": p, "nodesByState": {}} for p in c["partitions"]}
    return problem.build_problem({}, fresh, c["oldNodes"], [], c["oldNodes"], c["model"], **_config5_opts(c))


def config5_rebalance(fp1, res1, P=1048576, N=4096, hierarchy=False):
    """Config 5 proper: prevMap = partitionsToAssign = the plan `res1` of config5_initial's
    problem `fp1`; a tenth of the nodes removed, a tenth added."""
    c = rebalance_case(P=P, N=N, hierarchy=hierarchy)
    plan1, _ = problem.decode_result(fp1, res1)
    return problem.build_problem(plan1, plan1, c["nodesAll"], c["nodesToRemove"], c["nodesToAdd"], c["model"],
                                 **_config5_opts(c))


def replan_problem(fp, res):
    """The call PlanNextMap(prevMap = partitionsToAssign = the result `res` of `fp`, same nodes, nodesToRemove = nil,
    nodesToAdd = nil, same model / options) as a flat problem, without going through strings: the result's lists ARE
    the two input maps (plan.go:49-52 stores exactly them).  For problems whose partitions are all in
    partitionsToAssign and that carry no prevMap-only loads (the generators of this file)."""
    if int(fp.n_loads) != 0:
        raise ValueError("replan_problem: problems with prevMap-only partitions are not handled")
    P, M = int(fp.n_parts), int(fp.n_states)
    fp2 = abi.FlatProblem()
    fp2.scalars = dict(fp.scalars)
    for name, a in fp.arrays.items():
        fp2.set(name, a.copy())
    fp2.node_names, fp2.state_names, fp2.part_names = fp.node_names, fp.state_names, fp.part_names
    total = int(res.out_off[P * M]) if P * M else 0
    for pre in ("assign", "prev"):
        fp2.set(pre + "_off", np.asarray(res.out_off[:P * M + 1]))
        fp2.set(pre + "_nodes", np.asarray(res.out_nodes[:total]))
        fp2.set(pre + "_kind", np.asarray(res.out_kind[:P * M]))
    fp2.set("part_in_prev", np.ones(P, dtype=np.uint8))
    fp2.set("part_prev_never_equal", np.zeros(P, dtype=np.uint8))
    fp2.set("node_removed", np.zeros(len(fp.node_removed), dtype=np.uint8))
    fp2.set("node_added", np.zeros(len(fp.node_added), dtype=np.uint8))
    fp2.scalars.update(n_prev=P, nodes_to_add_nil=1)
    return fp2


# ---- config 3's general regime ("time the general hierarchical regime") ------------------------------
# The headline instance is the most regular one there is (fresh plan, numeric names, no weights).  Two more workloads
# of the same size are timed beside it by bench.py: (a) the rebalance of config 3's plan after every tenth node left,
# (b) config 3 with non-numeric, scrambled partition names and Zipf partition weights.

def _scrambled_names_hash(P):
    """name_i = "vb%08x" % h_i with h_i = (i * 2246822519 + 374761393) mod 2^32: distinct, fixed width, so the
    reference's name key (plan.go:519-540: Atoi fails -> the raw name) orders the partitions by h_i."""
    return (np.arange(P, dtype=np.uint64) * np.uint64(2246822519) + np.uint64(374761393)) & np.uint64(0xFFFFFFFF)


def _zipf_weights(P):
    """weight_i = clamp(1000 // rank_i, 1, 1000), rank a fixed pseudo-random permutation of 1..P (SURVEY.md 8d, config 5's law)."""
    h = (np.arange(P, dtype=np.uint64) * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
    rank = np.empty(P, dtype=np.int64)
    rank[np.argsort(h, kind="stable")] = np.arange(1, P + 1)
    return np.clip(1000 // rank, 1, 1000).astype(np.int32)


def config3_named_weighted_case(P, N):
    """API-level arguments of workload (b) (for reduced sizes: tests/test_synth.py checks it against the flat builder)."""
    c = config_case(3, P=P, N=N)
    names = ["vb%08x" % int(h) for h in _scrambled_names_hash(P)]
    w = _zipf_weights(P)
    c["partitionsToAssign"] = {n: {"name": n, "nodesByState": {}} for n in names}
    c["partitionWeights"] = {names[i]: int(w[i]) for i in range(P)}
    return c


def config3_named_weighted_flat(P=1048576, N=4096, max_iterations=10):
    """Workload (b) as a flat problem without going through strings.  Partition ids are the generator's indices (the
    interning layer numbers partitions in the order of the map it is given, problem.py); the static pass order is
    (weight descending, name) -- plan.go:519-540."""
    fp = config_flat(3, P=P, N=N, max_iterations=max_iterations)
    h = _scrambled_names_hash(P)
    w = _zipf_weights(P)
    fp.set("part_weight", w)
    fp.set("part_has_weight", np.ones(P, dtype=np.uint8))
    fp.set("part_order", np.lexsort((h, -w.astype(np.int64))).astype(np.int32))
    fp.scalars.update(partition_weights_nil=0)
    return fp


def config3_rebalance_flat(fp, res, every=10, which=3):
    """Workload (a): PlanNextMap(prevMap = partitionsToAssign = the plan `res` of `fp`, nodesToRemove = every `every`-th
    node (id % every == which), nodesToAdd = nil).  tests/golden/config3_full_size_properties.json holds the oracle's digest."""
    fp2 = replan_problem(fp, res)
    N = int(fp.n_nodes)
    rm = np.zeros(N, dtype=np.uint8)
    rm[np.arange(N) % every == which] = 1
    fp2.set("node_removed", rm)
    return fp2
