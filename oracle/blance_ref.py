"""ORACLE (test infrastructure only) -- literal, string-keyed CPU restatement of
couchbase/blance's PlanNextMap path.

This file is the *checker*, never the product: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it.  It follows the reference line by
line (citations are to /root/reference/<file>:<line>), keeps Go's observable
semantics (nil vs empty slices, nil vs empty maps, reflect.DeepEqual, map-key
presence) and uses pure-Python loops, so it is only meant for small cases.  The
fast id-based restatement lives in oracle/blance_oracle.c; the two are
cross-checked against each other and both are pinned by the reference's own 69
golden cases (tests/golden/planner_cases.json, transcribed from plan_test.go /
control_test.go by tools/extract_golden.py).

Go values are modelled as: []string -> list or None (nil); map -> dict or None;
*Partition -> Partition or None.
"""
import re

MAX_ITERATIONS_PER_PLAN = 10          # plan.go:21


class Partition:
    """api.go:28-36"""
    __slots__ = ("name", "nodes_by_state")

    def __init__(self, name, nodes_by_state):
        self.name = name
        self.nodes_by_state = nodes_by_state

    def to_json(self):
        return {"name": self.name, "nodesByState": self.nodes_by_state}

    def __repr__(self):
        return "Partition(%r, %r)" % (self.name, self.nodes_by_state)


def partition_map_from_json(d):
    if d is None:
        return None
    out = {}
    for k, v in d.items():
        nbs = v.get("nodesByState")
        out[k] = Partition(v.get("name", ""), None if nbs is None else
                           {s: (None if l is None else list(l)) for s, l in nbs.items()})
    return out


def partition_map_to_json(m):
    return None if m is None else {k: p.to_json() for k, p in m.items()}


# ---------------------------------------------------------------- misc.go

def strings_to_map(arr):                        # misc.go:13-22
    if arr is None:
        return None
    return {s: True for s in arr}


def strings_remove_strings(string_arr, remove_arr):   # misc.go:27-36
    rm = strings_to_map(remove_arr) or {}
    return [s for s in (string_arr or []) if not rm.get(s, False)]


def strings_intersect_strings(a, b):            # misc.go:40-51
    bm = strings_to_map(b) or {}
    seen = {}
    rv = []
    for s in (a or []):
        if bm.get(s, False) and not seen.get(s, False):
            seen[s] = True
            rv.append(s)
    return rv


def strings_deduplicate(a):                     # misc.go:55-66
    seen = set()
    rv = []
    for s in a:
        if s in seen:
            continue
        seen.add(s)
        rv.append(s)
    return rv


# ---------------------------------------------------------------- Go helpers

def go_append_copy(nodes):
    """append([]string(nil), nodes...) -- nil when nodes is empty."""
    if not nodes:
        return None
    return list(nodes)


_ATOI_RE = re.compile(r"^[+-]?[0-9]+$")


def go_atoi(s):
    """strconv.Atoi: optional sign, decimal digits, int64 range; else error."""
    if not _ATOI_RE.match(s):
        return None
    v = int(s)
    if v < -(1 << 63) or v > (1 << 63) - 1:
        return None
    return v


def go_pad10(n):
    return "%10d" % n                           # fmt.Sprintf("%10d", n): space padded


def deep_equal_partition(a, b):
    """reflect.DeepEqual on two *Partition (plan.go:38)."""
    if a is None or b is None:
        return a is None and b is None
    if a.name != b.name:
        return False
    ma, mb = a.nodes_by_state, b.nodes_by_state
    if ma is None or mb is None:
        return ma is None and mb is None
    if len(ma) != len(mb):
        return False
    for k, va in ma.items():
        if k not in mb:
            return False
        vb = mb[k]
        if va is None or vb is None:
            if not (va is None and vb is None):
                return False
            continue
        if va != vb:
            return False
    return True


# ---------------------------------------------------------------- plan.go bookkeeping

def copy_nodes_by_state(nbs):                   # plan.go:345-351
    rv = {}
    for state, nodes in (nbs or {}).items():
        rv[state] = go_append_copy(nodes)
    return rv


def to_array_copy(m):                           # plan.go:334-343
    return [Partition(p.name, copy_nodes_by_state(p.nodes_by_state)) for p in m.values()]


def adjust_state_node_counts(state_node_counts, state, nodes, amt):   # plan.go:353-363
    for node in (nodes or []):
        s = state_node_counts.get(state)
        if s is None:
            s = {}
            state_node_counts[state] = s
        s[node] = s.get(node, 0) + amt


def count_state_nodes(partition_map, partition_weights):   # plan.go:374-399
    rv = {}
    for pname, partition in (partition_map or {}).items():
        for state, nodes in (partition.nodes_by_state or {}).items():
            s = rv.get(state)
            if s is None:
                s = {}
                rv[state] = s
            for node in (nodes or []):
                w = 1
                if partition_weights is not None and pname in partition_weights:
                    w = partition_weights[pname]
                s[node] = s.get(node, 0) + w
    return rv


def remove_nodes_from_nodes_by_state(nbs, remove_nodes, cb):   # plan.go:408-421
    rv = {}
    for state, nodes in (nbs or {}).items():
        if cb is not None:
            cb(state, strings_intersect_strings(nodes, remove_nodes))
        rv[state] = strings_remove_strings(nodes, remove_nodes)
    return rv


def flatten_nodes_by_state(nbs):                # plan.go:425-431
    rv = []
    for nodes in (nbs or {}).values():
        rv.extend(nodes or [])
    return rv


# ---------------------------------------------------------------- orderings

def state_name_less(model, a, b):               # plan.go:459-470
    if model is not None and model.get(a) is not None and model.get(b) is not None \
            and model[a]["priority"] < model[b]["priority"]:
        return True
    return a < b


def state_order_is_consistent(model, names):
    """App. B-9: the comparator is only a strict weak order when priority order
    never contradicts name order.  Returns True when every pair is consistent."""
    for a in names:
        for b in names:
            if a != b and state_name_less(model, a, b) and state_name_less(model, b, a):
                return False
    return True


def sort_state_names(model):                    # plan.go:437-447
    """With a consistent comparator (see above) the order is unique; we use an
    insertion sort driven by the reference's Less, which is what Go's sort.Sort
    runs for <= 12 elements."""
    names = list((model or {}).keys())
    for i in range(1, len(names)):
        j = i
        while j > 0 and state_name_less(model, names[j], names[j - 1]):
            names[j], names[j - 1] = names[j - 1], names[j]
            j -= 1
    return names


def partition_sort_score(p, state_name, prev_map, nodes_to_remove, nodes_to_add,
                         partition_weights):    # plan.go:519-562
    name = p.name
    name_str = name
    n = go_atoi(name)
    if n is not None and n >= 0:
        name_str = go_pad10(n)
    w = 1
    if partition_weights is not None and name in partition_weights:
        w = partition_weights[name]
    w_str = go_pad10(999999999 - w)
    if prev_map is not None and nodes_to_remove is not None and len(nodes_to_remove) > 0:
        last = prev_map.get(name)
        if last is None:
            raise RuntimeError("reference panics: nil *Partition deref (plan.go:545)")
        lpnbs = (last.nodes_by_state or {}).get(state_name)
        if lpnbs is not None and len(strings_intersect_strings(lpnbs, nodes_to_remove)) > 0:
            return ["0", w_str, name_str]
    if nodes_to_add is not None:
        fnbs = flatten_nodes_by_state(p.nodes_by_state)
        if len(strings_intersect_strings(fnbs, nodes_to_add)) <= 0:
            return ["1", w_str, name_str]
    return ["2", w_str, name_str]


def sort_partitions(arr, state_name, prev_map, nodes_to_remove, nodes_to_add, partition_weights):
    """sort.Sort(&partitionSorter{...}) -- plan.go:495-513.  The comparator is a
    strict total order (names are unique), so any sort gives the same result."""
    def key(p):
        return (partition_sort_score(p, state_name, prev_map, nodes_to_remove, nodes_to_add,
                                     partition_weights), p.name)
    arr.sort(key=key)


# ---------------------------------------------------------------- hierarchy

def map_parents_to_map_children(map_parents):   # plan.go:703-717
    rv = {}
    for child in sorted((map_parents or {}).keys()):
        rv.setdefault(map_parents[child], []).append(child)
    return rv


def find_ancestor(node, map_parents, level):    # plan.go:755-762
    while level > 0:
        node = (map_parents or {}).get(node, "")
        level -= 1
    return node


def find_leaves(node, map_children):            # plan.go:764-774
    children = (map_children or {}).get(node)
    if not children:
        return [node]
    rv = []
    for c in children:
        rv.extend(find_leaves(c, map_children))
    return rv


def include_exclude_nodes(node, inc, exc, map_parents, map_children):   # plan.go:723-734
    inc_nodes = find_leaves(find_ancestor(node, map_parents, inc), map_children)
    exc_nodes = find_leaves(find_ancestor(node, map_parents, exc), map_children)
    return strings_remove_strings(inc_nodes, exc_nodes)


def include_exclude_nodes_intersect(nodes, inc, exc, map_parents, map_children):   # plan.go:738-753
    rv = None
    for node in nodes:
        res = include_exclude_nodes(node, inc, exc, map_parents, map_children)
        if rv is None or len(rv) == 0:
            rv = res
            continue
        rv = strings_intersect_strings(rv, res)
    return rv if rv is not None else []


# ---------------------------------------------------------------- booster

def booster_cbgt(w, stickiness):                # control_test.go:19-26
    score = float(-w)
    if score < stickiness:
        score = stickiness
    return score


BOOSTERS = {None: None, "cbgt": booster_cbgt}


# ---------------------------------------------------------------- the planner

class Options:
    """api.go:183-190"""

    def __init__(self, model_state_constraints=None, partition_weights=None,
                 state_stickiness=None, node_weights=None, node_hierarchy=None,
                 hierarchy_rules=None):
        self.model_state_constraints = model_state_constraints
        self.partition_weights = partition_weights
        self.state_stickiness = state_stickiness
        self.node_weights = node_weights
        self.node_hierarchy = node_hierarchy
        self.hierarchy_rules = hierarchy_rules


def node_score(node, state_name, partition, num_partitions, top_priority_node,
               state_node_counts, node_to_node_counts, node_partition_counts,
               node_weights, stickiness, booster):   # plan.go:634-689
    lower_priority_balance_factor = 0.0
    if node_to_node_counts is not None and num_partitions > 0:
        m = node_to_node_counts.get(top_priority_node)
        if m is not None:
            lower_priority_balance_factor = float(m.get(node, 0)) / float(num_partitions)
    filled_factor = 0.0
    if node_partition_counts is not None and num_partitions > 0:
        if node in node_partition_counts:
            filled_factor = (0.001 * float(node_partition_counts[node])) / float(num_partitions)
    current_factor = 0.0
    if partition is not None:
        for state_node in ((partition.nodes_by_state or {}).get(state_name) or []):
            if state_node == node:
                current_factor = stickiness
    r = 0.0
    if state_node_counts is not None:
        node_counts = state_node_counts.get(state_name)
        if node_counts is not None:
            r = float(node_counts.get(node, 0))
    r = r + lower_priority_balance_factor
    r = r + filled_factor
    if node_weights is not None and node in node_weights:
        w = node_weights[node]
        if w > 0:
            r = r / float(w)
        elif w < 0 and booster is not None:
            r += booster(w, current_factor)
    r = r - current_factor
    return r


def plan_next_map_inner(prev_map, partitions_to_assign, nodes_all, nodes_to_remove,
                        nodes_to_add, model, opts, booster):   # plan.go:60-331
    partition_warnings = {}
    node_positions = {}
    for i, node in enumerate(nodes_all or []):
        node_positions[node] = i
    nodes_next = strings_remove_strings(nodes_all, nodes_to_remove)
    hierarchy_children = map_parents_to_map_children(opts.node_hierarchy)

    next_partitions = to_array_copy(partitions_to_assign or {})
    for partition in next_partitions:
        partition.nodes_by_state = remove_nodes_from_nodes_by_state(
            partition.nodes_by_state, nodes_to_remove, None)
    sort_partitions(next_partitions, "", None, None, None, None)      # plan.go:89

    state_node_counts = count_state_nodes(prev_map, opts.partition_weights)
    num_partitions = len(prev_map or {})

    def find_best_nodes(partition, state_name, constraints, node_to_node_counts):   # plan.go:98-248
        stickiness = 1.5
        if opts.partition_weights is not None:
            if partition.name in opts.partition_weights:
                stickiness = float(opts.partition_weights[partition.name])
            elif opts.state_stickiness is not None:
                if state_name in opts.state_stickiness:
                    stickiness = float(opts.state_stickiness[state_name])

        node_partition_counts = {}
        for node_counts in state_node_counts.values():
            for node, c in node_counts.items():
                node_partition_counts[node] = node_partition_counts.get(node, 0) + c

        top_priority_state_name = ""
        for sname in sort_state_names(model):   # Go: map order; ties unspecified
            if top_priority_state_name == "" or \
                    model[sname]["priority"] < model[top_priority_state_name]["priority"]:
                top_priority_state_name = sname

        top_priority_node = ""
        tpsn = (partition.nodes_by_state or {}).get(top_priority_state_name)
        if tpsn:
            top_priority_node = tpsn[0]

        state_priority = model[state_name]["priority"]
        candidate_nodes = go_append_copy(nodes_next)

        def exclude_higher_priority_nodes(remaining):   # plan.go:146-154
            for sname, snodes in partition.nodes_by_state.items():
                if sname not in model:
                    raise RuntimeError("reference panics: nil model state deref (plan.go:148)")
                if model[sname]["priority"] < state_priority:
                    remaining = strings_remove_strings(remaining, snodes)
            return remaining

        candidate_nodes = exclude_higher_priority_nodes(candidate_nodes)

        def sort_nodes(nodes):                  # plan.go:158-172 / :198-212
            if not nodes:
                return nodes
            nodes.sort(key=lambda n: (node_score(
                n, state_name, partition, num_partitions, top_priority_node,
                state_node_counts, node_to_node_counts, node_partition_counts,
                opts.node_weights, stickiness, booster), node_positions.get(n, 0)))
            return nodes

        candidate_nodes = sort_nodes(candidate_nodes)

        if opts.hierarchy_rules is not None:    # plan.go:174-226
            hierarchy_nodes = []
            for rule in (opts.hierarchy_rules.get(state_name) or []):
                h = top_priority_node
                if h == "" and len(hierarchy_nodes) > 0:
                    h = hierarchy_nodes[0]
                for _ in range(constraints):
                    hc = include_exclude_nodes_intersect(
                        [h] + hierarchy_nodes, rule["includeLevel"], rule["excludeLevel"],
                        opts.node_hierarchy, hierarchy_children)
                    hc = strings_intersect_strings(hc, nodes_next)
                    hc = exclude_higher_priority_nodes(hc)
                    hc = sort_nodes(hc)
                    if len(hc) > 0:
                        hierarchy_nodes.append(hc[0])
                    elif candidate_nodes is not None and len(candidate_nodes) > 0:
                        hierarchy_nodes.append(candidate_nodes[0])
            candidate_nodes = hierarchy_nodes + (candidate_nodes or [])
            candidate_nodes = strings_deduplicate(candidate_nodes)

        if len(candidate_nodes or []) >= constraints:
            candidate_nodes = candidate_nodes[0:constraints]
        else:
            partition_warnings.setdefault(partition.name, []).append(
                "could not meet constraints: %d, stateName: %s, partitionName: %s"
                % (constraints, state_name, partition.name))

        for c in (candidate_nodes or []):       # plan.go:238-245
            m = node_to_node_counts.get(top_priority_node)
            if m is None:
                m = {}
                node_to_node_counts[top_priority_node] = m
            m[c] = m.get(c, 0) + 1
        return candidate_nodes

    def assign_state_to_partitions(state_name, constraints):   # plan.go:253-303
        arr = list(next_partitions)
        sort_partitions(arr, state_name, prev_map, nodes_to_remove, nodes_to_add,
                        opts.partition_weights)
        node_to_node_counts = {}
        for partition in arr:
            pw = 1
            if opts.partition_weights is not None and partition.name in opts.partition_weights:
                pw = opts.partition_weights[partition.name]

            def dec(sname, nodes, pw=pw):
                adjust_state_node_counts(state_node_counts, sname, nodes, -pw)

            nodes_to_assign = find_best_nodes(partition, state_name, constraints,
                                              node_to_node_counts)
            partition.nodes_by_state = remove_nodes_from_nodes_by_state(
                partition.nodes_by_state, partition.nodes_by_state.get(state_name), dec)
            partition.nodes_by_state = remove_nodes_from_nodes_by_state(
                partition.nodes_by_state, nodes_to_assign, dec)
            partition.nodes_by_state[state_name] = nodes_to_assign
            adjust_state_node_counts(state_node_counts, state_name, nodes_to_assign, pw)

    for state_name in sort_state_names(model):  # plan.go:307-324
        constraints = 0
        ms = (model or {}).get(state_name)
        if ms is not None:
            constraints = ms["constraints"]
        if opts.model_state_constraints is not None and state_name in opts.model_state_constraints:
            constraints = opts.model_state_constraints[state_name]
        if constraints > 0:
            assign_state_to_partitions(state_name, constraints)

    rv = {}
    for partition in next_partitions:
        rv[partition.name] = partition
    return rv, partition_warnings


def plan_next_map_ex(prev_map, partitions_to_assign, nodes_all, nodes_to_remove, nodes_to_add,
                     model, opts=None, booster=None, max_iterations=MAX_ITERATIONS_PER_PLAN,
                     info=None):
    """plan.go:23-58.  Mutates prev_map / partitions_to_assign like the reference.
    `booster` is a callable or a key of BOOSTERS.  `info`, if a dict, receives
    {"iterations": I, "converged": bool}."""
    if opts is None:
        opts = Options()
    if not callable(booster):
        booster = BOOSTERS[booster]
    next_map, warnings = None, None
    iterations, converged = 0, False
    for _ in range(max_iterations):
        next_map, warnings = plan_next_map_inner(prev_map, partitions_to_assign, nodes_all,
                                                 nodes_to_remove, nodes_to_add, model, opts,
                                                 booster)
        iterations += 1
        not_match = False
        for partition in next_map.values():
            if not deep_equal_partition(partition, (prev_map or {}).get(partition.name)):
                not_match = True
                break
        if not not_match:
            converged = True
            break
        for partition in next_map.values():
            prev_map[partition.name] = partition
            partitions_to_assign[partition.name] = partition
        nodes_all = strings_remove_strings(nodes_all, nodes_to_remove)
        nodes_to_remove = []
        nodes_to_add = []
    if info is not None:
        info["iterations"] = iterations
        info["converged"] = converged
    return next_map, warnings


def run_case(case, info=None):
    """Run one fixture of tests/golden/planner_cases.json; returns
    (result_json, warnings_dict)."""
    prev = partition_map_from_json(case["prevMap"])
    assign = prev if case.get("aliased") else partition_map_from_json(case["partitionsToAssign"])
    opts = Options(case.get("modelStateConstraints"), case.get("partitionWeights"),
                   case.get("stateStickiness"), case.get("nodeWeights"),
                   case.get("nodeHierarchy"), case.get("hierarchyRules"))
    r, w = plan_next_map_ex(prev, assign, case["nodesAll"], case["nodesToRemove"],
                            case["nodesToAdd"], case["model"], opts, case.get("booster"),
                            info=info)
    return partition_map_to_json(r), w


def count_warnings(warnings, mode):
    if mode == "messages":                      # plan_test.go:1599-1608
        return sum(len(v) for v in (warnings or {}).values())
    return len(warnings or {})                  # plan_test.go:1738
