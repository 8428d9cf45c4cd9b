"""Host-side interning: PlanNextMapEx() arguments -> the flat int32 SoA problem
of include/blance_hip.h, and the flat result back to a PartitionMap.

This is the work the reference does implicitly with Go maps keyed by strings
(api.go:24-190).  Everything that depends on *strings* is settled here, once:
state pass order (plan.go:437-474), the static part of the partition order
(plan.go:519-540), the hierarchy tree as DFS leaf intervals
(plan.go:703-717,:755-774).  Inputs the reference would panic on, or whose
result depends on Go's map iteration order / sort internals, raise
Unsupported -- the device never guesses.
"""
import re

import numpy as np

from . import abi


class Unsupported(Exception):
    """Input outside the supported envelope."""


_ATOI_RE = re.compile(r"\A[+-]?[0-9]+\Z")      # strconv.Atoi: no trailing newline either
INT32_MAX = (1 << 31) - 1


def _atoi(s):                      # strconv.Atoi (plan.go:525)
    if not _ATOI_RE.match(s):
        return None
    v = int(s)
    if v < -(1 << 63) or v > (1 << 63) - 1:
        return None
    return v


def _pname(p):
    return p.Name if hasattr(p, "Name") else p.get("name", "")


def _pnbs(p):
    return p.NodesByState if hasattr(p, "NodesByState") else p.get("nodesByState")


def _mprio(ms):
    return ms.Priority if hasattr(ms, "Priority") else ms["priority"]


def _mcons(ms):
    return ms.Constraints if hasattr(ms, "Constraints") else ms["constraints"]


def _rule(r):
    if hasattr(r, "IncludeLevel"):
        return int(r.IncludeLevel), int(r.ExcludeLevel)
    return int(r["includeLevel"]), int(r["excludeLevel"])


def state_less(model, a, b):       # stateNameSorter.Less, plan.go:459-470
    if model.get(a) is not None and model.get(b) is not None and \
            _mprio(model[a]) < _mprio(model[b]):
        return True
    return a < b


def sort_state_names(model):
    names = sorted(model.keys())
    for a in names:
        if model[a] is None:
            raise Unsupported("nil *PartitionModelState for state %r" % a)
    # App. B-9: the comparator must not contradict itself
    for a in names:
        for b in names:
            if a != b and state_less(model, a, b) and state_less(model, b, a):
                raise Unsupported("state priority order contradicts state name order "
                                  "(%r vs %r): reference result depends on sort internals" % (a, b))
    out = list(names)
    for i in range(1, len(out)):
        j = i
        while j > 0 and state_less(model, out[j], out[j - 1]):
            out[j], out[j - 1] = out[j - 1], out[j]
            j -= 1
    return out


def static_partition_order(names, weights):
    """Partition ids ordered by (weight key, name key, name): the part of
    partitionSorter.Score that does not change between passes (plan.go:519-540).
    Keys are compared as the reference's space-padded strings."""
    n = len(names)
    if n == 0:
        return np.zeros(0, dtype=np.int32)
    keys = []
    for i, name in enumerate(names):
        v = _atoi(name)
        nkey = "%10d" % v if (v is not None and v >= 0) else name
        w = 1
        if weights is not None and name in weights:
            w = weights[name]
        keys.append(("%10d" % (999999999 - w), nkey, name, i))
    keys.sort()
    return np.asarray([k[3] for k in keys], dtype=np.int32)


class _Intern:
    def __init__(self):
        self.ids = {}
        self.names = []

    def add(self, name):
        i = self.ids.get(name)
        if i is None:
            i = len(self.names)
            self.ids[name] = i
            self.names.append(name)
        return i


def build_hierarchy(node_ids, n_nodes_ext, node_hierarchy):
    """NodeHierarchy (child -> parent names) -> vertex arrays.  Vertices
    0..NX-1 are the nodes.  Returns (n_vertices, vertex_empty, parent, lo, hi,
    node_leaf_pos)."""
    if "" in node_ids.ids:
        raise Unsupported('"" used as a node name')
    v = _Intern()
    v.ids = dict(node_ids.ids)
    v.names = list(node_ids.names)
    assert len(v.names) == n_nodes_ext
    hier = node_hierarchy or {}
    for child, parent in hier.items():
        v.add(child)
        v.add(parent)
    empty = v.add("")
    VX = len(v.names)
    parent = np.full(VX, empty, dtype=np.int32)       # findAncestor: missing -> ""
    children = [[] for _ in range(VX)]
    has_parent = np.zeros(VX, dtype=bool)
    for child in sorted(hier.keys()):                 # plan.go:705-715: children sorted by name
        c, p = v.ids[child], v.ids[hier[child]]
        parent[c] = p
        children[p].append(c)
        has_parent[c] = True
    lo = np.full(VX, -1, dtype=np.int32)
    hi = np.full(VX, -1, dtype=np.int32)
    pos = 0
    for root in range(VX):
        if has_parent[root]:
            continue
        stack = [(root, 0)]
        while stack:
            u, ci = stack.pop()
            if ci == 0:
                lo[u] = pos
                if not children[u]:
                    pos += 1                          # findLeaves: childless vertex is its own leaf
                    hi[u] = pos
                    continue
            if ci < len(children[u]):
                stack.append((u, ci + 1))
                stack.append((children[u][ci], 0))
            else:
                hi[u] = pos
    if (lo < 0).any() or (hi < 0).any():
        raise Unsupported("cycle in NodeHierarchy (reference recurses forever)")
    leaf_pos = np.full(n_nodes_ext, -1, dtype=np.int32)
    for n in range(n_nodes_ext):
        if not children[n]:
            leaf_pos[n] = lo[n]
    return VX, empty, parent, lo, hi, leaf_pos


def build_problem(prev_map, partitions_to_assign, nodes_all, nodes_to_remove, nodes_to_add,
                  model, model_state_constraints=None, partition_weights=None,
                  state_stickiness=None, node_weights=None, node_hierarchy=None,
                  hierarchy_rules=None, booster=None, max_iterations=10):
    """Intern one PlanNextMapEx() call.  Maps may be dicts of dict-partitions
    ({"name", "nodesByState"}) or of objects with .Name/.NodesByState."""
    fp = abi.FlatProblem()
    sc = fp.scalars
    if prev_map is None:
        if partitions_to_assign:
            raise Unsupported("nil prevMap with partitions to assign (reference panics, plan.go:50)")
        prev_map = {}
    partitions_to_assign = partitions_to_assign or {}
    model = model or {}
    nodes_all = list(nodes_all or [])

    # ---- states
    states = sort_state_names(model)
    sid = {s: i for i, s in enumerate(states)}
    M = len(states)
    prios = [int(_mprio(model[s])) for s in states]
    cons = []
    for s in states:
        k = int(_mcons(model[s]))
        if model_state_constraints is not None and s in model_state_constraints:
            k = int(model_state_constraints[s])       # plan.go:314-319
        cons.append(k)
    any_pass = any(k > 0 for k in cons)
    top_state = 0
    if M:
        mn = min(prios)
        tops = [i for i, p in enumerate(prios) if p == mn]
        if len(tops) > 1 and any_pass:
            raise Unsupported("several states share the top priority: reference picks by Go map order")
        top_state = tops[0]
    # a ModelStateConstraints key that is not in the model never runs a pass (plan.go:307)

    # ---- nodes
    nodes = _Intern()
    for n in nodes_all:
        if n in nodes.ids:
            raise Unsupported("duplicate node name %r in nodesAll (App. B-13)" % n)
        nodes.add(n)
    N = len(nodes.names)

    def nid(name):
        return nodes.add(name)

    # ---- partitions
    pnames = list(partitions_to_assign.keys())
    P = len(pnames)
    for key in pnames:
        if _pname(partitions_to_assign[key]) != key:
            raise Unsupported("partition key %r != Partition.Name" % key)
    weights_nil = partition_weights is None
    part_weight = np.ones(P, dtype=np.int64)
    part_has_weight = np.zeros(P, dtype=np.uint8)
    if not weights_nil:
        for i, name in enumerate(pnames):
            if name in partition_weights:
                part_weight[i] = int(partition_weights[name])
                part_has_weight[i] = 1
    part_in_prev = np.zeros(P, dtype=np.uint8)
    never_equal = np.zeros(P, dtype=np.uint8)

    a_off = [0]; a_nodes = []; a_kind = []
    p_off = [0]; p_nodes = []; p_kind = []
    loads = []                                        # (state, node, weight, first_only)
    abs_load = 0

    removed_set = set(nodes_to_remove or [])
    for i, name in enumerate(pnames):
        nbs = _pnbs(partitions_to_assign[name]) or {}
        for s in nbs:
            if s not in sid:
                raise Unsupported("partition %r carries state %r that is not in the model "
                                  "(reference panics at plan.go:148 once a pass runs)" % (name, s))
        for s in states:
            if s in nbs:
                lst = nbs[s]
                if lst is not None and len(set(lst)) != len(lst):
                    raise Unsupported("duplicate node inside partition %r state %r" % (name, s))
                a_kind.append(abi.LIST_NIL if lst is None else abi.LIST_SET)
                a_nodes.extend(nid(x) for x in (lst or []))
            else:
                a_kind.append(abi.LIST_ABSENT)
            a_off.append(len(a_nodes))
        prev = prev_map.get(name)
        w = int(part_weight[i])
        if prev is None:
            if name in prev_map:
                raise Unsupported("nil *Partition in prevMap")
            if removed_set and any_pass:
                raise Unsupported("nodesToRemove non-empty but partition %r is not in prevMap "
                                  "(reference panics at plan.go:545)" % name)
            for s in states:
                p_kind.append(abi.LIST_ABSENT)
                p_off.append(len(p_nodes))
            continue
        part_in_prev[i] = 1
        pn = _pnbs(prev)
        if pn is None or _pname(prev) != name:
            never_equal[i] = 1
        pn = pn or {}
        for s in states:
            if s in pn:
                lst = pn[s]
                p_kind.append(abi.LIST_NIL if lst is None else abi.LIST_SET)
                p_nodes.extend(nid(x) for x in (lst or []))
                abs_load += abs(w) * len(lst or [])
            else:
                p_kind.append(abi.LIST_ABSENT)
            p_off.append(len(p_nodes))
        for s, lst in pn.items():
            if s not in sid:
                never_equal[i] = 1
                for x in (lst or []):
                    loads.append((M, nid(x), w, 1))
                    abs_load += abs(w)
    n_prev = len(prev_map)
    for name, prev in prev_map.items():               # partitions only in prevMap
        if name in partitions_to_assign:
            continue
        if prev is None:
            raise Unsupported("nil *Partition in prevMap")
        w = 1
        if not weights_nil and name in partition_weights:
            w = int(partition_weights[name])
        for s, lst in (_pnbs(prev) or {}).items():
            for x in (lst or []):
                loads.append((sid.get(s, M), nid(x), w, 0))
                abs_load += abs(w)
    # every sweep re-adds at most the result's K slots per partition
    abs_load += int(np.abs(part_weight).sum()) * max(1, sum(max(k, 0) for k in cons)) * 2
    if abs_load > INT32_MAX or (P and int(np.abs(part_weight).max()) > INT32_MAX):
        raise Unsupported("partition weights overflow the device's int32 load tables")

    # ---- node attributes (interning may still add ext names here)
    for x in (nodes_to_remove or []):
        nid(x)
    for x in (nodes_to_add or []):
        nid(x)
    if node_weights is not None:
        for x in node_weights:
            nid(x)
    hier_rules_nil = hierarchy_rules is None
    if not hier_rules_nil and node_hierarchy:
        pass                                          # hierarchy names are vertices, not nodes
    NX = len(nodes.names)
    node_removed = np.zeros(NX, dtype=np.uint8)
    node_added = np.zeros(NX, dtype=np.uint8)
    node_weight = np.zeros(NX, dtype=np.int64)
    node_has_weight = np.zeros(NX, dtype=np.uint8)
    for x in (nodes_to_remove or []):
        node_removed[nodes.ids[x]] = 1
    for x in (nodes_to_add or []):
        node_added[nodes.ids[x]] = 1
    if node_weights is not None:
        for x, wv in node_weights.items():
            node_weight[nodes.ids[x]] = int(wv)
            node_has_weight[nodes.ids[x]] = 1
        if len(node_weights) and int(np.abs(node_weight).max()) > INT32_MAX:
            raise Unsupported("node weight outside int32")

    # ---- stickiness
    st_val = np.zeros(M, dtype=np.int64)
    st_has = np.zeros(M, dtype=np.uint8)
    if state_stickiness is not None:
        for s, v in state_stickiness.items():
            if s in sid:
                st_val[sid[s]] = int(v)
                st_has[sid[s]] = 1

    # ---- hierarchy rules
    rule_off = [0]; rule_inc = []; rule_exc = []
    if not hier_rules_nil:
        for s in states:
            for r in (hierarchy_rules.get(s) or []):
                if r is None:
                    raise Unsupported("nil *HierarchyRule")
                inc, exc = _rule(r)
                rule_inc.append(max(inc, 0))          # findAncestor: `for level > 0`
                rule_exc.append(max(exc, 0))
            rule_off.append(len(rule_inc))
            k = cons[sid[s]]
            if k > 0 and (rule_off[-1] - rule_off[-2]) * k > 64:
                raise Unsupported("more than 64 hierarchy picks per partition and state")
        VX, v_empty, v_parent, v_lo, v_hi, leaf_pos = build_hierarchy(nodes, NX, node_hierarchy)
    else:
        rule_off = [0] * (M + 1)
        VX, v_empty = 0, 0
        v_parent = v_lo = v_hi = np.zeros(0, dtype=np.int32)
        leaf_pos = np.full(NX, -1, dtype=np.int32)

    if booster not in (None, "cbgt", abi.BOOSTER_NONE, abi.BOOSTER_CBGT):
        raise Unsupported("NodeScoreBooster other than the cbgt built-in")
    booster_kind = abi.BOOSTER_CBGT if booster in ("cbgt", abi.BOOSTER_CBGT) else abi.BOOSTER_NONE

    sc.update(n_nodes=N, n_nodes_ext=NX, n_states=M, n_parts=P, n_prev=n_prev,
              n_loads=len(loads), n_rules=len(rule_inc), n_vertices=VX,
              max_iterations=int(max_iterations),
              partition_weights_nil=int(weights_nil), nodes_to_add_nil=int(nodes_to_add is None),
              hierarchy_rules_nil=int(hier_rules_nil), booster_kind=booster_kind,
              top_state=top_state, vertex_empty=v_empty)
    fp.set("state_priority", prios)
    fp.set("state_constraints", cons)
    fp.set("state_stickiness", st_val)
    fp.set("state_has_stickiness", st_has)
    fp.set("node_removed", node_removed)
    fp.set("node_added", node_added)
    fp.set("node_weight", node_weight)
    fp.set("node_has_weight", node_has_weight)
    fp.set("part_order", static_partition_order(pnames, partition_weights))
    fp.set("part_weight", part_weight)
    fp.set("part_has_weight", part_has_weight)
    fp.set("part_in_prev", part_in_prev)
    fp.set("part_prev_never_equal", never_equal)
    fp.set("assign_off", a_off if P * M else [0])
    fp.set("assign_nodes", a_nodes)
    fp.set("assign_kind", a_kind)
    fp.set("prev_off", p_off if P * M else [0])
    fp.set("prev_nodes", p_nodes)
    fp.set("prev_kind", p_kind)
    fp.set("load_state", [l[0] for l in loads])
    fp.set("load_node", [l[1] for l in loads])
    fp.set("load_weight", [l[2] for l in loads])
    fp.set("load_first_sweep_only", [l[3] for l in loads])
    fp.set("rule_off", rule_off)
    fp.set("rule_inc", rule_inc)
    fp.set("rule_exc", rule_exc)
    fp.set("vertex_parent", v_parent)
    fp.set("vertex_leaf_lo", v_lo)
    fp.set("vertex_leaf_hi", v_hi)
    fp.set("node_leaf_pos", leaf_pos)
    fp.node_names = nodes.names
    fp.state_names = states
    fp.part_names = pnames
    return fp


def decode_result(fp, res):
    """Flat result -> ({name: {"name", "nodesByState"}}, {name: [warning strings]})."""
    M = fp.scalars["n_states"]
    out = {}
    lists = res.lists()
    for p, name in enumerate(fp.part_names):
        nbs = {}
        for m, s in enumerate(fp.state_names):
            kind, ids = lists[p][m]
            if kind == abi.LIST_ABSENT:
                continue
            nbs[s] = None if kind == abi.LIST_NIL else [fp.node_names[i] for i in ids.tolist()]
        out[name] = {"name": name, "nodesByState": nbs}
    cons = fp.arrays["state_constraints"]
    warnings = {}
    for p, m in res.warnings():
        name = fp.part_names[p]
        warnings.setdefault(name, []).append(                 # plan.go:231-234
            "could not meet constraints: %d, stateName: %s, partitionName: %s"
            % (int(cons[m]), fp.state_names[m], name))
    return out, warnings
