"""Host-side mirror of the reference's planner API (api.go:24-190) over the HIP
library: same names, argument meaning, warnings text and caller-visible
mutations -- PlanNextMapEx() here is what `plan_hip.go` does
in Go.  Partition maps are dicts {name: Partition}.

There is no CPU fallback in this package: inputs outside the device envelope
raise problem.Unsupported / hip.BlanceError (the Go shim would call the Go
planner there).
"""
from . import hip, problem

MaxIterationsPerPlan = 10            # plan.go:21
# plan.go:580: the hook that replaces the node sorter.  Anything but None (the default sorter) is a
# callback of the caller's language and cannot run on the device: PlanNextMapEx refuses (the Go shim
# runs plan.go then).  plan.go:693 NodeScoreBooster: only the cbgt booster (control_test.go:19-26) is
# built in -- pass booster="cbgt"; any other callable is refused the same way.
CustomNodeSorter = None


class Partition:
    """api.go:28-36"""
    __slots__ = ("Name", "NodesByState")

    def __init__(self, Name="", NodesByState=None):
        self.Name = Name
        self.NodesByState = NodesByState

    def __eq__(self, other):
        return isinstance(other, Partition) and self.Name == other.Name and self.NodesByState == other.NodesByState

    def __repr__(self):
        return "Partition(%r, %r)" % (self.Name, self.NodesByState)


class PartitionModelState:
    """api.go:46-62"""
    __slots__ = ("Priority", "Constraints")

    def __init__(self, Priority=0, Constraints=0):
        self.Priority = Priority
        self.Constraints = Constraints


class HierarchyRule:
    """api.go:96-105"""
    __slots__ = ("IncludeLevel", "ExcludeLevel")

    def __init__(self, IncludeLevel=0, ExcludeLevel=0):
        self.IncludeLevel = IncludeLevel
        self.ExcludeLevel = ExcludeLevel


class PlanNextMapOptions:
    """api.go:183-190; every field may be None (a nil map)."""

    def __init__(self, ModelStateConstraints=None, PartitionWeights=None, StateStickiness=None,
                 NodeWeights=None, NodeHierarchy=None, HierarchyRules=None):
        self.ModelStateConstraints = ModelStateConstraints
        self.PartitionWeights = PartitionWeights
        self.StateStickiness = StateStickiness
        self.NodeWeights = NodeWeights
        self.NodeHierarchy = NodeHierarchy
        self.HierarchyRules = HierarchyRules


_planner = None


def default_planner():
    global _planner
    if _planner is None:
        _planner = hip.Planner(device_id=0)
    return _planner


def PlanNextMapEx(prevMap, partitionsToAssign, nodesAll, nodesToRemove, nodesToAdd, model, options=None,
                  booster=None, planner=None):
    """api.go:147-157.  Returns (nextMap, warnings); mutates prevMap and
    partitionsToAssign the way planNextMapEx does (plan.go:49-52)."""
    options = options or PlanNextMapOptions()
    if CustomNodeSorter is not None:
        raise problem.Unsupported("CustomNodeSorter is not the default sorter (plan.go:580)")
    if booster is not None and booster != "cbgt":
        raise problem.Unsupported("NodeScoreBooster is an arbitrary callback (plan.go:693)")
    fp = problem.build_problem(prevMap, partitionsToAssign, nodesAll, nodesToRemove, nodesToAdd, model,
                               options.ModelStateConstraints, options.PartitionWeights,
                               options.StateStickiness, options.NodeWeights, options.NodeHierarchy,
                               options.HierarchyRules, booster, max_iterations=MaxIterationsPerPlan)
    res = (planner or default_planner()).plan(fp)
    if res.iterations == 0:                     # MaxIterationsPerPlan <= 0: (nil, nil)
        return None, None
    flat, warnings = problem.decode_result(fp, res)
    nextMap = {name: Partition(name, p["nodesByState"]) for name, p in flat.items()}
    # plan.go:49-52: every non-converged sweep stores its partitions into BOTH input maps.
    # The last such store holds the final map's content.
    # When the call converged (in sweep n > 1) the stored objects are sweep n - 1's: equal in content to the returned
    # ones but distinct objects (plan.go:334-343 makes fresh ones every sweep); at the iteration cap they are the returned
    # objects themselves.
    if res.iterations > 1 or not res.converged:
        for name, part in nextMap.items():
            stored = part
            if res.converged:
                stored = Partition(part.Name, None if part.NodesByState is None else
                                   {s: (None if l is None else list(l)) for s, l in part.NodesByState.items()})
            prevMap[name] = stored
            partitionsToAssign[name] = stored
    return nextMap, warnings


def PlanNextMap(prevMap, partitionsToAssign, nodesAll, nodesToRemove, nodesToAdd, model,
                modelStateConstraints=None, partitionWeights=None, stateStickiness=None, nodeWeights=None,
                nodeHierarchy=None, hierarchyRules=None, planner=None):
    """api.go:109-132 (deprecated positional wrapper)."""
    return PlanNextMapEx(prevMap, partitionsToAssign, nodesAll, nodesToRemove, nodesToAdd, model,
                         PlanNextMapOptions(modelStateConstraints, partitionWeights, stateStickiness, nodeWeights,
                                            nodeHierarchy, hierarchyRules), planner=planner)


class NodeStateOp:
    """moves.go:17-21"""
    __slots__ = ("Node", "State", "Op")

    def __init__(self, Node, State, Op):
        self.Node, self.State, self.Op = Node, State, Op

    def __eq__(self, other):
        return (self.Node, self.State, self.Op) == (other.Node, other.State, other.Op)

    def __repr__(self):
        return "NodeStateOp(%r, %r, %r)" % (self.Node, self.State, self.Op)


def CalcPartitionMovesBatch(states, begMap, endMap, favorMinNodes, planner=None):
    """CalcPartitionMoves (moves.go:41-119) for every partition of begMap / endMap at
    once -- what OrchestrateMoves does in a loop (orchestrate.go:273-287).  Maps are
    {name: nodesByState} or {name: Partition}; returns {name: [NodeStateOp]}."""
    import numpy as np
    from . import abi
    names = list(endMap.keys()) if endMap is not None else []
    for n in (begMap or {}):
        if n not in (endMap or {}):
            names.append(n)
    ids, node_names = {}, []

    def nid(x):
        i = ids.get(x)
        if i is None:
            i = ids[x] = len(node_names)
            node_names.append(x)
        return i

    def nbs_of(m, name):
        p = (m or {}).get(name)
        if p is None:
            return {}
        return (p.NodesByState if hasattr(p, "NodesByState") else p) or {}

    M = len(states)
    known = set(states)

    def csr(m):
        off, nodes = [0], []
        for name in names:
            nbs = nbs_of(m, name)
            for s in states:
                nodes.extend(nid(x) for x in (nbs.get(s) or []))
                off.append(len(nodes))
            for s, lst in nbs.items():               # keys outside `states` only feed flattenNodesByState
                if s not in known:
                    nodes.extend(nid(x) for x in (lst or []))
            off.append(len(nodes))
        return np.asarray(off, dtype=np.int32), np.asarray(nodes, dtype=np.int32)

    boff, bnod = csr(begMap)
    eoff, enod = csr(endMap)
    op_off, op_node, op_state, op_kind, _ = (planner or default_planner()).calc_moves(M, favorMinNodes, boff, bnod, eoff, enod)
    out = {}
    for i, name in enumerate(names):
        out[name] = [NodeStateOp(node_names[op_node[j]], "" if op_state[j] < 0 else states[op_state[j]],
                                 abi.OP_NAMES[op_kind[j]]) for j in range(op_off[i], op_off[i + 1])]
    return out


def CalcPartitionMoves(states, begNodesByState, endNodesByState, favorMinNodes, planner=None):
    """moves.go:41-119 for one partition."""
    return CalcPartitionMovesBatch(states, {"p": begNodesByState}, {"p": endNodesByState}, favorMinNodes, planner)["p"]


# misc.go:13-51, exported helpers callers may use
def StringsToMap(strs):
    return None if strs is None else {s: True for s in strs}


def StringsRemoveStrings(stringArr, removeArr):
    rm = StringsToMap(removeArr) or {}
    return [s for s in (stringArr or []) if s not in rm]


def StringsIntersectStrings(a, b):
    bm = StringsToMap(b) or {}
    rv, seen = [], set()
    for s in (a or []):
        if s in bm and s not in seen:
            seen.add(s)
            rv.append(s)
    return rv
