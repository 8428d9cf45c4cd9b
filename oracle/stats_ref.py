"""CPU restatement of the plan-quality numbers of blance_plan_stats_get:
countStateNodes (reference plan.go:374-399; partition weight default 1, plan.go:387-394) applied to
the RESULT map and reduced per state over nodesNext (plan.go:77), plus the unmet constraint slots
behind the warnings of plan.go:231-234.  TEST INFRASTRUCTURE ONLY.  The reference has no such
function (callers re-walk the map); the definition is this file, the ingredients are the reference's."""
import numpy as np


def plan_stats(fp, res):
    """fp: abi.FlatProblem, res: abi.FlatResult (or any object with out_off / out_nodes / out_kind)."""
    N, M, P = fp.n_nodes, fp.n_states, fp.n_parts
    removed = fp.arrays["node_removed"].astype(bool)
    alive = ~removed[:N] if N else np.zeros(0, dtype=bool)
    cons = fp.arrays["state_constraints"]
    w = np.ones(P, dtype=np.int64)
    if not fp.scalars["partition_weights_nil"]:
        has = fp.arrays["part_has_weight"].astype(bool)
        w[has] = fp.arrays["part_weight"][has]
    NX = fp.scalars["n_nodes_ext"]
    load = np.zeros((M, max(NX, 1)), dtype=np.int64)
    unmet = np.zeros(M, dtype=np.int64)
    off = res.out_off
    for p in range(P):
        for m in range(M):
            i = p * M + m
            lst = res.out_nodes[off[i]:off[i + 1]] if res.out_kind[i] != 0 else []
            for n in lst:
                load[m, n] += w[p]
            unmet[m] += max(0, max(int(cons[m]), 0) - len(lst))
    out = {"n_nodes_next": int(alive.sum())}
    sel = load[:, :N][:, alive] if N else np.zeros((M, 0), dtype=np.int64)
    any_nodes = sel.shape[1] > 0 and res.iterations > 0
    out["load_min"] = sel.min(axis=1) if any_nodes else np.zeros(M, dtype=np.int64)
    out["load_max"] = sel.max(axis=1) if any_nodes else np.zeros(M, dtype=np.int64)
    out["load_sum"] = sel.sum(axis=1) if any_nodes else np.zeros(M, dtype=np.int64)
    out["load_sumsq"] = (sel * sel).sum(axis=1) if any_nodes else np.zeros(M, dtype=np.int64)
    out["nodes_used"] = (sel > 0).sum(axis=1) if any_nodes else np.zeros(M, dtype=np.int64)
    out["unmet_slots"] = unmet if res.iterations > 0 else np.zeros(M, dtype=np.int64)
    out["rule_violations"] = rule_violations(fp, res) if res.iterations > 0 else np.zeros(M, dtype=np.int64)
    if res.iterations == 0:
        out["n_nodes_next"] = 0
    return out


def rule_violations(fp, res):
    """Per state: (partition, slot) pairs whose node breaks one of the state's hierarchy rules against the partition's top
    priority node or an earlier node of the same list -- outside leaves(findAncestor(a, IncludeLevel)) or inside
    leaves(findAncestor(a, ExcludeLevel)) of such an anchor a (plan.go:723-734, the anchors of plan.go:185-212; "" when the
    partition has no top priority node)."""
    M, P = fp.n_states, fp.n_parts
    out = np.zeros(M, dtype=np.int64)
    if fp.scalars["hierarchy_rules_nil"] or fp.scalars["n_rules"] == 0:
        return out
    parent, lo, hi = fp.arrays["vertex_parent"], fp.arrays["vertex_leaf_lo"], fp.arrays["vertex_leaf_hi"]
    leaf = fp.arrays["node_leaf_pos"]
    roff, rinc, rexc = fp.arrays["rule_off"], fp.arrays["rule_inc"], fp.arrays["rule_exc"]
    v_empty, top_state = fp.scalars["vertex_empty"], fp.scalars["top_state"]

    def anc(v, level):
        for _ in range(level):
            v = parent[v]
        return v

    def breaks(c, a, r):
        vi, ve = anc(a, int(rinc[r])), anc(a, int(rexc[r]))
        lp = leaf[c]
        return lp < lo[vi] or lp >= hi[vi] or (lo[ve] <= lp < hi[ve])

    off = res.out_off
    for p in range(P):
        ti = p * M + top_state
        tl = res.out_nodes[off[ti]:off[ti + 1]] if res.out_kind[ti] != 0 else []
        top = int(tl[0]) if len(tl) else v_empty
        for m in range(M):
            if roff[m + 1] <= roff[m]:
                continue
            i0 = p * M + m
            lst = [int(x) for x in (res.out_nodes[off[i0]:off[i0 + 1]] if res.out_kind[i0] != 0 else [])]
            for i, c in enumerate(lst):
                if any(breaks(c, a, r) for r in range(int(roff[m]), int(roff[m + 1])) for a in [top] + lst[:i]):
                    out[m] += 1
    return out
