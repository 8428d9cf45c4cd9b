"""Size-independent properties of a PlanNextMap result for BASELINE.json config 3's generator (synth.config_flat(3, ...)):
what the model and the hierarchy rule promise whatever the size -- checked WITHOUT the oracle, so that a full-size
result has a check of its own next to the digest comparison.

  model       primary x1, replica x2 (api.go:46-62): every list SET, of exactly that length, nodes of the cluster,
              no node twice in a partition (plan.go:146-154: higher priority nodes are no candidates);
  rule        replica {includeLevel 2, excludeLevel 1} (api.go:96-105, plan.go:174-226): both replicas in the primary's
              zone, the three nodes in three different racks -- whenever the zone has three racks with room;
  balance     the greedy of plan.go:253-303 levels counts: per state, the per-node counts of a fresh plan with equal
              weights differ by at most a small constant inside a region (measured on the oracle: see the test);
  idempotence PlanNextMap(prevMap = partitionsToAssign = result) converges in its first sweep on the same map
              (plan.go:32-57: the reference's own fixed point test).
"""
import numpy as np


def lists_of(res, P, M):
    off = np.asarray(res.out_off[:P * M + 1]).astype(np.int64)
    total = int(off[P * M])
    return off, np.asarray(res.out_nodes[:total]), np.asarray(res.out_kind[:P * M])


def config3_properties(fp, res, rack=16, racks_per_zone=8):
    """-> dict of measured properties; raises AssertionError on a violated promise."""
    from blance_amd import abi
    P, M, N = int(fp.n_parts), int(fp.n_states), int(fp.n_nodes)
    assert M == 2
    off, nodes, kind = lists_of(res, P, M)
    assert (kind == abi.LIST_SET).all()
    ln = np.diff(off).reshape(P, M)
    assert (ln[:, 0] == 1).all() and (ln[:, 1] == 2).all(), "a list of the wrong length"
    assert int(off[-1]) == 3 * P
    trip = nodes.reshape(P, 3).astype(np.int64)             # [primary, replica 1, replica 2] per partition
    assert (trip >= 0).all() and (trip < N).all()
    prim, r1, r2 = trip[:, 0], trip[:, 1], trip[:, 2]
    assert (prim != r1).all() and (prim != r2).all() and (r1 != r2).all(), "a node twice in one partition"
    zone = rack * racks_per_zone
    # promised when the primary's zone has at least three racks (else the rule's set runs empty for some copy and
    # plan.go:214-221 falls back to the best node anywhere)
    n_racks_in_zone = (np.minimum((prim // zone + 1) * zone, N) - (prim // zone) * zone + rack - 1) // rack
    wide = n_racks_in_zone >= 3
    in_zone = (r1 // zone == prim // zone) & (r2 // zone == prim // zone)
    assert in_zone[wide].all(), "a replica outside the primary's zone"
    rk = trip // rack
    ok3 = (rk[:, 0] != rk[:, 1]) & (rk[:, 0] != rk[:, 2]) & (rk[:, 1] != rk[:, 2])
    assert ok3[wide].all(), "two copies of a partition in one rack"
    cp = np.bincount(prim, minlength=N)
    cr = np.bincount(np.concatenate([r1, r2]), minlength=N)
    return {"primary_spread": int(cp.max() - cp.min()), "replica_spread": int(cr.max() - cr.min()),
            "primary_sum": int(cp.sum()), "replica_sum": int(cr.sum())}


def same_lists(a, b, P, M):
    oa, na, ka = lists_of(a, P, M)
    ob, nb, kb = lists_of(b, P, M)
    return np.array_equal(oa, ob) and np.array_equal(na, nb) and np.array_equal(ka, kb)


def plan_properties(fp, res):
    """Promises that hold for ANY problem whose live nodes suffice for every state's constraints and whose partitions
    carry no state outside the model (the generators of blance_amd/synth.py, config 5 included): every list SET and of
    its state's length (plan.go:228-235: no warning), nodes of nodesAll minus nodesToRemove (plan.go:70-81), no node
    twice in a partition (plan.go:146-154, :290-297), and the weighted load of a state adds up to sum(weights) x k."""
    from blance_amd import abi
    P, M, N = int(fp.n_parts), int(fp.n_states), int(fp.n_nodes)
    off, nodes, kind = lists_of(res, P, M)
    k = np.asarray(fp.state_constraints[:M]).astype(np.int64)
    assert int(res.n_warnings) == 0
    assert (kind == abi.LIST_SET).all()
    ln = np.diff(off).reshape(P, M)
    assert (ln == k[None, :]).all(), "a list of the wrong length"
    assert (nodes >= 0).all() and (nodes < N).all()
    removed = np.asarray(fp.node_removed[:N]).astype(bool)
    assert not removed[nodes].any(), "a removed node in the result"
    K = int(k.sum())
    per = nodes.reshape(P, K).astype(np.int64)
    srt = np.sort(per, axis=1)
    assert (srt[:, 1:] != srt[:, :-1]).all(), "a node twice in one partition"
    w = np.asarray(fp.part_weight[:P]).astype(np.int64)
    load = np.zeros((M, N), dtype=np.int64)
    col = 0
    for m in range(M):
        for j in range(int(k[m])):
            np.add.at(load[m], per[:, col], w)
            col += 1
    assert [int(load[m].sum()) for m in range(M)] == [int(w.sum()) * int(k[m]) for m in range(M)]
    return {"load_max": [int(load[m].max()) for m in range(M)], "load_min_live": [int(load[m][~removed].min()) for m in range(M)]}


def _csr_keys(P, M, N, off, nodes):
    """(partition, state, node) of every list entry of the model's states, and its (partition, node)."""
    off = np.asarray(off).astype(np.int64)
    nodes = np.asarray(nodes).astype(np.int64)
    ln = np.diff(off)
    slot = np.repeat(np.arange(P * (M + 1), dtype=np.int64), ln)
    p, s = slot // (M + 1), slot % (M + 1)
    keep = s < M
    return (p[keep] * M + s[keep]) * N + nodes[:len(slot)][keep], p[keep] * N + nodes[:len(slot)][keep]


def moves_round_trip(P, M, N, beg_off, beg_nodes, end_off, end_nodes, op_off, op_node, op_state, op_kind):
    """CalcPartitionMoves' contract without an oracle (moves.go:41-119): a node has at most one move per partition,
    and carrying the moves out on begMap -- add: the node enters the state; del: it leaves the partition; promote /
    demote: it changes to the state -- gives endMap's membership, state by state.  For maps that hold a node at most
    once per partition and only states of `states` (what a planner returns).  -> number of moves."""
    from blance_amd import abi
    op_off = np.asarray(op_off).astype(np.int64)
    n_ops = int(op_off[P])
    op_p = np.repeat(np.arange(P, dtype=np.int64), np.diff(op_off[:P + 1]))
    node = np.asarray(op_node[:n_ops]).astype(np.int64)
    state = np.asarray(op_state[:n_ops]).astype(np.int64)
    kind = np.asarray(op_kind[:n_ops])
    moved = op_p * N + node
    assert len(np.unique(moved)) == n_ops, "two moves for one node of a partition"
    assert (state[kind != abi.OP_DEL] >= 0).all() and (state[kind != abi.OP_DEL] < M).all()
    beg3, beg2 = _csr_keys(P, M, N, beg_off, beg_nodes)
    end3, _ = _csr_keys(P, M, N, end_off, end_nodes)
    stay = beg3[~np.isin(beg2, moved)]
    enters = kind != abi.OP_DEL
    new3 = (op_p[enters] * M + state[enters]) * N + node[enters]
    got = np.union1d(stay, new3)
    assert np.array_equal(got, np.unique(end3)), "begMap + moves != endMap"
    # and nothing superfluous: a node that keeps its state has no move
    same = np.intersect1d(beg3, end3)
    assert not np.isin(same % N + (same // N // M) * N, moved).any(), "a move for a node that keeps its state"
    return n_ops
