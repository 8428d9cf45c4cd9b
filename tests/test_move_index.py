"""MoveIndex (blance_amd/csrc/host/move_index.hpp, the compiled twin of
go/blance/orchestrate_index.go) against the reference's per-round rescan -- a randomised simulation of the
supply rounds of orchestrate.go:506-590 in which, after EVERY round, the index's buckets equal
findAvailableMovesUnlocked()'s (orchestrate.go:749-763) and its picks carry the op weights
filterNextPlausibleMovesForNode + LowestWeightPartitionMoveForNode would pick (orchestrate.go:482-504, :174-184)."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sim():
    import __graft_entry__ as g
    return g.build_move_index()


@pytest.mark.parametrize("seed,P,N,count", [(1, 500, 7, 2), (2, 3000, 40, 1), (3, 100, 3, 5), (4, 2000, 1, 3),
                                            (5, 1, 1, 1), (6, 20000, 300, 0), (7, 5000, 5000, 2), (8, 0, 4, 1)])
def test_index_equals_rescan(sim, seed, P, N, count):
    out = subprocess.run([sim, "check", str(seed), str(P), str(N), str(count)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr


def test_round_cost_does_not_grow_with_partitions(sim):
    """What the index is for: a supply round no longer walks every partition."""
    out = subprocess.run([sim, "time", "200000", "512", "1"], capture_output=True, text=True, timeout=600)
    r = json.loads(out.stdout)
    assert r["moves_offered_rescan"] == r["moves_offered_index"]
    assert r["index_ms_per_round"] * 5 < r["rescan_ms_per_round"], r


def test_go_text_has_the_same_operations():
    src = open(os.path.join(ROOT, "go", "blance", "orchestrate_index.go")).read()
    hdr = open(os.path.join(ROOT, "blance_amd", "csrc", "host", "move_index.hpp")).read()
    for go_name, cpp_name in (("func newMoveIndex(", "MoveIndex(int n_nodes"), ("func (ix *moveIndex) advanced(", "void advanced("),
                              ("func (ix *moveIndex) lowestWeight(", "void lowest_weight("), ("func (ix *moveIndex) bucket(", "void bucket("),
                              ("func (ix *moveIndex) snapshot(", "active_snapshot()"), ("func (ix *moveIndex) snapshotBuckets(", "void bucket(")):
        assert go_name in src and cpp_name in hdr, (go_name, cpp_name)
    # round 3's advisor: the supply loop runs after Unlock while moves complete -- only snapshots may leave the lock; an op
    # the reference's table does not know weighs 0 there; remove() of what was never filed changes nothing
    assert "LOCKING" in src and "func (ix *moveIndex) activeNodes(" not in src
    assert "moveOpUnknown" in src and "kOpUnknown" in hdr and "not filed" in src and "not filed" in hdr
    assert src.count("{") == src.count("}")
    assert "for _, nm := range b" not in src          # round 2's available() copied every bucket per round


def _orchestrate_cases():
    with open(os.path.join(ROOT, "tests", "golden", "orchestrate_cases.json")) as f:
        return json.load(f)["cases"]


def _case_moves(c):
    """CalcPartitionMoves (oracle/moves_ref.py, pinned by moves_test.go's own tables) for every partition of a case."""
    import sys
    sys.path.insert(0, ROOT)
    from oracle import moves_ref
    states = sorted(c["partitionModel"], key=lambda s: (c["partitionModel"][s].get("Priority", 0), s))     # sortStateNames
    favor = bool(c["options"].get("FavorMinNodes", False))
    out = {}
    for name, end in c["endMap"].items():
        beg = c["begMap"].get(name, {"NodesByState": {}})
        out[name] = [{"node": n, "state": st, "op": op} for n, st, op in
                     moves_ref.calc_partition_moves(states, beg.get("NodesByState") or {}, end.get("NodesByState") or {}, favor)]
    return states, out


def test_reference_orchestrator_fixtures_reproduced():
    """orchestrate_test.go:1049-1811 (TestOrchestrateMoves, 15 cases transcribed by tools/extract_orchestrate_golden.py): for
    every partition the sequence of (node, state) assignments the reference expects its orchestrator to make IS its
    CalcPartitionMoves list, in order (orchestrate.go:273-287, :684-691) -- the lists the move index is filled with."""
    n = 0
    for c in _orchestrate_cases():
        states, moves = _case_moves(c)
        for name, want in (c.get("expectAssignPartitions") or {}).items():
            got = [(m["node"], m["state"]) for m in moves[name]]
            assert got == [(r["node"], r.get("state", "")) for r in want], (c["label"], name)
            for r, m in zip(want, moves[name]):
                if r.get("op"):
                    assert r["op"] == m["op"], (c["label"], name)
            n += 1
        for name, ms in moves.items():                # partitions the table does not mention have nothing to do
            if name not in (c.get("expectAssignPartitions") or {}):
                assert ms == [], (c["label"], name)
    assert n == 21


def test_reference_orchestrator_fixtures_through_the_index(sim):
    """The same fixtures' move lists through MoveIndex (move_index_sim replay): every round equals the rescan of
    orchestrate.go:749-763, a node is offered at most MaxConcurrentPartitionMovesPerNode moves per round, every partition's
    moves are carried out in its list's order, each exactly once, and the index ends empty."""
    op_class = {"promote": 1, "demote": 2, "add": 3, "del": 4}
    for c in _orchestrate_cases():
        states, moves = _case_moves(c)
        names = sorted(moves)
        nodes = sorted({m["node"] for ms in moves.values() for m in ms})
        nid = {x: i for i, x in enumerate(nodes)}
        lines = ["%d %d" % (len(names), max(1, len(nodes)))]
        for name in names:
            ms = moves[name]
            lines.append(" ".join([str(len(ms))] + ["%d %d %d" % (nid[m["node"]], states.index(m["state"]) if m["state"] in states else -1,
                                                                  op_class.get(m["op"], 0)) for m in ms]))
        count = int(c["options"].get("MaxConcurrentPartitionMovesPerNode", 0))
        out = subprocess.run([sim, "replay", str(count)], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=60)
        assert out.returncode == 0, (c["label"], out.stdout, out.stderr)
        r = json.loads(out.stdout)
        assert "error" not in r and r["pending"] == 0, (c["label"], r)
        seen = {}
        for rnd in r["rounds"]:
            per_node = {}
            for item in rnd:
                p, pos = map(int, item.split(":"))
                assert seen.get(p, 0) == pos, (c["label"], item)          # in its list's order, none skipped, none twice
                seen[p] = pos + 1
                node = moves[names[p]][pos]["node"]
                per_node[node] = per_node.get(node, 0) + 1
            assert all(v <= max(count, 1) for v in per_node.values()), (c["label"], per_node)
        assert all(seen.get(i, 0) == len(moves[n]) for i, n in enumerate(names)), c["label"]
