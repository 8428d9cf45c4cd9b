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
                              ("func (ix *moveIndex) activeNodes(", "active_nodes()")):
        assert go_name in src and cpp_name in hdr, (go_name, cpp_name)
    assert src.count("{") == src.count("}")
    assert "for _, nm := range b" not in src          # round 2's available() copied every bucket per round
