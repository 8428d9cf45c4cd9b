#!/usr/bin/env python3
"""Transcribe the reference's TestOrchestrateMoves table (orchestrate_test.go:1049-1811) into
tests/golden/orchestrate_cases.json: per case the begin / end maps, the model, MaxConcurrentPartitionMovesPerNode and the
per-partition sequences of assignments the reference expects its orchestrator to make.  Reads (never copies) the Go test
source.  tests/test_move_index.py replays the sequences through the move index."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import extract_golden as E  # noqa: E402

E.NAMED["OrchestratorOptions"] = ("struct", {"MaxConcurrentPartitionMovesPerNode": E.INT, "FavorMinNodes": E.BOOL})
E.NAMED["assignPartitionRec"] = ("struct", {"partition": E.STRING, "node": E.STRING, "state": E.STRING, "op": E.STRING})
E.NAMED["error"] = E.STRING


def top_level_var(toks, name):
    for i in range(len(toks) - 2):
        if toks[i].val == "var" and toks[i + 1].val == name and toks[i + 2].val == "=":
            p = E.Parser(toks, {})
            p.i = i + 3
            return E.strip_lines(p.parse_expr())
    raise SyntaxError("no top-level var %s" % name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    a = ap.parse_args()
    toks = E.tokenize(open(os.path.join(a.ref, "orchestrate_test.go")).read())
    funcs = E.split_funcs(toks)
    env = {"mrPartitionModel": top_level_var(toks, "mrPartitionModel"), "options1": top_level_var(toks, "options1")}
    lo, hi = funcs["TestOrchestrateMoves"]
    j = lo
    while not (toks[j].val == "tests" and toks[j + 1].val == ":="):
        j += 1
    p = E.Parser(toks, env)
    p.i = j + 2
    cases = [E.strip_lines(c) for c in p.parse_composite(p.parse_type())]
    with open(os.path.join(a.out, "orchestrate_cases.json"), "w") as f:
        json.dump({"generator": "tools/extract_orchestrate_golden.py",
                   "reference": "couchbase/blance orchestrate_test.go TestOrchestrateMoves", "cases": cases}, f, indent=1, sort_keys=True)
    n_seq = sum(len(c.get("expectAssignPartitions") or {}) for c in cases)
    print("TestOrchestrateMoves cases: %d (%d skipped by the reference), expected partition sequences: %d" % (
        len(cases), sum(1 for c in cases if c.get("skip")), n_seq))


if __name__ == "__main__":
    main()
