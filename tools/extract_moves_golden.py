#!/usr/bin/env python3
"""Transcribe the reference's CalcPartitionMoves golden tables (moves_test.go) into
tests/golden/moves_cases.json.  Reads (never copies) the Go test source; the picture
DSL of TestCalcPartitionMoves is decoded the way the test's own harness does
(moves_test.go:370-480): every expected move becomes {node, state, ops allowed}."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import extract_golden as E  # noqa: E402


def convert_line(line, states):          # convertLineToNodesByState, moves_test.go:491-517
    nbs = {}
    line = line.strip(" ")
    while True:
        x = line.replace("  ", " ")
        if x == line:
            break
        line = x
    parts = line.split("|")
    for i, state in enumerate(states):
        if i >= len(parts):
            break
        part = parts[i].strip(" ")
        if part != "":
            nbs.setdefault(state, []).extend(part.split(" "))
    return nbs


def decode(test, states):
    before = convert_line(test["before"], states)
    after = convert_line(test["after"], states)
    exp = []
    if test["moves"] != "":
        for move_line in test["moves"].split("\n"):
            # the Go source indents continuation lines with tabs; Trim(" ") leaves them, so the
            # first token of such a line carries the tabs -- strip them as whitespace here
            m = convert_line(move_line.replace("\t", " "), states)
            found = None
            for si, state in enumerate(states):
                if found:
                    break
                for mv in m.get(state, []):
                    if found:
                        break
                    op = mv[0:1]
                    if op in "+-" and op:
                        node = mv[1:]
                        flip = ("-" if op == "+" else "+") + node
                        flip_state = ""
                        for j in range(si + 1, len(states)):
                            for x in m.get(states[j], []):
                                if x == flip:
                                    flip_state = states[j]
                        state_exp = state
                        if flip_state:
                            if op == "-":
                                state_exp = flip_state
                            ops = ["promote", "demote"]
                        else:
                            if op == "-":
                                state_exp = ""
                            ops = ["add" if op == "+" else "del"]
                        found = {"node": node, "state": state_exp, "ops": ops}
            if found is None:
                raise ValueError("move line without a move: %r" % move_line)
            exp.append(found)
    return {"states": states, "before": before, "after": after, "favorMinNodes": test["favorMinNodes"], "exp": exp}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    a = ap.parse_args()
    toks = E.tokenize(open(os.path.join(a.ref, "moves_test.go")).read())
    funcs = E.split_funcs(toks)
    fsc = E.extract_table(toks, funcs, "TestFindStateChanges")
    cpm = [decode(t, ["primary", "replica"]) for t in E.extract_table(toks, funcs, "TestCalcPartitionMoves")]
    with open(os.path.join(a.out, "moves_cases.json"), "w") as f:
        json.dump({"generator": "tools/extract_moves_golden.py", "reference": "couchbase/blance moves_test.go",
                   "findStateChanges": fsc, "calcPartitionMoves": cpm}, f, indent=1, sort_keys=True)
    print("findStateChanges rows: %d, calcPartitionMoves cases: %d" % (len(fsc), len(cpm)))


if __name__ == "__main__":
    main()
