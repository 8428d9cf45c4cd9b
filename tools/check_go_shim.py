#!/usr/bin/env python3
"""Keeps the Go shim (go/blance/*.go, which cannot be compiled in this image) in step with the C ABI:
every field of blance_problem / blance_result (inputs) / blance_moves_problem / blance_moves_result
in include/blance_hip.h must be assigned by the Go code through the matching variable, and the Go
code must not name a field the header does not have.  Exit status 0 = in step."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def struct_fields(header, name):
    m = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), header, re.S)
    body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        fields.append(re.split(r"[\s\*]+", decl)[-1].split("[")[0])
    return fields


def main():
    header = open(os.path.join(ROOT, "include", "blance_hip.h")).read()
    plan = open(os.path.join(ROOT, "go", "blance", "plan_hip.go")).read()
    moves = open(os.path.join(ROOT, "go", "blance", "moves_hip.go")).read()
    # result fields the library fills: not assigned by the caller
    outputs = {"blance_result": {"n_warnings", "iterations", "converged", "device_ms", "total_ms", "steps_total",
                                 "steps_sequential", "steps_batched", "kernel_launches", "pass_kernel_ms",
                                 "pass_kernel_launches", "flat_pass_ms", "flat_passes", "blank_pass_ms",
                                 "blank_pass_launches", "stay_pass_ms", "stay_pass_launches", "host_syncs"},
               "blance_moves_result": {"device_ms"}}
    bad = []
    for struct, var, text in (("blance_problem", "pb", plan), ("blance_result", "res", plan),
                              ("blance_moves_problem", "pb", moves), ("blance_moves_result", "res", moves)):
        fields = struct_fields(header, struct)
        assigned = set(re.findall(r"\b%s\.([a-z_0-9]+)\s*=[^=]" % var, text))
        used = set(re.findall(r"\b%s\.([a-z_0-9]+)\b" % var, text))
        for f in fields:
            if f not in assigned and f not in outputs.get(struct, ()):
                bad.append("%s.%s is never assigned by the Go shim" % (struct, f))
        for f in used:
            if f not in fields:
                bad.append("the Go shim names %s.%s, which %s does not have" % (var, f, struct))
    for c in re.findall(r"C\.(blance_[a-z_]+)\(", plan + moves):
        if not re.search(r"\b%s\(" % c, header):
            bad.append("the Go shim calls %s, which the header does not declare" % c)
    # the two host mirrors refuse the same inputs: every refusal of the compiled, tested C++ mirror (blance_api.cpp:
    # Unsupported{"..."}) has a counterpart among intern.go's unsupported("...") -- matched on the words of the message
    # (the Go text adds names with %q and range checks for its 64-bit ints)
    def words(msg):
        msg = re.sub(r"%[qdsv]", "", msg.lower())
        msg = re.sub(r"\(.*?\)", "", msg)
        return set(w for w in re.findall(r"[a-z*]+", msg) if w not in ("a", "the", "of", "for", "is", "in", "to", "that", "but", "with"))
    intern = open(os.path.join(ROOT, "go", "blance", "intern.go")).read()
    cpp = open(os.path.join(ROOT, "blance_amd", "csrc", "host", "blance_api.cpp")).read()
    go_msgs = re.findall(r'unsupported\("((?:[^"\\]|\\.)+)"', intern)
    for msg in set(re.findall(r'Unsupported\{"((?:[^"\\]|\\.)+)"', cpp)):
        w = words(msg)
        if not w:
            continue
        best = max((len(w & words(g)) / float(len(w | words(g))) for g in go_msgs), default=0.0)
        if best < 0.75:
            bad.append("the C++ mirror refuses %r; intern.go has no such refusal" % msg)
    for line in bad:
        print(line)
    print("go shim vs include/blance_hip.h and blance_api.cpp: %s" % ("IN STEP" if not bad else "%d problems" % len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
