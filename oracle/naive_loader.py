"""Build + run oracle/naive_proxy.cpp (test / baseline infrastructure only): the deliberately naive,
string-keyed restatement that stands in for "the reference Go path" in bench.py's cpu_baseline."""
import json
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_BIN = os.path.join(_HERE, "_build", "naive_proxy")


def build(force=False):
    src = os.path.join(_HERE, "naive_proxy.cpp")
    if force or not os.path.exists(_BIN) or os.path.getmtime(src) > os.path.getmtime(_BIN):
        os.makedirs(os.path.dirname(_BIN), exist_ok=True)
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-o", _BIN, src])
    return _BIN


def run(cfg, parts, nodes, quiet=False):
    """-> (lines "name|primary|replica" as node positions, stats dict)"""
    p = subprocess.run([build(), str(cfg), str(parts), str(nodes)] + (["quiet"] if quiet else []),
                       capture_output=True, text=True, check=True)
    stats = json.loads(p.stderr.strip().splitlines()[-1])
    return p.stdout.splitlines(), stats


def timed_sample(cfg, nodes, budget_s=12.0):
    """The proxy on the SAME node count, hierarchy and model as config `cfg`, with as many partitions as
    fit the time budget (its per-call cost is O(nodes log nodes), independent of the partition count);
    assignments/s is the figure, the partition count is a sample -- labelled as extrapolated."""
    if cfg not in (2, 3):
        return {"skipped": "the naive proxy generates configs 2 and 3 only"}
    _, probe = run(cfg, 48, nodes, quiet=True)
    per_call = probe["seconds"] / max(probe["find_best_nodes_calls"], 1)
    sweeps = max(probe["sweeps"], 2) + 1
    parts = int(max(64, min(65536 if cfg == 2 else 1 << 20, budget_s / (per_call * 2 * sweeps))))
    _, st = run(cfg, parts, nodes, quiet=True)
    return {"value": st["assignments_per_s"], "unit": "assignments/s", "cores": 1, "kind": "naive string-keyed proxy of the Go code path",
            "sample": "oracle/naive_proxy.cpp: full PlanNextMap (%d sweeps) on %d partitions x %d nodes, %.1f s, %.2f ms per "
                      "findBestNodes call; extrapolated linearly in the partition count" %
                      (st["sweeps"], parts, nodes, st["seconds"], 1e3 * st["seconds"] / st["find_best_nodes_calls"])}
