"""The C++ host mirror of the reference API (blance_amd/csrc/host): interning,
the C-ABI call, un-interning, warnings text and caller-visible mutations, on the
reference's golden inputs -- through the emulated kernels on CPU, and through
the real library on a GPU box."""
import copy
import json
import os
import subprocess

import pytest

from oracle import blance_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "blance_amd", "csrc", "host")
CLI = os.path.join(ROOT, "blance_amd", "lib", "blance_host_cli")


def build_cli():
    srcs = [os.path.join(HOST, "blance_host_cli.cpp"), os.path.join(HOST, "blance_api.cpp"), os.path.join(HOST, "call_arena.cpp")]
    deps = srcs + [os.path.join(HOST, "blance_api.hpp"), os.path.join(HOST, "call_arena.hpp"), os.path.join(ROOT, "include", "blance_hip.h")]
    if not os.path.exists(CLI) or any(os.path.getmtime(d) > os.path.getmtime(CLI) for d in deps):
        os.makedirs(os.path.dirname(CLI), exist_ok=True)
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-DBLANCE_CALL_ARENA", "-o", CLI] + srcs + ["-ldl"])
    return CLI


def _list(out, lst):
    if lst is None:
        out.append("NIL")
    else:
        out.append(str(len(lst)))
        out.extend(lst)


def _map(out, m):
    out.append(str(len(m)))
    for key, p in m.items():
        out.append(key)
        out.append(p.get("name", ""))
        nbs = p.get("nodesByState")
        if nbs is None:
            out.append("NIL")
        else:
            out.append(str(len(nbs)))
            for s, lst in nbs.items():
                out.append(s)
                _list(out, lst)


def _imap(out, m):
    if m is None:
        out.append("NIL")
    else:
        out.append(str(len(m)))
        for k, v in m.items():
            out.extend([k, str(v)])


def serialize(cases):
    out = [str(len(cases))]
    for c in cases:
        _map(out, c["prevMap"])
        if c.get("aliased"):
            out.append("ALIAS")
        else:
            out.append("OWN")
            _map(out, c["partitionsToAssign"])
        _list(out, c["nodesAll"])
        _list(out, c["nodesToRemove"])
        _list(out, c["nodesToAdd"])
        out.append(str(len(c["model"])))
        for s, ms in c["model"].items():
            out.extend([s, str(ms["priority"]), str(ms["constraints"])])
        for k in ("modelStateConstraints", "partitionWeights", "stateStickiness", "nodeWeights"):
            _imap(out, c.get(k))
        nh = c.get("nodeHierarchy")
        if nh is None:
            out.append("NIL")
        else:
            out.append(str(len(nh)))
            for k, v in nh.items():
                out.extend([k, v])
        hr = c.get("hierarchyRules")
        if hr is None:
            out.append("NIL")
        else:
            out.append(str(len(hr)))
            for s, rl in hr.items():
                out.extend([s, str(len(rl))])
                for r in rl:
                    out.extend([str(r["includeLevel"]), str(r["excludeLevel"])])
        out.append(c.get("booster") or "none")
    return "\n".join(out) + "\n"


def run_cli(lib, cases, eager=False, threads=0):
    env = dict(os.environ)
    if eager:
        env["BLANCE_HOST_EAGER_BULK"] = "1"
    if threads:                                  # the threaded paths of the mirror (blance_api.cpp) on maps of any size
        env["BLANCE_HOST_THREADS"] = str(threads)
        env["BLANCE_HOST_BLOCK"] = "3"
    p = subprocess.run([build_cli(), lib], input=serialize(cases), capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr
    return json.loads(p.stdout)


def _check(cases, results):
    for c, got in zip(cases, results):
        prev_o = R.partition_map_from_json(copy.deepcopy(c["prevMap"]))
        assign_o = prev_o if c.get("aliased") else R.partition_map_from_json(copy.deepcopy(c["partitionsToAssign"]))
        opts_o = R.Options(c.get("modelStateConstraints"), c.get("partitionWeights"), c.get("stateStickiness"),
                           c.get("nodeWeights"), c.get("nodeHierarchy"), c.get("hierarchyRules"))
        info = {}
        want, want_w = R.plan_next_map_ex(prev_o, assign_o, list(c["nodesAll"]), c["nodesToRemove"], c["nodesToAdd"],
                                          c["model"], opts_o, c.get("booster"), info=info)
        assert got["handled"], (c["source"], got["why"])
        assert got["nextMap"] == R.partition_map_to_json(want) == c["exp"], c["source"]
        assert got["warnings"] == (want_w or {}), c["source"]
        assert got["iterations"] == info["iterations"] and got["converged"] == info["converged"], c["source"]
        assert got["prevMap"] == R.partition_map_to_json(prev_o), c["source"]          # plan.go:49-52
        assert got["partitionsToAssign"] == R.partition_map_to_json(assign_o), c["source"]
        # object identity (plan.go:49-52 stores the objects of the last sweep that did NOT converge; :334-343 makes fresh
        # ones every sweep): a caller that edits nextMap[p] edits prevMap[p] only where the reference would
        if want is not None:
            ident = [sum(1 for n, p in want.items() if prev_o.get(n) is p), sum(1 for n, p in want.items() if assign_o.get(n) is p),
                     sum(1 for n in want if n in prev_o and n in assign_o and prev_o[n] is assign_o[n])]
            assert got["identity"] == ident, (c["source"], got["identity"], ident)


def test_cpp_mirror_on_golden_cases_emulated(golden_cases):
    from test_simt_emulated import build_emu
    _check(golden_cases, run_cli(build_emu(), golden_cases))
    _check(golden_cases, run_cli(build_emu(), golden_cases, eager=True))


def test_cpp_mirror_refuses_what_python_refuses():
    from test_simt_emulated import build_emu
    base = {"prevMap": {}, "partitionsToAssign": {"0": {"name": "0", "nodesByState": {}}}, "aliased": False,
            "nodesAll": ["a", "b"], "nodesToRemove": [], "nodesToAdd": [],
            "model": {"primary": {"priority": 0, "constraints": 1}}}
    bad = [dict(base, nodesAll=["a", "a"]),
           dict(base, partitionsToAssign={"0": {"name": "0", "nodesByState": {"dead": ["a"]}}}),
           dict(base, model={"a": {"priority": 1, "constraints": 1}, "b": {"priority": 0, "constraints": 1}}),
           dict(base, partitionsToAssign={"0": {"name": "zero", "nodesByState": {}}}),
           dict(base, booster="other"),          # a NodeScoreBooster that is not the cbgt one (plan.go:693)
           dict(base, booster="sorter")]         # a CustomNodeSorter (plan.go:580)
    for got in run_cli(build_emu(), bad):
        assert not got["handled"] and got["why"]


def _random_cases(lo, hi):
    from randgen import random_case
    cases = []
    for seed in range(lo, hi):
        c = random_case(seed)
        if c["partitionsToAssign"] is None:
            c["partitionsToAssign"] = {}
        c["nodesToRemove"] = c["nodesToRemove"] if c["nodesToRemove"] is not None else None
        c["source"] = "random seed %d" % seed
        cases.append(c)
    return cases


def _check_random(cases, results):
    """Random API-level inputs (maps that only partly overlap, weights for some partitions, names outside nodesAll,
    partitions only in prevMap ...): what the mirror's name-ordered joins see.  The literal oracle is the judge;
    inputs it refuses (the reference would panic) the mirror must refuse too."""
    from blance_amd import problem
    from helpers import build_from_case
    n = 0
    for c, got in zip(cases, results):
        try:
            build_from_case(c)
        except problem.Unsupported:
            assert not got["handled"], c["source"]
            continue
        prev_o = R.partition_map_from_json(copy.deepcopy(c["prevMap"]))
        assign_o = prev_o if c.get("aliased") else R.partition_map_from_json(copy.deepcopy(c["partitionsToAssign"]))
        opts_o = R.Options(c.get("modelStateConstraints"), c.get("partitionWeights"), c.get("stateStickiness"),
                           c.get("nodeWeights"), c.get("nodeHierarchy"), c.get("hierarchyRules"))
        info = {}
        want, want_w = R.plan_next_map_ex(prev_o, assign_o, list(c["nodesAll"]), c["nodesToRemove"], c["nodesToAdd"],
                                          c["model"], opts_o, c.get("booster"), info=info)
        assert got["handled"], (c["source"], got["why"])
        assert got["nextMap"] == R.partition_map_to_json(want), c["source"]
        assert got["warnings"] == (want_w or {}), c["source"]
        assert got["iterations"] == info["iterations"] and got["converged"] == info["converged"], c["source"]
        assert got["prevMap"] == R.partition_map_to_json(prev_o), c["source"]
        assert got["partitionsToAssign"] == R.partition_map_to_json(assign_o), c["source"]
        n += 1
    return n


def test_cpp_mirror_on_random_cases_emulated():
    from test_simt_emulated import build_emu
    cases = _random_cases(3000, 3160)
    assert _check_random(cases, run_cli(build_emu(), cases)) > 100


def test_cpp_mirror_threaded_paths_emulated(golden_cases):
    """What the mirror does with several threads at a million partitions -- the maps collected subtree by subtree, the
    partitions flattened in ranges, the static order by a threaded radix sort, the result built in blocks, the stores side by
    side, everything the result holds in the call's region (call_arena.cpp) -- forced onto the reference's tables, random
    cases and maps with names missing from prevMap: same results, same caller-visible mutations, same object identities."""
    from test_simt_emulated import build_emu
    for t in (2, 5):
        _check(golden_cases, run_cli(build_emu(), golden_cases, threads=t))
    cases = _random_cases(3200, 3320)
    assert _check_random(cases, run_cli(build_emu(), cases, threads=3)) > 70
    cases = _sparse_miss_cases(5200, 5280)
    assert _check_random(cases, run_cli(build_emu(), cases, threads=4)) > 10


def test_call_arena_on_its_own(tmp_path):
    """The region allocator behind the mirror's result objects (call_arena.cpp): built on four threads, deleted piecemeal on
    four others, chunks recycled and reused, large / over-aligned requests and everything outside a Scope left to malloc."""
    exe = str(tmp_path / "arena_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "tools", "arena_test.cpp"),
                           os.path.join(HOST, "call_arena.cpp")])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and p.stdout.strip().startswith("ok"), p.stdout + p.stderr


def _sparse_miss_cases(lo, hi):
    """Larger maps in which prevMap lacks one or two of the partitions to assign (and has one of its own): the store of
    plan.go:49-52 then fills recorded slots for most names and inserts the few others (blance_api.cpp: store)."""
    import random as _random
    from randgen import random_case
    cases = []
    for seed in range(lo, hi):
        c = random_case(seed, max_nodes=10, max_parts=72)
        if not c["partitionsToAssign"] or len(c["partitionsToAssign"]) < 34:
            continue
        rng = _random.Random(seed)
        prev = {}
        for name, part in c["partitionsToAssign"].items():
            old = (c["prevMap"] or {}).get(name)
            prev[name] = copy.deepcopy(old if old is not None else part)
        for name in rng.sample(sorted(prev), rng.choice([1, 2])):
            del prev[name]
        prev["only-in-prev"] = {"name": "only-in-prev", "nodesByState": {}}
        c["prevMap"] = prev
        c["aliased"] = False
        c["nodesToRemove"] = []
        c["source"] = "sparse-miss seed %d" % seed
        cases.append(c)
    return cases


def test_cpp_mirror_store_with_few_missing_names_emulated():
    from test_simt_emulated import build_emu
    cases = _sparse_miss_cases(5000, 5120)
    assert len(cases) > 20
    assert _check_random(cases, run_cli(build_emu(), cases)) > 15


@pytest.mark.gpu
def test_cpp_mirror_on_random_cases_gpu():
    from blance_amd import hip
    cases = _random_cases(3000, 3600)
    assert _check_random(cases, run_cli(hip.LIB_PATH, cases)) > 400


@pytest.mark.gpu
def test_cpp_mirror_on_golden_cases_gpu(golden_cases):
    from blance_amd import hip
    _check(golden_cases, run_cli(hip.LIB_PATH, golden_cases))
    _check(golden_cases, run_cli(hip.LIB_PATH, golden_cases, eager=True))


# ---- CalcPartitionMoves through the C++ mirror (blance::CalcPartitionMovesBatch)
def _nbs_map(out, m):
    out.append(str(len(m)))
    for name, nbs in m.items():
        out.append(name)
        out.append(str(len(nbs)))
        for state, lst in nbs.items():
            out.append(state)
            _list(out, lst)


def run_moves_cli(lib, cases):
    """cases: (favorMinNodes, states, begMap, endMap) with maps {partition: nodesByState}."""
    out = [str(len(cases))]
    for favor, states, beg, end in cases:
        out.append("1" if favor else "0")
        _list(out, states)
        _nbs_map(out, beg)
        _nbs_map(out, end)
    p = subprocess.run([build_cli(), lib, "moves"], input="\n".join(out) + "\n", capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads(p.stdout)


def _moves_cases():
    import random
    from test_moves import random_pair
    with open(os.path.join(ROOT, "tests", "golden", "moves_cases.json")) as f:
        golden = json.load(f)["calcPartitionMoves"]
    cases = []
    for favor in (False, True):
        rows = [c for c in golden if c["favorMinNodes"] == favor]
        cases.append((favor, rows[0]["states"], {str(i): c["before"] for i, c in enumerate(rows)},
                      {str(i): c["after"] for i, c in enumerate(rows)}))
    rng = random.Random(11)
    states = ["primary", "replica", "dead"]
    nodes = ["n%d" % i for i in range(7)]
    for favor in (False, True):
        beg, end = {}, {}
        for i in range(300):
            b, e = random_pair(rng, states, nodes)
            if rng.random() < 0.9:
                beg["p%d" % i] = b
            if rng.random() < 0.9:
                end["p%d" % i] = e
        cases.append((favor, states, beg, end))
    return cases


def _check_moves(results, cases):
    from oracle import moves_ref
    for got, (favor, states, beg, end) in zip(results, cases):
        assert "error" not in got, got
        names = set(beg) | set(end)
        assert set(got) == names
        for name in names:
            want = moves_ref.calc_partition_moves(states, beg.get(name, {}), end.get(name, {}), favor)
            assert [tuple(m) for m in got[name]] == [tuple(m) for m in want], (name, favor)


def test_cpp_moves_mirror_emulated():
    from test_simt_emulated import build_emu
    cases = _moves_cases()
    _check_moves(run_moves_cli(build_emu(), cases), cases)


@pytest.mark.gpu
def test_cpp_moves_mirror_gpu():
    from blance_amd import hip
    cases = _moves_cases()
    _check_moves(run_moves_cli(hip.LIB_PATH, cases), cases)
