#!/usr/bin/env python3
"""Benchmark of the PlanNextMap hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 3|2|5]

One "step" = one whole PlanNextMap call (every sweep of planNextMapEx, plan.go:23-58) on
BASELINE.json's config 3: 1,048,576 partitions x 4,096 nodes, primary + 2 replicas, 3-level
rack/zone/DC hierarchy with an exclusion rule.  The problem is uploaded once (inputs resident in
HBM when the timed region starts); value = partition-state assignments per second over the timed
steps.

N > 1 (one process per GPU; `--gpus N` launches the ranks itself through torch.distributed.run
when it is not already running under it):
  * value: every rank plans an instance of the same shape (replicas, weak scaling) -- all ranks'
    assignments / the slowest rank's time;
  * "sharded": ONE plan with its region chains sharded over the ranks -- per chain pass one RCCL sum
    all-reduce of [flags | load-vector change] and one all-gather of the output slices (BASELINE.json
    config 4) -- timed the same way and reported beside it, whatever it is (DESIGN.md "Multi-GPU").
N = 1 also reports "sharded_on_one_gpu": the same sharded plan with 8 ranks as 8 contexts of this one
device (collectives staged through the host) -- it proves the path on real kernels, it is not a speed --
"general_regime": config 3's size in its general regime (a rebalance after every tenth node left; scrambled
non-numeric partition names + Zipf partition weights), each checked against the oracle's digest -- and
roofline.traffic measured in this run (two rocprofv3 --pmc passes of one more PlanNextMap call).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

T_START = time.perf_counter()
T_BLOCKS = {}             # seconds per block of the line (block_seconds)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SHARDED_TIMEOUT_S = 180    # the sharded leg of N > 1 runs (a few plans of ~20-60 ms plus the communicator) never needs this long
SCLK_GHZ = 2.4            # MI355X_MICROARCH.md: 256 CU x 2.4 GHz
N_CUS, SIMDS_PER_CU = 256, 4
K_CW = 24                 # words of a compact chain step record (blance_kernels.h kCW)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_sample_config5(pl, parts, nodes):
    """The CPU oracle beside config 5 ("configs 3/5 are sampled ... labelled"): the rebalance PlanNextMap -- every
    state pass of all ten sweeps -- on 1/32 of the partitions over the SAME 4,096 nodes (a step is an O(nodes) scan whatever the
    partition count, so assignments/s carries over), one core.  The plan it starts from is made on the GPU (untimed)."""
    from blance_amd import synth
    from oracle import loader
    sample_parts = max(1024, parts // 32)
    fp1 = synth.config5_initial(sample_parts, nodes)
    fp = synth.config5_rebalance(fp1, pl.plan(fp1), sample_parts, nodes)
    t0 = time.perf_counter()
    res = loader.plan(fp)
    dt = time.perf_counter() - t0
    return {"value": synth.assignments(fp) / dt, "unit": "assignments/s", "cores": 1, "kind": "port", "extrapolated": True,
            "host_cpus": os.cpu_count(), "cpu_model": cpu_model(), "sample_seconds": dt,
            "sample": "oracle/blance_oracle.c, the rebalance PlanNextMap (%d sweeps, both state passes of every sweep) on %d partitions x "
                      "%d nodes = 1/32 of the partitions, same nodes / weights law / model; per-step cost is O(nodes), so assignments/s "
                      "carries over: EXTRAPOLATED from the sample (all %d partitions: about 5 minutes on this core), %.1f s"
                      % (res.iterations, sample_parts, nodes, parts, dt)}


def cpu_baseline(parts, nodes, cfg, full=False):
    """CPU legs, one core each (the Go planner is single threaded), on a bounded sample of the same
    workload: (i) "port": oracle/blance_oracle.c, the id-based restatement; (ii) "naive": the
    string-keyed proxy of the reference's Go code path (hash maps + comparison sort calling Score
    twice per compare, plan.go:617-689), on a smaller sample, extrapolated linearly in partitions."""
    from blance_amd import synth
    from oracle import loader
    info = {"unit": "assignments/s", "cores": 1, "kind": "port", "host_cpus": os.cpu_count(), "cpu_model": cpu_model()}
    if cfg == 5:
        # config 5 (weighted rebalance, 10 sweeps): 1/32 of the partitions; the plan it starts
        # from is made on the GPU (untimed), the oracle is timed on the rebalance only
        from blance_amd import hip
        sample_parts = max(1024, parts // 32)
        fp1 = synth.config5_initial(sample_parts, nodes)
        pl = hip.Planner()
        fp = synth.config5_rebalance(fp1, pl.plan(fp1), sample_parts, nodes)
        pl.close()
        what = "the rebalance PlanNextMap"
        frac = "1/32"
    else:
        sample_parts = parts if full else max(1024, parts // 4)
        fp = synth.config_flat(cfg, P=sample_parts, N=nodes)
        what = "full PlanNextMap"
        frac = "all" if full else "1/4"
    t0 = time.perf_counter()
    res = loader.plan(fp)
    dt = time.perf_counter() - t0
    info["value"] = synth.assignments(fp) / dt
    info["extrapolated"] = sample_parts != parts
    if sample_parts == parts:
        info["sample"] = "oracle/blance_oracle.c, %s (%d sweeps) on all %d partitions x %d nodes: NOT extrapolated, %.1f s" % (
            what, res.iterations, sample_parts, nodes, dt)
    else:
        info["sample"] = ("oracle/blance_oracle.c, %s (%d sweeps) on %d partitions x %d nodes (%s of the partitions, same "
                          "nodes/hierarchy/model; per-step cost is O(nodes), so assignments/s carries over: EXTRAPOLATED "
                          "from the sample -- bench.py without --cpu-sample runs all of them, %s), %.1f s"
                          % (what, res.iterations, sample_parts, nodes, frac,
                             {5: "8 minutes", 3: "about 2 minutes"}.get(cfg, "under a second"), dt))
    # config 2 in full (65,536 x 256, not sampled), same port, same core
    fp2 = synth.config_flat(2)
    t0 = time.perf_counter()
    res2 = loader.plan(fp2)
    dt2 = time.perf_counter() - t0
    info["config2_full"] = {"value": synth.assignments(fp2) / dt2, "unit": "assignments/s", "sweeps": res2.iterations,
                            "partitions": fp2.n_parts, "nodes": fp2.n_nodes, "seconds": dt2, "extrapolated": False}
    try:
        from oracle import naive_loader
        info["naive_proxy"] = naive_loader.timed_sample(cfg, nodes)
    except Exception as e:                                  # the proxy is optional test infrastructure
        info["naive_proxy"] = {"error": str(e)[:200]}
    return info


def profile_json(name):
    """A PMC summary committed under profiles/ -- only if it was taken from the kernel sources as they are now."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, "no %s" % name
    with open(path) as f:
        data = json.load(f)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import profile_summary
        now = profile_summary.source_hash()
    except Exception:
        now = None
    if now != data.get("source_hash"):
        return None, "%s was taken from other kernel sources (%s, now %s)" % (name, data.get("source_hash"), now)
    return data, "profiles/%s (git %s, kernel sources %s)" % (name, data.get("git_head"), data.get("source_hash"))


def kernel_counters(data, prefix):
    """Sum the counters of every kernel whose short name starts with one of `prefix`."""
    tot, calls = {}, 0
    for name, row in (data or {}).get("kernels", {}).items():
        if not any(name.startswith(p) for p in prefix):
            continue
        calls += row.get("calls", 0)
        for k, v in row.items():
            if k != "calls":
                tot[k] = tot.get(k, 0) + v
    return tot, calls


def live_pmc(args):
    """roofline.traffic measured IN THIS RUN: one more PlanNextMap call of the same configuration under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and again under `--pmc WRITE_SIZE` (separate passes: the TCC has no
    room for both; no other trace domain with --pmc), summed per kernel.  Returns ({"kernels": {short name:
    {"calls", "FETCH_SIZE", "WRITE_SIZE"}}, "plan_calls": 1}, description) or (None, why not)."""
    import glob
    import shutil
    import sqlite3
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "no rocprofv3 on this machine"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import profile_summary
    per = {}
    work = tempfile.mkdtemp(prefix="blance_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", os.path.join(work, counter), "-o", "p", "--",
                   sys.executable, os.path.abspath(__file__), "--config", str(args.config), "--steps", "1", "--warmup", "0",
                   "--no-cpu-baseline", "--no-sharded", "--no-extra", "--no-live-pmc", "--no-other-configs", "--no-transfers",
                   "--no-replicas", "--no-rccl-one-rank"]
            if args.parts:
                cmd += ["--parts", str(args.parts)]
            if args.nodes:
                cmd += ["--nodes", str(args.nodes)]
            if args.no_periodic:
                cmd += ["--no-periodic"]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=600)
            dbs = glob.glob(os.path.join(work, counter, "**", "*_results.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None, "rocprofv3 --pmc %s gave no result (rc %d): %s" % (counter, r.returncode, (r.stderr or r.stdout)[-200:])
            cur = sqlite3.connect(dbs[0]).cursor()
            g = profile_summary.tables(cur)
            rows = cur.execute(
                "select s.kernel_name, i.name, count(distinct d.id), sum(e.value) from %s e join %s i on e.pmc_id=i.id join %s d on "
                "e.event_id=d.event_id join %s s on d.kernel_id=s.id group by s.kernel_name, i.name"
                % (g("pmc_event"), g("info_pmc"), g("kernel_dispatch"), g("kernel_symbol"))).fetchall()
            for kname, cname, n, v in rows:
                row = per.setdefault(profile_summary.short(kname), {"calls": n})
                row[cname] = v
                row["calls"] = max(row["calls"], n)
    except Exception as e:
        return None, "live counters failed: %s: %s" % (type(e).__name__, str(e)[:200])
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return ({"kernels": per, "plan_calls": 1},
            "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes) around one more "
            "PlanNextMap call of this configuration; bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB (gfx950 counts half of a streaming "
            "read, MI355X_MICROARCH.md HBM), per launch")


def general_regime(pl, fp3, res3, steps):
    """Config 3's size outside its most regular instance: (a) the rebalance of the headline plan after every
    tenth node left, (b) scrambled non-numeric partition names + Zipf partition weights.  Each: uploaded, one warm-up,
    `steps` timed calls, digest against the oracle's (tests/golden/config3_full_size_properties.json, config3_general_regime.json)."""
    from blance_amd import synth
    out = []
    gold = {}
    for name in ("config3_full_size_properties.json", "config3_general_regime.json"):
        path = os.path.join(ROOT, "tests", "golden", name)
        if os.path.exists(path):
            with open(path) as f:
                gold.update(json.load(f))
    full = fp3.n_parts == 1 << 20 and fp3.n_nodes == 4096
    work = [("config 3's plan rebalanced after every tenth node left (prevMap = partitionsToAssign = the headline plan, "
             "nodesToRemove = node ids with id % 10 == 3)", lambda: synth.config3_rebalance_flat(fp3, res3), gold.get("rebalance")),
            ("config 3 with scrambled non-numeric partition names and Zipf partition weights (fresh plan)",
             lambda: synth.config3_named_weighted_flat(fp3.n_parts, fp3.n_nodes), gold.get("named_weighted"))]
    for label, make, want in work:
        try:
            fp = make()
            pl.upload(fp)
            pl.plan_resident()
            t0 = time.perf_counter()
            acc = {"pass_ms": 0.0, "flat_ms": 0.0, "device_ms": 0.0}
            r = None
            for _ in range(steps):
                r = pl.plan_resident()
                acc["pass_ms"] += r.pass_kernel_ms
                acc["flat_ms"] += r.flat_pass_ms
                acc["device_ms"] += r.device_ms
            dt = (time.perf_counter() - t0) / steps
            digest = pl.download().digest()
            a = synth.assignments(fp)
            out.append({"workload": label, "headline": False, "partitions": fp.n_parts, "nodes": fp.n_nodes, "steps": steps,
                        "ms_per_step": dt * 1e3, "value": a / dt, "unit": "assignments/s", "sweeps_per_call": r.iterations,
                        "converged": bool(r.converged), "device_ms_per_step": acc["device_ms"] / steps,
                        "pass_kernel_ms_per_step": acc["pass_ms"] / steps, "flat_pass_ms_per_step": acc["flat_ms"] / steps,
                        "steps_bulk": int(r.steps_batched), "steps_one_by_one": int(r.steps_sequential), "result_sha256": digest,
                        "matches_oracle_digest": (want["digest"] == digest) if (want and full) else None})
        except Exception as e:
            out.append({"workload": label, "error": "%s: %s" % (type(e).__name__, str(e)[:300])})
    return out


def other_configs(pl, steps, cpu=True, keep=None):
    """BASELINE.json's other single-GPU configurations under the same clock as the headline: config 2 in full
    (65,536 x 256) and config 5 at its named size -- the initial plan over the old nodes, then the rebalance after a tenth of
    the nodes left and a tenth joined (the configuration BASELINE names for 8 GPUs: its flat passes are ONE dependency chain
    per pass, so one GPU plans it).  Each: uploaded, timed resident calls, digest against the CPU oracle's
    (tests/golden/config_digests.json), and the whole call against the HBM roofline by SURVEY.md 8(d)'s dense-scan bytes."""
    from blance_amd import synth
    with open(os.path.join(ROOT, "tests", "golden", "config_digests.json")) as f:
        gold = json.load(f)
    out = []

    def one(label, cfg, fp, n_steps, warm, want):
        pl.upload(fp)
        for _ in range(warm):
            pl.plan_resident()
        t0 = time.perf_counter()
        dev = pm = fm = 0.0
        r = None
        for _ in range(n_steps):
            r = pl.plan_resident()
            dev += r.device_ms
            pm += r.pass_kernel_ms
            fm += r.flat_pass_ms
        dt = (time.perf_counter() - t0) / n_steps
        res = pl.download()
        digest = res.digest()
        a = synth.assignments(fp)
        dense = synth.algorithmic_bytes_per_sweep(fp) * r.iterations
        gbps = dense / (dev / n_steps * 1e-3) / 1e9
        out.append({"config": cfg, "workload": label, "partitions": fp.n_parts, "nodes": fp.n_nodes, "steps": n_steps, "warmup": warm,
                    "ms_per_step": dt * 1e3, "device_ms_per_step": dev / n_steps, "value": a / dt, "unit": "assignments/s",
                    "sweeps_per_call": r.iterations, "converged": bool(r.converged),
                    "pass_kernel_ms_per_step": pm / n_steps, "flat_pass_ms_per_step": fm / n_steps,
                    "steps_bulk": int(r.steps_batched), "steps_one_by_one": int(r.steps_sequential),
                    "roofline": {"bound": "hbm", "model": "SURVEY.md 8(d): the reference's dense scan, N x 16 B + 40 B per step and sweep, whole call",
                                 "bytes_per_call": float(dense), "achieved": gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBS},
                    "result_sha256": digest, "matches_oracle_digest": (want["digest"] == digest and want["iterations"] == r.iterations) if want else None})
        return res
    try:
        one("BASELINE.json config 2: 65536 partitions x 256 nodes, primary+1 replica, flat", 2, synth.config_flat(2), max(steps, 5), 1, gold.get("config2"))
    except Exception as e:
        out.append({"config": 2, "error": "%s: %s" % (type(e).__name__, str(e)[:300])})
    try:
        c5 = gold.get("config5") or {}
        fp1 = synth.config5_initial(1 << 20, 4096)
        r1 = one("BASELINE.json config 5, setup: the initial plan of 1048576 Zipf-weighted partitions over the 3686 old nodes (node "
                 "weights, stickiness; flat)", 5, fp1, 2, 1, c5.get("initial"))
        fp2 = synth.config5_rebalance(fp1, r1, 1 << 20, 4096)
        del fp1
        one("BASELINE.json config 5: the rebalance after a tenth of the nodes left and a tenth joined (prevMap = the initial plan)",
            5, fp2, 2, 1, c5.get("rebalance"))
        if keep is not None:
            keep[0] = fp2                               # (the replicas block plans the same problem in a child process)
        if cpu:
            try:
                out[-1]["cpu_baseline"] = cpu_sample_config5(pl, 1 << 20, 4096)
            except Exception as e:
                out[-1]["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    except Exception as e:
        out.append({"config": 5, "error": "%s: %s" % (type(e).__name__, str(e)[:300])})
    return out


def rccl_one_rank(fp, want_digest, device, steps, unsharded_ms):
    """The RCCL transport on the one GPU there is: a communicator of ONE rank whose chain passes take the
    sharded branch (blance_options.reserved[2] & 16384), so that collective A -- ncclAllReduce of [flags | load-vector change]
    -- and collective B -- ncclAllGather of the output slice -- really run on the planner's stream, timed by the event pairs
    of blance_comm_time_ms.  What it measures: the per-collective latency a sharded plan pays (launch + RCCL's kernel on one
    rank; no xGMI hop); not a multi-GPU speed."""
    from blance_amd import hip
    try:
        pl = hip.Planner(device_id=device, shard_one_rank=True)
        pl.comm_init_rccl_one_rank()
        pl.upload(fp)
        pl.plan_resident()                             # warm-up: the communicator's first collectives set up its channels
        calls0, words0 = pl.comm_stats()
        ms0 = pl.comm_time_ms()
        t0 = time.perf_counter()
        dev = 0.0
        r = None
        for _ in range(steps):
            r = pl.plan_resident()
            dev += r.device_ms
        dt = (time.perf_counter() - t0) / steps
        calls1, words1 = pl.comm_stats()
        ms1 = pl.comm_time_ms()
        digest = pl.download().digest()
        pl.close()
        n = calls1 - calls0
        return {"what": "one PlanNextMap on a RCCL communicator of ONE rank with the sharded branch forced: ncclAllReduce + ncclAllGather "
                        "execute per chain pass (the latency a sharded plan pays per collective; no xGMI hop on one rank)",
                "rccl_world_size": 1, "steps": steps, "ms_per_step": dt * 1e3, "device_ms_per_step": dev / steps,
                "unsharded_ms_per_step": unsharded_ms, "sweeps_per_call": r.iterations,
                "comm_calls_per_plan": n / float(steps), "comm_bytes_per_plan": 4.0 * (words1 - words0) / steps,
                "comm_device_ms_per_plan": (ms1 - ms0) / steps, "us_per_collective": (ms1 - ms0) * 1e3 / n if n else None,
                "comm_timing": "hipEvents on either side of every ncclAllReduce / ncclAllGather on the planner's stream",
                "same_digest_as_unsharded_plan": digest == want_digest}
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}


def replicas_worker(args):
    """`bench.py --replicas-per-gpu 1,4,16,64 [--config 5|3]`: R independent plans at once on ONE MI355X -- R contexts (own stream,
    buffers, staging) driven by R host threads of this process, the arrangement of an embedder that plans many indexes.  The
    sequential passes leave 97-99.6 % of the chip idle, so this is the lever exactness leaves: aggregate assignments/s, the
    slowest plan's latency and the host cores it takes.  Never the headline (parallelism: replicas xR on 1 GPU)."""
    import threading
    if args.sync_mode != "default":
        # how a host thread waits for its stream: the runtime's default spins (one core per planning thread); "blocking" /
        # "yield" let more contexts than cores plan at once (hipSetDeviceFlags, before the first context of the process)
        import ctypes
        rt = ctypes.CDLL("libamdhip64.so")
        rt.hipSetDevice(args.device)
        rc = rt.hipSetDeviceFlags({"blocking": 4, "yield": 2, "spin": 1}[args.sync_mode])
        if rc:
            print(json.dumps({"replicas_on_one_gpu": {"error": "hipSetDeviceFlags(%s) = %d" % (args.sync_mode, rc)}}), flush=True)
            return
    from blance_amd import abi, hip, synth
    Rs = sorted(set(int(x) for x in args.replicas_per_gpu.split(",") if x))
    cfg = args.config
    P, N = args.parts or 1 << 20, args.nodes or 4096
    with open(os.path.join(ROOT, "tests", "golden", "config_digests.json")) as f:
        gold = json.load(f)
    full = P == 1 << 20 and N == 4096
    t_start = time.perf_counter()
    if cfg == 5:
        if args.problem_npz:                            # (made by the parent: interning a million partitions takes a minute)
            fp = abi.FlatProblem.load_npz(args.problem_npz)
        else:
            fp1 = synth.config5_initial(P, N)
            pl0 = hip.Planner(device_id=args.device)
            fp = synth.config5_rebalance(fp1, pl0.plan(fp1), P, N)
            pl0.close()
            del fp1
        problems = lambda r: fp                         # noqa: E731 -- every context uploads and plans its own copy
        want = (gold.get("config5") or {}).get("rebalance") if full else None
        steps, warm = 1, 0                              # a plan is seconds: first-call allocations (ms) do not show
        instances = "the same config-5 rebalance in every context (the library shares nothing between contexts)"
    else:
        rot = {}

        def problems(r):                                # rotated instances: other per-node tables, the same plan as ids
            r %= 8
            if r not in rot:
                rot[r] = synth.config_flat(3, P=P, N=N, **({"rotate": (512 * r) % N} if r else {}))
            return rot[r]
        want = gold.get("config3") if full else None
        steps, warm = 5, 1
        instances = "config 3 rotated by 512 (r mod 8) node names in context r (other hierarchy tables, the same plan as ids)"
    planners, out = [], []
    a = None
    t1 = None
    speed_prev = 1.0
    for R in Rs:
        if t1 is not None:
            predicted = R * t1 * (steps + warm) / max(speed_prev, 1.0)
            if time.perf_counter() - t_start + predicted > args.replica_budget_s:
                out.append({"R": R, "skipped": "predicted %.0f s at the concurrency seen so far (x%.1f): beyond the %d s budget of this block"
                                               % (predicted, speed_prev, args.replica_budget_s)})
                continue
        while len(planners) < R:
            pl = hip.Planner(device_id=args.device)
            pl.upload(problems(len(planners)))
            planners.append(pl)
        a = synth.assignments(problems(0))
        bar = threading.Barrier(R + 1)
        lat = [[] for _ in range(R)]
        err = []

        def work(i):
            try:
                for _ in range(warm):
                    planners[i].plan_resident()
                bar.wait()
                for _ in range(steps):
                    t0 = time.perf_counter()
                    planners[i].plan_resident()
                    lat[i].append(time.perf_counter() - t0)
                bar.wait()
            except BaseException as e:                  # noqa: BLE001
                err.append(e)
                bar.abort()
        th = [threading.Thread(target=work, args=(i,)) for i in range(R)]
        for t in th:
            t.start()
        try:
            bar.wait()
            c0, w0 = sum(os.times()[:2]), time.perf_counter()
            bar.wait()
            wall = time.perf_counter() - w0
            cpu_s = sum(os.times()[:2]) - c0
        except threading.BrokenBarrierError:
            wall = cpu_s = float("nan")
        for t in th:
            t.join()
        if err:
            out.append({"R": R, "error": "%s: %s" % (type(err[0]).__name__, str(err[0])[:200])})
            break
        ok = None
        if want:
            ok = all(planners[i].download().digest() == want["digest"] for i in range(R))
        allv = [x for l in lat for x in l]
        if t1 is None:
            t1 = wall / steps * (1.0 / R if R > 1 else 1.0)      # (a list that does not start at 1: assume no concurrency)
        speed_prev = max(1.0, R * t1 / (wall / steps))           # plans in flight at once, as seen at this R
        out.append({"R": R, "parallelism": "replicas x%d on 1 GPU" % R, "steps_per_replica": steps, "wall_s": wall,
                    "aggregate_value": R * a * steps / wall, "unit": "assignments/s",
                    "plan_latency_ms": {"slowest": max(allv) * 1e3, "fastest": min(allv) * 1e3, "mean": sum(allv) / len(allv) * 1e3},
                    "speedup_vs_R1": (R * a * steps / wall) / out[0]["aggregate_value"] if out and "aggregate_value" in out[0] else 1.0,
                    "host_threads": R, "host_cores_used": cpu_s / wall if wall == wall and wall > 0 else None,
                    "every_digest_is_the_oracles": ok})
    for pl in planners:
        pl.close()
    print(json.dumps({"replicas_on_one_gpu": {
        "config": cfg, "partitions": P, "nodes": N, "instances": instances, "headline": False,
        "what": "R contexts of one process, one host thread each, planning at once on ONE MI355X; aggregate = R x assignments x steps / wall",
        "hw_queues_env": os.environ.get("GPU_MAX_HW_QUEUES"), "host_wait": args.sync_mode, "host_cpus": os.cpu_count(),
        "host_cpus_usable": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
        "block_seconds": time.perf_counter() - t_start, "runs": out}}), flush=True)


def replicas_block(args, cfg, budget_s, problem_npz=None, rs=None, sync_mode=None):
    """The replicas_on_one_gpu block of the default line: the worker above in a process of its own (GPU_MAX_HW_QUEUES must be in
    the environment before the HIP runtime starts: by default the streams of a process share 4 hardware queues)."""
    cmd = [sys.executable, os.path.abspath(__file__), "--replicas-worker", "--replicas-per-gpu", rs or args.replicas_per_gpu or "1,4,16,64",
           "--config", str(cfg), "--replica-budget-s", str(budget_s), "--device", str(int(os.environ.get("LOCAL_RANK", "0"))),
           "--sync-mode", sync_mode or args.sync_mode]
    if problem_npz:
        cmd += ["--problem-npz", problem_npz]
    if args.parts:
        cmd += ["--parts", str(args.parts)]
    if args.nodes:
        cmd += ["--nodes", str(args.nodes)]
    env = dict(os.environ, GPU_MAX_HW_QUEUES=os.environ.get("GPU_MAX_HW_QUEUES", str(args.hw_queues)))
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=budget_s + 240)
        line = [x for x in p.stdout.splitlines() if x.startswith("{")]
        return json.loads(line[-1])["replicas_on_one_gpu"] if line else {"error": (p.stdout + p.stderr)[-300:]}
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}


def self_launch(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks."""
    port = 29400 + os.getpid() % 500
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, help="BASELINE.json config: 3 (headline), 2, or 5 (weighted rebalance)")
    ap.add_argument("--parts", type=int, default=0, help="override partition count (not the headline)")
    ap.add_argument("--nodes", type=int, default=0, help="override node count (not the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sharded", action="store_true", help="skip the 8-contexts-on-one-GPU leg of N = 1")
    ap.add_argument("--no-periodic", action="store_true", help="the all-blank chain pass without its periodic form (blance_amd/csrc/"
                    "k_period.h, the default since round 4): every step walked; not the headline")
    ap.add_argument("--no-extra", action="store_true", help="skip the general_regime block (two more workloads of config 3's size)")
    ap.add_argument("--no-live-pmc", action="store_true", help="skip the two rocprofv3 --pmc passes that measure roofline.traffic in this run")
    ap.add_argument("--cpu-sample", action="store_true", help="cpu_baseline on a quarter of the partitions, extrapolated (14 s) -- the default "
                    "times the oracle on ALL partitions of config 3 (about a minute on the GPU box's EPYC, two here)")
    ap.add_argument("--cpu-full", action="store_true", help="(the default since round 5; kept so that old command lines still parse)")
    ap.add_argument("--no-transfers", action="store_true", help="skip the transfers block (the problem uploaded and the result downloaded again, "
                    "pageable and page-locked): profiles of the plan itself use this")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the other_configs block (BASELINE configs 2 and 5 timed with digests, about 40 s)")
    ap.add_argument("--replicas-per-gpu", default="", help="R[,R...]: ONLY the replicas_on_one_gpu block -- R contexts planning at once on one "
                    "GPU (config 5 by default with --config 5, else config 3): aggregate assignments/s, slowest plan, host cores; never the headline")
    ap.add_argument("--replica-budget-s", type=int, default=240, help="time budget of the replicas block (an R predicted to overrun it is skipped)")
    ap.add_argument("--hw-queues", type=int, default=32, help="GPU_MAX_HW_QUEUES for the replicas block (the runtime's default maps all streams "
                    "of a process onto 4 hardware queues)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--replicas-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--problem-npz", default="", help=argparse.SUPPRESS)
    ap.add_argument("--sync-mode", default="blocking", choices=["default", "spin", "yield", "blocking"],
                    help="replicas block: how a host thread waits for its stream (hipSetDeviceFlags).  blocking (the block's default): the "
                         "thread sleeps until the stream is done -- 64 planning threads on 6 cores; default / spin: the runtime's busy wait, "
                         "one core per planning thread (measured: 64 contexts on 16 cores plan 3 x slower than under blocking)")
    ap.add_argument("--time-budget-s", type=int, default=270, help="optional blocks of the default line are skipped (and say so) once the run "
                    "has taken this long: the headline, roofline, cpu_baseline and other_configs always run")
    ap.add_argument("--no-replicas", action="store_true", help="skip the replicas_on_one_gpu blocks of the default line")
    ap.add_argument("--no-rccl-one-rank", action="store_true", help="skip the one-rank RCCL leg (ncclAllReduce / ncclAllGather timed on this GPU)")
    args = ap.parse_args()

    if args.replicas_worker:
        return replicas_worker(args)
    if args.replicas_per_gpu and args.gpus == 1 and "WORLD_SIZE" not in os.environ:
        # the standalone form: the block alone, for config 3 or 5, in a child with the hardware-queue count raised
        print(json.dumps({"replicas_on_one_gpu": replicas_block(args, args.config, args.replica_budget_s)}), flush=True)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    # BLANCE_BENCH_REHEARSAL=<emulated library>: the multi-rank control flow of this file on a machine without a GPU (gloo,
    # kernels under the SIMT emulator, the collectives over host memory) -- tests/test_dist.py runs it so that the first
    # 8-GPU run is not the first execution of these lines.  Its line says "rehearsal": true; its numbers mean nothing.
    rehearsal = os.environ.get("BLANCE_BENCH_REHEARSAL")
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        world = dist.get_world_size()               # what RCCL's communicator reports

    from blance_amd import dist_util, hip, synth
    rot = 0
    if rehearsal:
        pl = hip.Planner(lib_path=rehearsal, chain_min_parts=8, periodic=not args.no_periodic)
    else:
        pl = hip.Planner(device_id=local_rank, periodic=not args.no_periodic)      # raises without the HIP library / a device
    if args.config == 5:                            # the rebalance starts from a plan over the old nodes (setup, untimed)
        fp1 = synth.config5_initial(args.parts or 1 << 20, args.nodes or 4096)
        fp = synth.config5_rebalance(fp1, pl.plan(fp1), args.parts or 1 << 20, args.nodes or 4096)
        del fp1
    else:
        # N > 1: the replicas plan DIFFERENT instances of the shape (a node serving several indexes): rank r's nodesAll
        # starts 512 r names further on -- other per-node hierarchy tables on every rank, the same plan as ids (zones stay
        # aligned blocks), so every rank's digest is still the oracle's
        rot = (512 * rank) % (args.nodes or 4096) if (world > 1 and args.config == 3) else 0
        fp = synth.config_flat(args.config, P=args.parts or None, N=args.nodes or None, **({"rotate": rot} if rot else {}))
    P, N = fp.n_parts, fp.n_nodes
    t0 = time.perf_counter()
    pl.upload(fp)
    upload_s = time.perf_counter() - t0

    def barrier():
        if dist is not None:
            dist.barrier()
        if not rehearsal:
            torch.cuda.synchronize()

    def timed(steps, warmup):
        for _ in range(warmup):
            pl.plan_resident()
        barrier()
        t0 = time.perf_counter()
        acc = {"pass_ms": 0.0, "pass_launches": 0, "flat_ms": 0.0, "flat_passes": 0, "device_ms": 0.0, "blank_ms": 0.0, "blank_launches": 0,
               "stay_ms": 0.0, "stay_launches": 0}
        r = None
        for _ in range(steps):
            r = pl.plan_resident()                  # returns after the device finished the call
            acc["pass_ms"] += r.pass_kernel_ms
            acc["pass_launches"] += r.pass_kernel_launches
            acc["flat_ms"] += r.flat_pass_ms
            acc["flat_passes"] += r.flat_passes
            acc["blank_ms"] += r.blank_pass_ms
            acc["blank_launches"] += r.blank_pass_launches
            acc["stay_ms"] += r.stay_pass_ms
            acc["stay_launches"] += r.stay_pass_launches
            acc["device_ms"] += r.device_ms
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            dt = dist_util.max_over_ranks(dt)
        return dt, acc, r

    T_BLOCKS["setup_and_first_upload"] = round(time.perf_counter() - T_START, 2)
    t_h = time.perf_counter()
    dt, acc, r = timed(args.steps, args.warmup)
    T_BLOCKS["headline"] = round(time.perf_counter() - t_h, 2)
    iterations, batched, sequential = r.iterations, r.steps_batched, r.steps_sequential
    assignments = synth.assignments(fp)
    value = assignments * args.steps * world / dt

    t1 = time.perf_counter()
    res = pl.download()                             # (every rank: the sharded leg compares with it)
    download_s = time.perf_counter() - t1
    digest_all = res.digest()
    replicas_ok = None
    if dist is not None:
        box_ = [None] * world
        dist.all_gather_object(box_, digest_all)
        replicas_ok = len(set(box_)) == 1
    # ---- what the boundary costs in steady state (reported beside the resident number, never as `value`):
    # the same problem uploaded again and the result downloaded again into buffers that exist -- pageable arrays (staged
    # through the context's page-locked buffer by a few threads) and arrays from blance_host_alloc (DMA where they lie)
    xfer = None
    t_x = time.perf_counter()
    if rank == 0 and not rehearsal and not args.no_transfers:
        try:
            def again(f, arena):
                r_ = None
                ups, downs = [], []
                for _ in range(3):
                    t_ = time.perf_counter()
                    pl.upload(f)
                    ups.append(time.perf_counter() - t_)
                    pl.plan_resident()
                    if r_ is None:
                        r_ = pl.download(arena)
                    t_ = time.perf_counter()
                    pl.download(arena, into=r_)
                    downs.append(time.perf_counter() - t_)
                return min(ups[1:]), min(downs[1:]), r_.digest()
            up_pg, down_pg, d_pg = again(fp, None)
            arena = hip.HostArena()
            fpp = fp.pin(arena)
            up_pin, down_pin, d_pin = again(fpp, arena)
            nb = sum(a.nbytes for a in fp.arrays.values())
            ob = res.out_off.nbytes + res.out_nodes.nbytes + res.out_kind.nbytes
            xfer = {"upload_bytes": nb, "download_bytes": ob,
                    "pageable": {"upload_s": up_pg, "download_s": down_pg, "what": "numpy arrays: staged through the context's page-locked buffer"},
                    "page_locked": {"upload_s": up_pin, "download_s": down_pin, "what": "arrays from blance_host_alloc: DMA where they lie",
                                    "upload_GBps": nb / up_pin / 1e9, "download_GBps": ob / down_pin / 1e9},
                    "same_digest_both_ways": d_pg == d_pin == digest_all,
                    "upload_includes": "the O(P) checks of blance_validate on the device and every table the planner derives at upload"}
            pl.upload(fp)                               # (the legs below plan the resident problem again)
            pl.plan_resident()
            arena.close()
        except Exception as e:
            xfer = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    T_BLOCKS["transfers"] = round(time.perf_counter() - t_x, 2)
    out = None
    if rank == 0:
        digest = digest_all
        M = fp.n_states
        k_by_state = [int(fp.state_constraints[m]) for m in range(M)]
        RW = 4 + M * (1 + max(k_by_state + [1]))
        kernel_states = synth.pass_kernel_states(fp)
        # ---- per kernel: the bytes its schedule has to move per launch (DESIGN.md "Measurement": every step reads its
        # record and writes its choice; nothing else leaves registers / LDS) over its average launch duration, measured
        # in this run with HIP events on the planner's stream
        headline_shape = not args.parts and not args.nodes
        # HBM traffic per kernel: measured in this run when rocprofv3 is here (N = 1, not the rehearsal), else quoted from a
        # committed profile of EXACTLY these kernel sources (hash checked, no exceptions), else null
        hbm, hbm_src = None, "not measured (--no-live-pmc)"
        if world == 1 and not rehearsal and not args.no_live_pmc:
            t_p = time.perf_counter()
            hbm, hbm_src = live_pmc(args)
            T_BLOCKS["live_pmc"] = round(time.perf_counter() - t_p, 2)
        if hbm is None:
            why = hbm_src
            hbm, hbm_src = profile_json("r6_pmc_hbm_config%d.json" % args.config)
            hbm_src = "%s -- a committed profile of the same kernel sources, NOT measured in this run (%s)" % (hbm_src, why) if hbm else \
                      "none: %s; %s" % (why, hbm_src)
        sq, sq_src = profile_json("r6_pmc_sq_config%d.json" % args.config)
        survey_state = synth.algorithmic_bytes_per_state(fp)          # the reference's dense scan, per pass of a state

        def kernel_line(label, prefix, words, ms, launches, chains, dense_bytes, walked=True, waves_per_chain=1):
            if not launches:
                return None
            bytes_per_launch = 4.0 * words * P
            avg_ms = ms / launches
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            line = {"kernel": label, "launches": launches, "avg_launch_ms": avg_ms, "bound": "hbm", "achieved": achieved,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "algorithmic_bytes_per_launch": bytes_per_launch,
                    "algorithmic_bytes_model": "SCHEDULE bytes: %d partitions x %d words x 4 B -- every step reads its record and writes its "
                                               "choice; load tables stay in registers / LDS (DESIGN.md 5)" % (P, words),
                    # the other model, side by side: what SURVEY.md 8(d) prices a pass at (the reference's dense per-step scan)
                    "survey_8d_bytes_per_launch": float(dense_bytes),
                    "survey_8d_GBps": dense_bytes / (avg_ms * 1e-3) / 1e9,
                    "survey_8d_frac": dense_bytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "traffic": None, "traffic_from": hbm_src}
            if line["survey_8d_frac"] > 1.0:
                # 8(d) prices the reference's dense scan of every node per step; above 1 it cannot be a fraction of anything:
                # the scan is NOT executed here (verified stays, region-local candidates, the periodic copy)
                line["dense_scan_executed"] = False
            if hbm and headline_shape:
                tot, calls = kernel_counters(hbm, prefix)
                if calls:
                    # gfx950: FETCH_SIZE (KB) counts half of a streaming read's bytes (MI355X_MICROARCH.md, HBM)
                    line["traffic"] = (2.0 * tot.get("FETCH_SIZE", 0) + tot.get("WRITE_SIZE", 0)) * 1024 / calls
                    line["traffic_launches_counted"] = calls
            if chains:
                chain_steps = -(-P // chains)
                ns = avg_ms * 1e6 / chain_steps
                line["occupancy"] = {"waves": chains * waves_per_chain, "cus": N_CUS, "simd_slots": N_CUS * SIMDS_PER_CU,
                                     "simd_slots_used_frac": chains * waves_per_chain / float(N_CUS * SIMDS_PER_CU),
                                     "waves_per_chain": waves_per_chain}
                crit = {"chains_per_launch": chains, "dependent_steps_per_chain": chain_steps,
                        "avg_ns_per_dependent_step": ns, "avg_cycles_per_dependent_step": ns * SCLK_GHZ}
                if not walked:
                    # the periodic form walks two periods per region and COPIES the rest: per-step instruction figures of a
                    # chain nobody walks mean nothing (round 5's line printed 0.0031 instructions per step)
                    crit = {"chains_per_launch": chains, "dependent_steps_per_chain": chain_steps,
                            "note": "periodic form: two periods per region walked, the periodic stretch copied (k_period.h); "
                                    "no per-step issue figures for steps that are not walked"}
                elif sq and headline_shape:
                    tot, calls = kernel_counters(sq, prefix)
                    if calls and tot.get("SQ_WAVES"):
                        instr = tot.get("SQ_INSTS_VALU", 0) + tot.get("SQ_INSTS_SALU", 0) + tot.get("SQ_INSTS_LDS", 0) + \
                            tot.get("SQ_INSTS_SMEM", 0) + tot.get("SQ_INSTS_VMEM_RD", 0) + tot.get("SQ_INSTS_VMEM_WR", 0)
                        per_step = instr / tot["SQ_WAVES"] / chain_steps
                        crit.update({
                            "instructions_per_step_per_wave": per_step,
                            "cycles_per_instruction": ns * SCLK_GHZ / per_step if per_step else None,
                            # one wave issues at most one instruction per ~4.5 cycles (tools/profile/lat_micro.hip, measured)
                            "issue_bound_ns_per_step": per_step * 4.5 / SCLK_GHZ,
                            "issue_bound_frac": (per_step * 4.5 / SCLK_GHZ) / ns if ns else None,
                            "wave_active_frac": tot.get("SQ_ACTIVE_INST_ANY", 0) / tot["SQ_WAVE_CYCLES"] if tot.get("SQ_WAVE_CYCLES") else None,
                            "wave_waiting_frac": tot.get("SQ_WAIT_ANY", 0) / tot["SQ_WAVE_CYCLES"] if tot.get("SQ_WAVE_CYCLES") else None,
                            "counters_from": sq_src + " (committed profile of the same kernel sources, not this run)"})
                line["critical_path"] = crit
            return line

        kernels = []
        kmax = max(k_by_state)
        rule_states = [m for m in kernel_states if survey_state[m] > 0]
        dense_pass = max([survey_state[m] for m in rule_states] or [0])          # the pass-kernel states' pass (config 3 / 5: replica)
        dense_flat = max([survey_state[m] for m in range(M) if m not in kernel_states] or [0])
        if args.config == 5:
            kernels.append(kernel_line("k_pass_queue (flat replica pass, one wave64: the candidates as a sorted window over the lanes)",
                                       ("k_pass_queue", "k_pass_tree"), RW + 1 + kmax, acc["pass_ms"], acc["pass_launches"], 1, dense_pass,
                                       waves_per_chain=8))           # (the walking wave and its seven helper waves)
        else:
            zones = -(-N // 128)                     # zones of 8 racks x 16 nodes: one chain each
            kernels.append(kernel_line("all-blank replica pass of the first sweep (k_pass_chain_planes: scalar bit-plane automaton, one wave64 per "
                                       "hierarchy region; with k_period.h two periods walked and the periodic stretch copied)",
                                       ("k_pass_chain_planes", "k_pass_chain_blank", "k_period"), K_CW + 1 + kmax, acc["blank_ms"],
                                       acc["blank_launches"], zones, dense_pass, walked=bool(args.no_periodic)))
            kernels.append(kernel_line("k_pass_chain<2,2,false> (the replica pass of a later sweep that still moves steps: one workgroup of eight "
                                       "waves per hierarchy region -- one walks the region's steps in order, all eight test 512 steps for stays at a time)",
                                       ("k_pass_chainI",), K_CW + 1 + kmax, acc["pass_ms"] - acc["blank_ms"] - acc["stay_ms"],
                                       acc["pass_launches"] - acc["blank_launches"] - acc["stay_launches"], zones, dense_pass,
                                       waves_per_chain=8))           # (k_pass_chain.h: kChainWaves -- the walking wave and seven helpers of the stay test)
            kernels.append(kernel_line("k_stay_by_top (the replica pass of a converged sweep: every step verified as a stay, a wave per top priority "
                                       "node and a step per lane; the steps' grouping by top node is made a sweep ahead on a second stream)",
                                       ("k_stay_by_top",), K_CW + 1 + kmax, acc["stay_ms"], acc["stay_launches"], None, dense_pass))
        kernels.append(kernel_line("flat driver passes (k_flat_*, k_fresh_*, k_sort_*: several launches per pass)",
                                   ("k_flat", "k_fresh", "k_sort"), RW + 2, acc["flat_ms"], acc["flat_passes"], None, dense_flat))
        kernels = [k for k in kernels if k]
        kernels.sort(key=lambda k: -k["avg_launch_ms"] * k["launches"])
        dom = dict(kernels[0]) if kernels else {"bound": "hbm", "achieved": 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 0.0, "traffic": None}
        # the whole call against the HBM peak: every kernel's measured traffic over this run's device time
        whole = None
        if hbm and headline_shape:
            tot, calls = kernel_counters(hbm, ("",))
            per_call = (2.0 * tot.get("FETCH_SIZE", 0) + tot.get("WRITE_SIZE", 0)) * 1024 / max(hbm.get("plan_calls", 1), 1)
            dev_s = acc["device_ms"] / args.steps * 1e-3
            whole = {"hbm_bytes_per_call": per_call, "GBps": per_call / dev_s / 1e9, "frac_of_hbm_peak": per_call / dev_s / 1e9 / HBM_PEAK_GBS,
                     "from": hbm_src + "; over this run's device time"}
        dense_call = synth.algorithmic_bytes_per_sweep(fp) * iterations
        dense = dense_call * args.steps / (acc["device_ms"] * 1e-3) / 1e9
        out = {
            "metric": "partition-state assignments/sec at 1M partitions x 4,096 nodes",
            "value": value, "unit": "assignments/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE.json config %d: %d partitions x %d nodes, %s"
                                   % (args.config, P, N,
                                      {3: "primary+2 replicas, 3-level rack/zone/DC hierarchy, rule replica{include 2, exclude 1}",
                                       5: "primary+2 replicas, flat, Zipf partition weights, node weights, stickiness; rebalance "
                                          "after removing and adding a tenth of the nodes, prevMap = the plan over the old nodes"}
                                      .get(args.config, "primary+1 replica, flat")),
                       "partitions": P, "nodes": N, "assignments_per_call": assignments,
                       "sweeps_per_call": iterations, "parallelism": "replicas x%d" % world,
                       **({"replicas": "every rank plans its own instance of the shape: rank r's nodesAll starts 512 r names further on "
                                       "(other per-node hierarchy tables, the same plan as ids); every rank's digest checked against the oracle's: %s"
                                       % replicas_ok} if world > 1 and args.config == 3 else {}),
                       "steps_bulk": int(batched), "steps_one_by_one": int(sequential),
                       "headline": bool(args.config == 3 and headline_shape and not args.no_periodic),
                       **({"not_default": "--no-periodic: the all-blank chain pass walks every step (the default copies the periodic stretch, k_period.h)"}
                          if args.no_periodic else {})},
            "roofline": dict(dom, **{
                "reference_dense_scan_equivalent_GBps": dense,
                "survey_8d_whole_call": dict({"bytes": float(dense_call), "GBps": dense, "frac": dense / HBM_PEAK_GBS},
                                             **({"dense_scan_executed": False} if dense / HBM_PEAK_GBS > 1.0 else {})),
                "whole_call": whole,
                "note": "two byte models, side by side.  frac = SCHEDULE bytes (what this implementation has to move per launch: one record in, "
                        "one choice out per step) over the launch time over 8 TB/s -- small, because the dominant kernel is bound by the "
                        "instruction issue of one dependent chain per wave (critical_path), not by HBM; traffic = the PMC counters' bytes for "
                        "the same launch.  survey_8d_frac = the bytes SURVEY.md 8(d) prices the same pass at (the reference's dense scan of "
                        "every node per step: N x (16 + 4k) + 40 bytes per step) over the same time: it EXCEEDS 1 because that scan is not "
                        "executed -- the kernels keep the load tables in registers / LDS and read one record per step -- so 8(d)'s model "
                        "does not describe this implementation's memory traffic (results are bit-identical by digest)"}),
            "roofline_per_kernel": kernels,
            "device_ms_per_step": acc["device_ms"] / args.steps,
            # (ABI 6) stream synchronisations the host makes inside one PlanNextMap: each reads a few flag words that decide what is
            # launched next; VERDICT r5 item 4 asked for <= 4 -- it is what it was
            "host_syncs_per_call": int(r.host_syncs),
            "pass_kernel_ms_per_step": acc["pass_ms"] / args.steps, "flat_pass_ms_per_step": acc["flat_ms"] / args.steps,
            "transfers": dict(xfer or {}, **{
                "first_upload_s": upload_s, "first_download_s": download_s,
                "first_call_note": "the first upload allocates every device buffer, the first download the result arrays: one-time costs",
                "value_incl_transfers": (assignments / (dt / args.steps + xfer["page_locked"]["upload_s"] + xfer["page_locked"]["download_s"])
                                         if xfer and "page_locked" in xfer else assignments / (dt / args.steps + upload_s + download_s)),
                "value_incl_transfers_pageable": (assignments / (dt / args.steps + xfer["pageable"]["upload_s"] + xfer["pageable"]["download_s"])
                                                  if xfer and "pageable" in xfer else None)}),
            "result_sha256": digest,
        }
        ref = os.path.join(ROOT, "tests", "golden", "config_digests.json")
        if os.path.exists(ref) and headline_shape:
            with open(ref) as f:
                want = json.load(f).get("config%d" % args.config)
            if want:
                out["matches_oracle_digest"] = (want["rebalance"] if args.config == 5 else want)["digest"] == digest
        # ---- the blocks beside the headline, in the order of their weight; each is timed (block_seconds), and the optional
        # ones are skipped -- and say so -- once the run has used its time budget (a default run must finish within minutes)
        blocks = out["block_seconds"] = dict(T_BLOCKS)

        def timed_block(name, fn, optional=True):
            if optional and time.perf_counter() - T_START > args.time_budget_s:
                return {"skipped": "the run had used %d s when this block was due (--time-budget-s %d); run it alone"
                                   % (time.perf_counter() - T_START, args.time_budget_s)}
            t_ = time.perf_counter()
            r_ = fn()
            blocks[name] = round(time.perf_counter() - t_, 2)
            return r_
        if world == 1:
            out["sharded"] = ("see sharded_on_one_gpu; N > 1 runs report the RCCL plan here" if args.config == 3 else
                              "not applicable: flat passes are one chain (DESIGN.md 7); only config 3's region chains shard")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = timed_block("cpu_baseline", lambda: cpu_baseline(
                args.parts or (1 << 20) if args.config == 5 else P, N, args.config, full=not args.cpu_sample), optional=False)
            if args.config != 3:
                pl.upload(fp)                           # (config 5's sample planned another problem on this context)
        fp5 = [None]
        if world == 1 and args.config == 3 and headline_shape and not args.no_other_configs and not rehearsal:
            out["other_configs"] = timed_block("other_configs", lambda: other_configs(
                pl, max(1, min(args.steps, 5)), cpu=not args.no_cpu_baseline, keep=fp5), optional=False)
            pl.upload(fp)                               # (what follows plans the headline problem)
            for oc in out["other_configs"]:             # the CPU figure NEXT TO every GPU figure
                if oc.get("config") == 2 and "error" not in oc and "cpu_baseline" in out:
                    c2 = out["cpu_baseline"].get("config2_full") or {}
                    oc["cpu_baseline"] = dict(c2, cores=1, kind="port", sample="oracle/blance_oracle.c on all of config 2, one core")
        if world == 1 and args.config == 3 and not args.no_extra:
            out["general_regime"] = timed_block("general_regime", lambda: general_regime(pl, fp, res, max(1, min(args.steps, 3))))
        if world == 1 and args.config == 3 and not args.no_rccl_one_rank and not rehearsal:
            out["rccl_one_rank"] = timed_block("rccl_one_rank", lambda: rccl_one_rank(
                fp, digest, local_rank, max(3, min(args.steps, 10)), dt * 1e3 / args.steps))
        if world == 1 and args.config == 3 and headline_shape and not args.no_replicas and not rehearsal:
            # (children of their own: GPU_MAX_HW_QUEUES has to be in the environment before the HIP runtime starts)
            def both():
                r3 = replicas_block(args, 3, 40)
                npz = None
                if fp5[0] is not None:
                    npz = "/dev/shm/blance_bench_config5_%d.npz" % os.getpid()
                    try:
                        fp5[0].save_npz(npz)
                    except Exception:
                        npz = None
                try:
                    r5 = replicas_block(args, 5, 90, problem_npz=npz)
                finally:
                    if npz and os.path.exists(npz):
                        os.remove(npz)
                return [r3, r5]
            out["replicas_on_one_gpu"] = timed_block("replicas_on_one_gpu", both)
        fp5[0] = None
        if world == 1 and args.config == 3 and not args.no_sharded:
            out["sharded_on_one_gpu"] = timed_block("sharded_on_one_gpu", lambda: sharded_on_one_gpu(fp, digest, local_rank, dt / args.steps))
        if world == 1 and not args.no_extra and not rehearsal:
            out["host_end_to_end"] = timed_block("host_end_to_end", lambda: host_end_to_end(args.config))
        blocks["total"] = round(time.perf_counter() - T_START, 2)
    # ---- one plan over all ranks (config 4).  The line of the replicas is ready before this starts: RCCL is bound at
    # run time inside the library and has never met this node, so a watchdog prints that line if the leg does not return.
    sharded = None
    if dist is not None:
        import threading

        def give_up():
            if rank == 0:
                out["sharded"] = {"error": "no answer within %d s (RCCL communicator / all-reduce inside the library)" % SHARDED_TIMEOUT_S}
                print(json.dumps(out), flush=True)
            os._exit(0)
        dog = threading.Timer(SHARDED_TIMEOUT_S, give_up)
        dog.daemon = True
        dog.start()
        try:
            replica_digest = digest_all
            if args.config != 3:
                # configs 2 and 5 have no hierarchy rule: their passes are flat -- one dependency chain per pass, nothing to shard
                sharded = "not applicable: flat passes are one chain (DESIGN.md 7); only config 3's region chains shard"
            else:
                if rehearsal:
                    ar, ag = dist_util.gloo_collectives(pl, dist)
                    pl.comm_set_callback(rank, world, ar, ag)
                else:
                    dist_util.shard_plan_rccl(pl, dist)
                if rot:                                 # one plan on all ranks: every rank holds the SAME problem
                    pl.upload(synth.config_flat(args.config, P=args.parts or None, N=args.nodes or None))
                calls0, words0 = pl.comm_stats()
                cms0 = 0.0 if rehearsal else pl.comm_time_ms()
                sdt, sacc, sr = timed(args.steps, args.warmup)
                calls1, words1 = pl.comm_stats()
                cms1 = 0.0 if rehearsal else pl.comm_time_ms()
                n_plans = args.steps + args.warmup
                sdig = pl.download().digest()
                box = [None] * world
                dist.all_gather_object(box, sdig)
                sharded = {"what": "one PlanNextMap with its region chains sharded over the ranks; per chain pass one RCCL int32 sum "
                                   "all-reduce of [flags | load-vector change] and one all-gather of the output slices",
                           "comm_calls_per_plan": (calls1 - calls0) / float(n_plans),
                           "comm_bytes_per_plan": 4.0 * (words1 - words0) / n_plans,
                           "rccl_world_size": world, "ms_per_step": sdt * 1e3 / args.steps,
                           "comm_device_ms_per_plan": (cms1 - cms0) / n_plans,
                           "us_per_collective": ((cms1 - cms0) * 1e3 / (calls1 - calls0)) if calls1 > calls0 else None,
                           "comm_timing": "hipEvents on either side of every ncclAllReduce / ncclAllGather on the planner's stream, rank 0"
                                          if not rehearsal else "not measured in the rehearsal (gloo on the host)",
                           "value": assignments * args.steps / sdt, "unit": "assignments/s", "scaling": "strong",
                           "device_ms_per_step": sacc["device_ms"] / args.steps,
                           "same_digest_on_every_rank": len(set(box)) == 1,
                           "same_digest_as_single_rank_plan": sdig == replica_digest,
                           "speedup_vs_one_rank_of_this_run": (dt / args.steps) / (sdt / args.steps)}
        except Exception as e:                      # the replicas line is still worth printing
            sharded = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        dog.cancel()

    if rank == 0:
        if sharded:
            out["sharded"] = sharded
        if rehearsal:
            out["rehearsal"] = True
            out["data"] = "REHEARSAL on the CPU (emulated kernels, gloo): not a measurement"
        print(json.dumps(out), flush=True)
    pl.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def sharded_on_one_gpu(fp, want_digest, device, one_rank_s, n_ranks=8):
    """BASELINE.json config 4 without a multi-GPU node: the SAME sharded code path (chains of a slice of the regions per
    rank, collective A = all-reduce of [flags | load change], collective B = all-gather of output slices) with the ranks as
    contexts of this one device, driven by threads; the collectives are staged through the host by the harness.  It shows
    that the path runs on the real kernels and gives the right answer; the time is NOT a multi-GPU time (the ranks share
    one GPU and the collectives are host copies) and is reported as such."""
    from blance_amd import dist_util, hip
    try:
        grp, planners = dist_util.local_sharded_planners(n_ranks, lambda: hip.Planner(device_id=device))

        def work(rank, pl):
            pl.plan(fp)                                 # warm-up (first touch of every buffer)
            t0 = time.perf_counter()
            res = pl.plan(fp)
            return time.perf_counter() - t0, res.digest(), res.struct.device_ms, pl.comm_stats()
        res = grp.run(planners, work)
        for pl in planners:
            pl.close()
        calls, words = res[0][3]
        return {"what": "one PlanNextMap sharded over %d ranks = %d contexts on ONE GPU, collectives staged through the host by the "
                        "test harness: a correctness run of the sharded path, not a multi-GPU speed" % (n_ranks, n_ranks),
                "ranks": n_ranks, "same_digest_on_every_rank": len({r[1] for r in res}) == 1,
                "same_digest_as_single_rank_plan": res[0][1] == want_digest,
                "ms_per_plan_host_to_host_slowest_rank": max(r[0] for r in res) * 1e3,
                "device_ms_slowest_rank": max(r[2] for r in res),
                "single_rank_ms_per_plan_resident": one_rank_s * 1e3,
                "comm_calls_per_plan": calls / 2.0, "comm_bytes_per_plan": 4.0 * words / 2.0}
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}


def host_end_to_end(cfg):
    """What a caller of the API sees: the C++ mirror of PlanNextMapEx from string maps to string maps
    (interning + blance_plan + un-interning), timed by its driver (blance_host_cli --bench)."""
    cli = os.path.join(ROOT, "blance_amd", "lib", "blance_host_cli")
    if cfg != 3 or not os.path.exists(cli):
        return None
    try:
        p = subprocess.run([cli, os.path.join(ROOT, "blance_amd", "lib", "libblance_hip.so"), "bench", "3"],
                           capture_output=True, text=True, timeout=600)
        line = [x for x in p.stdout.splitlines() if x.startswith("{")]
        return json.loads(line[-1]) if line else {"error": (p.stdout + p.stderr)[-300:]}
    except Exception as e:
        return {"error": str(e)[:200]}


if __name__ == "__main__":
    main()
