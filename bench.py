#!/usr/bin/env python3
"""Benchmark of the PlanNextMap hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one whole PlanNextMap call (every sweep of planNextMapEx,
plan.go:23-58) on BASELINE.json's config 3: 1,048,576 partitions x 4,096 nodes,
primary + 2 replicas, 3-level rack/zone/DC hierarchy with an exclusion rule.
The problem is uploaded once (inputs resident in HBM when the timed region
starts); value = partition-state assignments per second over the timed steps.

Multi-GPU (N > 1): replicas only -- every rank plans its own instance of the
same shape (see DESIGN.md "Multi-GPU"); value = all ranks' assignments / max
time over ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(parts, nodes, cfg):
    """The CPU oracle (a port of the reference's algorithm, 1 core -- the Go
    planner is single threaded) on a bounded sample: the SAME node count,
    hierarchy and model, 1/4 of the partitions, run to convergence.  Per-step
    cost is O(nodes), so assignments/s carries over to the full size."""
    from blance_amd import synth
    from oracle import loader
    if cfg == 5:
        # config 5 (weighted rebalance, 10 sweeps): 1/32 of the partitions; the plan it starts
        # from is made on the GPU (untimed), the oracle is timed on the rebalance only
        from blance_amd import hip
        sample_parts = max(1024, parts // 32)
        fp1 = synth.config5_initial(sample_parts, nodes)
        pl = hip.Planner()
        fp = synth.config5_rebalance(fp1, pl.plan(fp1), sample_parts, nodes)
        pl.close()
        t0 = time.perf_counter()
        res = loader.plan(fp)
        dt = time.perf_counter() - t0
        return {"value": synth.assignments(fp) / dt, "unit": "assignments/s", "cores": 1, "kind": "port",
                "sample": "oracle/blance_oracle.c, the rebalance PlanNextMap (%d sweeps) on %d partitions x %d nodes "
                          "(1/32 of the partitions, same generator), %.1f s" % (res.iterations, sample_parts, nodes, dt),
                "host_cpus": os.cpu_count()}
    sample_parts = max(1024, parts // 4)
    fp = synth.config_flat(cfg, P=sample_parts, N=nodes)
    t0 = time.perf_counter()
    res = loader.plan(fp)
    dt = time.perf_counter() - t0
    return {"value": synth.assignments(fp) / dt, "unit": "assignments/s", "cores": 1, "kind": "port",
            "sample": "oracle/blance_oracle.c, full PlanNextMap (%d sweeps) on %d partitions x %d nodes "
                      "(1/4 of the partitions, same nodes/hierarchy/model), %.1f s"
                      % (res.iterations, sample_parts, nodes, dt),
            "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, help="BASELINE.json config: 3 (headline), 2, or 5 (weighted rebalance; ~40 s per step)")
    ap.add_argument("--parts", type=int, default=0, help="override partition count (not the headline)")
    ap.add_argument("--nodes", type=int, default=0, help="override node count (not the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from blance_amd import dist_util, hip, synth
    pl = hip.Planner(device_id=local_rank)          # raises without the HIP library / a device
    if args.config == 5:                            # the rebalance starts from a plan over the old nodes (setup, untimed)
        fp1 = synth.config5_initial(args.parts or 1 << 20, args.nodes or 4096)
        fp = synth.config5_rebalance(fp1, pl.plan(fp1), args.parts or 1 << 20, args.nodes or 4096)
        del fp1
    else:
        fp = synth.config_flat(args.config, P=args.parts or None, N=args.nodes or None)
    P, N = fp.n_parts, fp.n_nodes
    t0 = time.perf_counter()
    pl.upload(fp)
    upload_s = time.perf_counter() - t0

    for _ in range(args.warmup):
        pl.plan_resident()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    pass_ms = flat_ms = device_ms = 0.0
    pass_launches = flat_passes = 0
    iterations = 0
    for _ in range(args.steps):
        r = pl.plan_resident()                      # returns after the device finished the call
        pass_ms += r.pass_kernel_ms
        pass_launches += r.pass_kernel_launches
        flat_ms += r.flat_pass_ms
        flat_passes += r.flat_passes
        device_ms += r.device_ms
        iterations = r.iterations
        batched, sequential = r.steps_batched, r.steps_sequential
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        dt = dist_util.max_over_ranks(dt)

    assignments = synth.assignments(fp)
    value = assignments * args.steps * world / dt

    out = None
    if rank == 0:
        t1 = time.perf_counter()
        res = pl.download()
        download_s = time.perf_counter() - t1
        digest = res.digest()
        # Dominant kernel: the state-pass kernel (k_pass_chain on config 3), one launch per
        # hierarchy-rule state pass, timed by hipEvents on the planner's stream.  Algorithmic
        # bytes per launch: SURVEY.md 8(d), P * (N * (16 + 4 k [rules]) + 40) for that state.
        per_state = synth.algorithmic_bytes_per_state(fp)
        kernel_states = synth.pass_kernel_states(fp)
        alg_per_launch = (sum(per_state[m] for m in kernel_states) / max(len(kernel_states), 1))
        kernel_label = "k_pass_chain / k_pass_chain_blank (state-pass kernel, one launch per replica pass)"
        dom_ms, dom_launches = pass_ms, pass_launches
        if args.config == 5:
            kernel_label = "k_pass_seq (workgroup pass, one launch per replica pass)"
        if not pass_launches and flat_passes:        # every pass went through the flat driver (config 2)
            kernel_label = "flat driver passes (k_flat_*, k_fresh_*, k_sort_*, flat chain; several launches per pass)"
            flat_states = [m for m in range(len(per_state)) if per_state[m] and m not in kernel_states]
            alg_per_launch = sum(per_state[m] for m in flat_states) / max(len(flat_states), 1)
            dom_ms, dom_launches = flat_ms, flat_passes
        avg_launch_ms = dom_ms / max(dom_launches, 1)
        achieved = alg_per_launch / (avg_launch_ms * 1e-3) / 1e9 if dom_launches else 0.0
        traffic = None
        prof = os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")
        if os.path.exists(prof) and args.config == 3 and not args.parts and not args.nodes:
            with open(prof) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
        whole = synth.algorithmic_bytes_per_sweep(fp) * iterations * args.steps / (device_ms * 1e-3) / 1e9
        # what actually bounds the kernel: the dependent chain of one region
        critical = None
        if args.config == 3 and dom_launches:
            regions = -(-N // 128)                                # zones of 8 racks x 16 nodes
            chain_steps = -(-P // regions)
            critical = {"regions": regions, "dependent_steps_per_launch": chain_steps,
                        "avg_ns_per_dependent_step": avg_launch_ms * 1e6 / chain_steps}
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
                       "steps_bulk": int(batched), "steps_one_by_one": int(sequential),
                       "headline": bool(args.config == 3 and not args.parts and not args.nodes)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": kernel_label, "launches": dom_launches,
                         "avg_launch_ms": avg_launch_ms, "algorithmic_bytes_per_launch": alg_per_launch,
                         "whole_call_algorithmic_GBps": whole, "critical_path": critical,
                         "note": "algorithmic bytes are what the reference's dense per-step scan reads "
                                 "(SURVEY.md 8d); the kernel keeps tables in registers/LDS and resolves "
                                 "verified stays in bulk, so measured HBM traffic is far below them"},
            "device_ms_per_step": device_ms / args.steps,
            "pass_kernel_ms_per_step": pass_ms / args.steps, "flat_pass_ms_per_step": flat_ms / args.steps,
            "transfers": {"upload_s": upload_s, "download_s": download_s,
                          "value_incl_transfers": assignments / (dt / args.steps + upload_s + download_s)},
            "result_sha256": digest,
        }
        ref = os.path.join(ROOT, "tests", "golden", "config_digests.json")
        if os.path.exists(ref) and not args.parts and not args.nodes:
            with open(ref) as f:
                want = json.load(f).get("config%d" % args.config)
            if want:
                out["matches_oracle_digest"] = (want["rebalance"] if args.config == 5 else want)["digest"] == digest
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.parts or (1 << 20) if args.config == 5 else P, N, args.config)
        print(json.dumps(out), flush=True)
    pl.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
