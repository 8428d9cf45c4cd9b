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
    hierarchy and model, 1/8 of the partitions, run to convergence.  Per-step
    cost is O(nodes), so assignments/s carries over to the full size."""
    from blance_amd import synth
    from oracle import loader
    sample_parts = max(1024, parts // 8)
    fp = synth.config_flat(cfg, P=sample_parts, N=nodes)
    t0 = time.perf_counter()
    res = loader.plan(fp)
    dt = time.perf_counter() - t0
    return {"value": synth.assignments(fp) / dt, "unit": "assignments/s", "cores": 1, "kind": "port",
            "sample": "oracle/blance_oracle.c, full PlanNextMap (%d sweeps) on %d partitions x %d nodes "
                      "(1/8 of the partitions, same nodes/hierarchy/model), %.1f s"
                      % (res.iterations, sample_parts, nodes, dt),
            "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, help="BASELINE.json config (2 or 3)")
    ap.add_argument("--parts", type=int, default=0, help="override partition count (not the headline)")
    ap.add_argument("--nodes", type=int, default=0, help="override node count (not the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", action="store_true", help="also compare the result with the oracle (slow)")
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

    from blance_amd import hip, synth
    fp = synth.config_flat(args.config, P=args.parts or None, N=args.nodes or None)
    P, N = fp.n_parts, fp.n_nodes
    pl = hip.Planner(device_id=local_rank)          # raises without the HIP library / a device
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
    pass_ms = 0.0
    pass_launches = 0
    device_ms = 0.0
    iterations = 0
    for _ in range(args.steps):
        r = pl.plan_resident()                      # returns after the device finished the call
        pass_ms += r.pass_kernel_ms
        pass_launches += r.pass_kernel_launches
        device_ms += r.device_ms
        iterations = r.iterations
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    assignments = synth.assignments(fp)
    value = assignments * args.steps * world / dt

    out = None
    if rank == 0:
        t1 = time.perf_counter()
        res = pl.download()
        download_s = time.perf_counter() - t1
        digest = res.digest()
        # dominant kernel: k_pass_seq, one launch per state pass; algorithmic bytes per
        # launch from SURVEY.md section 8(d) (all sweeps of all timed steps / launches)
        alg_bytes = synth.algorithmic_bytes_per_sweep(fp) * iterations * args.steps
        achieved = alg_bytes / (pass_ms * 1e-3) / 1e9 if pass_ms > 0 else 0.0
        out = {
            "metric": "partition-state assignments/sec at 1M partitions x 4,096 nodes",
            "value": value, "unit": "assignments/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64 scores / int32 tables",
            "data": "synthetic",
            "config": {"workload": "BASELINE.json config %d: %d partitions x %d nodes, %s"
                                   % (args.config, P, N,
                                      "primary+2 replicas, 3-level rack/zone/DC hierarchy, rule replica{include 2, exclude 1}"
                                      if args.config == 3 else "primary+1 replica, flat"),
                       "partitions": P, "nodes": N, "assignments_per_call": assignments,
                       "sweeps_per_call": iterations, "parallelism": "replicas x%d" % world,
                       "headline": bool(args.config == 3 and not args.parts and not args.nodes)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "k_pass_seq", "launches": pass_launches,
                         "avg_launch_ms": pass_ms / max(pass_launches, 1),
                         "algorithmic_bytes_per_launch": alg_bytes / max(pass_launches, 1)},
            "device_ms_per_step": device_ms / args.steps,
            "transfers": {"upload_s": upload_s, "download_s": download_s,
                          "value_incl_transfers": assignments / (dt / args.steps + upload_s + download_s)},
            "result_sha256": digest,
        }
        if args.verify:
            from oracle import loader
            out["matches_oracle"] = loader.plan(fp).digest() == digest
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(P, N, args.config)
        print(json.dumps(out), flush=True)
    pl.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
