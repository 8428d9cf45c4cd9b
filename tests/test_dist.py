"""The N > 1 paths without GPUs:
  * world size 2 over gloo: one plan sharded over two ranks (region chains split by region; per chain
    pass ONE int32 sum all-reduce of [flags | load-vector change] and ONE all-gather of the output
    slices) gives the SAME digest on both ranks as the single-rank plan and as the CPU oracle -- the
    kernels run on the SIMT emulator, the collectives are the embedder's of blance_comm_set;
  * the same with 2, 5 and 8 ranks as threads of one process (blance_amd.dist_util.LocalGroup -- the
    arrangement the GPU suite uses to run the sharded path on one MI355X), and without an all-gather
    hook (outputs summed instead);
  * the ranks agree on the slowest rank's time (bench.py's timing rule)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_world2(tmp_path, body, port):
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent(body))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    return out.stdout


def test_sharded_plan_world_size_2(tmp_path):
    from test_simt_emulated import build_emu
    emu = build_emu()
    out = _run_world2(tmp_path, """
        import os, sys
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
        import torch.distributed as dist
        from blance_amd import dist_util, hip, problem, synth
        from oracle import loader
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        from helpers import sharded_cases
        cases, c, opts = sharded_cases()
        single = hip.Planner(lib_path=%r, chain_min_parts=8)
        want = [single.plan(fp).digest() for fp in cases]
        assert want == [loader.plan(fp).digest() for fp in cases]
        sharded = hip.Planner(lib_path=%r, chain_min_parts=8)
        calls = []
        ar, ag = dist_util.gloo_collectives(sharded, dist)
        def counted(kind, inner):
            def f(ptr, count):
                calls.append((kind, count))
                inner(ptr, count)
            return f
        sharded.comm_set_callback(rank, world, counted("reduce", ar), counted("gather", ag))
        got = []
        for fp in cases:
            before = len(calls)
            got.append(sharded.plan(fp))
            # every chain pass: collective A (flags + loads) and B (output slices), nothing else;
            # at most one chain pass per sweep in these models (one state with a hierarchy rule)
            mine = calls[before:]
            assert len(mine) <= 2 * got[-1].iterations, (mine, got[-1].iterations)
            assert [k for k, _ in mine] == ["reduce", "gather"] * (len(mine) // 2), mine
        assert [g.digest() for g in got] == want, "sharded plan differs from the single-rank plan"
        assert got[0].struct.steps_batched > 0
        assert len(calls) >= 2 * (3 + 3 + 3 + 2), calls                 # every hierarchical case did shard
        assert sharded.comm_stats()[0] == len(calls)
        assert sharded.comm_time_ms() == 0.0                            # (an embedder's collectives run on the host: no RCCL time)
        # the rebalance from the sharded plan (events, nodes outside their region) as well
        plan1, _ = problem.decode_result(cases[-1], got[-1])
        fp2 = problem.build_problem(plan1, plan1, c["nodesAll"], c["nodesToRemove"], c["nodesToAdd"], c["model"], **opts)
        before = len(calls)
        assert sharded.plan(fp2).digest() == loader.plan(fp2).digest()
        assert len(calls) > before
        sharded.comm_clear()
        assert sharded.plan(cases[0]).digest() == want[0]
        dist.barrier()
        dist.destroy_process_group()
        print("rank", rank, "ok")
    """ % (ROOT, ROOT, emu, emu), 29519)
    assert out.count("ok") == 2


def test_sharded_plan_threads_one_process():
    """G ranks = G contexts driven by G threads (LocalGroup): digests equal the oracle's for every G, with
    the all-gather of output slices and with the summed outputs (no all-gather hook)."""
    from blance_amd import dist_util, hip, problem
    from oracle import loader
    from test_simt_emulated import build_emu
    emu = build_emu()
    from helpers import sharded_cases
    cases, c, opts = sharded_cases()
    want = [loader.plan(fp).digest() for fp in cases]
    for G, gather in ((2, True), (5, True), (8, True), (2, False)):
        grp, planners = dist_util.local_sharded_planners(G, lambda: hip.Planner(lib_path=emu, chain_min_parts=8))
        if not gather:
            for r, pl in enumerate(planners):
                pl.comm_set_callback(r, G, grp.rank_collectives(r)[0], None)

        def work(rank, pl):
            got = [pl.plan(fp) for fp in cases]
            # the rebalance from the sharded plan (events, nodes outside their region) as well
            plan1, _ = problem.decode_result(cases[-1], got[-1])
            fp2 = problem.build_problem(plan1, plan1, c["nodesAll"], c["nodesToRemove"], c["nodesToAdd"], c["model"], **opts)
            mid = pl.comm_stats()[0]
            return [g.digest() for g in got], pl.plan(fp2).digest(), fp2, mid, pl.comm_stats()[0], [g.iterations for g in got]
        res = grp.run(planners, work)
        for digests, d2, fp2, mid, end, iters in res:
            assert digests == want, G
            assert d2 == loader.plan(fp2).digest(), G
            # the 8-zone case shards for every G, two collectives per chain pass (= per sweep); so does its rebalance
            assert mid >= 2 * iters[3] and (end > mid or G > 5), (G, mid, end, iters)     # the weighted tree has 5 zones
            assert mid <= 2 * sum(iters)
        for pl in planners:
            pl.close()


def test_one_rank_takes_the_sharded_branch():
    """Planner(shard_one_rank=True): a communicator of ONE rank runs the sharded branch of every chain pass -- collective A
    (all-reduce of [flags | load change]) and B (all-gather of the one slice) are both made, in that order, and the plan is
    the single-rank plan.  On the device the same knob is how ncclAllReduce / ncclAllGather execute on a one-GPU box
    (tests/test_hip_parity.py::test_rccl_one_rank_runs_both_collectives)."""
    from blance_amd import dist_util, hip
    from oracle import loader
    from test_simt_emulated import build_emu
    from helpers import sharded_cases
    emu = build_emu()
    cases, _, _ = sharded_cases()
    pl = hip.Planner(lib_path=emu, chain_min_parts=8, shard_one_rank=True)
    grp = dist_util.LocalGroup(1, True)
    ar, ag = grp.rank_collectives(0)
    kinds = []
    pl.comm_set_callback(0, 1, lambda p, n: (kinds.append("reduce"), ar(p, n)), lambda p, n: (kinds.append("gather"), ag(p, n)))
    for i, fp in enumerate(cases):
        before = len(kinds)
        got = pl.plan(fp)
        assert got.digest() == loader.plan(fp).digest(), i
        mine = kinds[before:]
        assert mine == ["reduce", "gather"] * (len(mine) // 2), mine
        assert len(mine) <= 2 * got.iterations
    assert len(kinds) >= 2 * (3 + 3 + 3 + 2) and pl.comm_stats()[0] == len(kinds)
    pl.close()
    # without the knob a communicator of one rank makes no collective at all
    pl = hip.Planner(lib_path=emu, chain_min_parts=8)
    pl.comm_set_callback(0, 1, lambda p, n: kinds.append("x"), lambda p, n: kinds.append("x"))
    n = len(kinds)
    assert pl.plan(cases[0]).digest() == loader.plan(cases[0]).digest() and len(kinds) == n and pl.comm_stats()[0] == 0
    pl.close()


def test_max_over_ranks_world_size_2(tmp_path):
    out = _run_world2(tmp_path, """
        import os, sys
        sys.path.insert(0, %r)
        import torch.distributed as dist
        from blance_amd import dist_util
        dist.init_process_group("gloo")
        rank = dist.get_rank()
        got = dist_util.max_over_ranks(1.0 + rank)          # rank 1 is the slow one
        assert abs(got - 2.0) < 1e-12, got
        dist.barrier()
        dist.destroy_process_group()
        print("rank", rank, "ok")
    """ % ROOT, 29517)
    assert out.count("ok") == 2


def test_bench_multi_rank_flow_rehearsal():
    """bench.py's N > 1 path -- the lines the driver's 8-GPU run executes -- rehearsed with two gloo ranks and the
    emulated kernels (BLANCE_BENCH_REHEARSAL): replicas timed between barriers with the maximum over ranks, then the
    sharded plan with its two collectives per chain pass, one JSON line from rank 0."""
    import json
    from test_simt_emulated import build_emu
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", BLANCE_BENCH_REHEARSAL=build_emu())
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29531", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "2", "--warmup", "1", "--parts", "600", "--nodes", "256"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                                   # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d["rehearsal"] is True and d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1
    assert d["scaling"] == "weak" and "replicas x2" in d["config"]["parallelism"]
    # the whole job: both replicas' assignments over the slower rank's time
    per_call = d["value"] * d["ms_per_step"] * 1e-3 / 2
    assert abs(per_call - round(per_call)) < 1e-6 * per_call
    s = d["sharded"]
    assert "error" not in s, s
    assert s["rccl_world_size"] == 2 and s["same_digest_on_every_rank"] and s["same_digest_as_single_rank_plan"]
    assert s["scaling"] == "strong" and s["comm_calls_per_plan"] >= 2 and s["comm_bytes_per_plan"] > 0
    assert "cpu_baseline" not in d                                       # rank 0 at N = 1 only


import pytest  # noqa: E402


@pytest.mark.gpu
def test_rccl_two_ranks_when_two_devices(tmp_path):
    """RCCL between two GPUs (ncclAllReduce / ncclAllGather inside the library, bound at run time): one PlanNextMap of
    config 3's generator sharded over two ranks, one process per device -- same digest on both ranks as the single-rank
    plan and as the CPU oracle, two collectives per chain pass.  Skips unless this machine shows at least two devices
    (the development boxes show one): the first multi-GPU box runs it inside pytest, not only inside bench.py's watchdog."""
    import torch
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (found %d)" % (torch.cuda.device_count() if torch.cuda.is_available() else 0))
    script = tmp_path / "rccl2.py"
    script.write_text(textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        import torch, torch.distributed as dist
        from blance_amd import dist_util, hip, synth
        from oracle import loader
        local = int(os.environ["LOCAL_RANK"])
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        rank, world = dist.get_rank(), dist.get_world_size()
        fp = synth.config_flat(3, P=65536, N=4096)
        want = loader.plan(fp).digest()
        single = hip.Planner(device_id=local)
        assert single.plan(fp).digest() == want
        single.close()
        pl = hip.Planner(device_id=local)
        dist_util.shard_plan_rccl(pl, dist)
        calls0, words0 = pl.comm_stats()
        got = pl.plan(fp)
        calls1, words1 = pl.comm_stats()
        assert got.digest() == want, "sharded plan over RCCL differs from the single-rank plan"
        assert 0 < calls1 - calls0 <= 2 * got.iterations and words1 > words0
        box = [None] * world
        dist.all_gather_object(box, got.digest())
        assert len(set(box)) == 1
        pl.close()
        dist.barrier()
        dist.destroy_process_group()
        print("rank", rank, "ok")
    """ % ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("ok") == 2
