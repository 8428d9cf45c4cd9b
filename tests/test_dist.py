"""The N > 1 paths without GPUs, world size 2 over gloo:
  * one plan sharded over two ranks (region chains split by region, int32 sum all-reduce of the pass
    outputs and of the load-vector change after every such pass) gives the SAME digest on both
    ranks as the single-rank plan and as the CPU oracle -- the kernels run on the SIMT emulator,
    the collective is the caller-provided one of blance_comm_set;
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
        cases = [synth.config_flat(3, P=300, N=256), synth.config_flat(3, P=700, N=300), synth.config_flat(2, P=300, N=20)]
        c = synth.rebalance_case(P=300, N=64, hierarchy=True)
        fresh = {p: {"name": p, "nodesByState": {}} for p in c["partitions"]}
        opts = dict(partition_weights=c["partitionWeights"], state_stickiness=c["stateStickiness"],
                    node_weights=c["nodeWeights"], node_hierarchy=c["nodeHierarchy"], hierarchy_rules=c["hierarchyRules"])
        cases.append(problem.build_problem({}, fresh, c["oldNodes"], [], c["oldNodes"], c["model"], **opts))
        single = hip.Planner(lib_path=%r, chain_min_parts=8)
        want = [single.plan(fp).digest() for fp in cases]
        assert want == [loader.plan(fp).digest() for fp in cases]
        sharded = hip.Planner(lib_path=%r, chain_min_parts=8)
        calls = []
        inner = dist_util.gloo_allreduce(dist)
        def counted(ptr, count):
            calls.append(count)
            inner(ptr, count)
        sharded.comm_set_callback(rank, world, counted)
        got = [sharded.plan(fp) for fp in cases]
        assert [g.digest() for g in got] == want, "sharded plan differs from the single-rank plan"
        assert got[0].struct.steps_batched > 0
        assert len(calls) >= 3 * 3 and max(calls) == 700 * 3, calls      # flags, loads, outputs of every chain pass
        # the rebalance from the sharded plan (events, nodes outside their region) as well
        plan1, _ = problem.decode_result(cases[3], got[3])
        fp2 = problem.build_problem(plan1, plan1, c["nodesAll"], c["nodesToRemove"], c["nodesToAdd"], c["model"], **opts)
        assert sharded.plan(fp2).digest() == loader.plan(fp2).digest()
        sharded.comm_clear()
        assert sharded.plan(cases[0]).digest() == want[0]
        dist.barrier()
        dist.destroy_process_group()
        print("rank", rank, "ok")
    """ % (ROOT, ROOT, emu, emu), 29519)
    assert out.count("ok") == 2


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
