"""The N > 1 path of bench.py without GPUs: two gloo processes agree on the
slowest rank's time (the only cross-rank step -- DESIGN.md "Multi-GPU": replicas only)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_max_over_ranks_world_size_2(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        import torch.distributed as dist
        from blance_amd import dist_util
        dist.init_process_group("gloo")
        rank = dist.get_rank()
        got = dist_util.max_over_ranks(1.0 + rank)          # rank 1 is the slow one
        assert abs(got - 2.0) < 1e-12, got
        assert dist_util.replica_seed(0) != dist_util.replica_seed(1)
        dist.barrier()
        dist.destroy_process_group()
        print("rank", rank, "ok")
    """ % ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2
