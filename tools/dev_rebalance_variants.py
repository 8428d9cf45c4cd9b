import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from blance_amd import hip, synth
quiet = hip.Planner()
fp = synth.config_flat(3, 1 << 20, 4096)
res = quiet.plan(fp)
fp2 = synth.config3_rebalance_flat(fp, res)
quiet.close()
for q in ("on", "lean-cpp"):
    pl = hip.Planner(queue=q)
    for i in range(2):
        r = pl.plan(fp2)
    print(q, "device %.1f ms  flat passes %.1f ms  pass kernels %.1f ms" % (r.struct.device_ms, r.struct.flat_pass_ms, r.struct.pass_kernel_ms), flush=True)
    pl.close()
