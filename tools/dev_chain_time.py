"""Developer aid: config 3 at full size, resident; prints the per-pass times (BLANCE_TRACE) with the
all-blank chain pass on k_pass_chain_planes and on k_pass_chain_blank, and checks the digest."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blance_amd import hip, synth

want = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "config_digests.json")))["config3"]["digest"]
fp = synth.config_flat(3)
for planes in (True, False):
    pl = hip.Planner(device_id=0, planes=planes)
    pl.upload(fp)
    for i in range(3):
        r = pl.plan_resident()
    os.environ["BLANCE_TRACE"] = "1"
    pl2 = hip.Planner(device_id=0, planes=planes)
    pl2.upload(fp)
    pl2.plan_resident()
    r2 = pl2.plan_resident()
    del os.environ["BLANCE_TRACE"]
    t0 = time.time()
    for i in range(10):
        r = pl.plan_resident()
    dt = (time.time() - t0) / 10
    d = pl.download().digest()
    print("planes=%s device_ms=%.3f wall_ms=%.3f pass_kernel_ms=%.3f digest_ok=%s" % (planes, r.device_ms, dt * 1e3, r.pass_kernel_ms, d == want), flush=True)
    pl.close(); pl2.close()
