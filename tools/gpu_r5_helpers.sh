#!/bin/bash
# round 5: k_pass_queue with helper waves -- regime (b) and config 5 timed with digests, then the parity file of the GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
T=${1:-h1}
BLANCE_DEV_PRODUCT=1 BLANCE_QUEUE_STATS=1 timeout 300 python - > gpurun_out/r5/regime_b_$T.log 2>&1 <<'PY'
import json, os, sys, time
sys.path.insert(0, os.getcwd())
from blance_amd import hip, synth
want = json.load(open("tests/golden/config3_general_regime.json"))["named_weighted"]["digest"]
fp = synth.config3_named_weighted_flat(1 << 20, 4096)
for mode in (True, "exact-rebuild"):
    pl = hip.Planner(queue=mode)
    r = pl.plan(fp)
    print("regime (b) queue=%s: sweeps %d device %.1f ms digest ok %s" % (mode, r.iterations, r.struct.device_ms, r.digest() == want), flush=True)
    pl.close()
PY
tail -4 gpurun_out/r5/regime_b_$T.log
BLANCE_QUEUE_STATS=1 timeout 600 python tools/config5_gpu.py > gpurun_out/r5/config5_$T.log 2>&1
tail -6 gpurun_out/r5/config5_$T.log
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q > gpurun_out/r5/parity_$T.log 2>&1
tail -3 gpurun_out/r5/parity_$T.log
