#!/bin/bash
# round 5: the product library timed on regime (b) and config 5 (digests checked)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
T=${1:-q1}
BLANCE_DEV_PRODUCT=1 BLANCE_QUEUE_STATS=1 timeout 300 python - > gpurun_out/r5/regime_b_$T.log 2>&1 <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
from blance_amd import hip, synth
want = json.load(open("tests/golden/config3_general_regime.json"))["named_weighted"]["digest"]
fp = synth.config3_named_weighted_flat(1 << 20, 4096)
pl = hip.Planner()
r = pl.plan(fp)
print("regime (b): sweeps %d device %.1f ms digest ok %s" % (r.iterations, r.struct.device_ms, r.digest() == want), flush=True)
PY
tail -2 gpurun_out/r5/regime_b_$T.log
timeout 300 python tools/dev_rebalance_regime.py 2> /dev/null | tail -3
BLANCE_QUEUE_STATS=1 timeout 600 python tools/config5_gpu.py > gpurun_out/r5/config5_$T.log 2>&1
grep "k_pass_queue\|device\|matches" gpurun_out/r5/config5_$T.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-other-configs --no-extra --no-cpu-baseline --no-sharded --no-live-pmc > gpurun_out/r5/bench_short_$T.log 2>&1
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r5/bench_short_$T.log") if l.startswith("{")][-1])
print("ms_per_step", d["ms_per_step"], json.dumps(d["transfers"])[:1500])
PY
