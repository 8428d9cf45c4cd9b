#!/bin/bash
# round 5: the GPU suite, then a short bench line (config 3) and the quick timings of regime (b) / config 5
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
T=${1:-c1}
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r5/gputests_$T.log 2>&1
tail -3 gpurun_out/r5/gputests_$T.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-other-configs --no-extra --no-cpu-baseline --no-sharded --no-live-pmc > gpurun_out/r5/bench_short_$T.log 2>&1
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r5/bench_short_$T.log") if l.startswith("{")][-1])
print("ms_per_step", d["ms_per_step"], "device", d["device_ms_per_step"], "digest", d.get("matches_oracle_digest"), json.dumps(d["transfers"])[:400])
PY
bash tools/gpu_r5_quick.sh $T | head -8
