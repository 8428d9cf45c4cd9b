#!/bin/bash
# On the GPU box: round 4's evidence in one call -- the default bench line (live PMC traffic, general regime, CPU baseline,
# host end to end), configs 2 and 5, and the rocprofv3 summaries of all three (kernel trace, SQ and HBM counters).
#   gpurun --timeout 2400 -- 'bash tools/gpu_round4.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r4
mkdir -p "$out"
timeout 900 python bench.py --steps 20 --warmup 5 > "$out/bench_default.json" 2> "$out/bench_default.err"; tail -c 600 "$out/bench_default.err"
timeout 300 python bench.py --config 2 --steps 20 --warmup 5 > "$out/bench_config2.json" 2> "$out/bench_config2.err"
BLANCE_QUEUE_STATS=1 timeout 600 python bench.py --config 5 --steps 2 --warmup 0 > "$out/bench_config5.json" 2> "$out/bench_config5.err"
python - <<'PY'
import json, os
o = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r4")
for n in ("bench_default.json", "bench_config2.json", "bench_config5.json"):
    try:
        d = json.loads([l for l in open(os.path.join(o, n)) if l.startswith("{")][-1])
        print(n, "%.3f ms per call, %.1f M assignments/s, digest ok %s, traffic %s" % (d["ms_per_step"], d["value"] / 1e6, d.get("matches_oracle_digest"), d["roofline"].get("traffic")))
        for w in d.get("general_regime", []):
            print("   ", w.get("workload", "")[:60], w.get("ms_per_step"), w.get("sweeps_per_call"), w.get("matches_oracle_digest"), w.get("error"))
    except Exception as e:
        print(n, "no line:", e)
PY
for cfg in 3 2 5; do
  bash tools/gpu_profile.sh config$cfg --config $cfg > "$out/profile_config$cfg.log" 2>&1
  tail -3 "$out/profile_config$cfg.log"
done
timeout 1500 python -m pytest tests -q -m gpu > "$out/test_gpu_full.log" 2>&1; tail -3 "$out/test_gpu_full.log"
