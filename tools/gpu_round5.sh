#!/bin/bash
# On the GPU box: round 5's evidence in one call -- the default bench line (every BASELINE configuration, live PMC traffic,
# general regime, CPU oracle on all of config 3, transfers, host end to end), the lines of configs 2 and 5 by themselves, the
# rocprofv3 summaries of all three (kernel trace, SQ and HBM counters), kernel traces + HBM counters of the two general-regime
# workloads, and the phase clocks of k_pass_queue on regime (b) (devbuild/libblance_prof.so: tools/dev_build_prof.sh tu_queue).
#   gpurun --timeout 2700 -- 'bash tools/gpu_round5.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r5e
mkdir -p "$out"
timeout 900 python bench.py --steps 20 --warmup 5 > "$out/bench_default.json" 2> "$out/bench_default.err"; tail -c 300 "$out/bench_default.err"
timeout 300 python bench.py --config 2 --steps 20 --warmup 5 > "$out/bench_config2.json" 2> "$out/bench_config2.err"
BLANCE_QUEUE_STATS=1 timeout 600 python bench.py --config 5 --steps 2 --warmup 0 > "$out/bench_config5.json" 2> "$out/bench_config5.err"
python - <<'PY'
import json, os
o = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r5e")
for n in ("bench_default.json", "bench_config2.json", "bench_config5.json"):
    try:
        d = json.loads([l for l in open(os.path.join(o, n)) if l.startswith("{")][-1])
        print(n, "%.3f ms per call, %.1f M assignments/s, digest ok %s, traffic %s" % (d["ms_per_step"], d["value"] / 1e6, d.get("matches_oracle_digest"), d["roofline"].get("traffic")))
        for w in d.get("general_regime", []) + d.get("other_configs", []):
            print("   ", w.get("workload", "")[:60], w.get("ms_per_step"), w.get("sweeps_per_call"), w.get("matches_oracle_digest"), w.get("error"))
    except Exception as e:
        print(n, "no line:", e)
PY
for cfg in 3 2 5; do
  bash tools/gpu_profile.sh config$cfg --config $cfg > "$out/profile_config$cfg.log" 2>&1
  tail -3 "$out/profile_config$cfg.log"
done
bash tools/gpu_profile_general.sh > "$out/profile_general.log" 2>&1; tail -4 "$out/profile_general.log"
if [ -f devbuild/libblance_prof.so ]; then
  timeout 300 python tools/dev_general_regime.py > "$out/phase_general_b.log" 2>&1
  grep -c "queue\]" "$out/phase_general_b.log"
fi
timeout 1500 python -m pytest tests -q -m gpu > "$out/test_gpu_full.log" 2>&1; grep -E "passed|failed" "$out/test_gpu_full.log" | tail -2
bash tools/gpu_stress5.sh 150 71000 100 73000 | tail -8
