#!/bin/bash
# On the GPU box: round 6's evidence -- the default bench line as the driver runs it, the lines of configs 2 and 5 by themselves,
# rocprofv3 summaries (kernel trace, SQ and HBM counters) of configs 3 / 2 / 5 and of the two general-regime workloads, the
# register / spill table of the pass kernels, the full GPU suite and a stress run.  Summaries land in gpurun_out/r6e/; the ones
# that are judged are copied to profiles/r6_* by hand.
#   gpurun --timeout 3300 -- 'bash tools/profile/round6.sh [quick]'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r6e
mkdir -p "$out"
(time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5) > "$out/bench_default.json" 2> "$out/bench_default.err"; tail -c 300 "$out/bench_default.err"
timeout 300 python bench.py --config 2 --steps 20 --warmup 5 --no-replicas > "$out/bench_config2.json" 2> "$out/bench_config2.err"
BLANCE_QUEUE_STATS=1 timeout 600 python bench.py --config 5 --steps 2 --warmup 1 --no-replicas > "$out/bench_config5.json" 2> "$out/bench_config5.err"
python - <<'PY'
import json, os
o = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r6e")
for n in ("bench_default.json", "bench_config2.json", "bench_config5.json"):
    try:
        d = json.loads([l for l in open(os.path.join(o, n)) if l.startswith("{")][-1])
        print(n, "%.3f ms per call, %.1f M assignments/s, digest ok %s, traffic %s, blocks %s" % (
            d["ms_per_step"], d["value"] / 1e6, d.get("matches_oracle_digest"), d["roofline"].get("traffic"), d.get("block_seconds")))
        for w in d.get("general_regime", []) + d.get("other_configs", []):
            print("   ", w.get("workload", "")[:60], w.get("ms_per_step"), w.get("sweeps_per_call"), w.get("matches_oracle_digest"), w.get("error"))
    except Exception as e:
        print(n, "no line:", e)
PY
[ "$1" = quick ] && exit 0
for cfg in 3 2 5; do
  bash tools/profile/gpu_profile.sh config$cfg --config $cfg > "$out/profile_config$cfg.log" 2>&1
  tail -3 "$out/profile_config$cfg.log"
done
bash tools/profile/gpu_profile_general.sh > "$out/profile_general.log" 2>&1; tail -4 "$out/profile_general.log"
if [ -f devbuild/libblance_prof.so ]; then     # (tools/profile/build_prof.sh tu_chain: k_pass_chain's phase clocks at config 3)
  timeout 300 python tools/profile/chain_phases.py > "$out/phase_chain_config3.log" 2>&1
  grep -c "phase" "$out/phase_chain_config3.log"
fi
bash tools/profile/ab_speculate.sh > "$out/ab_speculate.txt" 2>&1; tail -12 "$out/ab_speculate.txt"
timeout 1500 python -m pytest tests -q -m gpu > "$out/test_gpu_full.log" 2>&1; grep -E "passed|failed" "$out/test_gpu_full.log" | tail -2
bash tools/profile/gpu_stress.sh 150 81000 100 83000 | tail -8
