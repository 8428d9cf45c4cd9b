#!/bin/bash
# On the GPU box (round 4's first call): the opt-in periodic pass of DESIGN.md 4.1c against the default planner --
#   gpurun --timeout 1500 -- 'bash tools/gpu_periodic.sh'
# 1. its GPU test (oracle digests; config 3 at full size), 2. bench.py with and without --periodic (same steps),
# 3. a kernel trace of the periodic run.  Everything lands in gpurun_out/periodic/.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/periodic
mkdir -p "$out"
timeout 900 python -m pytest tests/test_periodic_gpu.py -q -m gpu > "$out/test.log" 2>&1; tail -5 "$out/test.log"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sharded > "$out/bench_default.json" 2> "$out/bench_default.err"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sharded --periodic > "$out/bench_periodic.json" 2> "$out/bench_periodic.err"
python - <<'PY'
import json, os
o = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "periodic")
for n in ("bench_default.json", "bench_periodic.json"):
    try:
        d = json.loads([l for l in open(os.path.join(o, n)) if l.startswith("{")][-1])
        print(n, "%.3f ms per call, %.0f M assignments/s, digest %s" % (d["ms_per_step"], d["value"] / 1e6, str(d.get("result_sha256"))[:16]))
    except Exception as e:
        print(n, "no line:", e)
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_periodic -o kt -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-sharded --periodic > "$out/kt.log" 2>&1)
db=$(find /tmp/prof_periodic -name "*_results.db" | head -1)
[ -n "$db" ] && python tools/profile_summary.py trace "$db" 4 "$out/kernel_trace_stats_periodic.txt" "rocprofv3 --kernel-trace --stats -- python bench.py --periodic --steps 3 --warmup 1 (4 PlanNextMap calls)" > /dev/null
head -16 "$out/kernel_trace_stats_periodic.txt"
