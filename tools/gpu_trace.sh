#!/bin/bash
# On the GPU box: only the rocprofv3 kernel trace (+ stats summary) of one bench configuration.   gpu_trace.sh <tag> <bench args...>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p "$out"
work=/tmp/prof_$tag; rm -rf "$work"; mkdir -p "$work"
BENCH_ARGS="$* --steps 3 --warmup 1 --no-cpu-baseline --no-sharded"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$work/kt" -o kt -- python "$GRAFT_REPO_ROOT/bench.py" $BENCH_ARGS > "$out/kt.log" 2>&1)
db=$(find "$work/kt" -name "*_results.db" | head -1)
python tools/profile_summary.py trace "$db" 4 "$out/kernel_trace_stats.txt" "rocprofv3 --kernel-trace --stats -- python bench.py $BENCH_ARGS (4 PlanNextMap calls)" > /dev/null
grep "^{\"metric\"" "$out/kt.log" | tail -1 > "$out/bench_line_under_rocprof.json"
head -40 "$out/kernel_trace_stats.txt"
