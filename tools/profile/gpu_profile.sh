#!/bin/bash
# On the GPU box: rocprofv3 kernel trace + PMC passes (SQ, FETCH_SIZE, WRITE_SIZE; each with --kernel-trace
# only) of one bench configuration; summaries go to gpurun_out/prof_<tag>/.   tools/profile/gpu_profile.sh <tag> <bench args...>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p "$out"
work=/tmp/prof_$tag; rm -rf "$work"; mkdir -p "$work"
run() { # name, extra rocprof args
    local name=$1; shift
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace "$@" -d "$work/$name" -o "$name" -- python "$GRAFT_REPO_ROOT/bench.py" $BENCH_ARGS > "$out/$name.log" 2>&1)
    find "$work/$name" -name "*_results.db" | head -1
}
BENCH_ARGS="$* --steps 3 --warmup 1 --no-cpu-baseline --no-sharded --no-extra --no-live-pmc --no-other-configs --no-transfers --no-replicas --no-rccl-one-rank"
db=$(run kt --stats)
python tools/profile_summary.py trace "$db" 4 "$out/kernel_trace_stats.txt" "rocprofv3 --kernel-trace --stats -- python bench.py $BENCH_ARGS (4 PlanNextMap calls)" > /dev/null
grep "^{\"metric\"" "$out/kt.log" | tail -1 > "$out/bench_line_under_rocprof.json"
BENCH_ARGS="$* --steps 1 --warmup 0 --no-cpu-baseline --no-sharded --no-extra --no-live-pmc --no-other-configs --no-transfers --no-replicas --no-rccl-one-rank"
sq=$(run sq --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES)
sq2=$(run sq2 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE)
python tools/profile_summary.py pmc "$out/pmc_sq.txt" "$out/pmc_sq.json" "rocprofv3 --kernel-trace --pmc SQ_* (two passes) -- python bench.py $BENCH_ARGS (1 PlanNextMap call)" $sq $sq2 > /dev/null
f=$(run fetch --pmc FETCH_SIZE)
w=$(run write --pmc WRITE_SIZE)
python tools/profile_summary.py pmc "$out/pmc_hbm.txt" "$out/pmc_hbm.json" "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python bench.py $BENCH_ARGS (1 PlanNextMap call); unit KB as reported" $f $w > /dev/null
ls -la "$out"; head -14 "$out/kernel_trace_stats.txt"; head -8 "$out/pmc_sq.txt"; head -8 "$out/pmc_hbm.txt"
