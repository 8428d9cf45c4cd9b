#!/bin/bash
# On the GPU box: only the two SQ counter passes of one bench configuration (gpu_profile.sh does everything).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p "$out"
work=/tmp/prof_$tag; rm -rf "$work"; mkdir -p "$work"
run() {
    local name=$1; shift
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace "$@" -d "$work/$name" -o "$name" -- python "$GRAFT_REPO_ROOT/bench.py" $BENCH_ARGS > "$out/$name.log" 2>&1)
    find "$work/$name" -name "*_results.db" | head -1
}
BENCH_ARGS="$* --steps 1 --warmup 0 --no-cpu-baseline --no-sharded --no-extra --no-live-pmc"
sq=$(run sq --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES)
sq2=$(run sq2 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE)
python tools/profile_summary.py pmc "$out/pmc_sq.txt" "$out/pmc_sq.json" "rocprofv3 --kernel-trace --pmc SQ_* (two passes) -- python bench.py $BENCH_ARGS" $sq $sq2 > /dev/null
head -6 "$out/pmc_sq.txt"
