#!/bin/bash
# On the GPU box: rocprofv3 kernel trace of bench.py's two general-regime workloads (one PlanNextMap call each, product library):
# (a) config 3's plan rebalanced after every tenth node left, (b) config 3 with scrambled names and Zipf weights.
#   gpurun --timeout 900 -- 'bash tools/profile/gpu_profile_general.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_general
mkdir -p "$out"
for w in a b; do
  work=/tmp/prof_general_$w; rm -rf "$work"; mkdir -p "$work"
  if [ $w = a ]; then cmd="tools/profile/rebalance_regime.py"; what="config 3's plan (1 call, 3 ms) and its rebalance after every tenth node left (1 call)";
  else cmd="tools/profile/general_regime.py"; what="config 3 with scrambled non-numeric partition names and Zipf partition weights (1 call, 10 sweeps)"; fi
  (cd /tmp && BLANCE_DEV_PRODUCT=1 timeout 600 rocprofv3 --kernel-trace --stats -d "$work" -o kt -- python "$GRAFT_REPO_ROOT/$cmd" > "$out/kt_$w.log" 2>&1)
  db=$(find "$work" -name "*_results.db" | head -1)
  python tools/profile_summary.py trace "$db" 1 "$out/kernel_trace_stats_general_$w.txt" "rocprofv3 --kernel-trace --stats -- python $cmd: $what" > /dev/null
  tail -2 "$out/kt_$w.log"; head -16 "$out/kernel_trace_stats_general_$w.txt"
  # HBM traffic of the same call: FETCH_SIZE and WRITE_SIZE in separate passes, nothing but --kernel-trace beside --pmc
  dbs=""
  for counter in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && BLANCE_DEV_PRODUCT=1 timeout 600 rocprofv3 --kernel-trace --pmc $counter -d "$work/$counter" -o p -- python "$GRAFT_REPO_ROOT/$cmd" > "$out/pmc_${counter}_$w.log" 2>&1)
    dbs="$dbs $(find "$work/$counter" -name "*_results.db" | head -1)"
  done
  python tools/profile_summary.py pmc "$out/pmc_hbm_general_$w.txt" "$out/pmc_hbm_general_$w.json" "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python $cmd: $what; unit KB as reported" $dbs > /dev/null
  head -8 "$out/pmc_hbm_general_$w.txt"
done
