#!/bin/bash
# round 5: phase clocks of k_pass_queue (helper waves) on regime (b), then the product library timed on (b) and config 5
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
T=${1:-p1}
timeout 300 python tools/dev_general_regime.py > gpurun_out/r5/phase_b_$T.log 2>&1
grep -c "queue\]" gpurun_out/r5/phase_b_$T.log
BLANCE_DEV_PRODUCT=1 BLANCE_QUEUE_STATS=1 timeout 300 python tools/dev_general_regime.py > gpurun_out/r5/regime_b_$T.log 2>&1
tail -2 gpurun_out/r5/regime_b_$T.log
BLANCE_QUEUE_STATS=1 timeout 600 python tools/config5_gpu.py > gpurun_out/r5/config5_$T.log 2>&1
tail -5 gpurun_out/r5/config5_$T.log
