#!/bin/bash
# round 5, first GPU call: why does k_pass_queue's window fail to decide in the general regime (b)?  (BLANCE_QDIAG build), the
# GPU suite on the product library, and a kernel trace of regime (b) for k_ntn_bits
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
BLANCE_DEV_LIB=$PWD/devbuild/libblance_diag.so timeout 300 python tools/dev_general_regime.py > gpurun_out/r5/qdiag.log 2>&1
tail -3 gpurun_out/r5/qdiag.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5/gputests_1.log 2>&1
tail -3 gpurun_out/r5/gputests_1.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r5/trace_b -- env BLANCE_DEV_PRODUCT=1 python $GRAFT_REPO_ROOT/tools/dev_general_regime.py > $GRAFT_REPO_ROOT/gpurun_out/r5/trace_b.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r5/trace_b -name "*kernel_stats.csv" | head -1 | xargs -r head -30
