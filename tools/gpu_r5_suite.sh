#!/bin/bash
# round 5: the GPU suite on the product library, then transfer timings
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
T=${1:-s1}
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r5/gputests_$T.log 2>&1
tail -4 gpurun_out/r5/gputests_$T.log
timeout 300 python tools/dev_transfers.py > gpurun_out/r5/transfers_$T.log 2>&1
cat gpurun_out/r5/transfers_$T.log
