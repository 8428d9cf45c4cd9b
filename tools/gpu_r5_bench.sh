#!/bin/bash
# round 5: the default bench line as the driver runs it, and the transfer timings
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
T=${1:-b1}
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r5/bench_default_$T.log 2>&1
tail -c 600 gpurun_out/r5/bench_default_$T.log
timeout 300 python tools/dev_transfers.py > gpurun_out/r5/transfers_$T.log 2>&1
cat gpurun_out/r5/transfers_$T.log
