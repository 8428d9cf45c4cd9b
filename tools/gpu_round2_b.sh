#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "== config5 full"; BLANCE_TRACE=1 timeout 900 python tools/config5_gpu.py 2>&1 | grep -v "k_pass_tree state" | tail -60
} > gpurun_out/r2b.log 2>&1
tail -70 gpurun_out/r2b.log
