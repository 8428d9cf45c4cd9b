#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "== config5 full"; BLANCE_TRACE=1 timeout 900 python tools/config5_gpu.py 2>&1 | grep -v "k_pass_tree state" | grep "initial\|rebalance\|pass 0 \|pass 1 \|pass 3 \|pass 19\|oracle"
echo "== tests"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== bench config 3"; timeout 600 python bench.py 2>&1 | tail -1
} > gpurun_out/r2d.log 2>&1
cat gpurun_out/r2d.log
