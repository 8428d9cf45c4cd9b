#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "== bench config3"; BLANCE_TRACE=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -v "k_pass_tree state" | tail -14
echo "== gpu tests"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
} > gpurun_out/r2c.log 2>&1
tail -30 gpurun_out/r2c.log
