#!/bin/bash
# first GPU pass of round 2: tree kernel parity + timings
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "== tree parity tests"; timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "tree or wide_flat or miniature or config2 or golden" 2>&1 | tail -5
echo "== config5 full"; BLANCE_TRACE= timeout 900 python tools/config5_gpu.py 2>&1 | tail -12
echo "== bench config2"; timeout 300 python bench.py --config 2 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1
echo "== bench config3"; timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1
} > gpurun_out/r2a.log 2>&1
tail -40 gpurun_out/r2a.log
