#!/bin/bash
# On the GPU box: tests/tools/stress_gpu.py with two seed ranges (general and flat-heavy); the tail of each run goes to gpurun_out/r4b/stress_gpu2.txt
#   gpurun --timeout 1500 -- 'bash tools/gpu_stress4.sh 400 60000 300 62000'
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4b; export TMPDIR=/tmp
o=gpurun_out/r4b/stress_gpu2.txt
n1=${1:-200}; s1=${2:-51000}; n2=${3:-150}; s2=${4:-53000}
echo "python tests/tools/stress_gpu.py $n1 $s1" > $o
timeout 900 python tests/tools/stress_gpu.py $n1 $s1 2>&1 | tail -3 >> $o
echo "python tests/tools/stress_gpu.py $n2 $s2 --flat-heavy" >> $o
timeout 900 python tests/tools/stress_gpu.py $n2 $s2 --flat-heavy 2>&1 | tail -3 >> $o
cat $o
