#!/bin/bash
# On the GPU box: tests/tools/stress_gpu.py with two seed ranges (general and flat-heavy); the tail of each run goes to gpurun_out/r6e/stress_gpu.txt
#   gpurun --timeout 1500 -- 'bash tools/profile/gpu_stress.sh 150 71000 100 73000'
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6e; export TMPDIR=/tmp
o=gpurun_out/r6e/stress_gpu.txt
n1=${1:-200}; s1=${2:-51000}; n2=${3:-150}; s2=${4:-53000}
echo "python tests/tools/stress_gpu.py $n1 $s1" > $o
timeout 900 python tests/tools/stress_gpu.py $n1 $s1 2>&1 | tail -3 >> $o
echo "python tests/tools/stress_gpu.py $n2 $s2 --flat-heavy" >> $o
timeout 900 python tests/tools/stress_gpu.py $n2 $s2 --flat-heavy 2>&1 | tail -3 >> $o
cat $o
