#!/bin/bash
# On the GPU box: tests/tools/stress_gpu.py under the host's modes (default, every assumption forced to fail, none taken,
# k_stay_by_top tried in every pass, the chain kernel on four waves); the tails go to gpurun_out/r6e/stress_gpu_modes.txt
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6e; export TMPDIR=/tmp
o=gpurun_out/r6e/stress_gpu_modes.txt; : > $o
run() { echo "$1" >> $o; shift; env "$@" 2>&1 | tail -1 >> $o; }
run "python tests/tools/stress_gpu.py 400 181000" X=1 timeout 900 python tests/tools/stress_gpu.py 400 181000
run "python tests/tools/stress_gpu.py 200 183000 --flat-heavy" X=1 timeout 900 python tests/tools/stress_gpu.py 200 183000 --flat-heavy
run "BLANCE_SPECULATE=fail python tests/tools/stress_gpu.py 150 185000" BLANCE_SPECULATE=fail timeout 900 python tests/tools/stress_gpu.py 150 185000
run "BLANCE_SPECULATE=0 python tests/tools/stress_gpu.py 100 187000" BLANCE_SPECULATE=0 timeout 900 python tests/tools/stress_gpu.py 100 187000
run "STRESS_PLANNER_KW={stay_top: force} python tests/tools/stress_gpu.py 150 189000" STRESS_PLANNER_KW='{"stay_top": "force"}' timeout 900 python tests/tools/stress_gpu.py 150 189000
run "BLANCE_CHAIN_WAVES=4 python tests/tools/stress_gpu.py 100 191000" BLANCE_CHAIN_WAVES=4 timeout 900 python tests/tools/stress_gpu.py 100 191000
run "BLANCE_CHAIN_WAVES=8 python tests/tools/stress_gpu.py 100 193000" BLANCE_CHAIN_WAVES=8 timeout 900 python tests/tools/stress_gpu.py 100 193000
cat $o
