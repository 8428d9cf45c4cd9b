cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4b; export TMPDIR=/tmp
o=gpurun_out/r4b/stress_gpu.txt
echo "python tests/tools/stress_gpu.py 200 51000" > $o
timeout 600 python tests/tools/stress_gpu.py 200 51000 2>&1 | tail -4 >> $o
echo "python tests/tools/stress_gpu.py 150 53000 --flat-heavy" >> $o
timeout 600 python tests/tools/stress_gpu.py 150 53000 --flat-heavy 2>&1 | tail -4 >> $o
cat $o
