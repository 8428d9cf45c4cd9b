#!/bin/bash
# On the GPU box: k_pass_queue (flat passes, sorted window) -- parity subset, then config 5 with its statistics.
#   gpurun --timeout 1500 -- 'bash tools/gpu_queue.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/queue
mkdir -p "$out"
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "golden or random_instances or config2 or config5 or tree_pass or edge or flat" > "$out/test.log" 2>&1; tail -5 "$out/test.log"
BLANCE_QUEUE_STATS=1 timeout 600 python bench.py --config 5 --steps 2 --warmup 0 --no-cpu-baseline > "$out/bench_config5.json" 2> "$out/bench_config5.err"
grep "k_pass_queue" "$out/bench_config5.err" | tail -4
python - <<'PY'
import json, os
o = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "queue")
try:
    d = json.loads([l for l in open(os.path.join(o, "bench_config5.json")) if l.startswith("{")][-1])
    print("config 5: %.1f ms per call, %.2f M assignments/s, digest ok %s, pass kernels %.1f ms, flat passes %.1f ms" % (
        d["ms_per_step"], d["value"] / 1e6, d.get("matches_oracle_digest"), d["pass_kernel_ms_per_step"], d["flat_pass_ms_per_step"]))
except Exception as e:
    print("no line:", e)
PY
