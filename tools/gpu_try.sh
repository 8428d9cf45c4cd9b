cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "golden or random_instances or config2 or config5 or tree_pass or edge or flat" > gpurun_out/r4/test_subset.log 2>&1; tail -3 gpurun_out/r4/test_subset.log
BLANCE_QUEUE_STATS=1 timeout 600 python bench.py --config 5 --steps 2 --warmup 0 --no-cpu-baseline --no-live-pmc > gpurun_out/r4/bench5_try.json 2> gpurun_out/r4/bench5_try.err
grep "k_pass_queue:" gpurun_out/r4/bench5_try.err | tail -2
python - <<'PY'
import json
for n in ("bench5_try.json",):
    try:
        d = json.loads([l for l in open("gpurun_out/r4/" + n) if l.startswith("{")][-1])
        print(n, "%.3f ms per call, %.2f M/s, digest ok %s, pass %.1f flat %.1f" % (d["ms_per_step"], d["value"] / 1e6, d.get("matches_oracle_digest"), d["pass_kernel_ms_per_step"], d["flat_pass_ms_per_step"]))
    except Exception as e:
        print(n, "no line:", e)
PY
BLANCE_QUEUE_STATS=1 timeout 600 python tools/dev_tree_profile.py 1048576 4096 > gpurun_out/r4/prof5_5.log 2>&1; grep "rebalance\|initial" gpurun_out/r4/prof5_5.log
