cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_planner_api.py tests/test_host_cpp.py tests/test_stats.py -q -m gpu -x -k "golden or random_instances or config2 or config5 or tree_pass or edge or flat or stats or mirror" > gpurun_out/r4/test_subset.log 2>&1; tail -3 gpurun_out/r4/test_subset.log
BLANCE_QUEUE_STATS=1 timeout 600 python bench.py --config 5 --steps 2 --warmup 0 --no-cpu-baseline --no-live-pmc > gpurun_out/r4/bench5_try.json 2> gpurun_out/r4/bench5_try.err
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-live-pmc --no-sharded > gpurun_out/r4/bench_try3.json 2> gpurun_out/r4/bench_try3.err
python - <<'PY'
import json
for n in ("bench5_try.json", "bench_try3.json",):
    try:
        d = json.loads([l for l in open("gpurun_out/r4/" + n) if l.startswith("{")][-1])
        print(n, "%.3f ms per call, %.2f M/s, digest ok %s, pass %.1f flat %.1f" % (d["ms_per_step"], d["value"] / 1e6, d.get("matches_oracle_digest"), d["pass_kernel_ms_per_step"], d["flat_pass_ms_per_step"]))
        for w in d.get("general_regime", []):
            print("   ", w.get("workload", "")[:50], "ms", w.get("ms_per_step"), "sweeps", w.get("sweeps_per_call"), "ok", w.get("matches_oracle_digest"), "pass", w.get("pass_kernel_ms_per_step"), "flat", w.get("flat_pass_ms_per_step"), w.get("error"))
    except Exception as e:
        print(n, "no line:", e)
PY
