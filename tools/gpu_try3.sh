cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4b; export TMPDIR=/tmp
o=gpurun_out/r4b
timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "golden or random_instances or config2 or rowless or config3_rebalance or config5_miniature or tree_pass_flat_weighted or edge" > $o/test_subset6.log 2>&1; tail -3 $o/test_subset6.log
BLANCE_QUEUE_STATS=1 timeout 600 python bench.py --config 5 --steps 2 --warmup 0 --no-cpu-baseline --no-live-pmc > $o/bench5_try6.json 2> $o/bench5_try6.err
grep "k_pass_queue:" $o/bench5_try6.err | tail -2
timeout 600 python tools/dev_general_regime.py > $o/general_b9.log 2>&1; tail -2 $o/general_b6.log
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-live-pmc --no-sharded > $o/bench_try6.json 2> $o/bench_try6.err
python - <<'PY'
import json
for n in ("bench5_try6.json", "bench_try6.json",):
    try:
        d = json.loads([l for l in open("gpurun_out/r4b/" + n) if l.startswith("{")][-1])
        print(n, "%.3f ms per call, %.2f M/s, digest ok %s, pass %.1f flat %.1f" % (d["ms_per_step"], d["value"] / 1e6, d.get("matches_oracle_digest"), d["pass_kernel_ms_per_step"], d["flat_pass_ms_per_step"]))
        for w in d.get("general_regime", []):
            print("   ", w.get("workload", "")[:50], "ms", w.get("ms_per_step"), "sweeps", w.get("sweeps_per_call"), "ok", w.get("matches_oracle_digest"), "pass", w.get("pass_kernel_ms_per_step"), "flat", w.get("flat_pass_ms_per_step"), w.get("error"))
        if d.get("host_end_to_end"): print("    host", {k: v for k, v in d["host_end_to_end"].items() if k.endswith("_ms") or k in ("threads", "error")})
    except Exception as e:
        print(n, "no line:", e)
PY
