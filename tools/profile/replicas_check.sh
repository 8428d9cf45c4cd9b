cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 600 python bench.py --steps 5 --warmup 2 --no-extra --no-other-configs --no-cpu-baseline --no-sharded --no-transfers --no-live-pmc --no-rccl-one-rank > gpurun_out/repl_check.json 2> gpurun_out/repl_check.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/repl_check.json') if l.startswith('{')][-1])
print(d["ms_per_step"])
for blk in d["replicas_on_one_gpu"]:
    print(blk["config"], [(x["R"], round(x["aggregate_value"]/1e6,1), round(x["plan_latency_ms"]["mean"],2)) for x in blk["runs"]])
PY
