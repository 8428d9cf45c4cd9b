#!/bin/bash
# On the GPU box: configs 3 and 2 with the host's shortcuts (assumed classification, deferred verdicts, the opening pass known
# from the upload) off and on, three runs each; `full`: then the parity suite in both modes and with every speculation forced
# to fail.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
F="--no-extra --no-other-configs --no-replicas --no-cpu-baseline --no-sharded --no-transfers --no-live-pmc --no-rccl-one-rank"
mkdir -p gpurun_out/ab
for cfg in 3 2; do
for r in 1 2 3; do
for s in 0 1; do
  BLANCE_SPECULATE=$s timeout 300 python bench.py --config $cfg --steps 40 --warmup 5 $F > gpurun_out/ab/c${cfg}_s${s}_$r.json 2> gpurun_out/ab/c${cfg}_s${s}_$r.err
  python -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/ab/c${cfg}_s${s}_$r.json') if l.startswith('{')][-1])
print('config $cfg spec $s run $r: %.4f ms, syncs %s, digest %s' % (d['ms_per_step'], d.get('host_syncs_per_call'), d.get('matches_oracle_digest')))"
done; done; done
[ "$1" = full ] || exit 0
timeout 1200 python -m pytest tests/test_hip_parity.py -q -m gpu -x > gpurun_out/ab/parity_default.log 2>&1; grep -E "passed|failed" gpurun_out/ab/parity_default.log | tail -1
BLANCE_SPECULATE=fail timeout 1200 python -m pytest tests/test_hip_parity.py -q -m gpu -x > gpurun_out/ab/parity_fail.log 2>&1; grep -E "passed|failed" gpurun_out/ab/parity_fail.log | tail -1
BLANCE_SPECULATE=0 timeout 1200 python -m pytest tests/test_hip_parity.py -q -m gpu -x > gpurun_out/ab/parity_off.log 2>&1; grep -E "passed|failed" gpurun_out/ab/parity_off.log | tail -1
