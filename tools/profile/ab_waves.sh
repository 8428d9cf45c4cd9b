#!/bin/bash
# On the GPU box: config 3 with k_pass_chain on four and on eight waves per region, three runs each.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
F="--no-extra --no-other-configs --no-replicas --no-cpu-baseline --no-sharded --no-transfers --no-live-pmc --no-rccl-one-rank"
mkdir -p gpurun_out/ab
for r in 1 2 3; do
for w in 4 8; do
  BLANCE_CHAIN_WAVES=$w timeout 300 python bench.py --steps 40 --warmup 5 $F > gpurun_out/ab/c3_w${w}_$r.json 2> gpurun_out/ab/c3_w${w}_$r.err
  python -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/ab/c3_w${w}_$r.json') if l.startswith('{')][-1])
print('waves $w run $r: %.4f ms, syncs %s, digest %s' % (d['ms_per_step'], d.get('host_syncs_per_call'), d.get('matches_oracle_digest')))"
done; done
