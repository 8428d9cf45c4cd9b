#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/dev_tree_profile.py $1 $2 > gpurun_out/treeprof.log 2>&1
tail -120 gpurun_out/treeprof.log
