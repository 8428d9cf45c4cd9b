#!/bin/bash
# developer aid: devbuild/libblance_parstats.so = the product objects with tu_par.hip and tu_pool.hip rebuilt under
# -DBLANCE_PAR_STATS (k_pass_par / k_pass_pool print their counters per launch); with an argument: also under
# -DBLANCE_PHASE_PROF (shader-clock totals per phase of k_pass_pool)
set -e
cd "$(dirname "$0")/.."
mkdir -p devbuild
EXTRA=""; [ -n "$1" ] && EXTRA="-DBLANCE_PHASE_PROF"
for tu in par; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DBLANCE_PAR_STATS $EXTRA -c -o devbuild/tu_${tu}_stats.o blance_amd/csrc/tu_${tu}.hip
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o devbuild/libblance_parstats.so blance_amd/lib/obj/blance_hip.o blance_amd/lib/obj/tu_seq.o blance_amd/lib/obj/tu_tree.o blance_amd/lib/obj/tu_chain.o devbuild/tu_par_stats.o
