#!/bin/bash
# developer aid: libblance_prof.so = the product objects with one translation unit (default tu_tree) rebuilt under
# -DBLANCE_PHASE_PROF (per-phase shader-clock totals printed by the kernels):  tools/profile/build_prof.sh [tu_tree|tu_chain|tu_seq|tu_queue]
set -e
cd "$(dirname "$0")/../.."
tu=${1:-tu_tree}
mkdir -p devbuild
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DBLANCE_PHASE_PROF -c -o devbuild/${tu}_prof.o blance_amd/csrc/${tu}.hip
objs=""
for t in blance_hip tu_seq tu_chain tu_tree tu_queue; do
    if [ "$t" = "$tu" ]; then objs="$objs devbuild/${tu}_prof.o"; else objs="$objs blance_amd/lib/obj/$t.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o devbuild/libblance_prof.so $objs
