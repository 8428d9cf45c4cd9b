#!/bin/bash
# developer aid: libblance_prof.so = the product objects with tu_tree.hip rebuilt under -DBLANCE_PHASE_PROF
set -e
cd "$(dirname "$0")/.."
mkdir -p devbuild
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DBLANCE_PHASE_PROF -c -o devbuild/tu_tree_prof.o blance_amd/csrc/tu_tree.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o devbuild/libblance_prof.so blance_amd/lib/obj/blance_hip.o blance_amd/lib/obj/tu_seq.o blance_amd/lib/obj/tu_chain.o devbuild/tu_tree_prof.o
