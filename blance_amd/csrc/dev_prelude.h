// Common prelude of every translation unit of libblance_hip.so (and of the SIMT emulator build, which
// includes them all into one): launch macros, the parameter blocks, the device helpers.
#pragma once
#ifndef BLANCE_SIMT_EMU
#include <hip/hip_runtime.h>
#define BLANCE_LAUNCH(kern, grid, block, lds, stream, ...) \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, stream, __VA_ARGS__)
#define BLANCE_LAUNCH_NOSYNC BLANCE_LAUNCH   /* kernel has no barrier / cross-lane op */
#define BLANCE_DYN_LDS(ptr)                                        \
    extern __shared__ __align__(16) unsigned char blance_lds_[];   \
    unsigned char* ptr = blance_lds_
// a wave64 runs in lockstep: LDS writes of one lane are seen by the other lanes'
// later reads without a barrier; this only pins the compiler's schedule
#define BLANCE_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
// s_waitcnt vmcnt(0) (gfx9 encoding: vmcnt in bits 3:0 and 15:14; expcnt 7 and lgkmcnt 15 = no wait)
#define BLANCE_WAIT_VMEM() __builtin_amdgcn_s_waitcnt(0x0F70)
#endif

#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/blance_hip.h"
#include "blance_kernels.h"
#include "dev_common.h"

namespace blance {
// ---- launch wrappers: one translation unit per pass-kernel family (they compile in parallel)
// k_pass_seq (tu_seq.hip): 0, or -1 when the cluster is too wide for the register-resident pass
int launch_pass_seq(hipStream_t stream, PassParams q, int force_threads, bool allow_spec);
// k_pass_tree (tu_tree.hip): false when the pass is outside its envelope (nothing launched)
bool launch_pass_tree(hipStream_t stream, PassParams q, int knobs);
// k_pass_queue (tu_queue.hip): flat passes with k <= 2 on one wave64, candidates as a sorted window; false: outside its envelope
bool launch_pass_queue(hipStream_t stream, PassParams q);
size_t queue_bits_words(int NX);   // words of PassParams::ntn_bits for a cluster of NX node names
// k_pass_chain (tu_chain.hip): one wave64 per region; false when the shape has no variant
bool launch_chain(hipStream_t stream, ChainParams& q, int max_size, bool fast);
// ... whether launch_chain would keep the regions' nodeToNodeCounts rows in LDS (then the matrix in HBM is not touched)
bool chain_rows_in_lds(const ChainParams& q, int max_size);
// k_pass_chain_blank (tu_chain.hip): the lean first-sweep kernel
void launch_chain_blank(hipStream_t stream, const ChainParams& q, int max_size);
// k_pass_chain_planes (tu_chain.hip): the all-blank pass as a scalar bit-plane automaton; false: shape outside it
bool launch_chain_planes(hipStream_t stream, const ChainParams& q, int max_size);
// k_stay_by_top (tu_chain.hip): a chain pass of stays, one thread per top priority node; false: shape outside it
bool launch_stay_by_top(hipStream_t stream, const StayParams& q, int n_wgs, int max_size);
}  // namespace blance
