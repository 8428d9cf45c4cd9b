// MI355X (gfx950) implementation of blance's planNextMapEx (plan.go:23-58) behind
// the C ABI of include/blance_hip.h.  One blance_plan() call runs the whole
// convergence loop on the device; the host only sequences kernels and reads
// one convergence word per sweep.  See DESIGN.md for the kernel inventory.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC
// (fp64 scores must keep the reference's operation order, plan.go:634-689).
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
#endif

#include <limits.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <type_traits>
#include <string>
#include <vector>

#include "../../include/blance_hip.h"
#include "blance_kernels.h"

namespace blance {

// ============================================================================
// Device helpers
// ============================================================================

// nodeSorter.Score, plan.go:634-689, in the reference's operation order.
// Absent map keys are zeros here: x + 0.0 and x - 0.0 are exact.
__device__ __forceinline__ double node_score(int cnt, int ntn, int tot, int hasw, int w, int NP,
                                             double cf, int booster) {
    double lp = 0.0, ff = 0.0;
    if (NP > 0) {
        lp = (double)ntn / (double)NP;              // plan.go:638-644
        ff = (0.001 * (double)tot) / (double)NP;    // plan.go:647-652
    }
    double r = (double)cnt;                         // plan.go:664-670
    r = r + lp;
    r = r + ff;
    if (hasw) {                                     // plan.go:675-684
        if (w > 0) {
            r = r / (double)w;
        } else if (w < 0 && booster == BLANCE_BOOSTER_CBGT) {
            double b = (double)(-w);                // control_test.go:19-26
            if (b < cf) b = cf;
            r = r + b;
        }
    }
    r = r - cf;                                     // plan.go:686
    return r;
}

// nodeSorter.Less, plan.go:617-628: (score, position) ascending, strict total order.
__device__ __forceinline__ bool better(double s1, int n1, double s2, int n2) {
    return s1 < s2 || (s1 == s2 && n1 < n2);
}

__device__ __forceinline__ double pos_inf() { return __longlong_as_double(0x7ff0000000000000LL); }

struct RedSlot { double s; int n; int pad; };

// Lexicographic (score, position) argmin over a workgroup of T threads.
// One barrier per call; slots are double-buffered by call parity.
template <int T>
__device__ __forceinline__ int block_argmin(double s, int n, RedSlot* red, int& round) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        double s2 = __shfl_xor(s, off, 64);
        int n2 = __shfl_xor(n, off, 64);
        if (better(s2, n2, s, n)) { s = s2; n = n2; }
    }
    constexpr int W = T / 64;
    if (W == 1) return n;
    RedSlot* slot = red + (round & 1) * W;
    round++;
    if ((threadIdx.x & 63) == 0) {
        slot[threadIdx.x >> 6].s = s;
        slot[threadIdx.x >> 6].n = n;
    }
    __syncthreads();
    double bs = slot[0].s;
    int bn = slot[0].n;
#pragma unroll
    for (int j = 1; j < W; j++) {
        double s2 = slot[j].s;
        int n2 = slot[j].n;
        if (better(s2, n2, bs, bn)) { bs = s2; bn = n2; }
    }
    return bn;
}

// Running value of includeExcludeNodesIntersect (plan.go:738-753) as leaf-interval
// algebra: one include interval minus a few excluded sub-intervals.  Leaf
// intervals of tree vertices are laminar (nested or disjoint), which keeps
// every intermediate in this form (DESIGN.md "Hierarchy masks").
template <int XN>
struct FoldT {
    int empty;
    int ilo, ihi;
    int nx;
    int xlo[XN], xhi[XN];
};

template <int XN>
__device__ __forceinline__ void fold_reset(FoldT<XN>& f) {
    f.empty = 1; f.ilo = 0; f.ihi = 0; f.nx = 0;
#pragma unroll
    for (int j = 0; j < XN; j++) { f.xlo[j] = 0; f.xhi[j] = 0; }
}

template <int XN>
__device__ __forceinline__ void fold_push_x(FoldT<XN>& f, int lo, int hi, int* err) {
    // clip to the include interval (laminar: disjoint, inside, or covering)
    if (hi <= f.ilo || lo >= f.ihi) return;
    if (lo <= f.ilo && hi >= f.ihi) { f.empty = 1; return; }
    bool dup = false;
#pragma unroll
    for (int j = 0; j < XN; j++)
        if (j < f.nx && f.xlo[j] == lo && f.xhi[j] == hi) dup = true;
    if (dup) return;
    if (f.nx >= XN) { *err = 1; return; }
#pragma unroll
    for (int j = 0; j < XN; j++)
        if (j == f.nx) { f.xlo[j] = lo; f.xhi[j] = hi; }
    f.nx++;
}

template <int XN>
__device__ __forceinline__ void fold_check_empty(FoldT<XN>& f) {
    if (f.empty) return;
    int covered = 0;
#pragma unroll
    for (int i = 0; i < XN; i++) {
        if (i >= f.nx) continue;
        bool nested = false;
#pragma unroll
        for (int j = 0; j < XN; j++)
            if (j < f.nx && j != i && f.xlo[j] <= f.xlo[i] && f.xhi[i] <= f.xhi[j]) nested = true;
        if (!nested) covered += f.xhi[i] - f.xlo[i];
    }
    if (covered >= f.ihi - f.ilo) f.empty = 1;
}

// One step of the fold: rv = (len(rv) == 0) ? set(a) : rv ∩ set(a)   (plan.go:744-750)
template <int XN>
__device__ __forceinline__ void fold_step(FoldT<XN>& f, AnchorSet a, int* err) {
    bool set_empty = (a.blo <= a.alo && a.bhi >= a.ahi);   // exclude covers include
    if (f.empty) {
        f.empty = set_empty ? 1 : 0;
        f.ilo = a.alo; f.ihi = a.ahi; f.nx = 0;
        if (!set_empty) fold_push_x(f, a.blo, a.bhi, err);
        return;
    }
    if (set_empty) { f.empty = 1; return; }
    // include ∩ include
    int lo = f.ilo > a.alo ? f.ilo : a.alo;
    int hi = f.ihi < a.ahi ? f.ihi : a.ahi;
    if (lo >= hi) { f.empty = 1; return; }
    if (lo != f.ilo || hi != f.ihi) {      // the include interval shrank: re-clip the exclusions
        int onx = f.nx;
        int olo[XN], ohi[XN];
#pragma unroll
        for (int j = 0; j < XN; j++) { olo[j] = f.xlo[j]; ohi[j] = f.xhi[j]; }
        f.ilo = lo; f.ihi = hi; f.nx = 0;
#pragma unroll
        for (int j = 0; j < XN; j++)
            if (j < onx && !f.empty) fold_push_x(f, olo[j], ohi[j], err);
    }
    if (!f.empty) fold_push_x(f, a.blo, a.bhi, err);
    fold_check_empty(f);
}

template <int XN>
__device__ __forceinline__ bool fold_contains(const FoldT<XN>& f, int pos) {
    if (f.empty || pos < f.ilo || pos >= f.ihi) return false;
    bool in = true;
#pragma unroll
    for (int j = 0; j < XN; j++)
        if (j < f.nx && pos >= f.xlo[j] && pos < f.xhi[j]) in = false;
    return in;
}

using Fold = FoldT<kMaxAnchors>;

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ============================================================================
// The sequential state pass: assignStateToPartitions (plan.go:253-303) with
// findBestNodes (plan.go:98-248) inlined.  ONE workgroup walks the partitions
// in pass order; thread t owns nodes t, t+T, ... and keeps their load counts,
// total counts, weights and partition-independent scores in registers, so a
// step costs one barrier per argmin and no table traffic.
// ============================================================================
template <int T, int NPT>
__global__ __launch_bounds__(T) void k_pass_seq(PassParams q) {
    BLANCE_DYN_LDS(lds);
    RedSlot* red = (RedSlot*)lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int N = q.N, NX = q.NX, M = q.M, L = q.L, NP = q.NP, s = q.s, k = q.k;
    const int SW = 1 + L;                        // words per state inside a record
    int round = 0;

    int cntv[NPT], totv[NPT], wv[NPT], lpos[NPT];
    unsigned alive_m = 0, hasw_m = 0;
    double g[NPT];
#pragma unroll
    for (int i = 0; i < NPT; i++) {
        int n = tid + i * T;
        cntv[i] = 0; totv[i] = 0; wv[i] = 0; lpos[i] = -1; g[i] = 0.0;
        if (n < NX) {
            cntv[i] = q.cnt[s * NX + n];
            int tsum = 0;
            for (int t = 0; t <= M; t++) tsum += q.cnt[t * NX + n];   // plan.go:118-124
            totv[i] = tsum;
            wv[i] = q.node_weight[n];
            if (q.node_has_weight[n]) hasw_m |= 1u << i;
            if (n < N && q.alive[n]) alive_m |= 1u << i;
            lpos[i] = q.node_leaf_pos[n];
            g[i] = node_score(cntv[i], 0, totv[i], (hasw_m >> i) & 1, wv[i], NP, 0.0, q.booster_kind);
        }
    }

    // step record of the current partition: lane j of every wave holds word j
    int recw = 0, recw_next = 0;
    if (q.beg < q.end && lane < q.RW) recw_next = q.rec[(size_t)q.beg * q.RW + lane];

    for (int oi = q.beg; oi < q.end; oi++) {
        recw = recw_next;
        if (oi + 1 < q.end && lane < q.RW) recw_next = q.rec[(size_t)(oi + 1) * q.RW + lane];
#define REC(i) __builtin_amdgcn_readlane(recw, (i))
        const int p = REC(0);
        const int w = REC(1);
        const double stick = __hiloint2double(REC(3), REC(2));
        // topPriorityNode, plan.go:134-138
        int top = -1;
        {
            int hdr = REC(kRecHead + q.top_state * SW);
            if ((hdr >> 16) != kListAbsent && (hdr & 0xffff) > 0) top = REC(kRecHead + q.top_state * SW + 1);
        }
        const int row = top < 0 ? NX : top;

        // nodeToNodeCounts row of the top priority node (only read when NP > 0, plan.go:638)
        int ntnv[NPT];
#pragma unroll
        for (int i = 0; i < NPT; i++) {
            int n = tid + i * T;
            ntnv[i] = (NP > 0 && n < N) ? q.ntn[(size_t)row * N + n] : 0;
        }

        // membership of my nodes in the higher-priority lists (plan.go:146-154)
        // and in this state's current list (plan.go:654-662)
        unsigned inh_m = 0, own_m = 0;
        int any_higher_key = 0;
        for (int t = 0; t < M; t++) {
            int hdr = REC(kRecHead + t * SW);
            if ((hdr >> 16) == kListAbsent) continue;
            int len = hdr & 0xffff;
            bool higher = (q.higher_mask >> t) & 1;
            if (higher) any_higher_key = 1;
            if (!higher && t != s) continue;
            for (int j = 0; j < len; j++) {
                int x = REC(kRecHead + t * SW + 1 + j);
#pragma unroll
                for (int i = 0; i < NPT; i++) {
                    if (x == tid + i * T) {
                        if (higher) inh_m |= 1u << i;
                        if (t == s) own_m |= 1u << i;
                    }
                }
            }
        }
        const unsigned elig_m = alive_m & ~inh_m;

        double sc[NPT];
#pragma unroll
        for (int i = 0; i < NPT; i++) {
            bool own = (own_m >> i) & 1;
            if (own || ntnv[i] != 0)
                sc[i] = node_score(cntv[i], ntnv[i], totv[i], (hasw_m >> i) & 1, wv[i], NP,
                                   own ? stick : 0.0, q.booster_kind);
            else
                sc[i] = g[i];
        }

        int chosen[kMaxK];
#pragma unroll
        for (int j = 0; j < kMaxK; j++) chosen[j] = -1;
        int n_out = 0;
        unsigned emitted_m = 0;                    // my nodes already in the output list

        if (q.hier) {                              // plan.go:174-226
            int hn[kMaxAnchors];
#pragma unroll
            for (int j = 0; j < kMaxAnchors; j++) hn[j] = -1;
            int n_hn = 0;
            int cand0 = -2;                        // candidateNodes[0], computed lazily
            int err = 0;
            for (int r = q.rule_begin; r < q.rule_end; r++) {
                const AnchorSet* tab = q.anchors + (size_t)r * (NX + 1);
                int h = top < 0 ? q.vertex_empty_anchor : top;
                if (top < 0 && n_hn > 0) h = hn[0];
                Fold f;
                fold_reset(f);
                {
                    AnchorSet a = tab[h];
                    a.alo = uni(a.alo); a.ahi = uni(a.ahi); a.blo = uni(a.blo); a.bhi = uni(a.bhi);
                    fold_step(f, a, &err);
                }
#pragma unroll
                for (int j = 0; j < kMaxAnchors; j++) {
                    if (j < n_hn) {
                        AnchorSet a = tab[hn[j]];
                        a.alo = uni(a.alo); a.ahi = uni(a.ahi); a.blo = uni(a.blo); a.bhi = uni(a.bhi);
                        fold_step(f, a, &err);
                    }
                }
                for (int i = 0; i < k; i++) {
                    // best node of the rule's set ∩ nodesNext − higher priority nodes (plan.go:185-212)
                    double bs = pos_inf();
                    int bn = INT_MAX;
#pragma unroll
                    for (int u = 0; u < NPT; u++) {
                        if (((elig_m >> u) & 1) && lpos[u] >= 0 && fold_contains(f, lpos[u]) &&
                            better(sc[u], tid + u * T, bs, bn)) {
                            bs = sc[u]; bn = tid + u * T;
                        }
                    }
                    int best = uni(block_argmin<T>(bs, bn, red, round));
                    int pick = -1;
                    if (best != INT_MAX) {
                        pick = best;
                    } else {                        // plan.go:216-218
                        if (cand0 == -2) {
                            double cs = pos_inf();
                            int cn = INT_MAX;
#pragma unroll
                            for (int u = 0; u < NPT; u++) {
                                if (((elig_m >> u) & 1) && better(sc[u], tid + u * T, cs, cn)) {
                                    cs = sc[u]; cn = tid + u * T;
                                }
                            }
                            cand0 = uni(block_argmin<T>(cs, cn, red, round));
                            if (cand0 == INT_MAX) cand0 = -1;
                        }
                        pick = cand0;
                    }
                    if (pick >= 0) {
                        if (n_hn >= kMaxAnchors - 1) { err = 1; }
                        else {
#pragma unroll
                            for (int j = 0; j < kMaxAnchors; j++) if (j == n_hn) hn[j] = pick;
                            n_hn++;
                            AnchorSet a = tab[pick];
                            a.alo = uni(a.alo); a.ahi = uni(a.ahi); a.blo = uni(a.blo); a.bhi = uni(a.bhi);
                            fold_step(f, a, &err);
                        }
                    }
                }
            }
            if (err && tid == 0) *q.err = 1;
            // candidateNodes = dedupe(hierarchyNodes ++ candidateNodes), plan.go:224-225
#pragma unroll
            for (int j = 0; j < kMaxAnchors; j++) {
                if (j < n_hn && n_out < k) {
                    int x = hn[j];
                    bool dup = false;
#pragma unroll
                    for (int c = 0; c < kMaxK; c++) if (c < n_out && chosen[c] == x) dup = true;
                    if (!dup) {
#pragma unroll
                        for (int c = 0; c < kMaxK; c++) if (c == n_out) chosen[c] = x;
                        n_out++;
#pragma unroll
                        for (int u = 0; u < NPT; u++) if (x == tid + u * T) emitted_m |= 1u << u;
                    }
                }
            }
        }
        // the sorted candidate list consumed lazily (plan.go:171-172, :228-235)
        while (n_out < k) {
            double bs = pos_inf();
            int bn = INT_MAX;
#pragma unroll
            for (int u = 0; u < NPT; u++) {
                if (((elig_m & ~emitted_m) >> u) & 1) {
                    if (better(sc[u], tid + u * T, bs, bn)) { bs = sc[u]; bn = tid + u * T; }
                }
            }
            int best = uni(block_argmin<T>(bs, bn, red, round));
            if (best == INT_MAX) break;
#pragma unroll
            for (int c = 0; c < kMaxK; c++) if (c == n_out) chosen[c] = best;
            n_out++;
#pragma unroll
            for (int u = 0; u < NPT; u++) if (best == tid + u * T) emitted_m |= 1u << u;
        }

        // ---- commit (plan.go:238-245, :290-301); every thread updates the nodes it owns
        unsigned changed_m = 0;
        for (int t = 0; t < M; t++) {
            int hdr = REC(kRecHead + t * SW);
            if ((hdr >> 16) == kListAbsent) continue;
            int len = hdr & 0xffff;
            for (int j = 0; j < len; j++) {
                int x = REC(kRecHead + t * SW + 1 + j);
                bool hit = (t == s);
                if (!hit) {
                    // x also held this state (plan.go:290-293) or was chosen now (plan.go:294-297)
                    int hs = REC(kRecHead + s * SW);
                    if ((hs >> 16) != kListAbsent) {
                        int ls = hs & 0xffff;
                        for (int jj = 0; jj < ls; jj++)
                            if (REC(kRecHead + s * SW + 1 + jj) == x) hit = true;
                    }
#pragma unroll
                    for (int c = 0; c < kMaxK; c++) if (c < n_out && chosen[c] == x) hit = true;
                }
                if (!hit) continue;
#pragma unroll
                for (int u = 0; u < NPT; u++) {
                    if (x == tid + u * T) {
                        totv[u] -= w;
                        if (t == s) cntv[u] -= w;
                        changed_m |= 1u << u;
                    }
                }
                if (t != s && tid == 0) q.cnt[t * NX + x] -= w;
            }
        }
#pragma unroll
        for (int c = 0; c < kMaxK; c++) {
            if (c < n_out) {
                int x = chosen[c];
#pragma unroll
                for (int u = 0; u < NPT; u++) {
                    if (x == tid + u * T) {
                        cntv[u] += w;
                        totv[u] += w;
                        changed_m |= 1u << u;
                        if (NP > 0) q.ntn[(size_t)row * N + x] = ntnv[u] + 1;   // plan.go:238-245
                    }
                }
            }
        }
        if (changed_m) {
#pragma unroll
            for (int u = 0; u < NPT; u++)
                if ((changed_m >> u) & 1)
                    g[u] = node_score(cntv[u], 0, totv[u], (hasw_m >> u) & 1, wv[u], NP, 0.0, q.booster_kind);
        }
        if (tid == 0) {
            int is_nil = (n_out == 0 && q.n_alive == 0 && !any_higher_key && !q.hier);
            int* o = q.out + (size_t)oi * q.OW;
            o[0] = n_out | (is_nil << 16);
#pragma unroll
            for (int c = 0; c < kMaxK; c++) if (c < k) o[1 + c] = chosen[c];
            if (n_out < k) {                       // plan.go:230-235
                int wi = *q.warn_count;
                q.warn_part[wi] = p;
                q.warn_state[wi] = s;
                *q.warn_count = wi + 1;
            }
        }
#undef REC
    }

#pragma unroll
    for (int i = 0; i < NPT; i++) {
        int n = tid + i * T;
        if (n < NX) q.cnt[s * NX + n] = cntv[i];
    }
}

// ============================================================================
// Region chains.  When the state's hierarchy rule cuts the cluster into regions
// (every node's include set is the same leaf interval as its neighbours'), a
// step whose top priority node and current nodes all live in one region reads
// and writes only that region's counters.  Steps of different regions commute,
// so each region's steps run as an independent in-order chain on one wave64:
// lanes own the region's leaves, the region's slice of nodeToNodeCounts sits
// in LDS, and the argmin is a DPP reduction -- no barrier, no global traffic
// on the critical path.  A chain that would have to look outside its region
// (fallback to candidateNodes[0], unmet constraints) raises flags[1] and the
// host redoes the whole pass with k_pass_seq.
// ============================================================================
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
}

template <int CTRL>
__device__ __forceinline__ void argmin_stage(double& s, int& n) {
    int lo2 = dpp_mov<CTRL>(__double2loint(s));
    int hi2 = dpp_mov<CTRL>(__double2hiint(s));
    int n2 = dpp_mov<CTRL>(n);
    double s2 = __hiloint2double(hi2, lo2);
    if (better(s2, n2, s, n)) { s = s2; n = n2; }
}

// (score, position) argmin over one wave64; result is wave-uniform.
__device__ __forceinline__ int wave_argmin(double s, int n) {
    argmin_stage<0xB1>(s, n);     // quad_perm [1,0,3,2]
    argmin_stage<0x4E>(s, n);     // quad_perm [2,3,0,1]
    argmin_stage<0x141>(s, n);    // row_half_mirror
    argmin_stage<0x140>(s, n);    // row_mirror: every row of 16 now agrees
    double bs = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s), 0),
                                 __builtin_amdgcn_readlane(__double2loint(s), 0));
    int bn = __builtin_amdgcn_readlane(n, 0);
#pragma unroll
    for (int r = 16; r < 64; r += 16) {
        double s2 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s), r),
                                     __builtin_amdgcn_readlane(__double2loint(s), r));
        int n2 = __builtin_amdgcn_readlane(n, r);
        if (better(s2, n2, bs, bn)) { bs = s2; bn = n2; }
    }
    return bn;
}

#ifdef BLANCE_PHASE_PROF     // developer build only: per-phase shader-clock totals of chain 0
#define PH_DECL unsigned long long ph_acc[12] = {0}, ph_t0 = clock64(), ph_t1
#define PH(i) do { ph_t1 = clock64(); ph_acc[i] += ph_t1 - ph_t0; ph_t0 = ph_t1; } while (0)
#define PH_DUMP(steps) do { if (blockIdx.x == 0 && threadIdx.x == 0) { for (int i_ = 0; i_ < 12; i_++) \
    printf("[phase %d] %.1f cycles/step\n", i_, (double)ph_acc[i_] / (double)(steps)); } } while (0)
#elif defined(BLANCE_ASM_MARKS)   // developer build only: phase markers as comments in the ISA
#define PH_DECL
#define PH(i) asm volatile("; PHASE_MARK " #i)
#define PH_DUMP(steps)
#else
#define PH_DECL
#define PH(i)
#define PH_DUMP(steps)
#endif

// nodeSorter.Score with the two quotients that do not depend on the node taken
// from LDS tables filled by the same expressions (bit-identical by construction).
__device__ __forceinline__ double chain_score(int cnt, int ntn, int tot, int hasw, int w, int NP, double cf,
                                              int booster, const double* lp_tab, const double* ff_tab) {
    double lp = 0.0, ff = 0.0;
    if (NP > 0) {
        bool lin = (unsigned)ntn < (unsigned)kLpTab, fin = (unsigned)tot < (unsigned)kFfTab;
        lp = lp_tab[lin ? ntn : 0];
        ff = ff_tab[fin ? tot : 0];
        if (!lin) lp = (double)ntn / (double)NP;
        if (!fin) ff = (0.001 * (double)tot) / (double)NP;
    }
    double r = (double)cnt;
    r = r + lp;
    r = r + ff;
    if (hasw) {
        if (w > 0) {
            r = r / (double)w;
        } else if (w < 0 && booster == BLANCE_BOOSTER_CBGT) {
            double b = (double)(-w);
            if (b < cf) b = cf;
            r = r + b;
        }
    }
    r = r - cf;
    return r;
}

// 32-bit minimum over one wave64 (wave-uniform result)
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    unsigned t;
    t = (unsigned)dpp_mov<0xB1>((int)v);  v = t < v ? t : v;
    t = (unsigned)dpp_mov<0x4E>((int)v);  v = t < v ? t : v;
    t = (unsigned)dpp_mov<0x141>((int)v); v = t < v ? t : v;
    t = (unsigned)dpp_mov<0x140>((int)v); v = t < v ? t : v;
    unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)v, 0), r1 = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    unsigned r2 = (unsigned)__builtin_amdgcn_readlane((int)v, 32), r3 = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    r0 = r1 < r0 ? r1 : r0;
    r2 = r3 < r2 ? r3 : r2;
    return r2 < r0 ? r2 : r0;
}

// FAST = the pass has NP == 0 and no node weights: nodeSorter.Score is then
// double(count) - currentFactor with currentFactor in {1.5, integers}, so
// 2 * score is an exact small integer and (score, position) packs into one
// 32-bit key: [ 2*count - 2*currentFactor + 2^17 | node id (13 bits) ].  Lanes
// whose counters leave the representable range make the chain escape.
constexpr int kKeyBias = 1 << 17;
constexpr unsigned kKeyNone = 0xffffffffu;

template <int NPTC, int KM, bool FAST>
__global__ __launch_bounds__(64) void k_pass_chain(ChainParams q) {
    BLANCE_DYN_LDS(lds);
    if (q.flags[0]) return;
    const int lane = threadIdx.x;
    const int rg = blockIdx.x;
    const int lo = q.reg_lo[rg], hi = q.reg_hi[rg], size = hi - lo;
    const int cbeg = q.reg_off[rg], cend = q.reg_off[rg + 1];
    if (cbeg >= cend) return;
    const int N = q.N, NX = q.NX, M = q.M, NP = q.NP, s = q.s, k = q.k;
    // LDS: quotient tables, mirrors of the per-leaf registers (read by the stay
    // validators), a 64-step staging area for records and outputs (no global memory
    // operation inside the step loop), the region's nodeToNodeCounts rows
    double* lp_tab = (double*)lds;                   // [kLpTab]
    double* ff_tab = lp_tab + kLpTab;                // [kFfTab]
    double* gL = ff_tab + kFfTab;                    // [size]
    int* cntL = (int*)(gL + size);                   // [size]
    int* totL = cntL + size;
    int* nidL = totL + size;
    int* wgtL = nidL + size;
    int* flgL = wgtL + size;                         // bit 0 alive (in nodesNext), bit 1 has weight
    int* clsL = flgL + size;                         // exclude class of the leaf's node, -1 if none
    int* cszL = clsL + size;                         // leaves covered by class c
    int* recbuf = cszL + size;                       // [64][kCW]
    int* outbuf = recbuf + 64 * kCW;                 // [64][OW]
    int* ntn_l = outbuf + 64 * q.OW;                 // [size][ST] nodeToNodeCounts rows, padded stride
    const int ST = size + 1;
    if (!FAST && NP > 0) {
        for (int i = lane; i < kLpTab; i += 64) lp_tab[i] = (double)i / (double)NP;
        for (int i = lane; i < kFfTab; i += 64) ff_tab[i] = (0.001 * (double)i) / (double)NP;
        if (q.ntn_in_lds)
            for (int i = lane; i < (size + 1) * ST; i += 64) ntn_l[i] = 0;    // last row: "" (flat mode)
    }
    for (int i = lane; i < size; i += 64) cszL[i] = q.cls_size[lo + i];
    __syncthreads();

    // lane l owns leaves lo + l + 64 u
    int nid[NPTC], cntv[NPTC], totv[NPTC], wv[NPTC], cls[NPTC];
    unsigned alive_m = 0, hasw_m = 0;
    double g[NPTC];
    bool range_bad = false;
#pragma unroll
    for (int u = 0; u < NPTC; u++) {
        const int pos = lo + lane + 64 * u;
        nid[u] = -2; cntv[u] = 0; totv[u] = 0; wv[u] = 0; cls[u] = -1; g[u] = 0.0;
        if (pos < hi) {
            int n = q.leaf_node[pos];
            if (n >= 0) {
                nid[u] = n;
                cntv[u] = q.cnt[s * NX + n];
                int tsum = 0;
                for (int t = 0; t <= M; t++) tsum += q.cnt[t * NX + n];
                totv[u] = tsum;
                wv[u] = q.node_weight[n];
                if (q.node_has_weight[n]) hasw_m |= 1u << u;
                if (n < N && q.alive[n]) alive_m |= 1u << u;
                g[u] = chain_score(cntv[u], 0, totv[u], (hasw_m >> u) & 1, wv[u], NP, 0.0, q.booster_kind, lp_tab, ff_tab);
                cls[u] = q.leaf_cls[pos];
                if (FAST && (cntv[u] >= (1 << 15) || cntv[u] <= -(1 << 15))) range_bad = true;
            }
            const int i = lane + 64 * u;
            gL[i] = g[u]; cntL[i] = cntv[u]; totL[i] = totv[u]; nidL[i] = nid[u]; wgtL[i] = wv[u];
            flgL[i] = ((alive_m >> u) & 1) | (((hasw_m >> u) & 1) << 1);
            clsL[i] = cls[u];
        }
    }
    bool escaped = __ballot(range_bad) != 0;
    int stop_at = cbeg;                            // flat mode: first step this launch did not do
    bool stop_range = escaped;
    // stay speculation: tried again whenever the last general step turned out to be a stay
    const bool spec_ok = q.ntn_in_lds || NP == 0;
    bool try_spec = true, gmin_dirty = true;
    double gmin_s = 0.0;
    int gmin_n = INT_MAX;
    int spec_steps = 0, spec_batches = 0;          // statistics (lane 0)
    PH_DECL;

    for (int base = cbeg; base < cend && !escaped; base += 64) {
      const int nb = cend - base < 64 ? cend - base : 64;
      PH(0);
      for (int i = lane; i < nb * kCW; i += 64) recbuf[i] = q.crec[(size_t)base * kCW + i];
      __syncthreads();
      int b = 0;
      while (b < nb) {
        // ---- Speculate that the next (up to 64) steps keep their nodes.  A stay
        // changes no counter, so under that hypothesis every step sees the state as
        // it is now and lane a can check step b + a on its own: the partition's
        // nodes, scored exactly (stickiness, nodeToNodeCounts row), must beat a lower
        // bound of every other candidate -- the smallest partition-independent score
        // of the region (the terms it leaves out are >= 0 and IEEE add / divide /
        // subtract are monotone).  The verified prefix is committed; the first step
        // that is not a certain stay takes the general step below.
        if (spec_ok && try_spec) {
            if (gmin_dirty) {                      // smallest (g, node) over the region's live leaves
                double ms = pos_inf();
                int mn = INT_MAX;
#pragma unroll
                for (int u = 0; u < NPTC; u++) {
                    const bool ok = (alive_m >> u) & 1;
                    const bool take = ok && better(g[u], nid[u], ms, mn);
                    ms = take ? g[u] : ms;
                    mn = take ? nid[u] : mn;
                }
                gmin_n = wave_argmin(ms, mn);
                gmin_s = pos_inf();
#pragma unroll
                for (int u = 0; u < NPTC; u++) {
                    unsigned long long bm = __ballot(nid[u] == gmin_n);
                    if (bm) {
                        int wl = __ffsll((long long)bm) - 1;
                        gmin_s = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(g[u]), wl),
                                                  __builtin_amdgcn_readlane(__double2loint(g[u]), wl));
                    }
                }
                gmin_dirty = false;
            }
            const int a = lane;
            const int sb = b + a;
            const bool active = sb < nb;
            const int* rp = recbuf + (active ? sb : b) * kCW;
            bool fail = false;
            const double vstick = __hiloint2double(rp[3], rp[2]);
            const int vtl = rp[4];
            const int cn = rp[5];
            if (!((cn >> 24) & 1) || (cn & 0xff) != k) fail = true;      // must hold exactly k nodes
            int oi[KM], oc[KM + 1];
            int on[KM];
            double so[KM];
            oc[0] = rp[6];
#pragma unroll
            for (int j = 0; j < KM; j++) {
                on[j] = -3; so[j] = 0.0; oi[j] = 0; oc[j + 1] = -1;
                if (j < k) {
                    int li = rp[kCOwn + j];
                    if (li < 0 || li >= size) { fail = true; li = 0; }
                    oi[j] = li;
                    on[j] = nidL[li];
                    oc[j + 1] = clsL[li];
                }
            }
            // anchors top, own_0 .. own_{k-2}: their exclude classes must leave candidates,
            // and own_j must not sit in a class excluded before its slot
            {
                int cov = 0;
#pragma unroll
                for (int j = 0; j < KM; j++) {
                    if (j < k) {
                        if (oc[j] < 0 && !(q.flat && j == 0)) fail = true;
                        bool dup = false;
#pragma unroll
                        for (int e = 0; e < KM; e++) if (e < j && oc[e] == oc[j]) dup = true;
                        if (!dup && oc[j] >= 0) cov += cszL[oc[j]];
                        if (cov >= size) fail = true;
#pragma unroll
                        for (int e = 0; e < KM; e++) if (e <= j && oc[e] >= 0 && oc[e] == oc[j + 1]) fail = true;
                    }
                }
            }
            // the partition's own nodes: candidates, in list order, below the bound
#pragma unroll
            for (int j = 0; j < KM; j++) {
                if (j < k) {
                    const int li = oi[j];
                    if (!(flgL[li] & 1)) fail = true;
                    const int nt = (!FAST && NP > 0) ? ntn_l[vtl * ST + li] : 0;
                    so[j] = chain_score(cntL[li], nt, totL[li], (flgL[li] >> 1) & 1, wgtL[li], NP, vstick,
                                        q.booster_kind, lp_tab, ff_tab);
                    if (j > 0 && !better(so[j - 1], on[j - 1], so[j], on[j])) fail = true;
                    if (!better(so[j], on[j], gmin_s, gmin_n)) fail = true;
                }
            }
            // an own node also listed in a higher priority state is no candidate (the
            // record keeps such leaves under "higher"; gather refuses nodes held twice)
            // an earlier step of the batch with the same top priority node would have bumped my row
            if (NP > 0) {
                for (int e = 0; e < 64; e++) {
                    const int t2 = __builtin_amdgcn_readlane(vtl, e);
                    if (e < a && t2 == vtl) fail = true;
                }
            }
            if (!active) fail = false;
            const unsigned long long fm = __ballot(fail);
            int nok = fm ? __ffsll((long long)fm) - 1 : 64;
            if (nok > nb - b) nok = nb - b;
            if (a < nok) {
                int* op = outbuf + sb * q.OW;
                op[0] = k;
#pragma unroll
                for (int j = 0; j < KM; j++) {
                    if (j < k) {
                        op[1 + j] = on[j];
                        if (!FAST && NP > 0) ntn_l[vtl * ST + oi[j]] += 1;      // plan.go:238-245
                    }
                }
            }
            BLANCE_WAVE_SYNC();
            if (lane == 0) { spec_steps += nok; spec_batches++; }
            b += nok;
            if (b >= nb) break;
            if (nok == 64) continue;
        }
        // ---- FAST mode, runs of blank steps (a partition that holds no node in any
        // state, apart from higher priority nodes already inside the top node's exclude
        // class): nothing to match, demote or un-count, so a step is k masked minima over
        // the packed keys plus two counter bumps.  Lane a pre-scans step b + a; the run
        // is walked with everything in registers.
        if (FAST) {
            const int sb = b + lane;
            const bool active = sb < nb;
            const int* rp = recbuf + (active ? sb : b) * kCW;
            const int w0 = recbuf[b * kCW + 1];
            const int tcv = rp[6];
            const int tcsz = cszL[tcv < 0 ? 0 : tcv];       // leaves covered by the top node's exclude class
            // no own node, no lower priority node; higher priority nodes are just masked out
            const bool blank = active && (rp[5] & 0xff00ff) == 0 && rp[1] == w0 && (tcv >= 0 || q.flat);
            int hv[kChainHigh];
#pragma unroll
            for (int j = 0; j < kChainHigh; j++) hv[j] = rp[kCHigh + j];
            const bool any_high = __ballot(blank && (rp[5] & 0xff00) != 0) != 0;
            const unsigned long long nm = __ballot(!blank);
            int run = nm ? __ffsll((long long)nm) - 1 : 64;
            if (run > nb - b) run = nb - b;
            if (run > 0 && w0 > 0 && w0 < (1 << 14)) {
                unsigned key[NPTC];
                int mycsz[NPTC];
#pragma unroll
                for (int u = 0; u < NPTC; u++) {
                    key[u] = ((alive_m >> u) & 1) ? (((unsigned)(2 * cntv[u] + kKeyBias) << 13) | (unsigned)nid[u]) : kKeyNone;
                    mycsz[u] = cszL[cls[u] < 0 ? 0 : cls[u]];
                }
                const unsigned bump = (unsigned)(2 * w0) << 13;
                bool esc = false;
                int r = 0;
                // two instances of the loop: the common one carries no code for higher priority nodes
                auto walk = [&](auto with_high) {
                for (; r < run; r++) {
                    int acls = __builtin_amdgcn_readlane(tcv, r), acsz = __builtin_amdgcn_readlane(tcsz, r);
                    int ec[KM];
#pragma unroll
                    for (int j = 0; j < KM; j++) ec[j] = -2;
                    int covered = 0;
                    unsigned excl_m = 0;
                    if (decltype(with_high)::value) {   // plan.go:146-154
#pragma unroll
                        for (int j = 0; j < kChainHigh; j++) {
                            const int hj = __builtin_amdgcn_readlane(hv[j], r);
#pragma unroll
                            for (int u = 0; u < NPTC; u++) excl_m |= (hj == lane + 64 * u ? 1u : 0u) << u;
                        }
                    }
                    int chosen[KM];
#pragma unroll
                    for (int j = 0; j < KM; j++) chosen[j] = -1;
                    unsigned picked_m = 0;
#pragma unroll
                    for (int slot = 0; slot < KM; slot++) {
                        if (slot < k) {
                            bool dup = false;
#pragma unroll
                            for (int j = 0; j < KM; j++) if (j < slot && ec[j] == acls) dup = true;
                            if (acls < 0 && !(q.flat && slot == 0)) esc = true;
                            if (!dup && acls >= 0) {
                                ec[slot] = acls;
                                covered += acsz;
#pragma unroll
                                for (int u = 0; u < NPTC; u++) excl_m |= (cls[u] == acls ? 1u : 0u) << u;
                            }
                            if (covered >= size) esc = true;
                            unsigned km = kKeyNone;
#pragma unroll
                            for (int u = 0; u < NPTC; u++) {
                                const unsigned kv = ((excl_m >> u) & 1) ? kKeyNone : key[u];
                                km = kv < km ? kv : km;
                            }
                            const unsigned kb = wave_min_u32(km);
                            if (kb == kKeyNone) esc = true;
                            int wcls = -1, wcsz = 0;
#pragma unroll
                            for (int u = 0; u < NPTC; u++) {
                                const bool mine = key[u] == kb && kb != kKeyNone;
                                unsigned long long bm = __ballot(mine);
                                if (bm) {
                                    const int wl = __ffsll((long long)bm) - 1;
                                    wcls = __builtin_amdgcn_readlane(cls[u], wl);
                                    wcsz = __builtin_amdgcn_readlane(mycsz[u], wl);
                                }
                                picked_m |= (mine ? 1u : 0u) << u;
                            }
                            // a duplicate pick is impossible: the winner's own class is excluded next,
                            // unless it has none (wcls < 0 -> escape)
                            chosen[slot] = (int)(kb & 0x1fff);
                            acls = wcls;
                            acsz = wcsz;
                        }
                    }
                    if (__ballot(esc)) break;
#pragma unroll
                    for (int u = 0; u < NPTC; u++) {
                        if ((picked_m >> u) & 1) {
                            key[u] += bump;
                            cntv[u] += w0;
                            totv[u] += w0;
                            if (cntv[u] >= (1 << 15)) range_bad = true;
                        }
                    }
                    if (lane == 0) {
                        int* o = outbuf + (b + r) * q.OW;
                        o[0] = k;
#pragma unroll
                        for (int c = 0; c < KM; c++) if (c < k) o[1 + c] = chosen[c];
                    }
                    if (__ballot(range_bad)) { r++; break; }
                }
                };
                if (any_high) walk(std::true_type{}); else walk(std::false_type{});
                // refresh the mirrors and the partition-independent scores of my leaves
#pragma unroll
                for (int u = 0; u < NPTC; u++) {
                    const int i = lane + 64 * u;
                    if (i < size) {
                        g[u] = (double)cntv[u];
                        gL[i] = g[u]; cntL[i] = cntv[u]; totL[i] = totv[u];
                    }
                }
                BLANCE_WAVE_SYNC();
                gmin_dirty = true;
                if (r > 0) try_spec = false;          // the run's steps were moves
                b += r;
                if (__ballot(range_bad)) { escaped = true; stop_range = true; break; }
                if (b >= nb) break;
                if (r > 0 && !__ballot(esc)) continue;
                // an escape inside the run: let the general step decide (it escapes the same way)
            }
        }
        // ---- general step: findBestNodes (plan.go:98-248) + commit (plan.go:290-301)
        PH(1);
        const int recw = lane < kCW ? recbuf[b * kCW + lane] : 0;
#define REC(i) __builtin_amdgcn_readlane(recw, (i))
        const int w = REC(1);
        const double stick = __hiloint2double(REC(3), REC(2));
        const int tl = REC(4);
        const int cn = REC(5);
        const int n_low = (cn >> 16) & 0xff;
        bool esc = false;
        int ntnv[NPTC];
#pragma unroll
        for (int u = 0; u < NPTC; u++) {
            ntnv[u] = 0;
            if (!FAST && NP > 0) {
                if (q.ntn_in_lds) { if (lane + 64 * u < size) ntnv[u] = ntn_l[tl * ST + lane + 64 * u]; }
                else if (nid[u] >= 0 && nid[u] < N) ntnv[u] = q.ntn[(size_t)(tl < size ? nidL[tl] : NX) * N + nid[u]];
            }
        }
        PH(2);
        unsigned inh_m = 0, own_m = 0;
        if (cn & 0xff) {
#pragma unroll
            for (int j = 0; j < kChainOwn; j++) {
                const int oj = REC(kCOwn + j);
#pragma unroll
                for (int u = 0; u < NPTC; u++) own_m |= (oj == lane + 64 * u ? 1u : 0u) << u;
            }
        }
        if (cn & 0xff00) {
#pragma unroll
            for (int j = 0; j < kChainHigh; j++) {
                const int hj = REC(kCHigh + j);
#pragma unroll
                for (int u = 0; u < NPTC; u++) inh_m |= (hj == lane + 64 * u ? 1u : 0u) << u;
            }
        }
        PH(3);
        const unsigned elig_m = alive_m & ~inh_m;
        double sc[NPTC];
        unsigned key[NPTC];
        if (FAST) {
            // 2 * stickiness: 3 or an even integer; out of range -> let the sequential pass do it
            const double s2 = stick + stick;
            const int stick2 = (s2 >= 0.0 && s2 < 32768.0) ? (int)s2 : 0;
            if (!(s2 >= 0.0 && s2 < 32768.0) || (double)stick2 != s2) esc = true;
#pragma unroll
            for (int u = 0; u < NPTC; u++) {
                const int v = 2 * cntv[u] - (((own_m >> u) & 1) ? stick2 : 0) + kKeyBias;
                key[u] = ((elig_m >> u) & 1) ? (((unsigned)v << 13) | (unsigned)nid[u]) : kKeyNone;
                sc[u] = 0.0;
            }
        } else {
            unsigned need_m = own_m;
#pragma unroll
            for (int u = 0; u < NPTC; u++) { if (ntnv[u] != 0) need_m |= 1u << u; key[u] = 0; }
            if (__ballot(need_m != 0)) {
#pragma unroll
                for (int u = 0; u < NPTC; u++) {
                    double full = chain_score(cntv[u], ntnv[u], totv[u], (hasw_m >> u) & 1, wv[u], NP,
                                              ((own_m >> u) & 1) ? stick : 0.0, q.booster_kind, lp_tab, ff_tab);
                    sc[u] = ((need_m >> u) & 1) ? full : g[u];
                }
            } else {
#pragma unroll
                for (int u = 0; u < NPTC; u++) sc[u] = g[u];
            }
        }
        PH(4);
        // The rule's k picks (plan.go:177-223).  Every anchor's include set is this
        // region, so the running set is the region minus the anchors' exclude classes;
        // an empty running set (plan.go:746 would reset it), an anchor without a
        // proper class, a fallback to candidateNodes[0] or a duplicate pick escape.
        int ec[KM];
#pragma unroll
        for (int j = 0; j < KM; j++) ec[j] = -2;
        int covered = 0;
        unsigned excl_m = 0;
        int chosen[KM], chosen_l[KM];
#pragma unroll
        for (int j = 0; j < KM; j++) { chosen[j] = -1; chosen_l[j] = -1; }
        int n_out = 0;
        int acls = REC(6);                           // exclude class of the current anchor
        PH(5);
#pragma unroll
        for (int slot = 0; slot < KM; slot++) {
            if (slot < k) {
                bool dup = false;
#pragma unroll
                for (int j = 0; j < KM; j++) if (j < slot && ec[j] == acls) dup = true;
                if (acls < 0 && !(q.flat && slot == 0)) esc = true;
                if (!dup && acls >= 0) {
                    ec[slot] = acls;
                    covered += cszL[acls];
#pragma unroll
                    for (int u = 0; u < NPTC; u++) excl_m |= (cls[u] == acls ? 1u : 0u) << u;
                }
                if (covered >= size) esc = true;
                int best;
                if (FAST) {
                    unsigned km = kKeyNone;
#pragma unroll
                    for (int u = 0; u < NPTC; u++) {
                        const unsigned kv = ((excl_m >> u) & 1) ? kKeyNone : key[u];
                        km = kv < km ? kv : km;
                    }
                    PH(6);
                    const unsigned kb = wave_min_u32(km);
                    best = kb == kKeyNone ? INT_MAX : (int)(kb & 0x1fff);
                } else {
                    double bs = pos_inf();
                    int bn = INT_MAX;
#pragma unroll
                    for (int u = 0; u < NPTC; u++) {
                        const bool ok = ((elig_m & ~excl_m) >> u) & 1;
                        const bool take = ok && better(sc[u], nid[u], bs, bn);
                        bs = take ? sc[u] : bs;
                        bn = take ? nid[u] : bn;
                    }
                    PH(6);
                    best = wave_argmin(bs, bn);
                }
                PH(7);
                if (best == INT_MAX) esc = true;
                int wcls = -1, wloc = -1;
#pragma unroll
                for (int u = 0; u < NPTC; u++) {
                    unsigned long long bm = __ballot(nid[u] == best);
                    if (bm) {
                        int wl = __ffsll((long long)bm) - 1;
                        wcls = __builtin_amdgcn_readlane(cls[u], wl);
                        wloc = wl + 64 * u;
                    }
                }
#pragma unroll
                for (int c = 0; c < KM; c++) if (c < slot && chosen[c] == best) esc = true;   // duplicate pick
                chosen[slot] = best;
                chosen_l[slot] = wloc;
                n_out = slot + 1;
                acls = wcls;
                PH(8);
            }
        }
        if (__ballot(esc)) { escaped = true; break; }

        // ---- commit: the owner lane of a leaf updates it
        int dc[NPTC], dt[NPTC];
#pragma unroll
        for (int u = 0; u < NPTC; u++) { dc[u] = 0; dt[u] = 0; }
#pragma unroll
        for (int u = 0; u < NPTC; u++) {             // old nodes of this state leave it (plan.go:290-293)
            const int d = ((own_m >> u) & 1) ? w : 0;
            dc[u] -= d; dt[u] -= d;
        }
#pragma unroll
        for (int c = 0; c < KM; c++) {               // chosen nodes enter it (plan.go:299-301)
            if (c < k) {
#pragma unroll
                for (int u = 0; u < NPTC; u++) {
                    const bool mine = chosen_l[c] == lane + 64 * u;
                    dc[u] += mine ? w : 0;
                    dt[u] += mine ? w : 0;
                    if (!FAST && NP > 0 && mine) {
                        if (q.ntn_in_lds) ntn_l[tl * ST + lane + 64 * u] = ntnv[u] + 1;      // plan.go:238-245
                        else q.ntn[(size_t)(tl < size ? nidL[tl] : NX) * N + nid[u]] = ntnv[u] + 1;
                    }
                }
            }
        }
        if (n_low > 0) {                             // a chosen node leaves its lower priority state (plan.go:294-297)
            for (int e = 0; e < kChainLow; e++) {
                const int le = REC(kCLow + e), lt = REC(kCLowState + e);
                if (le < 0) continue;
#pragma unroll
                for (int c = 0; c < KM; c++) {
                    if (c < k && chosen_l[c] == le) {
#pragma unroll
                        for (int u = 0; u < NPTC; u++) dt[u] -= le == lane + 64 * u ? w : 0;
                        if (lane == 0) q.cnt[lt * NX + chosen[c]] -= w;
                    }
                }
            }
        }
        PH(9);
        {
            unsigned changed_m = 0;
#pragma unroll
            for (int u = 0; u < NPTC; u++) {
                cntv[u] += dc[u];
                totv[u] += dt[u];
                if (dc[u] | dt[u]) changed_m |= 1u << u;
                if (FAST && (cntv[u] >= (1 << 15) || cntv[u] <= -(1 << 15))) range_bad = true;
            }
            if (__ballot(changed_m != 0)) {
#pragma unroll
                for (int u = 0; u < NPTC; u++) {
                    double gn = FAST ? (double)cntv[u]       // no quotients, no weights: plan.go:664-670 only
                                     : chain_score(cntv[u], 0, totv[u], (hasw_m >> u) & 1, wv[u], NP, 0.0,
                                                   q.booster_kind, lp_tab, ff_tab);
                    g[u] = ((changed_m >> u) & 1) ? gn : g[u];
                    if ((changed_m >> u) & 1) {
                        const int i = lane + 64 * u;
                        gL[i] = g[u]; cntL[i] = cntv[u]; totL[i] = totv[u];
                    }
                }
            }
        }
        PH(10);
        {
            // did this step keep its nodes?  then the next ones probably do, too
            bool same = ((cn >> 24) & 1) && (cn & 0xff) == n_out;
            if (same) {
#pragma unroll
                for (int c = 0; c < KM; c++) if (c < n_out && REC(kCOwn + c) != chosen_l[c]) same = false;
            }
            try_spec = same;
            gmin_dirty = true;
        }
        if (lane == 0) {
            int* o = outbuf + b * q.OW;
            o[0] = n_out;
#pragma unroll
            for (int c = 0; c < KM; c++) if (c < k) o[1 + c] = chosen[c];
        }
        PH(11);
#undef REC
        BLANCE_WAVE_SYNC();
        b++;
        if (__ballot(range_bad)) { escaped = true; stop_range = true; break; }
      }
      __syncthreads();
      stop_at = base + b;
      // flat mode keeps the steps done before a stop; a region chain's pass is redone as a whole
      const int n_done = (!escaped || q.flat) ? b : 0;
      for (int i = lane; i < n_done * q.OW; i += 64) q.out[(size_t)base * q.OW + i] = outbuf[i];
    }
    PH_DUMP(cend - cbeg);
    if (lane == 0 && spec_batches) { atomicAdd(&q.flags[2], spec_steps); atomicAdd(&q.flags[3], spec_batches); }
    if (escaped) {
        if (lane == 0) { q.flags[1] = 1; q.flags[4] = stop_at; q.flags[5] = stop_range ? 1 : 0; }
        if (!q.flat) return;
        // the rest of the pass continues from global memory: hand over the LDS rows
        if (!FAST && NP > 0 && q.ntn_in_lds) {
            __syncthreads();
            for (int i = lane; i < (size + 1) * size; i += 64) {
                const int row = i / size, col = i - row * size;
                const int cn = nidL[col];
                if (cn >= 0 && cn < N) q.ntn[(size_t)(row < size ? nidL[row] : NX) * N + cn] = ntn_l[row * ST + col];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < NPTC; u++)
        if (nid[u] >= 0) q.cnt[s * NX + nid[u]] = cntv[u];
}

// ============================================================================
// Flat bulk engine: passes of a state without hierarchy rules.  Two kinds of
// steps are resolved without walking them one by one, both exactly:
//
//  * certain stays (k = 1): the partition keeps its node if even the largest
//    score it could have (nodeToNodeCounts term bounded from above) beats the
//    smallest partition-independent score of every other candidate, which is a
//    lower bound of that candidate's true score (the skipped terms are >= 0
//    and IEEE add / divide / subtract are monotone).  Stays change no counter,
//    so a whole run of them is validated by independent threads.
//  * fresh identical partitions (k = 1): partitions without nodes, exclusions
//    or own rows all see the same candidate scores; each node's score grows
//    with every pick it receives, so the greedy's picks are the R smallest
//    elements of the merged per-node score sequences, in sorted order
//    ((score, position) ascending).  A threshold search finds how many picks
//    each node gets, a stable radix sort orders them.
//
// Everything else goes through k_pass_seq on a sub-range of the pass.
// ============================================================================

__device__ __forceinline__ unsigned long long sortable_key(double v) {
    if (v == 0.0) v = 0.0;                         // -0.0 and +0.0 compare equal
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

// tot[n], g[n] and the kTopList smallest (g, n) over nodesNext.  One workgroup.
__global__ __launch_bounds__(1024) void k_flat_prepare(FlatParams q, int32_t* tot, double* g, double* top_g,
                                                       int32_t* top_n) {
    BLANCE_DYN_LDS(lds);
    RedSlot* red = (RedSlot*)lds;
    int round = 0;
    const int tid = threadIdx.x;
    for (int n = tid; n < q.NX; n += 1024) {
        int tsum = 0;
        for (int t = 0; t <= q.M; t++) tsum += q.cnt[t * q.NX + n];
        tot[n] = tsum;
        g[n] = node_score(q.cnt[q.s * q.NX + n], 0, tsum, q.node_has_weight[n], q.node_weight[n], q.NP, 0.0,
                          q.booster_kind);
    }
    __syncthreads();
    double last_s = 0.0;
    int last_n = -1;
    for (int r = 0; r < kTopList; r++) {
        double bs = pos_inf();
        int bn = INT_MAX;
        for (int n = tid; n < q.N; n += 1024) {
            if (!q.alive[n]) continue;
            double v = g[n];
            if (r > 0 && !better(last_s, last_n, v, n)) continue;     // already listed
            if (better(v, n, bs, bn)) { bs = v; bn = n; }
        }
        int best = block_argmin<1024>(bs, bn, red, round);
        if (best == INT_MAX) {
            if (tid == 0) { top_g[r] = pos_inf(); top_n[r] = INT_MAX; }
            last_s = pos_inf(); last_n = INT_MAX;
        } else {
            last_s = g[best]; last_n = best;
            if (tid == 0) { top_g[r] = last_s; top_n[r] = best; }
        }
    }
}

// steps of the pass per nodeToNodeCounts row: an upper bound of any entry of that row
__global__ void k_flat_row_count(FlatParams q, int32_t* row_count) {
    int oi = blockIdx.x * blockDim.x + threadIdx.x;
    if (oi >= q.P) oi = -1;
    const int32_t* r = q.rec + (size_t)(oi < 0 ? 0 : oi) * q.RW;
    int hdr = r[kRecHead + q.top_state * (1 + q.L)];
    int top = ((hdr >> 16) != kListAbsent && (hdr & 0xffff) > 0) ? r[kRecHead + q.top_state * (1 + q.L) + 1] : -1;
    if (oi >= 0 && top >= 0) atomicAdd(&row_count[top], 1);
    unsigned long long none = __ballot(oi >= 0 && top < 0);    // the "" row: one atomic per wave
    if (none && (int)(threadIdx.x & 63) == __ffsll((long long)none) - 1) atomicAdd(&row_count[q.NX], __popcll(none));
}

// Classify the steps [beg, end): record the first one that is not a certain
// stay and the first one that is not "fresh identical" to step beg.
__global__ void k_flat_scan(FlatParams q, int beg, int end) {
    int oi = beg + blockIdx.x * blockDim.x + threadIdx.x;
    if (oi >= end) return;
    const int SW = 1 + q.L;
    const int32_t* r = q.rec + (size_t)oi * q.RW;
    const int w = r[1];
    const double stick = __hiloint2double(r[3], r[2]);
    int hT = r[kRecHead + q.top_state * SW];
    int top = ((hT >> 16) != kListAbsent && (hT & 0xffff) > 0) ? r[kRecHead + q.top_state * SW + 1] : -1;
    int hs = r[kRecHead + q.s * SW];
    int own_len = (hs >> 16) == kListAbsent ? 0 : (hs & 0xffff);
    int all_len = 0;                                   // nodes the partition holds in any state
    for (int t = 0; t < q.M; t++) {
        int h = r[kRecHead + t * SW];
        if ((h >> 16) != kListAbsent) all_len += h & 0xffff;
    }
    // ---- fresh identical to step beg?
    {
        const int32_t* r0 = q.rec + (size_t)beg * q.RW;
        // no node anywhere: nothing to exclude, demote or promote (plan.go:290-297)
        bool fresh = q.k == 1 && all_len == 0 && w > 0 && w == r0[1] && top < 0;
        if (!fresh) atomicMin(&q.scan[1], oi);
    }
    // ---- certain stay?
    bool stay = false;
    if (q.k == 1 && own_len == 1) {
        int o = r[kRecHead + q.s * SW + 1];
        bool ok = o < q.N && q.alive[o];
        for (int t = 0; t < q.M && ok; t++) {           // o in another list of the partition: promoted / excluded
            if (t == q.s) continue;
            int h = r[kRecHead + t * SW];
            if ((h >> 16) == kListAbsent) continue;
            for (int j = 0; j < (h & 0xffff); j++) if (r[kRecHead + t * SW + 1 + j] == o) ok = false;
        }
        if (ok) {
            int ub = q.NP > 0 ? q.row_count[top < 0 ? q.NX : top] : 0;
            double s_hi = node_score(q.cnt[q.s * q.NX + o], ub, q.tot[o], q.node_has_weight[o], q.node_weight[o],
                                     q.NP, stick, q.booster_kind);
            // smallest other candidate: first listed node that is neither o nor excluded
            bool found = false;
            for (int e = 0; e < kTopList && !found; e++) {
                int n = q.top_n[e];
                if (n == INT_MAX) { found = true; stay = true; break; }        // no other candidate at all
                bool excl = n == o;
                for (int t = 0; t < q.M && !excl; t++) {
                    int h = r[kRecHead + t * SW];
                    if ((h >> 16) == kListAbsent || !((q.higher_mask >> t) & 1)) continue;
                    for (int j = 0; j < (h & 0xffff); j++) if (r[kRecHead + t * SW + 1 + j] == n) excl = true;
                }
                if (excl) continue;
                found = true;
                stay = better(s_hi, o, q.top_g[e], n);
            }
        }
    }
    if (!stay) atomicMin(&q.scan[0], oi);
}

// commit a run of certain stays: the lists do not change; nodeToNodeCounts does (plan.go:238-245)
__global__ void k_flat_commit_stay(FlatParams q, int beg, int end) {
    int oi = beg + blockIdx.x * blockDim.x + threadIdx.x;
    if (oi >= end) return;
    const int SW = 1 + q.L;
    const int32_t* r = q.rec + (size_t)oi * q.RW;
    int o = r[kRecHead + q.s * SW + 1];
    int* out = q.out + (size_t)oi * q.OW;
    out[0] = 1;
    out[1] = o;
    if (q.NP > 0) {
        int hT = r[kRecHead + q.top_state * SW];
        int top = ((hT >> 16) != kListAbsent && (hT & 0xffff) > 0) ? r[kRecHead + q.top_state * SW + 1] : -1;
        atomicAdd(&q.ntn[(size_t)(top < 0 ? q.NX : top) * q.N + o], 1);
    }
}

// ---- fresh identical run: how many of the R picks each node receives --------
__device__ __forceinline__ unsigned long long fresh_key(const FlatParams& q, int n, int cnt0, int tot0, int ntn0,
                                                        int w, int c) {
    return sortable_key(node_score(cnt0 + c * w, ntn0 + c, tot0 + c * w, q.node_has_weight[n], q.node_weight[n],
                                   q.NP, 0.0, q.booster_kind));
}

// #{c in [0, R): key(n, c) <= tau}; the keys grow with c
__device__ __forceinline__ int fresh_count_le(const FlatParams& q, int n, int cnt0, int tot0, int ntn0, int w, int R,
                                              unsigned long long tau) {
    int lo = 0, hi = R;                              // first c in [0, R] with key > tau
    while (lo < hi) {
        int mid = lo + (hi - lo) / 2;
        if (fresh_key(q, n, cnt0, tot0, ntn0, w, mid) <= tau) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(1024) void k_fresh_threshold(FlatParams q, int beg, int R, int32_t* m_out,
                                                          int32_t* m_off) {
    BLANCE_DYN_LDS(lds);
    long long* part = (long long*)lds;               // [1024] partial sums, then [1024] scan scratch
    const int tid = threadIdx.x;
    const int w = q.rec[(size_t)beg * q.RW + 1];
    unsigned long long lo = 0, hi = ~0ull;           // smallest tau with total(tau) >= R
    while (lo < hi) {
        unsigned long long mid = lo + (hi - lo) / 2;
        long long sum = 0;
        for (int n = tid; n < q.N; n += 1024) {
            if (!q.alive[n]) continue;
            int ntn0 = q.NP > 0 ? q.ntn[(size_t)q.NX * q.N + n] : 0;
            sum += fresh_count_le(q, n, q.cnt[q.s * q.NX + n], q.tot[n], ntn0, w, R, mid);
        }
        part[tid] = sum;
        __syncthreads();
        for (int off = 512; off >= 1; off >>= 1) {
            if (tid < off) part[tid] += part[tid + off];
            __syncthreads();
        }
        long long total = part[0];
        __syncthreads();
        if (total >= R) hi = mid; else lo = mid + 1;
    }
    const unsigned long long tau = lo;
    // picks strictly below tau, then the ties at tau in node order
    const int per = (q.N + 1023) / 1024;
    const int nb = tid * per, ne = nb + per < q.N ? nb + per : q.N;
    long long below = 0, ties = 0;
    for (int n = nb; n < ne; n++) {
        int lt = 0, le = 0;
        if (q.alive[n]) {
            int ntn0 = q.NP > 0 ? q.ntn[(size_t)q.NX * q.N + n] : 0;
            int c0 = q.cnt[q.s * q.NX + n], t0 = q.tot[n];
            lt = tau == 0 ? 0 : fresh_count_le(q, n, c0, t0, ntn0, w, R, tau - 1);
            le = fresh_count_le(q, n, c0, t0, ntn0, w, R, tau);
        }
        m_out[n] = lt;
        m_off[n] = le - lt;                          // ties of node n, consumed below
        below += lt;
        ties += le - lt;
    }
    // exclusive scans over threads (contiguous node slices keep node order)
    long long* sb = part;
    long long* st = part + 1024;
    sb[tid] = below; st[tid] = ties;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        long long a = tid >= off ? sb[tid - off] : 0, b = tid >= off ? st[tid - off] : 0;
        __syncthreads();
        sb[tid] += a; st[tid] += b;
        __syncthreads();
    }
    long long total_below = sb[1023];
    long long rem = (long long)R - total_below;      // ties to hand out, in node order
    long long ties_before = st[tid] - ties;
    long long elems_before = sb[tid] - below;        // picks below tau of earlier nodes
    for (int n = nb; n < ne; n++) {
        long long avail = m_off[n];
        long long left = rem - ties_before;
        long long take = left <= 0 ? 0 : (avail < left ? avail : left);
        int m = m_out[n] + (int)take;
        ties_before += avail;
        m_out[n] = m;
    }
    __syncthreads();
    // element offsets: exclusive scan of the final counts
    long long mine = 0;
    for (int n = nb; n < ne; n++) mine += m_out[n];
    sb[tid] = mine;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        long long a = tid >= off ? sb[tid - off] : 0;
        __syncthreads();
        sb[tid] += a;
        __syncthreads();
    }
    long long acc = sb[tid] - mine;
    for (int n = nb; n < ne; n++) { m_off[n] = (int)acc; acc += m_out[n]; }
    if (tid == 1023) m_off[q.N] = (int)sb[1023];
    (void)elems_before;
}

// the R picked (score, node) elements in node-major order
__global__ void k_fresh_emit(FlatParams q, int beg, int R, const int32_t* m_off, unsigned long long* keys,
                             int32_t* vals) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= R) return;
    int lo = 0, hi = q.N;                            // last n with m_off[n] <= e
    while (hi - lo > 1) {
        int mid = (lo + hi) / 2;
        if (m_off[mid] <= e) lo = mid; else hi = mid;
    }
    int n = lo, c = e - m_off[n];
    int w = q.rec[(size_t)beg * q.RW + 1];
    int ntn0 = q.NP > 0 ? q.ntn[(size_t)q.NX * q.N + n] : 0;
    keys[e] = fresh_key(q, n, q.cnt[q.s * q.NX + n], q.tot[n], ntn0, w, c);
    vals[e] = n;
}

__global__ void k_fresh_commit_steps(FlatParams q, int beg, int R, const int32_t* sorted_nodes) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= R) return;
    int* out = q.out + (size_t)(beg + j) * q.OW;
    out[0] = 1;
    out[1] = sorted_nodes[j];
}

__global__ void k_fresh_commit_nodes(FlatParams q, int beg, const int32_t* m, int32_t* cnt) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= q.N) return;
    int w = q.rec[(size_t)beg * q.RW + 1];
    if (m[n] == 0) return;
    cnt[q.s * q.NX + n] += m[n] * w;                 // plan.go:299-301
    if (q.NP > 0) q.ntn[(size_t)q.NX * q.N + n] += m[n];
}

// ---- stable LSD radix sort of (64-bit key, 32-bit value) pairs, 8 bits per pass.
// One wave64 per tile of kSortTile elements; ranks inside a wave come from ballots.
constexpr int kSortTile = 2048;

__device__ __forceinline__ unsigned long long same_digit_lanes(int digit, bool valid) {
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        unsigned long long m = __ballot((digit >> b) & 1);
        peers &= ((digit >> b) & 1) ? m : ~m;
    }
    return peers;
}

__global__ __launch_bounds__(64) void k_sort_hist(int n, int shift, const unsigned long long* keys, int n_tiles,
                                                  int32_t* hist /* [256][n_tiles] */) {
    BLANCE_DYN_LDS(lds);
    int* cnt = (int*)lds;                            // [256]
    const int lane = threadIdx.x, tile = blockIdx.x;
    for (int i = lane; i < 256; i += 64) cnt[i] = 0;
    __syncthreads();
    int beg = tile * kSortTile, end = beg + kSortTile < n ? beg + kSortTile : n;
    for (int base = beg; base < end; base += 64) {
        int i = base + lane;
        bool valid = i < end;
        int digit = valid ? (int)((keys[i] >> shift) & 0xff) : 0;
        unsigned long long peers = same_digit_lanes(digit, valid);
        if (valid && (peers & ((1ull << lane) - 1)) == 0) cnt[digit] += __popcll(peers);   // lowest peer adds
        __syncthreads();
    }
    for (int i = lane; i < 256; i += 64) hist[(size_t)i * n_tiles + tile] = cnt[i];
}

__global__ __launch_bounds__(64) void k_sort_scatter(int n, int shift, const unsigned long long* keys_in,
                                                     const int32_t* vals_in, unsigned long long* keys_out,
                                                     int32_t* vals_out, int n_tiles, const int32_t* offsets) {
    BLANCE_DYN_LDS(lds);
    int* pos = (int*)lds;                            // [256] next output slot per digit
    const int lane = threadIdx.x, tile = blockIdx.x;
    for (int i = lane; i < 256; i += 64) pos[i] = offsets[(size_t)i * n_tiles + tile];
    __syncthreads();
    int beg = tile * kSortTile, end = beg + kSortTile < n ? beg + kSortTile : n;
    for (int base = beg; base < end; base += 64) {
        int i = base + lane;
        bool valid = i < end;
        unsigned long long key = valid ? keys_in[i] : 0;
        int digit = valid ? (int)((key >> shift) & 0xff) : 0;
        unsigned long long peers = same_digit_lanes(digit, valid);
        unsigned long long lower = peers & ((1ull << lane) - 1);
        int dst = 0;
        if (valid) dst = pos[digit] + __popcll(lower);
        __syncthreads();
        if (valid && lower == 0) pos[digit] += __popcll(peers);
        __syncthreads();
        if (valid) { keys_out[dst] = key; vals_out[dst] = vals_in[i]; }
    }
}

// ============================================================================
// Data-parallel kernels around the pass
// ============================================================================

struct DevProblem {   // device pointers + sizes shared by the elementwise kernels
    int32_t N, NX, M, L, P;
    int32_t weights_nil;
    const uint8_t* node_removed;   // view of this sweep (all zero after sweep 1)
    const uint8_t* node_added;
    const int32_t* part_weight;
    const uint8_t* part_has_weight;
    int32_t* live; int32_t* live_len; uint8_t* live_kind;
    int32_t* prv;  int32_t* prv_len;  uint8_t* prv_kind;
    uint8_t* in_prev; uint8_t* never_equal;
};

// nextPartitions = copy of partitionsToAssign minus nodesToRemove (plan.go:83-88)
__global__ void k_live_init(DevProblem d, const int32_t* a_off, const int32_t* a_nodes,
                            const uint8_t* a_kind, const int32_t* p_off, const int32_t* p_nodes,
                            const uint8_t* p_kind) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= d.P * d.M) return;
    int len = 0;
    for (int i = a_off[idx]; i < a_off[idx + 1]; i++) {
        int n = a_nodes[i];
        if (!d.node_removed[n]) d.live[(size_t)idx * d.L + len++] = n;
    }
    d.live_len[idx] = len;
    d.live_kind[idx] = a_kind[idx] == kListAbsent ? kListAbsent : kListSet;
    len = 0;
    for (int i = p_off[idx]; i < p_off[idx + 1]; i++) d.prv[(size_t)idx * d.L + len++] = p_nodes[i];
    d.prv_len[idx] = len;
    d.prv_kind[idx] = p_kind[idx];
}

// sweeps >= 2: every present key is a non-nil slice again (plan.go:418)
__global__ void k_live_refresh(DevProblem d) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= d.P * d.M) return;
    if (d.live_kind[idx] != kListAbsent) d.live_kind[idx] = kListSet;
}

// countStateNodes (plan.go:374-399): extra loads ...
__global__ void k_count_loads(int n_loads, int NX, int later_sweep, const int32_t* st, const int32_t* nd,
                              const int32_t* wt, const uint8_t* first_only, int32_t* cnt) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_loads) return;
    if (later_sweep && first_only[i]) return;
    atomicAdd(&cnt[st[i] * NX + nd[i]], wt[i]);
}

// ... and the prevMap view of the partitions being assigned
__global__ void k_count_prev(DevProblem d, int32_t* cnt) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= d.P * d.M) return;
    int p = idx / d.M, m = idx % d.M;
    if (!d.in_prev[p]) return;
    int w = (!d.weights_nil && d.part_has_weight[p]) ? d.part_weight[p] : 1;
    for (int i = 0; i < d.prv_len[idx]; i++) atomicAdd(&cnt[m * d.NX + d.prv[(size_t)idx * d.L + i]], w);
}

// partitionSorter category (plan.go:542-561)
__global__ void k_category(DevProblem d, int m, int any_removed, int add_nil, uint8_t* cat) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= d.P) return;
    int cv = 2;
    bool is0 = false;
    if (any_removed && d.in_prev[p]) {
        int idx = p * d.M + m;
        if (d.prv_kind[idx] == kListSet)
            for (int i = 0; i < d.prv_len[idx]; i++)
                if (d.node_removed[d.prv[(size_t)idx * d.L + i]]) { is0 = true; break; }
    }
    if (is0) cv = 0;
    else if (!add_nil) {
        bool hit = false;
        for (int t = 0; t < d.M && !hit; t++) {
            int idx = p * d.M + t;
            if (d.live_kind[idx] == kListAbsent) continue;
            for (int i = 0; i < d.live_len[idx]; i++)
                if (d.node_added[d.live[(size_t)idx * d.L + i]]) { hit = true; break; }
        }
        if (!hit) cv = 1;
    }
    cat[p] = (uint8_t)cv;
}

// Stable partition of a sequence by a small key (the per-pass category of
// partitionSorter, plan.go:519-562; the region of a step): per-chunk bucket
// counts -> exclusive scan (bucket major) -> stable scatter.  One wave64 per
// chunk of kPartChunk elements; ranks inside a round of 64 come from ballots.
constexpr int kPartChunk = 1024;

__device__ __forceinline__ int part_key(const int32_t* key32, const uint8_t* key8, const int32_t* index, int i) {
    int j = index ? index[i] : i;
    return key8 ? (int)key8[j] : key32[j];
}

__global__ __launch_bounds__(64) void k_part_count(int n, const int32_t* key32, const uint8_t* key8,
                                                   const int32_t* index, int n_chunks, int B,
                                                   int32_t* counts /* [B][n_chunks] */) {
    BLANCE_DYN_LDS(lds);
    int* hist = (int*)lds;                           // [B]
    const int lane = threadIdx.x, chunk = blockIdx.x;
    for (int i = lane; i < B; i += 64) hist[i] = 0;
    __syncthreads();
    int beg = chunk * kPartChunk, end = beg + kPartChunk < n ? beg + kPartChunk : n;
    for (int base = beg; base < end; base += 64) {
        int i = base + lane;
        if (i < end) atomicAdd(&hist[part_key(key32, key8, index, i)], 1);
    }
    __syncthreads();
    for (int i = lane; i < B; i += 64) counts[(size_t)i * n_chunks + chunk] = hist[i];
}

__global__ __launch_bounds__(64) void k_part_scatter(int n, const int32_t* key32, const uint8_t* key8,
                                                     const int32_t* index, const int32_t* values, int n_chunks,
                                                     int B, int nbits, const int32_t* offsets, int32_t* out) {
    BLANCE_DYN_LDS(lds);
    int* pos = (int*)lds;                            // [B] next output slot per bucket
    const int lane = threadIdx.x, chunk = blockIdx.x;
    for (int i = lane; i < B; i += 64) pos[i] = offsets[(size_t)i * n_chunks + chunk];
    __syncthreads();
    int beg = chunk * kPartChunk, end = beg + kPartChunk < n ? beg + kPartChunk : n;
    for (int base = beg; base < end; base += 64) {
        int i = base + lane;
        bool valid = i < end;
        int key = valid ? part_key(key32, key8, index, i) : 0;
        unsigned long long peers = __ballot(valid);
        for (int bit = 0; bit < nbits; bit++) {
            unsigned long long m = __ballot((key >> bit) & 1);
            peers &= ((key >> bit) & 1) ? m : ~m;
        }
        unsigned long long lower = peers & ((1ull << lane) - 1);
        int dst = valid ? pos[key] + __popcll(lower) : 0;
        __syncthreads();
        if (valid && lower == 0) pos[key] += __popcll(peers);
        __syncthreads();
        if (valid) out[dst] = values[i];
    }
}

// Region of every step of the pass, or flags[0] if some step is not region-local:
// its top priority node and the nodes it currently holds in this state must sit
// in one region (their counters are then owned by that region's chain).
__global__ void k_chain_classify(DevProblem d, int m, int top_state, const int32_t* order,
                                 const int32_t* node_region, int32_t* regid, int32_t* flags) {
    int oi = blockIdx.x * blockDim.x + threadIdx.x;
    if (oi >= d.P) return;
    int p = order[oi];
    int idxT = p * d.M + top_state;
    int top = (d.live_kind[idxT] != kListAbsent && d.live_len[idxT] > 0) ? d.live[(size_t)idxT * d.L] : -1;
    int rg = top >= 0 ? node_region[top] : -1;
    if (rg >= 0) {
        int idx = p * d.M + m;
        if (d.live_kind[idx] != kListAbsent)
            for (int i = 0; i < d.live_len[idx]; i++)
                if (node_region[d.live[(size_t)idx * d.L + i]] != rg) rg = -1;
    }
    if (rg < 0) { flags[0] = 1; rg = 0; }
    regid[oi] = rg;
}

// Compact chain records (layout: blance_kernels.h): the step's nodes as leaf
// indices local to its region.  Steps the chain kernel cannot represent raise flags[0].
__global__ void k_gather_chain(DevProblem d, int m, int top_state, int higher_mask, const int32_t* chain_order,
                               const int32_t* state_stickiness, const uint8_t* state_has_stickiness,
                               const int32_t* node_leaf_pos, const int32_t* node_region, const int32_t* reg_lo,
                               const int32_t* leaf_cls, int flat, int32_t* crec, int32_t* flags) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.P) return;
    int p = chain_order[i];
    int32_t* r = crec + (size_t)i * kCW;
    for (int j = 0; j < kCW; j++) r[j] = -1;
    int w = 1;
    double stick = 1.5;
    if (!d.weights_nil) {
        if (d.part_has_weight[p]) { w = d.part_weight[p]; stick = (double)w; }
        else if (state_has_stickiness[m]) stick = (double)state_stickiness[m];
    }
    r[0] = p; r[1] = w; r[2] = __double2loint(stick); r[3] = __double2hiint(stick);
    int idxT = p * d.M + top_state;
    int top = (d.live_kind[idxT] != kListAbsent && d.live_len[idxT] > 0) ? d.live[(size_t)idxT * d.L] : -1;
    int rg = flat ? 0 : (top >= 0 ? node_region[top] : -1);
    if (rg < 0) { flags[0] = 1; r[4] = 0; r[5] = 0; r[6] = -1; return; }
    const int lo = reg_lo[rg];
    if (flat) {
        r[4] = top >= 0 ? top : d.NX;              // the "" row when there is no top priority node
        r[6] = -1;                                 // no anchor: nothing is excluded in the first slot
    } else {
        r[4] = node_leaf_pos[top] - lo;
        r[6] = leaf_cls[node_leaf_pos[top]];
    }
    bool bad = false;
    int own_nodes[kChainOwn];
    int n_own = 0, n_h = 0, n_low = 0, present = 0;
    int idx = p * d.M + m;
    if (d.live_kind[idx] != kListAbsent) {
        present = 1;
        for (int j = 0; j < d.live_len[idx]; j++) {
            int x = d.live[(size_t)idx * d.L + j];
            if (n_own >= kChainOwn || node_region[x] != rg) { bad = true; break; }
            own_nodes[n_own] = x;
            r[kCOwn + n_own++] = node_leaf_pos[x] - lo;
        }
    }
    for (int t = 0; t < d.M && !bad; t++) {
        if (t == m) continue;
        int ix = p * d.M + t;
        if (d.live_kind[ix] == kListAbsent) continue;
        const bool higher = (higher_mask >> t) & 1;
        for (int j = 0; j < d.live_len[ix]; j++) {
            int x = d.live[(size_t)ix * d.L + j];
            for (int e = 0; e < n_own; e++) if (own_nodes[e] == x) bad = true;   // a node held in two states
            if (node_region[x] != rg) continue;      // never a candidate of this region's chain
            int loc = node_leaf_pos[x] - lo;
            if (higher) {
                // the top priority node's exclude class is excluded in every slot anyway
                if (r[6] >= 0 && leaf_cls[node_leaf_pos[x]] == r[6]) continue;
                if (n_h >= kChainHigh) { bad = true; break; }
                r[kCHigh + n_h++] = loc;
            } else {
                if (n_low >= kChainLow) { bad = true; break; }
                r[kCLow + n_low] = loc;
                r[kCLowState + n_low++] = t;
            }
        }
    }
    r[5] = n_own | (n_h << 8) | (n_low << 16) | (present << 24);
    if (bad) flags[0] = 1;
}

// exclusive scan of n ints by one workgroup of 1024 threads: tiles of 8192
// elements, 8 contiguous per thread (coalesced), carry across tiles
__global__ __launch_bounds__(1024) void k_scan_excl(int n, int32_t* data) {
    BLANCE_DYN_LDS(lds);
    int* wsum = (int*)lds;                       // [16] wave totals, [16] carry
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int carry = 0;
    for (int base = 0; base < n; base += 8192) {
        int v[8];
        int sum = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            int i = base + tid * 8 + j;
            v[j] = i < n ? data[i] : 0;
            sum += v[j];
        }
        int incl = sum;                          // inclusive scan of the thread sums inside the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            int t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int wbase = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) { int x = wsum[w]; if (w < wave) wbase += x; total += x; }
        int acc = carry + wbase + incl - sum;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            int i = base + tid * 8 + j;
            if (i < n) data[i] = acc;
            acc += v[j];
        }
        carry += total;
        __syncthreads();
    }
}

__global__ void k_region_offsets(int B, int n_chunks, int P, const int32_t* offsets, int32_t* reg_off) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > B) return;
    reg_off[b] = b == B ? P : offsets[(size_t)b * n_chunks];
}

// Step records in pass order: what findBestNodes needs to know about its partition.
__global__ void k_gather(DevProblem d, int m, int top_state, int RW, const int32_t* order,
                         const int32_t* state_stickiness, const uint8_t* state_has_stickiness,
                         int32_t* rec) {
    int oi = blockIdx.x * blockDim.x + threadIdx.x;
    if (oi >= d.P) return;
    int p = order[oi];
    int32_t* r = rec + (size_t)oi * RW;
    int w = 1;                                         // plan.go:269-275
    double stick = 1.5;                                // plan.go:104-115
    if (!d.weights_nil) {
        if (d.part_has_weight[p]) { w = d.part_weight[p]; stick = (double)w; }
        else if (state_has_stickiness[m]) stick = (double)state_stickiness[m];
    }
    r[0] = p; r[1] = w;
    r[2] = __double2loint(stick); r[3] = __double2hiint(stick);
    for (int t = 0; t < d.M; t++) {
        int idx = p * d.M + t;
        int32_t* rs = r + kRecHead + t * (1 + d.L);
        int len = d.live_kind[idx] == kListAbsent ? 0 : d.live_len[idx];
        rs[0] = len | ((int)d.live_kind[idx] << 16);
        for (int i = 0; i < d.L; i++) rs[1 + i] = i < len ? d.live[(size_t)idx * d.L + i] : -1;
    }
}

// Apply the pass's choices to the live lists (plan.go:290-299); list edits only
// touch the step's own partition, so this runs in parallel after the pass.
__global__ void k_scatter(DevProblem d, int m, int RW, int OW, const int32_t* order, const int32_t* rec,
                          const int32_t* out) {
    int oi = blockIdx.x * blockDim.x + threadIdx.x;
    if (oi >= d.P) return;
    int p = order[oi];
    const int32_t* r = rec + (size_t)oi * RW;
    const int32_t* o = out + (size_t)oi * OW;
    int n_out = o[0] & 0xffff, is_nil = o[0] >> 16;
    const int32_t* old_s = r + kRecHead + m * (1 + d.L);
    int n_old = (old_s[0] >> 16) == kListAbsent ? 0 : (old_s[0] & 0xffff);
    for (int t = 0; t < d.M; t++) {
        int idx = p * d.M + t;
        if (t == m) continue;
        if (d.live_kind[idx] == kListAbsent) continue;
        int len = d.live_len[idx], w = 0;
        int32_t* lst = d.live + (size_t)idx * d.L;
        for (int i = 0; i < len; i++) {
            int x = lst[i];
            bool rm = false;
            for (int j = 0; j < n_old; j++) rm |= old_s[1 + j] == x;
            for (int j = 0; j < n_out; j++) rm |= o[1 + j] == x;
            if (!rm) lst[w++] = x;
        }
        d.live_len[idx] = w;
        d.live_kind[idx] = kListSet;
    }
    int idx = p * d.M + m;
    for (int j = 0; j < n_out; j++) d.live[(size_t)idx * d.L + j] = o[1 + j];
    d.live_len[idx] = n_out;
    d.live_kind[idx] = is_nil ? kListNil : kListSet;
}

// Convergence test (plan.go:36-45) fused with the write-back prevMap[name] =
// partitionsToAssign[name] = nextMap[name] (plan.go:49-52).
__global__ void k_converge(DevProblem d, int32_t* not_match) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= d.P) return;
    bool diff = !d.in_prev[p] || d.never_equal[p];
    for (int m = 0; m < d.M; m++) {
        int idx = p * d.M + m;
        int len = d.live_len[idx];
        if (d.live_kind[idx] != d.prv_kind[idx] || len != d.prv_len[idx]) diff = true;
        for (int i = 0; i < len; i++) {
            int x = d.live[(size_t)idx * d.L + i];
            if (!diff && d.prv[(size_t)idx * d.L + i] != x) diff = true;
            d.prv[(size_t)idx * d.L + i] = x;
        }
        d.prv_len[idx] = len;
        d.prv_kind[idx] = d.live_kind[idx];
    }
    d.in_prev[p] = 1;
    d.never_equal[p] = 0;
    if (diff) atomicOr(not_match, 1);
}

}  // namespace blance

// ============================================================================
// Host side: context, upload, the sweep driver, download
// ============================================================================
using namespace blance;

static thread_local std::string g_last_error;

static int fail(int code, const char* fmt, const char* a = "", long b = 0) {
    char buf[512];
    snprintf(buf, sizeof buf, fmt, a, b);
    g_last_error = buf;
    return code;
}

#define HIPTRY(expr)                                                                  \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess)                                                         \
            return fail(BLANCE_ERR_DEVICE, "%s failed: line %ld", hipGetErrorString(e_), __LINE__); \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap && p) return 0;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes < 256 ? 256 : bytes;
        if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return -1; }
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return (T*)p; }
};

struct blance_ctx {
    int device = 0;
    int engine = BLANCE_ENGINE_AUTO;
    int force_threads = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::mutex mu;
    bool uploaded = false;
    bool planned = false;

    // host copy of the small parts of the problem
    blance_problem h{};
    std::vector<int32_t> state_priority, state_constraints, rule_off;
    int L = 1, np_later = 0, n_alive = 0, any_removed = 0;
    int chain_min_parts = 2048;
    int any_node_weight = 0;
    bool no_fast_keys = false;      // a chain left the packed keys' range during this pass
    struct RuleRegions {           // regions the rule cuts the leaves into (chains), if it does
        bool ok = false;
        int n_regions = 0, max_size = 0;
        DevBuf node_region, reg_lo, reg_hi, leaf_cls, cls_size;
    };
    std::vector<RuleRegions> rule_regions;
    DevBuf leaf_node, regid, chain_order, bucket_counts, reg_off, cnt_save, crec;
    DevBuf fl_iota, fl_zero, fl_one, fl_reglo, fl_reghi;   // the whole cluster as one region (flat single chain)
    bool flat_chain_ok = false;
    DevBuf f_tot, f_g, f_top_g, f_top_n, f_row_count, f_m, f_moff, f_keys_a, f_keys_b, f_vals_a, f_vals_b, f_hist;
    int64_t steps_batched = 0;
    int64_t out_capacity = 0;

    // device: problem
    DevBuf node_removed, node_added, node_weight, node_has_weight, alive, zeros_nx, node_leaf_pos;
    DevBuf part_order, part_weight, part_has_weight, part_in_prev, part_never_equal;
    DevBuf a_off, a_nodes, a_kind, p_off, p_nodes, p_kind;
    DevBuf load_state, load_node, load_weight, load_first;
    DevBuf rule_inc, rule_exc, vparent, vlo, vhi, anchors;
    DevBuf state_stick, state_has_stick;
    // device: working state
    DevBuf live, live_len, live_kind, prv, prv_len, prv_kind, in_prev, never_equal;
    DevBuf cnt, ntn, cat, order, chunk_counts, rec, out, warn_part, warn_state, scalars;
    // scalars: [0] warn_count, [1] not_match, [2] err
    int32_t iterations = 0, converged = 0;
    int64_t n_warnings = 0, steps_total = 0, kernel_launches = 0, pass_launches = 0;
    double device_ms = 0.0, pass_ms = 0.0;
    std::vector<hipEvent_t> pass_events;     // begin/end pairs around every pass kernel
    std::vector<int> pass_kind;              // 0 = one pass kernel, 1 = flat bulk driver
    double flat_ms = 0.0;
    int64_t flat_passes = 0;

    void free_all() {
        DevBuf* all[] = {&node_removed, &node_added, &node_weight, &node_has_weight, &alive, &zeros_nx,
                         &node_leaf_pos, &part_order, &part_weight, &part_has_weight, &part_in_prev,
                         &part_never_equal, &a_off, &a_nodes, &a_kind, &p_off, &p_nodes, &p_kind,
                         &load_state, &load_node, &load_weight, &load_first, &rule_inc, &rule_exc,
                         &vparent, &vlo, &vhi, &anchors, &state_stick, &state_has_stick, &live,
                         &live_len, &live_kind, &prv, &prv_len, &prv_kind, &in_prev, &never_equal,
                         &cnt, &ntn, &cat, &order, &chunk_counts, &rec, &out, &warn_part,
                         &warn_state, &scalars};
        for (DevBuf* b : all) b->release();
        for (auto& rr : rule_regions) { rr.node_region.release(); rr.reg_lo.release(); rr.reg_hi.release(); rr.leaf_cls.release(); rr.cls_size.release(); }
        rule_regions.clear();
        DevBuf* more[] = {&leaf_node, &regid, &chain_order, &bucket_counts, &reg_off, &cnt_save, &crec, &fl_iota, &fl_zero,
                          &fl_one, &fl_reglo, &fl_reghi, &f_tot, &f_g,
                          &f_top_g, &f_top_n, &f_row_count, &f_m, &f_moff, &f_keys_a, &f_keys_b, &f_vals_a,
                          &f_vals_b, &f_hist};
        for (DevBuf* b : more) b->release();
    }
};

extern "C" int blance_abi_version(void) { return BLANCE_ABI_VERSION; }
extern "C" const char* blance_last_error(void) { return g_last_error.c_str(); }

extern "C" int64_t blance_result_capacity(const blance_problem* pb) {
    if (!pb || !pb->assign_off || !pb->state_constraints) return 0;
    int64_t cap = 0;
    const int64_t PM = (int64_t)pb->n_parts * pb->n_states;
    for (int64_t idx = 0; idx < PM; idx++) {
        int len = pb->assign_off[idx + 1] - pb->assign_off[idx];
        int k = pb->state_constraints[idx % pb->n_states];
        cap += len > k ? len : k;
    }
    return cap;
}

extern "C" int blance_validate(const blance_problem* pb) {
    if (!pb) return fail(BLANCE_ERR_BAD_ARG, "null problem");
    const int N = pb->n_nodes, NX = pb->n_nodes_ext, M = pb->n_states, P = pb->n_parts;
    if (N < 0 || NX < N || M < 0 || P < 0 || pb->n_prev < 0 || pb->n_loads < 0 || pb->n_rules < 0 ||
        pb->max_iterations < 0)
        return fail(BLANCE_ERR_BAD_ARG, "negative or inconsistent sizes");
    if ((int64_t)P * (M > 0 ? M : 1) > (int64_t)INT32_MAX / 4) return fail(BLANCE_ERR_UNSUPPORTED, "P*M too large");
    if (M > kMaxStates) return fail(BLANCE_ERR_UNSUPPORTED, "more than 16 model states");
    if (M > 0 && (pb->top_state < 0 || pb->top_state >= M)) return fail(BLANCE_ERR_BAD_ARG, "top_state out of range");
    const void* need[] = {pb->state_priority, pb->state_constraints, pb->state_stickiness, pb->state_has_stickiness,
                          pb->node_removed, pb->node_added, pb->node_weight, pb->node_has_weight, pb->part_order,
                          pb->part_weight, pb->part_has_weight, pb->part_in_prev, pb->part_prev_never_equal,
                          pb->assign_off, pb->assign_nodes, pb->assign_kind, pb->prev_off, pb->prev_nodes,
                          pb->prev_kind, pb->load_state, pb->load_node, pb->load_weight, pb->load_first_sweep_only,
                          pb->rule_off, pb->rule_inc, pb->rule_exc, pb->node_leaf_pos};
    for (const void* q : need) if (!q) return fail(BLANCE_ERR_BAD_ARG, "null array pointer");
    const int64_t PM = (int64_t)P * M;
    if (pb->assign_off[0] != 0 || pb->prev_off[0] != 0) return fail(BLANCE_ERR_BAD_ARG, "CSR offsets must start at 0");
    for (int64_t i = 0; i < PM; i++) {
        if (pb->assign_off[i + 1] < pb->assign_off[i] || pb->prev_off[i + 1] < pb->prev_off[i])
            return fail(BLANCE_ERR_BAD_ARG, "CSR offsets not monotone");
        if (pb->assign_kind[i] > BLANCE_LIST_SET || pb->prev_kind[i] > BLANCE_LIST_SET)
            return fail(BLANCE_ERR_BAD_ARG, "bad list kind");
        if (pb->assign_off[i + 1] - pb->assign_off[i] > 0xffff || pb->prev_off[i + 1] - pb->prev_off[i] > 0xffff)
            return fail(BLANCE_ERR_UNSUPPORTED, "state list longer than 65535");
    }
    for (int64_t i = 0; i < pb->assign_off[PM]; i++)
        if (pb->assign_nodes[i] < 0 || pb->assign_nodes[i] >= NX) return fail(BLANCE_ERR_BAD_ARG, "assign node id out of range");
    for (int64_t i = 0; i < pb->prev_off[PM]; i++)
        if (pb->prev_nodes[i] < 0 || pb->prev_nodes[i] >= NX) return fail(BLANCE_ERR_BAD_ARG, "prev node id out of range");
    {
        std::vector<uint8_t> seen((size_t)P, 0);
        for (int i = 0; i < P; i++) {
            int p = pb->part_order[i];
            if (p < 0 || p >= P || seen[p]) return fail(BLANCE_ERR_BAD_ARG, "part_order is not a permutation");
            seen[p] = 1;
        }
    }
    for (int i = 0; i < pb->n_loads; i++)
        if (pb->load_state[i] < 0 || pb->load_state[i] > M || pb->load_node[i] < 0 || pb->load_node[i] >= NX)
            return fail(BLANCE_ERR_BAD_ARG, "load entry out of range");
    for (int m = 0; m < M; m++) {
        int k = pb->state_constraints[m];
        if (k > kMaxK) return fail(BLANCE_ERR_UNSUPPORTED, "constraints > 8 for a state");
        if (pb->rule_off[m + 1] < pb->rule_off[m]) return fail(BLANCE_ERR_BAD_ARG, "rule_off not monotone");
        if (!pb->hierarchy_rules_nil && k > 0 && (pb->rule_off[m + 1] - pb->rule_off[m]) * k > kMaxAnchors - 1)
            return fail(BLANCE_ERR_UNSUPPORTED, "more than 8 hierarchy picks per partition and state");
    }
    if (M > 0 && pb->rule_off[M] > pb->n_rules) return fail(BLANCE_ERR_BAD_ARG, "rule_off exceeds n_rules");
    if (!pb->hierarchy_rules_nil) {
        const int VX = pb->n_vertices;
        if (VX <= NX || !pb->vertex_parent || !pb->vertex_leaf_lo || !pb->vertex_leaf_hi)
            return fail(BLANCE_ERR_BAD_ARG, "hierarchy arrays missing");
        if (pb->vertex_empty < 0 || pb->vertex_empty >= VX) return fail(BLANCE_ERR_BAD_ARG, "vertex_empty out of range");
        for (int v = 0; v < VX; v++) {
            if (pb->vertex_parent[v] < 0 || pb->vertex_parent[v] >= VX) return fail(BLANCE_ERR_BAD_ARG, "vertex_parent out of range");
            if (pb->vertex_leaf_lo[v] < 0 || pb->vertex_leaf_hi[v] <= pb->vertex_leaf_lo[v])
                return fail(BLANCE_ERR_BAD_ARG, "vertex leaf interval empty");
        }
        for (int r = 0; r < pb->n_rules; r++)
            if (pb->rule_inc[r] < 0 || pb->rule_exc[r] < 0 || pb->rule_inc[r] > 64 || pb->rule_exc[r] > 64)
                return fail(BLANCE_ERR_UNSUPPORTED, "hierarchy rule level outside 0..64");
    }
    if (pb->booster_kind != BLANCE_BOOSTER_NONE && pb->booster_kind != BLANCE_BOOSTER_CBGT)
        return fail(BLANCE_ERR_UNSUPPORTED, "unknown booster kind");
    if (NX > 1024 * 8) return fail(BLANCE_ERR_UNSUPPORTED, "more than 8192 node names (register-resident tables)");
    int L = 1;
    for (int m = 0; m < M; m++) if (pb->state_constraints[m] > L) L = pb->state_constraints[m];
    for (int64_t i = 0; i < PM; i++) {
        int a = pb->assign_off[i + 1] - pb->assign_off[i], b = pb->prev_off[i + 1] - pb->prev_off[i];
        if (a > L) L = a;
        if (b > L) L = b;
    }
    if (kRecHead + M * (1 + L) > 64) return fail(BLANCE_ERR_UNSUPPORTED, "step record wider than 64 words (states x list length)");
    if ((int64_t)(NX + 1) * (N > 0 ? N : 1) * 4 > (int64_t)64 << 30) return fail(BLANCE_ERR_UNSUPPORTED, "nodeToNodeCounts matrix > 64 GiB");
    return BLANCE_OK;
}

extern "C" int blance_ctx_create(const blance_options* opt, blance_ctx** out) {
    if (!out) return fail(BLANCE_ERR_BAD_ARG, "null out");
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(BLANCE_ERR_NO_DEVICE, "no HIP device visible");
    int dev = opt ? opt->device_id : 0;
    if (dev < 0 || dev >= n_dev) return fail(BLANCE_ERR_BAD_ARG, "device_id out of range");
    HIPTRY(hipSetDevice(dev));
    blance_ctx* c = new blance_ctx();
    c->device = dev;
    c->engine = opt ? opt->engine : BLANCE_ENGINE_AUTO;
    c->force_threads = opt ? opt->reserved[0] : 0;
    if (opt && opt->reserved[1] > 0) c->chain_min_parts = opt->reserved[1];
    if (hipStreamCreate(&c->stream) != hipSuccess || hipEventCreate(&c->ev0) != hipSuccess ||
        hipEventCreate(&c->ev1) != hipSuccess) {
        delete c;
        return fail(BLANCE_ERR_DEVICE, "stream/event creation failed");
    }
    *out = c;
    return BLANCE_OK;
}

extern "C" void blance_ctx_destroy(blance_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->free_all();
    for (hipEvent_t e : c->pass_events) (void)hipEventDestroy(e);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

template <class T>
static int put(blance_ctx* c, DevBuf& b, const T* src, size_t n) {
    if (b.reserve(n * sizeof(T))) return fail(BLANCE_ERR_DEVICE, "hipMalloc failed");
    if (n) HIPTRY(hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return 0;
}
#define PUT(buf, src, n) do { int e__ = put(c, c->buf, src, (size_t)(n)); if (e__) return e__; } while (0)
#define RESERVE(buf, bytes) do { if (c->buf.reserve((size_t)(bytes))) return fail(BLANCE_ERR_DEVICE, "hipMalloc failed"); } while (0)

static inline int cdiv(int64_t a, int b) { return (int)((a + b - 1) / b); }

static int upload_locked(blance_ctx* c, const blance_problem* pb) {
    int st = blance_validate(pb);
    if (st) return st;
    HIPTRY(hipSetDevice(c->device));
    c->uploaded = false;
    c->planned = false;
    c->h = *pb;
    const int N = pb->n_nodes, NX = pb->n_nodes_ext, M = pb->n_states, P = pb->n_parts;
    const int64_t PM = (int64_t)P * M;
    c->state_priority.assign(pb->state_priority, pb->state_priority + M);
    c->state_constraints.assign(pb->state_constraints, pb->state_constraints + M);
    c->rule_off.assign(pb->rule_off, pb->rule_off + M + 1);
    int L = 1;
    for (int m = 0; m < M; m++) if (pb->state_constraints[m] > L) L = pb->state_constraints[m];
    for (int64_t i = 0; i < PM; i++) {
        int a = pb->assign_off[i + 1] - pb->assign_off[i], b = pb->prev_off[i + 1] - pb->prev_off[i];
        if (a > L) L = a;
        if (b > L) L = b;
    }
    c->L = L;
    int fresh = 0;
    for (int p = 0; p < P; p++) if (!pb->part_in_prev[p]) fresh++;
    c->np_later = pb->n_prev + fresh;                      // plan.go:50
    std::vector<uint8_t> alive((size_t)NX + 1, 0);
    c->n_alive = 0;
    c->any_removed = 0;
    for (int n = 0; n < NX; n++) {
        if (pb->node_removed[n]) c->any_removed = 1;
        if (n < N && !pb->node_removed[n]) { alive[n] = 1; c->n_alive++; }
    }
    c->out_capacity = blance_result_capacity(pb);

    PUT(node_removed, pb->node_removed, NX);
    PUT(node_added, pb->node_added, NX);
    PUT(node_weight, pb->node_weight, NX);
    PUT(node_has_weight, pb->node_has_weight, NX);
    PUT(alive, alive.data(), NX);
    PUT(node_leaf_pos, pb->node_leaf_pos, NX);
    RESERVE(zeros_nx, NX + 1);
    HIPTRY(hipMemsetAsync(c->zeros_nx.p, 0, (size_t)NX + 1, c->stream));
    PUT(part_order, pb->part_order, P);
    PUT(part_weight, pb->part_weight, P);
    PUT(part_has_weight, pb->part_has_weight, P);
    PUT(part_in_prev, pb->part_in_prev, P);
    PUT(part_never_equal, pb->part_prev_never_equal, P);
    PUT(a_off, pb->assign_off, PM + 1);
    PUT(a_nodes, pb->assign_nodes, pb->assign_off[PM]);
    PUT(a_kind, pb->assign_kind, PM);
    PUT(p_off, pb->prev_off, PM + 1);
    PUT(p_nodes, pb->prev_nodes, pb->prev_off[PM]);
    PUT(p_kind, pb->prev_kind, PM);
    PUT(load_state, pb->load_state, pb->n_loads);
    PUT(load_node, pb->load_node, pb->n_loads);
    PUT(load_weight, pb->load_weight, pb->n_loads);
    PUT(load_first, pb->load_first_sweep_only, pb->n_loads);
    PUT(state_stick, pb->state_stickiness, M);
    PUT(state_has_stick, pb->state_has_stickiness, M);
    PUT(rule_inc, pb->rule_inc, pb->n_rules);
    PUT(rule_exc, pb->rule_exc, pb->n_rules);
    for (auto& rr : c->rule_regions) { rr.node_region.release(); rr.reg_lo.release(); rr.reg_hi.release(); rr.leaf_cls.release(); rr.cls_size.release(); }
    c->rule_regions.clear();
    c->any_node_weight = 0;
    for (int n = 0; n < NX; n++) if (pb->node_has_weight[n]) c->any_node_weight = 1;
    if (!pb->hierarchy_rules_nil) {
        // leaf-interval table of every (rule, anchor): plan.go:723-734, :755-774
        const int R = pb->n_rules;
        std::vector<AnchorSet> tab((size_t)(R > 0 ? R : 1) * (NX + 1));
        for (int r = 0; r < R; r++)
            for (int a = 0; a <= NX; a++) {
                int v = a == NX ? pb->vertex_empty : a;
                int vi = v, ve = v;
                for (int l = pb->rule_inc[r]; l > 0; l--) vi = pb->vertex_parent[vi];   // findAncestor
                for (int l = pb->rule_exc[r]; l > 0; l--) ve = pb->vertex_parent[ve];
                AnchorSet st;
                st.alo = pb->vertex_leaf_lo[vi]; st.ahi = pb->vertex_leaf_hi[vi];
                st.blo = pb->vertex_leaf_lo[ve]; st.bhi = pb->vertex_leaf_hi[ve];
                tab[(size_t)r * (NX + 1) + a] = st;
            }
        PUT(anchors, tab.data(), tab.size());
        int n_leaves = 1;
        for (int v = 0; v < pb->n_vertices; v++) if (pb->vertex_leaf_hi[v] > n_leaves) n_leaves = pb->vertex_leaf_hi[v];
        std::vector<int32_t> leaf_node((size_t)n_leaves, -1);
        for (int a = 0; a < NX; a++)
            if (pb->node_leaf_pos[a] >= 0 && pb->node_leaf_pos[a] < n_leaves) leaf_node[pb->node_leaf_pos[a]] = a;
        PUT(leaf_node, leaf_node.data(), leaf_node.size());
        // Does the rule cut the leaves into regions?  Every node whose leaf lies in
        // a region must have exactly that region as its include set.
        c->rule_regions.resize(R);
        for (int r = 0; r < R; r++) {
            blance_ctx::RuleRegions& rr = c->rule_regions[r];
            const AnchorSet* t = &tab[(size_t)r * (NX + 1)];
            std::vector<std::pair<int, int>> iv;
            for (int a = 0; a < NX; a++) {
                int lp = pb->node_leaf_pos[a];
                if (lp >= 0 && t[a].alo <= lp && lp < t[a].ahi) iv.emplace_back(t[a].alo, t[a].ahi);
            }
            std::sort(iv.begin(), iv.end());
            iv.erase(std::unique(iv.begin(), iv.end()), iv.end());
            bool ok = iv.size() >= 2;
            for (size_t i = 1; i < iv.size() && ok; i++) if (iv[i].first < iv[i - 1].second) ok = false;
            std::vector<int32_t> node_region((size_t)NX, -1), rlo, rhi;
            int max_size = 0;
            if (ok) {
                for (auto& x : iv) {
                    rlo.push_back(x.first); rhi.push_back(x.second);
                    if (x.second - x.first > max_size) max_size = x.second - x.first;
                }
                for (int a = 0; a < NX && ok; a++) {
                    int lp = pb->node_leaf_pos[a];
                    if (lp < 0) continue;
                    size_t j = std::upper_bound(rlo.begin(), rlo.end(), lp) - rlo.begin();
                    if (j == 0 || lp >= rhi[j - 1]) continue;
                    if (t[a].alo != rlo[j - 1] || t[a].ahi != rhi[j - 1]) ok = false;
                    node_region[a] = (int)j - 1;
                }
            }
            if (max_size > 256) ok = false;
            // Exclude classes: inside a region the anchors' exclude intervals must be
            // pairwise disjoint (racks inside a zone), so "leaf is excluded by anchor a"
            // is "leaf has a's class".  Intervals that cover the region get class -1.
            std::vector<int32_t> leaf_cls((size_t)n_leaves, -1), cls_size((size_t)n_leaves, 0);
            for (size_t g = 0; g < rlo.size() && ok; g++) {
                std::vector<std::pair<int, int>> cl;
                for (int lp = rlo[g]; lp < rhi[g]; lp++) {
                    int a = leaf_node[lp];
                    if (a < 0) continue;
                    int bl = t[a].blo, bh = t[a].bhi;
                    if (rlo[g] <= bl && bh <= rhi[g] && bh - bl < rhi[g] - rlo[g]) cl.emplace_back(bl, bh);
                }
                std::sort(cl.begin(), cl.end());
                cl.erase(std::unique(cl.begin(), cl.end()), cl.end());
                for (size_t i = 1; i < cl.size() && ok; i++) if (cl[i].first < cl[i - 1].second) ok = false;
                for (size_t i = 0; i < cl.size() && ok; i++) cls_size[rlo[g] + i] = cl[i].second - cl[i].first;
                for (int lp = rlo[g]; lp < rhi[g] && ok; lp++) {
                    int a = leaf_node[lp];
                    if (a < 0) continue;
                    // the class whose interval holds this leaf must be the node's own exclude interval
                    size_t j = std::upper_bound(cl.begin(), cl.end(), std::make_pair(lp, INT_MAX)) - cl.begin();
                    if (j > 0 && lp < cl[j - 1].second) {
                        if (t[a].blo != cl[j - 1].first || t[a].bhi != cl[j - 1].second) ok = false;
                        leaf_cls[lp] = (int)j - 1;
                    }
                }
            }
            rr.ok = ok;
            rr.n_regions = ok ? (int)rlo.size() : 0;
            rr.max_size = max_size;
            if (ok) {
                if (put(c, rr.node_region, node_region.data(), node_region.size())) return BLANCE_ERR_DEVICE;
                if (put(c, rr.reg_lo, rlo.data(), rlo.size())) return BLANCE_ERR_DEVICE;
                if (put(c, rr.reg_hi, rhi.data(), rhi.size())) return BLANCE_ERR_DEVICE;
                if (put(c, rr.leaf_cls, leaf_cls.data(), leaf_cls.size())) return BLANCE_ERR_DEVICE;
                if (put(c, rr.cls_size, cls_size.data(), cls_size.size())) return BLANCE_ERR_DEVICE;
            }
        }
    }
    const int RW = kRecHead + M * (1 + L);       // header + per-state lists
    int kmax = 1;
    for (int m = 0; m < M; m++) if (pb->state_constraints[m] > kmax) kmax = pb->state_constraints[m];
    RESERVE(live, sizeof(int32_t) * (size_t)(PM * L + 1));
    RESERVE(live_len, sizeof(int32_t) * (size_t)(PM + 1));
    RESERVE(live_kind, (size_t)PM + 1);
    RESERVE(prv, sizeof(int32_t) * (size_t)(PM * L + 1));
    RESERVE(prv_len, sizeof(int32_t) * (size_t)(PM + 1));
    RESERVE(prv_kind, (size_t)PM + 1);
    RESERVE(in_prev, (size_t)P + 1);
    RESERVE(never_equal, (size_t)P + 1);
    RESERVE(cnt, sizeof(int32_t) * (size_t)(M + 1) * (NX + 1));
    RESERVE(ntn, sizeof(int32_t) * (size_t)(NX + 1) * (N > 0 ? N : 1));
    RESERVE(cat, (size_t)P + 1);
    RESERVE(order, sizeof(int32_t) * ((size_t)P + 1));
    RESERVE(chunk_counts, sizeof(int32_t) * 3 * (size_t)(cdiv(P, kPartChunk) + 1));
    RESERVE(rec, sizeof(int32_t) * ((size_t)P * RW + 64));
    RESERVE(out, sizeof(int32_t) * ((size_t)P * (1 + kmax) + 1));
    RESERVE(warn_part, sizeof(int32_t) * (size_t)(PM + 1));
    RESERVE(warn_state, sizeof(int32_t) * (size_t)(PM + 1));
    RESERVE(scalars, 64);
    {
        int maxB = 1;
        for (auto& rr : c->rule_regions) if (rr.ok && rr.n_regions > maxB) maxB = rr.n_regions;
        RESERVE(regid, sizeof(int32_t) * ((size_t)P + 1));
        RESERVE(chain_order, sizeof(int32_t) * ((size_t)P + 1));
        RESERVE(bucket_counts, sizeof(int32_t) * ((size_t)maxB * (cdiv(P, kPartChunk) + 1) + 1));
        RESERVE(reg_off, sizeof(int32_t) * ((size_t)maxB + 2));
        RESERVE(cnt_save, sizeof(int32_t) * (size_t)(M + 1) * (NX + 1));
        c->flat_chain_ok = NX >= 1 && NX <= 256 && L <= kChainOwn;
        if (maxB > 1 || c->flat_chain_ok) RESERVE(crec, sizeof(int32_t) * ((size_t)P * kCW + 64));
        if (c->flat_chain_ok) {
            std::vector<int32_t> iota((size_t)NX + 1), zero((size_t)NX + 1, 0), one((size_t)NX + 1, 1);
            for (int i = 0; i <= NX; i++) iota[i] = i;
            int32_t lo0 = 0, hi0 = NX;
            PUT(fl_iota, iota.data(), iota.size());
            PUT(fl_zero, zero.data(), zero.size());
            PUT(fl_one, one.data(), one.size());
            PUT(fl_reglo, &lo0, 1);
            PUT(fl_reghi, &hi0, 1);
        }
        RESERVE(f_tot, sizeof(int32_t) * ((size_t)NX + 1));
        RESERVE(f_g, sizeof(double) * ((size_t)NX + 1));
        RESERVE(f_top_g, sizeof(double) * kTopList);
        RESERVE(f_top_n, sizeof(int32_t) * kTopList);
        RESERVE(f_row_count, sizeof(int32_t) * ((size_t)NX + 2));
        RESERVE(f_m, sizeof(int32_t) * ((size_t)N + 2));
        RESERVE(f_moff, sizeof(int32_t) * ((size_t)N + 2));
        RESERVE(f_keys_a, sizeof(unsigned long long) * ((size_t)P + 1));
        RESERVE(f_keys_b, sizeof(unsigned long long) * ((size_t)P + 1));
        RESERVE(f_vals_a, sizeof(int32_t) * ((size_t)P + 1));
        RESERVE(f_vals_b, sizeof(int32_t) * ((size_t)P + 1));
        RESERVE(f_hist, sizeof(int32_t) * 256 * ((size_t)cdiv(P, kSortTile) + 1));
    }
    HIPTRY(hipStreamSynchronize(c->stream));
    // the caller's arrays are not retained: drop the host pointers
    blance_problem& h = c->h;
    h.state_priority = h.state_constraints = h.state_stickiness = nullptr;
    h.state_has_stickiness = h.node_removed = h.node_added = nullptr;
    c->uploaded = true;
    return BLANCE_OK;
}

template <int T, int NPT>
static void launch_pass(blance_ctx* c, const PassParams& q) {
    size_t lds = sizeof(RedSlot) * 2 * (T / 64) + 64;
    auto kern = k_pass_seq<T, NPT>;
    BLANCE_LAUNCH(kern, 1, T, lds, c->stream, q);
}

static int dispatch_pass(blance_ctx* c, const PassParams& q) {
    // T threads own NPT nodes each (register resident); one workgroup runs the pass.
    const int NX = q.NX > 0 ? q.NX : 1;
    int T = c->force_threads;
    if (T != 64 && T != 256 && T != 1024) T = NX <= 256 ? 64 : (NX <= 1024 ? 256 : 1024);
    if (T == 64 && NX > 256) T = 256;
    if (T == 256 && NX > 1024) T = 1024;
    const int npt = cdiv(NX, T);
    if (T == 64) {
        if (npt <= 1) launch_pass<64, 1>(c, q);
        else launch_pass<64, 4>(c, q);
    } else if (T == 256) {
        if (npt <= 1) launch_pass<256, 1>(c, q);
        else launch_pass<256, 4>(c, q);
    } else {
        if (npt <= 2) launch_pass<1024, 2>(c, q);
        else if (npt <= 4) launch_pass<1024, 4>(c, q);
        else if (npt <= 8) launch_pass<1024, 8>(c, q);
        else return fail(BLANCE_ERR_UNSUPPORTED, "too many nodes for the register-resident pass");
    }
    return 0;
}

// stable LSD radix sort of n (key, value) pairs; result ends in the *_a buffers
static int radix_sort_pairs(blance_ctx* c, int n, int64_t* launches) {
    const int n_tiles = cdiv(n, kSortTile);
    unsigned long long* ka = c->f_keys_a.as<unsigned long long>();
    unsigned long long* kb = c->f_keys_b.as<unsigned long long>();
    int32_t* va = c->f_vals_a.as<int32_t>();
    int32_t* vb = c->f_vals_b.as<int32_t>();
    for (int shift = 0; shift < 64; shift += 8) {
        BLANCE_LAUNCH(k_sort_hist, n_tiles, 64, 1024 + 64, c->stream, n, shift, ka, n_tiles, c->f_hist.as<int32_t>());
        BLANCE_LAUNCH(k_scan_excl, 1, 1024, 256, c->stream, 256 * n_tiles, c->f_hist.as<int32_t>());
        BLANCE_LAUNCH(k_sort_scatter, n_tiles, 64, 1024 + 64, c->stream, n, shift, ka, va, kb, vb, n_tiles,
                      c->f_hist.as<int32_t>());
        std::swap(ka, kb);
        std::swap(va, vb);
        *launches += 3;
    }
    return 0;                                       // 8 passes: the data is back in the *_a buffers
}

static int dispatch_pass(blance_ctx* c, const PassParams& q);
static bool dispatch_chain(blance_ctx* c, ChainParams& q, int max_size);

// Steps [beg, end) of a flat pass on ONE wave64 (clusters of <= 256 node names): the
// chain kernel with the whole cluster as its region and every node its own exclude
// class.  Where the chain cannot go on exactly it stops; k_pass_seq does a few steps
// and the chain resumes.  crec must hold the pass's compact records in pass order.
static int run_flat_chain(blance_ctx* c, PassParams q, int beg, int end, bool lds_rows, int32_t* scal,
                          int64_t* launches) {
    hipStream_t sm = c->stream;
    ChainParams cq;
    memset(&cq, 0, sizeof cq);
    cq.N = q.N; cq.NX = q.NX; cq.M = q.M; cq.L = q.L; cq.s = q.s; cq.k = q.k; cq.NP = q.NP; cq.OW = q.OW;
    cq.booster_kind = q.booster_kind;
    cq.n_regions = 1; cq.flat = 1;
    cq.reg_lo = c->fl_reglo.as<int32_t>(); cq.reg_hi = c->fl_reghi.as<int32_t>();
    cq.reg_off = c->reg_off.as<int32_t>();
    cq.leaf_node = c->fl_iota.as<int32_t>(); cq.leaf_cls = c->fl_iota.as<int32_t>(); cq.cls_size = c->fl_one.as<int32_t>();
    cq.alive = q.alive; cq.node_weight = q.node_weight; cq.node_has_weight = q.node_has_weight;
    cq.cnt = q.cnt; cq.ntn = q.ntn; cq.crec = c->crec.as<int32_t>(); cq.out = q.out; cq.flags = scal + 4;
    int pos = beg;
    while (pos < end) {
        int32_t range[2] = {pos, end};
        HIPTRY(hipMemcpyAsync(c->reg_off.p, range, sizeof range, hipMemcpyHostToDevice, sm));
        HIPTRY(hipMemsetAsync(scal + 4, 0, 32, sm));
        cq.ntn_in_lds = lds_rows ? 1 : 0;
        if (!dispatch_chain(c, cq, q.NX)) return fail(BLANCE_ERR_UNSUPPORTED, "flat chain shape");
        int32_t fl[8] = {0};
        HIPTRY(hipMemcpyAsync(fl, scal + 4, sizeof fl, hipMemcpyDeviceToHost, sm));
        HIPTRY(hipStreamSynchronize(sm));
        *launches += 1;
        lds_rows = false;                          // a stopped chain handed its rows to global memory
        if (fl[0]) return 1;                       // a step the compact record cannot hold: caller falls back
        if (!fl[1]) break;                         // ran to the end
        if (fl[5]) c->no_fast_keys = true;
        int stop = fl[4];
        int nseq = end - stop < 16 ? end - stop : 16;
        q.beg = stop; q.end = stop + nseq;
        int e = dispatch_pass(c, q);
        if (e) return e;
        *launches += 1;
        pos = stop + nseq;
    }
    return 0;
}

// A flat pass (no hierarchy rule for the state): runs of certain stays and of
// fresh identical partitions are resolved in bulk, the rest by k_pass_seq in
// order on sub-ranges.  See the "Flat bulk engine" comment above the kernels.
static int run_flat_pass(blance_ctx* c, PassParams q, int32_t* scal, int64_t* launches, int64_t* batched,
                         bool chain_ok) {
    hipStream_t sm = c->stream;
    const int P = q.P;
    FlatParams fq;
    memset(&fq, 0, sizeof fq);
    fq.N = q.N; fq.NX = q.NX; fq.M = q.M; fq.L = q.L; fq.P = P; fq.s = q.s; fq.k = q.k; fq.top_state = q.top_state;
    fq.NP = q.NP; fq.RW = q.RW; fq.OW = q.OW; fq.higher_mask = q.higher_mask; fq.booster_kind = q.booster_kind;
    fq.alive = q.alive; fq.node_weight = q.node_weight; fq.node_has_weight = q.node_has_weight;
    fq.cnt = q.cnt; fq.tot = c->f_tot.as<int32_t>(); fq.g = c->f_g.as<double>();
    fq.top_g = c->f_top_g.as<double>(); fq.top_n = c->f_top_n.as<int32_t>();
    fq.row_count = c->f_row_count.as<int32_t>();
    fq.ntn = q.ntn; fq.rec = q.rec; fq.out = q.out; fq.scan = scal + 8;
    if (q.NP > 0) {                                 // only read by the stay test when NP > 0
        HIPTRY(hipMemsetAsync(c->f_row_count.p, 0, sizeof(int32_t) * ((size_t)q.NX + 1), sm));
        BLANCE_LAUNCH(k_flat_row_count, cdiv(P, 256), 256, 0, sm, fq, c->f_row_count.as<int32_t>());
        *launches += 1;
    }
    // bulk paths have fixed costs (a host round trip, a sort): short runs stay sequential
    const int kMinStayRun = c->chain_min_parts < 64 ? c->chain_min_parts : 64;
    const int kMinFreshRun = c->chain_min_parts < 512 ? c->chain_min_parts : 512;
    int pos = 0, seq_batch = 256;
    bool dirty = true;
    while (pos < P) {
        if (dirty) {
            BLANCE_LAUNCH(k_flat_prepare, 1, 1024, sizeof(RedSlot) * 32 + 64, sm, fq, c->f_tot.as<int32_t>(),
                          c->f_g.as<double>(), c->f_top_g.as<double>(), c->f_top_n.as<int32_t>());
            dirty = false;
            *launches += 1;
        }
        int32_t init[2] = {INT_MAX, INT_MAX};
        HIPTRY(hipMemcpyAsync(scal + 8, init, sizeof init, hipMemcpyHostToDevice, sm));
        BLANCE_LAUNCH_NOSYNC(k_flat_scan, cdiv(P - pos, 256), 256, 0, sm, fq, pos, P);
        int32_t got[2] = {0, 0};
        HIPTRY(hipMemcpyAsync(got, scal + 8, sizeof got, hipMemcpyDeviceToHost, sm));
        HIPTRY(hipStreamSynchronize(sm));
        *launches += 1;
        int first_nonstay = got[0] > P ? P : got[0], first_nonfresh = got[1] > P ? P : got[1];
        if (first_nonstay - pos >= kMinStayRun || (first_nonstay == P && first_nonstay > pos)) {
            BLANCE_LAUNCH_NOSYNC(k_flat_commit_stay, cdiv(first_nonstay - pos, 256), 256, 0, sm, fq, pos, first_nonstay);
            *launches += 1;
            *batched += first_nonstay - pos;
            pos = first_nonstay;
            continue;
        }
        if (first_nonfresh - pos >= kMinFreshRun && c->n_alive > 0) {
            const int R = first_nonfresh - pos;
            BLANCE_LAUNCH(k_fresh_threshold, 1, 1024, 16384 + 64, sm, fq, pos, R, c->f_m.as<int32_t>(),
                          c->f_moff.as<int32_t>());
            BLANCE_LAUNCH_NOSYNC(k_fresh_emit, cdiv(R, 256), 256, 0, sm, fq, pos, R, c->f_moff.as<int32_t>(),
                                 c->f_keys_a.as<unsigned long long>(), c->f_vals_a.as<int32_t>());
            int e = radix_sort_pairs(c, R, launches);
            if (e) return e;
            BLANCE_LAUNCH_NOSYNC(k_fresh_commit_steps, cdiv(R, 256), 256, 0, sm, fq, pos, R, c->f_vals_a.as<int32_t>());
            BLANCE_LAUNCH_NOSYNC(k_fresh_commit_nodes, cdiv(q.N, 256), 256, 0, sm, fq, pos, c->f_m.as<int32_t>(), q.cnt);
            *launches += 4;
            *batched += R;
            pos += R;
            dirty = true;
            seq_batch = 256;
            continue;
        }
        int B = P - pos < seq_batch ? P - pos : seq_batch;
        int e;
        if (chain_ok) {                               // small cluster: one wave64 walks the batch
            e = run_flat_chain(c, q, pos, pos + B, false, scal, launches);
            if (e < 0) return e;
        } else {
            q.beg = pos; q.end = pos + B;
            e = dispatch_pass(c, q);
            if (e) return e;
            *launches += 1;
        }
        pos += B;
        dirty = true;
        if (seq_batch < (1 << 20)) seq_batch *= 2;
    }
    return 0;
}

template <int NPTC, int KM, bool FAST>
static void launch_chain(blance_ctx* c, const ChainParams& q, size_t lds) {
    auto kern = k_pass_chain<NPTC, KM, FAST>;
    BLANCE_LAUNCH(kern, q.n_regions, 64, lds, c->stream, q);
}

template <int NPTC, int KM>
static void launch_chain_mode(blance_ctx* c, const ChainParams& q, size_t lds, bool fast) {
    if (fast) launch_chain<NPTC, KM, true>(c, q, lds);
    else launch_chain<NPTC, KM, false>(c, q, lds);
}

// one wave64 per region; lanes own NPTC leaves each, k <= KM picks per step
static bool dispatch_chain(blance_ctx* c, ChainParams& q, int max_size) {
    int nptc = cdiv(max_size, 64);
    size_t ntn_bytes = sizeof(int32_t) * (size_t)(max_size + 1) * (max_size + 1);
    if (!q.flat || q.ntn_in_lds) q.ntn_in_lds = ntn_bytes <= 100 * 1024;   // flat mode may insist on global rows
    const bool fast = q.NP == 0 && !c->any_node_weight && !c->no_fast_keys;
    size_t lds = sizeof(double) * (kLpTab + kFfTab + (size_t)max_size) + sizeof(int32_t) * 7 * (size_t)max_size +
                 sizeof(int32_t) * 64 * (size_t)(kCW + q.OW) + (q.NP > 0 && q.ntn_in_lds ? ntn_bytes : 0) + 64;
    if (q.k <= 2) {
        if (nptc <= 2) launch_chain_mode<2, 2>(c, q, lds, fast);
        else if (nptc <= 4) launch_chain_mode<4, 2>(c, q, lds, fast);
        else return false;
    } else if (q.k <= 4) {
        if (nptc <= 2) launch_chain_mode<2, 4>(c, q, lds, fast);
        else if (nptc <= 4) launch_chain_mode<4, 4>(c, q, lds, fast);
        else return false;
    } else {
        return false;
    }
    return true;
}

static int plan_locked(blance_ctx* c, blance_result* res) {
    if (!c->uploaded) return fail(BLANCE_ERR_BAD_ARG, "no problem uploaded");
    HIPTRY(hipSetDevice(c->device));
    const blance_problem& h = c->h;
    const int N = h.n_nodes, NX = h.n_nodes_ext, M = h.n_states, P = h.n_parts, L = c->L;
    const int64_t PM = (int64_t)P * M;
    const int RW = kRecHead + M * (1 + L);       // header + per-state lists
    hipStream_t sm = c->stream;
    int32_t* scal = c->scalars.as<int32_t>();
    int64_t launches = 0, steps = 0, batched = 0;
    int n_pass = 0;
    unsigned chain_gave_up = 0;     // states whose chain pass had to be redone sequentially

    DevProblem d;
    d.N = N; d.NX = NX; d.M = M; d.L = L; d.P = P;
    d.weights_nil = h.partition_weights_nil;
    d.part_weight = c->part_weight.as<int32_t>();
    d.part_has_weight = c->part_has_weight.as<uint8_t>();
    d.live = c->live.as<int32_t>(); d.live_len = c->live_len.as<int32_t>(); d.live_kind = c->live_kind.as<uint8_t>();
    d.prv = c->prv.as<int32_t>(); d.prv_len = c->prv_len.as<int32_t>(); d.prv_kind = c->prv_kind.as<uint8_t>();
    d.in_prev = c->in_prev.as<uint8_t>(); d.never_equal = c->never_equal.as<uint8_t>();

    HIPTRY(hipEventRecord(c->ev0, sm));
    HIPTRY(hipMemsetAsync(scal, 0, 64, sm));
    if (P > 0) {
        HIPTRY(hipMemcpyAsync(d.in_prev, c->part_in_prev.p, (size_t)P, hipMemcpyDeviceToDevice, sm));
        HIPTRY(hipMemcpyAsync(d.never_equal, c->part_never_equal.p, (size_t)P, hipMemcpyDeviceToDevice, sm));
    }
    int iterations = 0, converged = 0;
    for (int it = 0; it < h.max_iterations; it++) {                 // plan.go:32
        const bool first = it == 0;
        d.node_removed = first ? c->node_removed.as<uint8_t>() : c->zeros_nx.as<uint8_t>();   // plan.go:53-55
        d.node_added = first ? c->node_added.as<uint8_t>() : c->zeros_nx.as<uint8_t>();
        const int add_nil = first ? h.nodes_to_add_nil : 0;
        const int any_removed = first ? c->any_removed : 0;
        const int NP = first ? h.n_prev : c->np_later;
        HIPTRY(hipMemsetAsync(scal, 0, 8, sm));                     // warn_count, not_match
        if (PM > 0) {
            if (first)
                BLANCE_LAUNCH_NOSYNC(k_live_init, cdiv(PM, 256), 256, 0, sm, d, c->a_off.as<int32_t>(),
                                     c->a_nodes.as<int32_t>(), c->a_kind.as<uint8_t>(), c->p_off.as<int32_t>(),
                                     c->p_nodes.as<int32_t>(), c->p_kind.as<uint8_t>());
            else
                BLANCE_LAUNCH_NOSYNC(k_live_refresh, cdiv(PM, 256), 256, 0, sm, d);
            launches++;
        }
        // stateNodeCounts = countStateNodes(prevMap), plan.go:94
        HIPTRY(hipMemsetAsync(c->cnt.p, 0, sizeof(int32_t) * (size_t)(M + 1) * (NX + 1), sm));
        if (h.n_loads > 0) {
            BLANCE_LAUNCH_NOSYNC(k_count_loads, cdiv(h.n_loads, 256), 256, 0, sm, h.n_loads, NX, first ? 0 : 1,
                                 c->load_state.as<int32_t>(), c->load_node.as<int32_t>(),
                                 c->load_weight.as<int32_t>(), c->load_first.as<uint8_t>(), c->cnt.as<int32_t>());
            launches++;
        }
        if (PM > 0) {
            BLANCE_LAUNCH_NOSYNC(k_count_prev, cdiv(PM, 256), 256, 0, sm, d, c->cnt.as<int32_t>());
            launches++;
        }
        for (int m = 0; m < M; m++) {                               // plan.go:307-324
            const int k = c->state_constraints[m];
            if (k <= 0 || P == 0) continue;
            const int n_chunks = cdiv(P, kPartChunk);
            BLANCE_LAUNCH_NOSYNC(k_category, cdiv(P, 256), 256, 0, sm, d, m, any_removed, add_nil, c->cat.as<uint8_t>());
            BLANCE_LAUNCH(k_part_count, n_chunks, 64, 64, sm, P, (const int32_t*)nullptr, c->cat.as<uint8_t>(),
                          c->part_order.as<int32_t>(), n_chunks, 3, c->chunk_counts.as<int32_t>());
            BLANCE_LAUNCH(k_scan_excl, 1, 1024, 256, sm, 3 * n_chunks, c->chunk_counts.as<int32_t>());
            BLANCE_LAUNCH(k_part_scatter, n_chunks, 64, 64, sm, P, (const int32_t*)nullptr, c->cat.as<uint8_t>(),
                          c->part_order.as<int32_t>(), c->part_order.as<int32_t>(), n_chunks, 3, 2,
                          c->chunk_counts.as<int32_t>(), c->order.as<int32_t>());
            if (NP > 0)                                             // nodeToNodeCounts := fresh, plan.go:266
                HIPTRY(hipMemsetAsync(c->ntn.p, 0, sizeof(int32_t) * (size_t)(NX + 1) * (N > 0 ? N : 1), sm));
            const int OW = 1 + k;
            int higher_mask = 0;
            for (int t = 0; t < M; t++)
                if (c->state_priority[t] < c->state_priority[m]) higher_mask |= 1 << t;
            const int r0 = c->rule_off[m], r1 = c->rule_off[m + 1];
            while (c->pass_events.size() < 2 * (size_t)(n_pass + 2)) {
                hipEvent_t ev;
                HIPTRY(hipEventCreate(&ev));
                c->pass_events.push_back(ev);
            }

            // ---- region chains, when the state's single hierarchy rule allows them
            bool done = false;
            if (c->engine != BLANCE_ENGINE_SEQUENTIAL && !h.hierarchy_rules_nil && r1 - r0 == 1 &&
                c->rule_regions[r0].ok && P >= c->chain_min_parts && k <= 4 && !((chain_gave_up >> m) & 1)) {
                blance_ctx::RuleRegions& rr = c->rule_regions[r0];
                const int B = rr.n_regions, nbc = cdiv(P, kPartChunk);
                HIPTRY(hipMemsetAsync(scal + 4, 0, 16, sm));
                BLANCE_LAUNCH_NOSYNC(k_chain_classify, cdiv(P, 256), 256, 0, sm, d, m, h.top_state, c->order.as<int32_t>(),
                                     rr.node_region.as<int32_t>(), c->regid.as<int32_t>(), scal + 4);
                int nbits = 1;
                while ((1 << nbits) < B) nbits++;
                BLANCE_LAUNCH(k_part_count, nbc, 64, sizeof(int32_t) * B + 64, sm, P, c->regid.as<int32_t>(),
                              (const uint8_t*)nullptr, (const int32_t*)nullptr, nbc, B, c->bucket_counts.as<int32_t>());
                BLANCE_LAUNCH(k_scan_excl, 1, 1024, 256, sm, B * nbc, c->bucket_counts.as<int32_t>());
                BLANCE_LAUNCH_NOSYNC(k_region_offsets, cdiv(B + 1, 64), 64, 0, sm, B, nbc, P,
                                     c->bucket_counts.as<int32_t>(), c->reg_off.as<int32_t>());
                BLANCE_LAUNCH(k_part_scatter, nbc, 64, sizeof(int32_t) * B + 64, sm, P, c->regid.as<int32_t>(),
                              (const uint8_t*)nullptr, (const int32_t*)nullptr, c->order.as<int32_t>(), nbc, B, nbits,
                              c->bucket_counts.as<int32_t>(), c->chain_order.as<int32_t>());
                BLANCE_LAUNCH_NOSYNC(k_gather, cdiv(P, 256), 256, 0, sm, d, m, h.top_state, RW, c->chain_order.as<int32_t>(),
                                     c->state_stick.as<int32_t>(), c->state_has_stick.as<uint8_t>(), c->rec.as<int32_t>());
                BLANCE_LAUNCH_NOSYNC(k_gather_chain, cdiv(P, 256), 256, 0, sm, d, m, h.top_state, higher_mask,
                                     c->chain_order.as<int32_t>(), c->state_stick.as<int32_t>(),
                                     c->state_has_stick.as<uint8_t>(), c->node_leaf_pos.as<int32_t>(),
                                     rr.node_region.as<int32_t>(), rr.reg_lo.as<int32_t>(), rr.leaf_cls.as<int32_t>(), 0,
                                     c->crec.as<int32_t>(), scal + 4);
                HIPTRY(hipMemcpyAsync(c->cnt_save.p, c->cnt.p, sizeof(int32_t) * (size_t)(M + 1) * NX,
                                      hipMemcpyDeviceToDevice, sm));
                ChainParams cq;
                memset(&cq, 0, sizeof cq);
                cq.N = N; cq.NX = NX; cq.M = M; cq.L = L; cq.s = m; cq.k = k;
                cq.NP = NP; cq.OW = OW; cq.booster_kind = h.booster_kind;
                cq.n_regions = B;
                cq.reg_lo = rr.reg_lo.as<int32_t>(); cq.reg_hi = rr.reg_hi.as<int32_t>();
                cq.reg_off = c->reg_off.as<int32_t>();
                cq.leaf_node = c->leaf_node.as<int32_t>();
                cq.leaf_cls = rr.leaf_cls.as<int32_t>();
                cq.cls_size = rr.cls_size.as<int32_t>();
                cq.alive = c->alive.as<uint8_t>();
                cq.node_weight = c->node_weight.as<int32_t>();
                cq.node_has_weight = c->node_has_weight.as<uint8_t>();
                cq.cnt = c->cnt.as<int32_t>(); cq.ntn = c->ntn.as<int32_t>();
                cq.crec = c->crec.as<int32_t>(); cq.out = c->out.as<int32_t>();
                cq.flags = scal + 4;
                HIPTRY(hipEventRecord(c->pass_events[2 * n_pass], sm));
                bool launched = dispatch_chain(c, cq, rr.max_size);
                HIPTRY(hipEventRecord(c->pass_events[2 * n_pass + 1], sm));
                launches += 8;
                if (launched) {
                    c->pass_kind.resize(n_pass + 1);
                    c->pass_kind[n_pass] = 0;
                    n_pass++;
                    int32_t fl[4] = {0, 0, 0, 0};
                    HIPTRY(hipMemcpyAsync(fl, scal + 4, sizeof fl, hipMemcpyDeviceToHost, sm));
                    HIPTRY(hipStreamSynchronize(sm));
                    if (getenv("BLANCE_TRACE"))
                        fprintf(stderr, "[blance] chain pass state %d: %d of %d steps committed as verified stays in %d batches\n",
                                m, fl[2], P, fl[3]);
                    if (!fl[0] && !fl[1]) {
                        BLANCE_LAUNCH_NOSYNC(k_scatter, cdiv(P, 256), 256, 0, sm, d, m, RW, OW, c->chain_order.as<int32_t>(),
                                             c->rec.as<int32_t>(), c->out.as<int32_t>());
                        launches++;
                        batched += P;
                        done = true;
                    } else {                                        // not region-local after all: redo in order
                        chain_gave_up |= 1u << m;                  // and do not try again in later sweeps
                        if (NP > 0)                                 // chains of big regions keep their rows in global memory
                            HIPTRY(hipMemsetAsync(c->ntn.p, 0, sizeof(int32_t) * (size_t)(NX + 1) * (N > 0 ? N : 1), sm));
                        HIPTRY(hipMemcpyAsync(c->cnt.p, c->cnt_save.p, sizeof(int32_t) * (size_t)(M + 1) * NX,
                                              hipMemcpyDeviceToDevice, sm));
                    }
                }
            }
            if (!done) {
            BLANCE_LAUNCH_NOSYNC(k_gather, cdiv(P, 256), 256, 0, sm, d, m, h.top_state, RW, c->order.as<int32_t>(),
                                 c->state_stick.as<int32_t>(), c->state_has_stick.as<uint8_t>(), c->rec.as<int32_t>());
            PassParams q;
            memset(&q, 0, sizeof q);
            q.N = N; q.NX = NX; q.M = M; q.L = L; q.P = P; q.s = m; q.k = k; q.top_state = h.top_state;
            q.NP = NP; q.RW = RW; q.OW = OW;
            q.higher_mask = higher_mask;
            q.hier = !h.hierarchy_rules_nil;
            q.rule_begin = r0; q.rule_end = r1;
            q.booster_kind = h.booster_kind;
            q.n_alive = c->n_alive;
            q.vertex_empty_anchor = NX;
            q.alive = c->alive.as<uint8_t>();
            q.node_weight = c->node_weight.as<int32_t>();
            q.node_has_weight = c->node_has_weight.as<uint8_t>();
            q.node_leaf_pos = c->node_leaf_pos.as<int32_t>();
            q.anchors = c->anchors.as<AnchorSet>();
            q.cnt = c->cnt.as<int32_t>();
            q.ntn = c->ntn.as<int32_t>();
            q.rec = c->rec.as<int32_t>();
            q.out = c->out.as<int32_t>();
            q.warn_part = c->warn_part.as<int32_t>();
            q.warn_state = c->warn_state.as<int32_t>();
            q.warn_count = scal + 0;
            q.err = scal + 2;
            q.beg = 0; q.end = P;
            // a flat pass (no rule for the state) of a small cluster can run on one wave64
            const bool flat_state = h.hierarchy_rules_nil || r1 == r0;
            bool flat_chain = c->engine != BLANCE_ENGINE_SEQUENTIAL && flat_state && c->flat_chain_ok && k <= 4 &&
                              P >= c->chain_min_parts;
            c->no_fast_keys = false;
            if (flat_chain) {
                HIPTRY(hipMemsetAsync(scal + 4, 0, 32, sm));
                BLANCE_LAUNCH_NOSYNC(k_gather_chain, cdiv(P, 256), 256, 0, sm, d, m, h.top_state, higher_mask,
                                     c->order.as<int32_t>(), c->state_stick.as<int32_t>(),
                                     c->state_has_stick.as<uint8_t>(), c->fl_iota.as<int32_t>(),
                                     c->fl_zero.as<int32_t>(), c->fl_reglo.as<int32_t>(), c->fl_iota.as<int32_t>(), 1,
                                     c->crec.as<int32_t>(), scal + 4);
                int32_t bad = 0;
                HIPTRY(hipMemcpyAsync(&bad, scal + 4, sizeof bad, hipMemcpyDeviceToHost, sm));
                HIPTRY(hipStreamSynchronize(sm));
                launches++;
                if (bad) flat_chain = false;       // some step does not fit the compact record
            }
            HIPTRY(hipEventRecord(c->pass_events[2 * n_pass], sm));
            int e;
            c->pass_kind.resize(n_pass + 1);
            if (c->engine != BLANCE_ENGINE_SEQUENTIAL && flat_state && k == 1 && P >= c->chain_min_parts) {
                c->pass_kind[n_pass] = 1;
                e = run_flat_pass(c, q, scal, &launches, &batched, flat_chain);
            } else if (flat_chain) {
                c->pass_kind[n_pass] = 0;
                const size_t rows = sizeof(int32_t) * (size_t)(NX + 1) * (NX + 1);
                e = run_flat_chain(c, q, 0, P, NP > 0 && rows <= 100 * 1024, scal, &launches);
                if (e > 0) e = fail(BLANCE_ERR_DEVICE, "flat chain refused a checked pass");
                if (!e) batched += P;
            } else {
                c->pass_kind[n_pass] = 0;
                e = dispatch_pass(c, q);
            }
            if (e) return e;
            HIPTRY(hipEventRecord(c->pass_events[2 * n_pass + 1], sm));
            n_pass++;
            BLANCE_LAUNCH_NOSYNC(k_scatter, cdiv(P, 256), 256, 0, sm, d, m, RW, q.OW, c->order.as<int32_t>(),
                                 c->rec.as<int32_t>(), c->out.as<int32_t>());
            }
            launches += 7;
            steps += P;
        }
        iterations++;
        // convergence (plan.go:36-45) + write-back (plan.go:49-52)
        if (P > 0) {
            BLANCE_LAUNCH_NOSYNC(k_converge, cdiv(P, 256), 256, 0, sm, d, scal + 1);
            launches++;
        }
        int32_t hs[4] = {0, 0, 0, 0};
        HIPTRY(hipMemcpyAsync(hs, scal, sizeof hs, hipMemcpyDeviceToHost, sm));
        HIPTRY(hipStreamSynchronize(sm));
        HIPTRY(hipGetLastError());
        if (hs[2]) return fail(BLANCE_ERR_UNSUPPORTED, "hierarchy fold overflowed the device's interval budget");
        c->n_warnings = hs[0];
        if (!hs[1]) { converged = 1; break; }
    }
    HIPTRY(hipEventRecord(c->ev1, sm));
    HIPTRY(hipEventSynchronize(c->ev1));
    float ms = 0.f;
    HIPTRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    double pass_ms = 0.0, flat_ms = 0.0;
    int n_kernel_pass = 0, n_flat = 0;
    for (int i = 0; i < n_pass; i++) {
        float pm = 0.f;
        HIPTRY(hipEventElapsedTime(&pm, c->pass_events[2 * i], c->pass_events[2 * i + 1]));
        if (c->pass_kind[i] == 0) { pass_ms += pm; n_kernel_pass++; } else { flat_ms += pm; n_flat++; }
        if (getenv("BLANCE_TRACE"))
            fprintf(stderr, "[blance] pass %d (%s): %.3f ms\n", i, c->pass_kind[i] ? "flat bulk driver" : "pass kernel", pm);
    }
    c->pass_ms = pass_ms;
    c->pass_launches = n_kernel_pass;
    c->flat_ms = flat_ms;
    c->flat_passes = n_flat;
    c->iterations = iterations;
    c->converged = converged;
    c->device_ms = ms;
    c->steps_total = steps;
    c->steps_batched = batched;
    c->kernel_launches = launches;
    c->planned = true;
    if (iterations == 0) c->n_warnings = 0;
    if (res) {
        res->iterations = iterations;
        res->converged = converged;
        res->device_ms = ms;
        res->total_ms = ms;
        res->steps_total = steps;
        res->steps_sequential = steps - batched;
        res->steps_batched = batched;
        res->kernel_launches = launches;
        res->n_warnings = c->n_warnings;
        res->pass_kernel_ms = pass_ms;
        res->pass_kernel_launches = n_kernel_pass;
        res->flat_pass_ms = flat_ms;
        res->flat_passes = n_flat;
    }
    return BLANCE_OK;
}

static int download_locked(blance_ctx* c, blance_result* res) {
    if (!c->planned) return fail(BLANCE_ERR_BAD_ARG, "nothing planned yet");
    if (!res || !res->out_off || !res->out_nodes || !res->out_kind || !res->warn_part || !res->warn_state)
        return fail(BLANCE_ERR_BAD_ARG, "null result buffers");
    HIPTRY(hipSetDevice(c->device));
    const blance_problem& h = c->h;
    const int M = h.n_states, P = h.n_parts, L = c->L;
    const size_t PM = (size_t)P * M;
    if (c->n_warnings > res->warn_capacity) return fail(BLANCE_ERR_CAPACITY, "warn_capacity too small");
    std::vector<int32_t> live(PM * L + 1), len(PM + 1);
    std::vector<uint8_t> kind(PM + 1);
    if (c->iterations == 0) {
        // MaxIterationsPerPlan <= 0: planNextMapEx returns (nil, nil) -- nothing to report (plan.go:32-58)
        for (size_t i = 0; i <= PM; i++) res->out_off[i] = 0;
        for (size_t i = 0; i < PM; i++) res->out_kind[i] = BLANCE_LIST_ABSENT;
        res->n_warnings = 0;
        res->iterations = 0;
        res->converged = 0;
        return BLANCE_OK;
    }
    if (PM) {
        HIPTRY(hipMemcpyAsync(live.data(), c->live.p, sizeof(int32_t) * PM * L, hipMemcpyDeviceToHost, c->stream));
        HIPTRY(hipMemcpyAsync(len.data(), c->live_len.p, sizeof(int32_t) * PM, hipMemcpyDeviceToHost, c->stream));
        HIPTRY(hipMemcpyAsync(kind.data(), c->live_kind.p, PM, hipMemcpyDeviceToHost, c->stream));
    }
    if (c->n_warnings) {
        HIPTRY(hipMemcpyAsync(res->warn_part, c->warn_part.p, sizeof(int32_t) * (size_t)c->n_warnings, hipMemcpyDeviceToHost, c->stream));
        HIPTRY(hipMemcpyAsync(res->warn_state, c->warn_state.p, sizeof(int32_t) * (size_t)c->n_warnings, hipMemcpyDeviceToHost, c->stream));
    }
    HIPTRY(hipStreamSynchronize(c->stream));
    int64_t off = 0;
    for (size_t idx = 0; idx < PM; idx++) {
        res->out_off[idx] = (int32_t)off;
        res->out_kind[idx] = kind[idx];
        int n = kind[idx] == BLANCE_LIST_ABSENT ? 0 : len[idx];
        if (off + n > res->out_capacity) return fail(BLANCE_ERR_CAPACITY, "out_capacity too small");
        for (int i = 0; i < n; i++) res->out_nodes[off + i] = live[idx * L + i];
        off += n;
    }
    res->out_off[PM] = (int32_t)off;
    res->n_warnings = c->n_warnings;
    res->iterations = c->iterations;
    res->converged = c->converged;
    res->device_ms = c->device_ms;
    res->steps_total = c->steps_total;
    res->steps_sequential = c->steps_total - c->steps_batched;
    res->steps_batched = c->steps_batched;
    res->kernel_launches = c->kernel_launches;
    res->pass_kernel_ms = c->pass_ms;
    res->pass_kernel_launches = c->pass_launches;
    res->flat_pass_ms = c->flat_ms;
    res->flat_passes = c->flat_passes;
    return BLANCE_OK;
}

extern "C" int blance_upload(blance_ctx* c, const blance_problem* pb) {
    if (!c) return fail(BLANCE_ERR_BAD_ARG, "null ctx");
    std::lock_guard<std::mutex> g(c->mu);
    return upload_locked(c, pb);
}

extern "C" int blance_plan_resident(blance_ctx* c, blance_result* res) {
    if (!c) return fail(BLANCE_ERR_BAD_ARG, "null ctx");
    std::lock_guard<std::mutex> g(c->mu);
    return plan_locked(c, res);
}

extern "C" int blance_download(blance_ctx* c, blance_result* res) {
    if (!c) return fail(BLANCE_ERR_BAD_ARG, "null ctx");
    std::lock_guard<std::mutex> g(c->mu);
    return download_locked(c, res);
}

extern "C" int blance_plan(blance_ctx* c, const blance_problem* pb, blance_result* res) {
    if (!c) return fail(BLANCE_ERR_BAD_ARG, "null ctx");
    if (!res) return fail(BLANCE_ERR_BAD_ARG, "null result");
    std::lock_guard<std::mutex> g(c->mu);
    hipEvent_t t0, t1;
    HIPTRY(hipSetDevice(c->device));
    HIPTRY(hipEventCreate(&t0));
    HIPTRY(hipEventCreate(&t1));
    HIPTRY(hipEventRecord(t0, c->stream));
    int st = upload_locked(c, pb);
    if (!st) st = plan_locked(c, res);
    if (!st) st = download_locked(c, res);
    if (!st) {
        (void)hipEventRecord(t1, c->stream);
        (void)hipEventSynchronize(t1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, t0, t1);
        res->total_ms = ms;
    }
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    return st;
}
