// Device helpers: nodeSorter.Score / Less, workgroup argmin, hierarchy masks as leaf-interval algebra.
// Part of blance_hip.hip (one translation unit); see DESIGN.md section 4.
#pragma once

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

struct RedSlot { unsigned hi, lo; int n; int pad; };

#ifdef BLANCE_PHASE_PROF     // developer build only: per-phase shader-clock totals of chain 0
#define PH_DECL unsigned long long ph_acc[20] = {0}, ph_t0 = clock64(), ph_t1
#define PH(i) do { ph_t1 = clock64(); ph_acc[i] += ph_t1 - ph_t0; ph_t0 = ph_t1; } while (0)
#define PH_DUMP(steps) do { if (blockIdx.x == 0 && threadIdx.x == 0) { for (int i_ = 0; i_ < 20; i_++) \
    printf("[phase %d] %.1f cycles/step\n", i_, (double)ph_acc[i_] / (double)(steps)); } } while (0)
#elif defined(BLANCE_ASM_MARKS)   // developer build only: phase markers as comments in the ISA
#define PH_DECL
#define PH(i) asm volatile("; PHASE_MARK " #i)
#define PH_DUMP(steps)
#define PHM(name) asm volatile("; MARK " #name)
#else
#define PH_DECL
#define PH(i)
#define PH_DUMP(steps)
#endif
#ifndef PHM
#define PHM(name)            /* BLANCE_ASM_MARKS: a comment in the ISA */
#endif

constexpr unsigned kKeyNoneV = 0xffffffffu;

// v_writelane_b32: lane `lane` of `old` takes the wave-uniform value v (both selectors come from the scalar unit)
__device__ __forceinline__ int write_lane(int v, int lane, int old) {
#ifndef BLANCE_SIMT_EMU
    // gfx9 reads one SGPR per VALU instruction (constant bus): the lane select goes through M0
    asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(old) : "s"(v), "s"(lane) : "m0");
    return old;
#else
    return (int)(threadIdx.x & 63) == lane ? v : old;
#endif
}

template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
}

// 32-bit minimum over one wave64 (wave-uniform result): the cross-row part by row broadcasts, the
// minimum taken by the DPP instruction itself -- six v_min_u32_dpp and one v_readlane
__device__ __forceinline__ unsigned wave_min_u32_bcast(unsigned v) {
#ifndef BLANCE_SIMT_EMU
    asm volatile(
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 0"
        : "+v"(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
#else
    unsigned t;
    t = (unsigned)dpp_mov<0xB1>((int)v);  v = t < v ? t : v;
    t = (unsigned)dpp_mov<0x4E>((int)v);  v = t < v ? t : v;
    t = (unsigned)dpp_mov<0x141>((int)v); v = t < v ? t : v;
    t = (unsigned)dpp_mov<0x140>((int)v); v = t < v ? t : v;
    t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v = t < v ? t : v;
    t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    v = t < v ? t : v;
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
#endif
}

// 32-bit minimum over aligned groups of G = 4 or 16 lanes (every lane gets its group's minimum)
template <int G>
__device__ __forceinline__ unsigned row_min_u32(unsigned v) {
    static_assert(G == 4 || G == 16, "group of 4 or 16 lanes");
#ifndef BLANCE_SIMT_EMU
    asm volatile(
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0"
        : "+v"(v));
    if (G == 16)
        asm volatile(
            "s_nop 1\n\t"
            "v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 0"
            : "+v"(v));
#else
    unsigned t;
    t = (unsigned)dpp_mov<0xB1>((int)v);  v = t < v ? t : v;
    t = (unsigned)dpp_mov<0x4E>((int)v);  v = t < v ? t : v;
    if (G == 16) {
        t = (unsigned)dpp_mov<0x141>((int)v); v = t < v ? t : v;
        t = (unsigned)dpp_mov<0x140>((int)v); v = t < v ? t : v;
    }
#endif
    return v;
}

// order-preserving integer image of a non-NaN double (-0.0 == +0.0)
__device__ __forceinline__ unsigned long long sortable_bits(double s) {
    if (s == 0.0) s = 0.0;
    unsigned long long b = (unsigned long long)__double_as_longlong(s);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every
// outstanding global load and store of the wave (vmcnt(0)) -- that would put the latency of
// the record / row prefetches, issued precisely to overlap with the step, on every argmin.
__device__ __forceinline__ void lds_barrier() {
#ifndef BLANCE_SIMT_EMU
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    __syncthreads();
#endif
}

// Lexicographic (score, position) argmin over a workgroup of T threads; callers pass
// (+inf, INT_MAX) for "no candidate".  Per wave: three 32-bit DPP minima (high word,
// low word of the score's integer image, position) -- the same total order as
// better().  One barrier per call; slots are double-buffered by call parity.
template <int T>
__device__ __forceinline__ int block_argmin(double s, int n, RedSlot* red, int& round) {
    const unsigned long long b = sortable_bits(s);
    const unsigned hi = (unsigned)(b >> 32), lo = (unsigned)b;
    const unsigned mh = wave_min_u32_bcast(hi);
    const bool ok2 = hi == mh;
    const unsigned ml = wave_min_u32_bcast(ok2 ? lo : kKeyNoneV);
    const bool ok3 = ok2 && lo == ml;
    const unsigned mn = wave_min_u32_bcast(ok3 ? (unsigned)n : kKeyNoneV);
    constexpr int W = T / 64;
    if constexpr (W == 1) {
        return (int)mn;
    } else {
    RedSlot* slot = red + (round & 1) * W;
    round++;
    if ((threadIdx.x & 63) == 0) {
        slot[threadIdx.x >> 6].hi = mh;
        slot[threadIdx.x >> 6].lo = ml;
        slot[threadIdx.x >> 6].n = (int)mn;
    }
    lds_barrier();
    // second stage: lane l takes wave (l mod W)'s slot; W <= 16 slots sit in one DPP row
    constexpr int G = W <= 4 ? 4 : 16;             // W = 8: every slot sits twice in a row of 16
    const RedSlot mine = slot[threadIdx.x & (W - 1)];
    const unsigned bh = row_min_u32<G>(mine.hi);
    const bool k2 = mine.hi == bh;
    const unsigned bl = row_min_u32<G>(k2 ? mine.lo : kKeyNoneV);
    const bool k3 = k2 && mine.lo == bl;
    return (int)row_min_u32<G>(k3 ? (unsigned)mine.n : kKeyNoneV);
    }
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


}  // namespace blance
