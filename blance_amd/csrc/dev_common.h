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


}  // namespace blance
