// k_pass_queue: the exact sequential state pass of a state WITHOUT hierarchy rules on one wave64, the
// candidates kept as a SORTED WINDOW across the lanes (a systolic priority queue) -- new in round 4.
// Part of tu_queue.hip; see DESIGN.md section 4.3a.
#pragma once

namespace blance {

// ============================================================================
// assignStateToPartitions (plan.go:253-303) with findBestNodes (plan.go:98-248) for a state that has no
// hierarchy rule and k <= 2.  Same facts as k_pass_tree.h: a node n that is not one
// of the partition's own has exact score (plan.go:634-689) >= g[n], its partition-independent score, with
// equality bit for bit when the partition's nodeToNodeCounts entry is 0; a step changes g of at most
// (old + chosen) nodes.  What differs is the structure that orders the candidates:
//
//  * The WINDOW: lane i holds the i-th smallest (g, node) over all nodes -- the <= 64 nodes below a bound
//    THETA, sorted.  Invariant: the window holds exactly the nodes whose (g, node) is below THETA.  A node
//    whose g changed is removed (lanes above it move down one: one DPP wave shift per register) and, if
//    its new key is below THETA, inserted at its place (lanes at and above move up one); a 65th entry is
//    dropped and becomes THETA.  The smallest candidate is lane 0, no reduction over the wave -- k_pass_tree
//    paid three to four wave minima (two DPP chains each) and a group rescan per moving step.  A window that has
//    run dry, or whose entries cannot settle a step, is REBUILT from the keys in LDS -- since round 5 by the helper
//    waves (topl_part below: three waves select the 32 smallest of a third of the nodes each, the lists are merged
//    below the smallest of their last entries).  Measured on BASELINE config 5: one rebuild per ~500 moving steps;
//    with many nodes of one load (a rebalance after nodes left, Zipf weights) one per 64 to 120.
//  * ROW BIT MAPS: "is nodeToNodeCounts[row][n] zero" for the 64 rows of a batch sits in LDS (ntn_bits:
//    one bit per matrix entry, maintained next to the matrix), so a candidate whose bit is clear has its
//    exact score = its window key without touching the matrix -- k_pass_tree waited ~1 us for an entry
//    in half of its moving steps.  Only candidates with the bit set, in front of the k-th clean one, are
//    read from the matrix and scored exactly.
//  * A step that keeps its nodes is validated by its lane against the front of the window, as in
//    k_pass_tree; every other step resolves as: k best of (own nodes with their exact scores, the first
//    eligible window entries), valid if all of them lie below THETA -- otherwise the window is rebuilt
//    (from the g keys kept in LDS) and the step repeated.
//
//  * FOLDED ROW: a batch whose steps all belong to partitions without a top priority node (they lost their
//    primary: plan.go:134-138 gives them row "") keeps that row in LDS as part of the keys; no step of it reads
//    the matrix.  PROMOTIONS (the taken node holds the partition in a lower priority state, plan.go:294-297) are
//    settled where they occur: that state's counter of the node drops, its total does not grow.
//  * The step loop for the plain case is hand-written assembly (k_queue_walk.h); its C++ twin below is what the
//    SIMT emulator runs and what takes the steps the assembly leaves.
//
// What the kernel does not do it does not guess: a step that needs the general machinery of k_pass_tree
// (a node held in two lists, more higher-priority nodes than it keeps, fewer candidates than constraints,
// a general step inside a folded batch) STOPS the
// launch there -- everything before it is committed, q.stop[0] says where -- and the host lets
// k_pass_tree do a few steps before it relaunches this kernel.
// ============================================================================
constexpr int kQueueMaxNodes = 4096;
constexpr int kQStopNone = 0, kQStopShape = 1, kQStopShort = 3, kQStopPromote = 4;

#ifndef BLANCE_SIMT_EMU
#define BLANCE_QLD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define BLANCE_QFENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent")
// This kernel is one wave on one CU: its atomics and its loads meet in the same L2, so "my bumps are done before I read
// again" needs no cache maintenance, only (a) the bumps acknowledged (vmcnt) and (b) reads that do not hit a stale L1 line
// -- the scoped loads above and the 16-byte row loads below (sc0 sc1: past the L1).  The full fence (L2 write-back and
// invalidate) stays where it is rare.
#define BLANCE_QWAIT_BUMPS() __builtin_amdgcn_s_waitcnt(0x0F70)      /* vmcnt(0): gfx9 counts returnless atomics there too */
typedef int qv4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ qv4 q_load_row16(const qv4* p) {
    qv4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
#define BLANCE_QROWS_ARRIVED(v) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]))
#else
#define BLANCE_QLD(p) (*(p))
#define BLANCE_QFENCE()
#define BLANCE_QWAIT_BUMPS()
typedef int4 qv4;
static inline qv4 q_load_row16(const qv4* p) { return *p; }
#define BLANCE_QROWS_ARRIVED(v)
#endif

// lane i takes lane i - 1's value (lane 0: fill) / lane i + 1's value (lane 63: fill): DPP wave shifts
__device__ __forceinline__ int wave_from_prev(int v, int fill) {
    return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false);       // wave_shr:1
}
__device__ __forceinline__ int wave_from_next(int v, int fill) {
    return __builtin_amdgcn_update_dpp(fill, v, 0x130, 0xf, 0xf, false);       // wave_shl:1
}

__device__ __forceinline__ bool qless(unsigned long long a, int an, unsigned long long b, int bn) {
    return a < b || (a == b && an < bn);              // nodeSorter.Less on sortable images, plan.go:617-628
}

// nodeSorter.Score (plan.go:634-689): the reference's operations in the reference's order; the two NumPartitions
// quotients from LDS tables filled by the same expressions, the division by a power-of-two weight as an exponent shift
__device__ __forceinline__ double queue_score(int cnt, int nt, int tot, int hasw, int w, int NP, double cf,
                                              int booster, const double* lpT, const double* ffT) {
    double r = (double)cnt;                           // plan.go:664-670
    if (NP > 0) {
        const double lp = (unsigned)nt < (unsigned)kLpTab ? lpT[nt] : (double)nt / (double)NP;      // :638-644
        const double ff = (unsigned)tot < (unsigned)kFfTab ? ffT[tot] : (0.001 * (double)tot) / (double)NP;   // :647-652
        r = r + lp;
        r = r + ff;
    }
    if (hasw) {                                       // plan.go:675-684
        if (w > 0) {
            if ((w & (w - 1)) == 0) r = ldexp(r, -__builtin_ctz((unsigned)w));
            else r = r / (double)w;
        } else if (w < 0 && booster == BLANCE_BOOSTER_CBGT) {
            double b = (double)(-w);                  // control_test.go:19-26
            if (b < cf) b = cf;
            r = r + b;
        }
    }
    r = r - cf;                                       // plan.go:686
    return r;
}

// the compiler's divergence analysis gives up on values carried around the step loop: say that they are wave uniform
__device__ __forceinline__ unsigned long long uni64(unsigned long long v) {
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
}

struct QMin { unsigned hi, lo; int node; };

// minimum of (key, node) over the wave (lanes that pass ~0 / INT_MAX never win): three v_min_u32 DPP chains
__device__ __forceinline__ QMin wave_min_key_node(unsigned long long key, int node) {
    QMin r;
    const unsigned hi = (unsigned)(key >> 32), lo = (unsigned)key;
    r.hi = wave_min_u32_bcast(hi);
    const bool a = hi == r.hi;
    r.lo = wave_min_u32_bcast(a ? lo : kKeyNoneV);
    const bool b = a && lo == r.lo;
    r.node = (int)wave_min_u32_bcast(b ? (unsigned)node : 0x7fffffffu);
    return r;
}

}  // namespace blance
#include "k_queue_walk.h"
namespace blance {

// HELPER WAVES (round 5).  The walk is one wave; what it cannot do alone at a reasonable price is look at EVERY node: the
// dense step (a row whose entries cover the whole window: ~17 K cycles of one wave for 4,096 nodes) and the window rebuild.
// The kernel therefore runs as ONE workgroup of kQueueWaves waves -- one per SIMD of the CU, so the walk keeps the register
// file it has (a fifth wave would halve it).  Waves 1.. park on s_barrier, which costs the walk nothing; wave 0 posts a
// command in LDS, joins the barrier, every wave scans its share of the nodes out of the LDS tables (keys, row bits,
// counters -- all of them already there), leaves its k best (key, node) in LDS, second barrier, wave 0 merges.
constexpr int kQueueWaves = 4;
constexpr int kQCmdExit = 0, kQCmdDense = 1, kQCmdTopL = 2, kQCmdExact = 3, kQCmdStripe = 4, kQCmdBits = 5;
constexpr int kQStripeMin = 40;                     // entries a striped rebuild has to leave to be taken
constexpr int kQTopL = 48;                          // entries each worker selects for the window's rebuild
constexpr int kQScratch = 5632;                    // bytes the cooperative rebuild needs (aliases rowTag, which only a batch's validation uses)
constexpr int kQCmdWords = 32, kQResWords = 4;      // a command block; one (key hi, key lo, node, -) result per wave and pick

template <int KM>
__global__ __launch_bounds__(64 * kQueueWaves) void k_pass_queue(PassParams q) {
    static_assert(KM == 2, "k <= 2");
    typedef unsigned long long u64;
    constexpr int KH = 2;                            // higher priority nodes a step may carry
    BLANCE_DYN_LDS(lds);
    const int lane = threadIdx.x & 63;
    const int wave = uni((int)(threadIdx.x >> 6)), NW = uni((int)(blockDim.x >> 6));
    const int N = q.N, NX = q.NX, M = q.M, L = q.L, NP = q.NP, s = q.s, k = q.k, RW = q.RW;
    const int SW = 1 + L;
    const int G = (NX + 63) >> 6, NXp = G << 6, BW = ((NXp >> 5) + 3) & ~3;     // words per row bit map (16-byte rows)
    const int OWs = q.OW;

    u64* gB = (u64*)lds;                             // [NXp] sortable image of g; ~0: no candidate
    int* cntL = (int*)(gB + NXp);                    // [NXp] stateNodeCounts[s]
    int* totL = cntL + NXp;                          // [NXp] nodePartitionCounts (plan.go:118-124)
    int* wL = totL + NXp;                            // [NXp] node weights
    double* lpT = (double*)(wL + NXp);               // [kLpTab] c / NP
    double* ffT = lpT + kLpTab;                      // [kFfTab] (0.001 * t) / NP
    int* recS = (int*)(ffT + kFfTab);                // [64 * RW] step records of the batch
    int* outS = recS + 64 * RW;                      // [64 * (KM + 1)] the batch's outputs
    unsigned* bitsL = (unsigned*)(outS + 64 * (KM + 1));   // [64 * BW] row bit maps of the batch's steps
    unsigned char* flL = (unsigned char*)(bitsL + 64 * BW);       // [NXp] 1: in nodesNext, 2: has a weight
    unsigned char* rowTag = flL + NXp;               // [NXp + 64] a lane of the batch with this row (any of them)
    const int RT = NXp + 64 > kQScratch ? NXp + 64 : kQScratch;
    unsigned char* shL = rowTag + RT;                // [NXp] e when the node's score is divided by 2^e (no weight, weight 0: e = 0)
    unsigned short* ntL = (unsigned short*)(shL + NXp);      // [NXp] folded mode: row "" of nodeToNodeCounts
    // the cooperative rebuild's scratch, over rowTag (no rebuild runs between a batch's writes and reads of rowTag)
    u64* runK = (u64*)rowTag;                        // [(kQueueWaves - 1) * 64] every worker's sorted list: keys ...
    int* runN = (int*)(runK + (kQueueWaves - 1) * 64);         // ... and nodes
    u64* winK = (u64*)(runN + (kQueueWaves - 1) * 64);         // [64] the new window
    int* winN = (int*)(winK + 64);
    int* th65 = winN + 64;                           // [4] the 65th smallest entry (hi, lo, node)
    int* cntw = th65 + 4;                            // [2 * kQueueWaves] per worker: entries of its list; entries below the bound
    int* thT = cntw + 2 * kQueueWaves;               // [4] the bound T* (hi, lo, node)
    static_assert((kQueueWaves - 1) * 64 * 12 + 64 * 12 + 16 + 8 * kQueueWaves + 16 <= kQScratch, "scratch");
    int* hcmd = (int*)(ntL + NXp);                   // [kQCmdWords] wave 0's command to the helper waves
    int* hres = hcmd + kQCmdWords;                   // [kQueueWaves * KM * kQResWords] their answers
    int* rowS = hres + kQueueWaves * KM * kQResWords;   // [64] the rows of the batch's steps (kQCmdBits)

    // ---- the dense step's share of one wave: the k best (key, node) among the candidates 64 i + lane, i in this wave's
    // columns, for the step the command describes -- clean entries by their keys in LDS, entries with their bit set read
    // from the matrix unless even an entry of 1 would put them behind `bound` (an upper bound of the step's k-th pick).
    // Every wave of the workgroup runs this between the two barriers of a dense step; results in hres[wave].
    auto dense_part = [&]() {
        const int f = hcmd[1], kk = hcmd[2], o0 = hcmd[3], o1 = hcmd[4], h0 = hcmd[5], h1 = hcmd[6], hb = hcmd[7], rowf = hcmd[8];
        const u64 U = ((u64)(unsigned)hcmd[10] << 32) | (unsigned)hcmd[11];
        const int NH = NW, hw = wave;                // every wave of the workgroup takes a share, the walking wave too
        const int CJ = (G + NH - 1) / NH, ib = hw * CJ, ie = ib + CJ < G ? ib + CJ : G;
        u64 lb[KM];
        int ln[KM];
#pragma unroll
        for (int j = 0; j < KM; j++) { lb[j] = ~0ull; ln[j] = INT_MAX; }
        auto keep_local = [&](u64 b, int n) {               // the lane's own k best, ascending
            if (!qless(b, n, lb[KM - 1], ln[KM - 1])) return;       // (most nodes: not among them)
#pragma unroll
            for (int j = KM - 1; j >= 0; j--) {
                const bool here = qless(b, n, lb[j], ln[j]);
                const bool above = j > 0 && qless(b, n, lb[j - 1], ln[j - 1]);
                if (here) {
                    if (above) { lb[j] = lb[j - 1]; ln[j] = ln[j - 1]; }
                    else { lb[j] = b; ln[j] = n; }
                }
            }
        };
        if (NP > 0 && hb) {
            // One pass over the lane's nodes, no branch: the row's bits and the keys come in eights; a node that is no
            // candidate (its key is ~0, or it is one of the step's own / higher priority nodes) or has its bit set takes part
            // with the key ~0; the lane's two smallest are kept by strict less-than -- a lane's nodes ascend, so of equal keys
            // the smaller node stays in front, the order of plan.go:617-628.
            u64 b0 = ~0ull, b1 = ~0ull, dmask = 0;
            int n0 = INT_MAX, n1 = INT_MAX;
            for (int i0 = ib; i0 < ie; i0 += 8) {
                unsigned wv[8];
                u64 kv[8];
#pragma unroll
                for (int u = 0; u < 8; u++) wv[u] = i0 + u < ie ? bitsL[f * BW + 2 * (i0 + u) + (lane >> 5)] : 0u;
#pragma unroll
                for (int u = 0; u < 8; u++) kv[u] = i0 + u < ie ? gB[(i0 + u) * 64 + lane] : ~0ull;
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    // (the step's own and higher priority nodes have their bits set in this row copy -- the walking wave did
                    // that before it posted the command --, a node outside nodesNext has the key ~0: no test per node)
                    const int n = (i0 + u) * 64 + lane;
                    const bool d = ((wv[u] >> (lane & 31)) & 1u) != 0;
                    if (d) dmask |= 1ull << (i0 + u - ib);
                    const u64 key = d ? ~0ull : kv[u];
                    const bool lt0 = key < b0;
                    if (kk > 1) {                       // (wave uniform; k = 1 keeps one)
                        const bool lt1 = key < b1;
                        b1 = lt0 ? b0 : (lt1 ? key : b1);
                        n1 = lt0 ? n0 : (lt1 ? n : n1);
                    }
                    b0 = lt0 ? key : b0;
                    n0 = lt0 ? n : n0;
                }
            }
            lb[0] = b0; ln[0] = n0;
            static_assert(KM == 2, "the lane keeps a pair");
            lb[1] = b1; ln[1] = n1;
            // The wave's k best CLEAN nodes first: the k-th of them bounds the step's k-th pick from above (so does U, the k-th
            // of the step's own nodes), and an entry with its bit set is out of the race when even an entry of 1 (plan.go:638-644
            // is monotone in the entry) puts it behind that bound -- in the tie regime all of them are, and nobody reads the
            // matrix.  (Round 5's first form bounded by U and the LANE's clean keys only: U carries the own node's entry, which
            // stays have raised, and hundreds of set entries per row went to the matrix, a round trip each.)
            u64 rK[KM];
            int rN[KM];
            {
                u64 tb[KM];
                int tn[KM];
#pragma unroll
                for (int j = 0; j < KM; j++) { tb[j] = lb[j]; tn[j] = ln[j]; rK[j] = ~0ull; rN[j] = INT_MAX; }
                for (int j = 0; j < kk; j++) {
                    const QMin m = wave_min_key_node(tb[0], tn[0]);
#pragma unroll
                    for (int e = 0; e < KM; e++) if (e == j) { rK[e] = m.node == INT_MAX ? ~0ull : (((u64)m.hi << 32) | m.lo); rN[e] = m.node; }
                    if (tn[0] == m.node) {
#pragma unroll
                        for (int e = 0; e + 1 < KM; e++) { tb[e] = tb[e + 1]; tn[e] = tn[e + 1]; }
                        tb[KM - 1] = ~0ull; tn[KM - 1] = INT_MAX;
                    }
                }
            }
            const u64 cK = kk > 1 ? rK[1] : rK[0];
            const u64 bound = U < cK ? U : cK;
            bool read_any = false;
            for (u64 dd = dmask; dd; dd &= dd - 1) {
                const int n = (ib + __ffsll((long long)dd) - 1) * 64 + lane;
                if (gB[n] > bound || n >= N || !(flL[n] & 1) || n == o0 || n == o1 || n == h0 || n == h1) continue;
                if (shL[n] != 255) {
                    const int tt = totL[n];
                    double r = (double)cntL[n];
                    r = r + lpT[1];
                    r = r + ((unsigned)tt < (unsigned)kFfTab ? ffT[tt] : (0.001 * (double)tt) / (double)NP);
                    r = ldexp(r, -(int)shL[n]);
                    r = r - 0.0;
                    if (sortable_bits(r) > bound) continue;
                }
                const int nt = BLANCE_QLD(q.ntn + (size_t)rowf * N + n);
                const u64 b = nt ? sortable_bits(queue_score(cntL[n], nt, totL[n], (flL[n] >> 1) & 1, wL[n], NP, 0.0,
                                                             q.booster_kind, lpT, ffT)) : gB[n];
                keep_local(b, n);
                read_any = true;
            }
            if (__ballot(read_any) == 0) {           // the common case: the clean minima are the wave's answer
                if (lane == 0) {
#pragma unroll
                    for (int j = 0; j < KM; j++) {
                        if (j < kk) {
                            int* r = hres + (hw * KM + j) * kQResWords;
                            r[0] = (int)(unsigned)(rK[j] >> 32); r[1] = (int)(unsigned)rK[j]; r[2] = rN[j];
                        }
                    }
                }
                return;
            }
        } else {
            // (no bit map for this step -- its row was bumped inside the batch -- or no NumPartitions terms at all: the row
            // itself, eight entries of the lane in flight at a time)
            for (int i0 = ib; i0 < ie; i0 += 8) {
                int ntv[8];
                bool cv[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int n = (i0 + u) * 64 + lane;
                    cv[u] = i0 + u < ie && n < N && (flL[n] & 1) && n != o0 && n != o1 && n != h0 && n != h1;
                    ntv[u] = (NP > 0 && cv[u]) ? BLANCE_QLD(q.ntn + (size_t)rowf * N + n) : 0;
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int n = (i0 + u) * 64 + lane;
                    if (cv[u]) {
                        const u64 b = ntv[u] ? sortable_bits(queue_score(cntL[n], ntv[u], totL[n], (flL[n] >> 1) & 1, wL[n], NP, 0.0,
                                                                         q.booster_kind, lpT, ffT)) : gB[n];
                        keep_local(b, n);
                    }
                }
            }
        }
        // the wave's k best, ascending, into its slots
        for (int j = 0; j < kk; j++) {
            const QMin m = wave_min_key_node(lb[0], ln[0]);
            if (lane == 0) {
                int* r = hres + (hw * KM + j) * kQResWords;
                r[0] = (int)m.hi; r[1] = (int)m.lo; r[2] = m.node;
            }
            if (ln[0] == m.node) {
#pragma unroll
                for (int e = 0; e + 1 < KM; e++) { lb[e] = lb[e + 1]; ln[e] = ln[e + 1]; }
                lb[KM - 1] = ~0ull; ln[KM - 1] = INT_MAX;
            }
        }
    };
    // ---- the cooperative rebuild (the helper waves between the barriers of a kQCmdTopL command).  Worker w owns a third of the
    // columns and selects the kQTopL smallest (key, node) of ITS nodes one by one -- the successive minima of round 4's exact
    // selection, but 32 of them on each of three waves side by side instead of 65 on one.  With t_w the last of worker w's list
    // (a worker that ran out of nodes has none), T* = the smallest t_w: a node of worker w that is not in w's list is >= t_w >= T*,
    // so EVERY node below T* is in one of the lists.  The window = the (at most 64) smallest list entries below T*, THETA = the
    // 65th of them, else T* -- the invariant, exactly, and never fewer than kQTopL - 1 entries while there are that many nodes
    // (in practice nearly always all 64: three lists of 32 reach about the 85th smallest node).  The lists are sorted by
    // construction; an entry's rank among all of them is its place in its own list plus, by binary search, the entries of the
    // other two in front of it.  (Round 5's first cooperative form -- every lane's three smallest nodes, the window cut at the
    // first (wave, lane) cell that holds three of the 65 smallest -- came out with fewer than 40 entries more than half of the
    // time on scattered keys and then fell back to the 73 K-cycle selection on one wave; this form has no fallback.)
    auto topl_part = [&]() {
        const int NH = NW - 1, hw = wave - 1;
        // worker hw owns the columns hw, hw + NH, hw + 2 NH, ...: blocks of 64 consecutive node ids go round the workers, so the
        // workers' lists reach about equally deep into the order whatever the ids of the lowest nodes (contiguous thirds of the id
        // range did not: with the lowest nodes in one third T* was that worker's 32nd entry and the window came out half full)
        const int CJ = (G - hw + NH - 1) / NH;      // columns of this worker (local index j: column hw + j NH)
        constexpr int RC = 4;
        u64 taken = 0;                               // bit j: node 64 (hw + j NH) + lane is in the list already
        u64 ca[RC];                                  // the lane's four smallest untaken keys, ascending ...
        int cm[RC];                                  // ... and their nodes: a column is scanned again only when all four are gone
        bool exhausted = false;
        auto scan4 = [&]() {
#pragma unroll
            for (int j = 0; j < RC; j++) { ca[j] = ~0ull; cm[j] = INT_MAX; }
            for (int j0 = 0; j0 < CJ; j0 += 8) {     // (8 independent LDS reads at a time)
                u64 kv[8];
#pragma unroll
                for (int u = 0; u < 8; u++) kv[u] = j0 + u < CJ ? gB[(hw + (j0 + u) * NH) * 64 + lane] : ~0ull;
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    u64 v = kv[u];
                    int n = (hw + (j0 + u) * NH) * 64 + lane;
                    if (j0 + u >= CJ || ((taken >> (j0 + u)) & 1) || v == ~0ull) continue;
                    bool placed = false;             // ascending columns: ties keep the lower node in front; what follows moves down
#pragma unroll
                    for (int j = 0; j < RC; j++) {
                        if (placed || v < ca[j]) { const u64 tv = ca[j]; const int tn = cm[j]; ca[j] = v; cm[j] = n; v = tv; n = tn; placed = true; }
                    }
                }
            }
            if (cm[0] == INT_MAX) exhausted = true;
        };
        scan4();
        int found = 0;
        u64 myK = ~0ull;                             // lane e keeps the e-th of the list
        int myN = INT_MAX;
        for (int e = 0; e < kQTopL; e++) {
            const QMin m = wave_min_key_node(ca[0], cm[0]);
            if (m.node == INT_MAX) break;            // the worker's nodes are used up
            if (lane == e) { myK = ((u64)m.hi << 32) | m.lo; myN = m.node; }
            found = e + 1;
            if (cm[0] == m.node) {
                taken |= 1ull << (((m.node >> 6) - hw) / NH);
#pragma unroll
                for (int j = 0; j + 1 < RC; j++) { ca[j] = ca[j + 1]; cm[j] = cm[j + 1]; }
                ca[RC - 1] = ~0ull; cm[RC - 1] = INT_MAX;
            }
            const bool dry = cm[0] == INT_MAX && !exhausted;
            if (__ballot(dry)) { if (dry) scan4(); }
        }
        found = uni(found);
        runK[hw * 64 + lane] = myK;                  // (lanes >= found: (~0, INT_MAX), behind everything)
        runN[hw * 64 + lane] = myN;
        if (lane == 0) cntw[hw] = found;
        lds_barrier();                               // (the walking wave passes this one too)
        u64 tk = ~0ull;                              // T*: the smallest last entry of the full lists
        int tn = INT_MAX;
        for (int w2 = 0; w2 < NH; w2++) {
            if (cntw[w2] < kQTopL) continue;         // (that worker's list holds all of its nodes: no bound from it)
            const u64 xk = runK[w2 * 64 + kQTopL - 1];
            const int xn = runN[w2 * 64 + kQTopL - 1];
            if (qless(xk, xn, tk, tn)) { tk = xk; tn = xn; }
        }
        int rank = lane;
        for (int w2 = 0; w2 < NH; w2++) {
            if (w2 == hw) continue;
            const u64* rk = runK + w2 * 64;
            const int* rn = runN + w2 * 64;
            int pos = 0;                             // entries of list w2 in front of this one: lower bound over 64 sorted slots
#pragma unroll
            for (int st = 32; st >= 1; st >>= 1) pos += qless(rk[pos + st - 1], rn[pos + st - 1], myK, myN) ? st : 0;
            pos += qless(rk[pos], rn[pos], myK, myN) ? 1 : 0;
            rank += pos;
        }
        const bool in = myN != INT_MAX && (tn == INT_MAX || qless(myK, myN, tk, tn));
        const int cw = __popcll(__ballot(in));
        if (in && rank < 64) { winK[rank] = myK; winN[rank] = myN; }
        if (in && rank == 64) { th65[0] = (int)(unsigned)(myK >> 32); th65[1] = (int)(unsigned)myK; th65[2] = myN; }
        BLANCE_WAVE_SYNC();
        if (lane == 0) cntw[kQueueWaves + hw] = cw;  // (cntw[0 .. NH) are still read by the other workers: the counts go behind them)
        if (hw == 0 && lane == 0) { thT[0] = (int)(unsigned)(tk >> 32); thT[1] = (int)(unsigned)tk; thT[2] = tn; }     // the bound T*
    };
    // ---- the exact selection (worker 0 alone; a test knob since topl_part: an independent form of the same rebuild): 64 + 1 successive minima of the keys
    // in LDS; lane l owns nodes l, 64 + l, ... (bit i of `taken`: node 64 i + l) and keeps its four smallest untaken keys, so
    // that a column is scanned again only when all four are gone (64 minima over 64 columns: a column with five is rare).
    // Results: winK / winN, cntw[0] = entries, th65 = THETA ((~0, INT_MAX): fewer than 65 candidates).
    auto exact_part = [&]() {
        constexpr int RC = 4;
        u64 taken = 0;
        int wc = 0;
        u64 tK = ~0ull;
        int tN = INT_MAX;
        u64 ca[RC];
        int cm[RC];
        bool exhausted = false;
        auto scan4 = [&]() {
#pragma unroll
            for (int j = 0; j < RC; j++) { ca[j] = ~0ull; cm[j] = INT_MAX; }
            for (int i0 = 0; i0 < G; i0 += 8) {      // (8 independent LDS reads at a time)
                u64 kv[8];
#pragma unroll
                for (int u = 0; u < 8; u++) kv[u] = i0 + u < G ? gB[(i0 + u) * 64 + lane] : ~0ull;
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    u64 v = kv[u];
                    int n = (i0 + u) * 64 + lane;
                    if (((taken >> (i0 + u)) & 1) || v == ~0ull) continue;
                    bool placed = false;             // ascending i: ties keep the lower node in front; what follows moves down
#pragma unroll
                    for (int j = 0; j < RC; j++) {
                        if (placed || v < ca[j]) { const u64 tv = ca[j]; const int tn = cm[j]; ca[j] = v; cm[j] = n; v = tv; n = tn; placed = true; }
                    }
                }
            }
            if (cm[0] == INT_MAX) exhausted = true;
        };
        scan4();
        for (int e = 0; e <= 64; e++) {
            const QMin m = wave_min_key_node(ca[0], cm[0]);
            if (m.node == INT_MAX) break;            // fewer than 65 candidates: THETA stays infinite
            const u64 mk = ((u64)m.hi << 32) | m.lo;
            if (e < 64) {
                if (lane == 0) { winK[e] = mk; winN[e] = m.node; }
                wc = e + 1;
            } else { tK = mk; tN = m.node; }
            if (cm[0] == m.node) {
                taken |= 1ull << (m.node >> 6);
#pragma unroll
                for (int j = 0; j + 1 < RC; j++) { ca[j] = ca[j + 1]; cm[j] = cm[j + 1]; }
                ca[RC - 1] = ~0ull; cm[RC - 1] = INT_MAX;
            }
            const bool dry = cm[0] == INT_MAX && !exhausted;
            if (__ballot(dry)) { if (dry) scan4(); }
        }
        if (lane == 0) { cntw[0] = wc; th65[0] = (int)(unsigned)(tK >> 32); th65[1] = (int)(unsigned)tK; th65[2] = tN; }
    };
    // ---- the striped rebuild (worker 0 alone; folded batches): the window = every lane's smallest (key, node) below
    // THETA' = the smallest of the lanes' SECOND smallest -- exactly the nodes below THETA' (a node outside is a lane's minimum
    // >= THETA', or lies behind its lane's second) --, ranked by comparison against the 64 candidates in LDS.  One pass over the
    // keys and 64 compares per lane instead of 48 successive wave minima on three waves: it comes out full when the smallest keys
    // sit in different lanes -- many nodes of one load with consecutive ids, the regime in which a folded batch takes the
    // window's front step after step and drains it every 64 steps -- and short otherwise (the caller then asks for topl_part).
    // Results as exact_part's: winK / winN, cntw[0] = entries, th65 = THETA ((~0, INT_MAX): everything is in the window).
    auto stripe_part = [&]() {
        u64 m1 = ~0ull, m2 = ~0ull;
        int n1 = INT_MAX, n2 = INT_MAX;
        for (int i0 = 0; i0 < G; i0 += 8) {          // (8 independent LDS reads at a time; a lane's nodes ascend: ties keep the lower node in front)
            u64 kv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) kv[u] = i0 + u < G ? gB[(i0 + u) * 64 + lane] : ~0ull;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const u64 v = kv[u];
                const int n = (i0 + u) * 64 + lane;
                const bool lt1 = v < m1, lt2 = v < m2;
                m2 = lt1 ? m1 : (lt2 ? v : m2);
                n2 = lt1 ? n1 : (lt2 ? n : n2);
                m1 = lt1 ? v : m1;
                n1 = lt1 ? n : n1;
            }
        }
        if (m1 == ~0ull) n1 = INT_MAX;               // (no candidate in this lane)
        if (m2 == ~0ull) n2 = INT_MAX;
        const QMin t = wave_min_key_node(m2, n2);
        const u64 tK = t.node == INT_MAX ? ~0ull : (((u64)t.hi << 32) | t.lo);
        const int tN = t.node;
        const bool in = n1 != INT_MAX && (tN == INT_MAX || qless(m1, n1, tK, tN));
        runK[lane] = in ? m1 : ~0ull;
        runN[lane] = in ? n1 : INT_MAX;
        BLANCE_WAVE_SYNC();
        int rank = 0;
        for (int j0 = 0; j0 < 64; j0 += 8) {
            u64 ck[8];
            int cn[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { ck[u] = runK[j0 + u]; cn[u] = runN[j0 + u]; }
#pragma unroll
            for (int u = 0; u < 8; u++) rank += qless(ck[u], cn[u], m1, n1) ? 1 : 0;
        }
        if (in) { winK[rank] = m1; winN[rank] = n1; }
        const int cw = __popcll(__ballot(in));
        if (lane == 0) { cntw[0] = cw; th65[0] = (int)(unsigned)(tK >> 32); th65[1] = (int)(unsigned)tK; th65[2] = tN; }
    };
    // ---- the row bit maps of a batch's steps into LDS (round 6; the helper waves, while wave 0 tags dirty rows and scores the own
    // nodes): helper h takes every third pair of rows; lanes 0..31 / 32..63 copy one 4 BW-byte row each per round (16-byte loads
    // past the L1, all of a helper's rounds in flight at once).  Wave 0 has waited for the last batch's bumps before it posted.
    auto bits_part = [&]() {
        const int NH = NW - 1, hw = wave - 1;
        const int Bb = uni(hcmd[1]);
        const int BQ = BW >> 2, c = lane & 31, half = lane >> 5;
        qv4 va[8], vb[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int g = hw + NH * u, r2 = 2 * g;
            const int myrow = (g < 32 && r2 + half < Bb) ? rowS[r2 + half] : 0;
            va[u] = q_load_row16((const qv4*)q.ntn_bits + (size_t)((g < 32 && r2 + half < Bb && c < BQ) ? myrow : 0) * BQ + (c < BQ ? c : 0));
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int g = hw + NH * (8 + u), r2 = 2 * g;
            const int myrow = (g < 32 && r2 + half < Bb) ? rowS[r2 + half] : 0;
            vb[u] = q_load_row16((const qv4*)q.ntn_bits + (size_t)((g < 32 && r2 + half < Bb && c < BQ) ? myrow : 0) * BQ + (c < BQ ? c : 0));
        }
        BLANCE_QROWS_ARRIVED(va);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int g = hw + NH * u, rr = 2 * g + half;
            if (g < 32 && rr < Bb && c < BQ) ((qv4*)bitsL)[rr * BQ + c] = va[u];
        }
        BLANCE_QROWS_ARRIVED(vb);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int g = hw + NH * (8 + u), rr = 2 * g + half;
            if (g < 32 && rr < Bb && c < BQ) ((qv4*)bitsL)[rr * BQ + c] = vb[u];
        }
    };
    if (wave != 0) {
        // ---- a helper wave: wait for a command, do its share, wait again
        for (;;) {
            lds_barrier();                           // (1) the command is posted, the tables are as the step sees them
            const int op = uni(hcmd[0]);
            if (op == kQCmdExit) break;
            if (op == kQCmdDense) dense_part();
            if (op == kQCmdTopL) topl_part();
            if (op == kQCmdExact && wave == 1) exact_part();
            if (op == kQCmdStripe && wave == 1) stripe_part();
            if (op == kQCmdBits) bits_part();
            lds_barrier();                           // (2) the answers are in
        }
        return;
    }

    for (int i = lane; i < kLpTab; i += 64) lpT[i] = NP > 0 ? (double)i / (double)NP : 0.0;
    for (int i = lane; i < kFfTab; i += 64) ffT[i] = NP > 0 ? (0.001 * (double)i) / (double)NP : 0.0;
    BLANCE_WAVE_SYNC();
    // Folded mode (as in k_pass_tree): while every step of a batch is a partition that holds nothing in this state and has
    // no top priority node -- the steps all read and bump row "" of nodeToNodeCounts (plan.go:134-138, :238-245) -- that
    // row lives in LDS and is part of the window keys: a candidate's exact score is its key, the matrix is not read.
    bool fold = false;
    auto gkey = [&](int n) -> u64 {
        return (flL[n] & 1) ? sortable_bits(queue_score(cntL[n], fold ? (int)ntL[n] : 0, totL[n], (flL[n] >> 1) & 1, wL[n], NP, 0.0,
                                                        q.booster_kind, lpT, ffT)) : ~0ull;
    };
    bool odd_weight = false;
    for (int i = 0; i < G; i++) {
        const int n = i * 64 + lane;
        int c = 0, t = 0, w = 0, fl = 0;
        if (n < NX) {
            c = q.cnt[s * NX + n];
            for (int tt = 0; tt <= M; tt++) t += q.cnt[tt * NX + n];
            w = q.node_weight[n];
            fl = ((n < N && q.alive[n]) ? 1 : 0) | (q.node_has_weight[n] ? 2 : 0);
        }
        cntL[n] = c; totL[n] = t; wL[n] = w; flL[n] = (unsigned char)fl;
        // the lean walk divides by a power of two only (plan.go:675-684; node weights 1 / 2 / 4 ... and none at all)
        int sh = 0;
        if ((fl & 2) && w > 0) sh = (w & (w - 1)) == 0 ? __builtin_ctz((unsigned)w) : 255;
        if ((fl & 2) && w < 0 && q.booster_kind != BLANCE_BOOSTER_NONE) sh = 255;
        shL[n] = (unsigned char)sh;
        if ((fl & 1) && sh == 255) odd_weight = true;
    }
    u64 alivecol = 0;                                // bit i: node 64 i + lane is in nodesNext (and inside nodesAll)
    for (int i = 0; i < G; i++) if (i * 64 + lane < N && (flL[i * 64 + lane] & 1)) alivecol |= 1ull << i;
    const bool lean_ok = __ballot(odd_weight) == 0 && !(q.spec & 8);      // (q.spec & 8: test knob, every step through the general code)
    BLANCE_WAVE_SYNC();
    for (int i = 0; i < G; i++) gB[i * 64 + lane] = gkey(i * 64 + lane);
    BLANCE_WAVE_SYNC();

    // ---- the window: lane i = the i-th smallest (g, node); lanes >= wcnt hold (~0, INT_MAX)
    u64 wk = ~0ull;
    int wn = INT_MAX;
    int wcnt = 0;
    u64 thK = ~0ull;                                 // THETA: every node outside the window has (g, node) >= it
    int thN = INT_MAX;
    long long n_rebuild = 0;
#ifdef BLANCE_PHASE_PROF
    long long rb_cycles = 0, rb_scans = 0, rb_scan_cycles = 0;
#endif
    int rb_wcnt = 0;                                 // entries the last rebuild left: a window that has not drained since is not rebuilt again
    long long n_coop = 0, n_stripe = 0;
    int stripe_skip = 0;                             // rebuilds that do not try the striped form (the last try came out short)
    int dense_streak = 0, dense_probe = 0;           // general steps in a row that ended in the dense scan; steps that went there directly
    // The window is rebuilt by the helper waves (topl_part): this wave posts the command, passes the barriers, and loads the
    // new window out of LDS.
    auto rebuild = [&]() {
#ifdef BLANCE_PHASE_PROF
        const long long rb_t0 = clock64();
#endif
        bool striped = false;
        if (fold && stripe_skip == 0 && !(q.spec & 64) && NW > 1) {
            // a folded batch drains the window from its front: try the one-pass striped form first (stripe_part)
            if (lane == 0) hcmd[0] = kQCmdStripe;
            lds_barrier();
            lds_barrier();
            const int wc = uni(cntw[0]);
            const int tn = uni(th65[2]);
            if (wc >= kQStripeMin || tn == INT_MAX) {
                wk = lane < wc ? winK[lane] : ~0ull;
                wn = lane < wc ? winN[lane] : INT_MAX;
                wcnt = wc;
                thK = uni64(((u64)(unsigned)th65[0] << 32) | (unsigned)th65[1]);
                thN = tn;
                striped = true;
                n_stripe++;
            } else stripe_skip = 16;                 // (the smallest keys share lanes: the next rebuilds do not try)
            BLANCE_WAVE_SYNC();
        } else if (stripe_skip > 0) stripe_skip--;
        if (striped) {
        } else if (!(q.spec & 64)) {
            if (lane == 0) hcmd[0] = kQCmdTopL;
            lds_barrier();                           // (1) posted
            lds_barrier();                           //     (the workers' own: their lists are in LDS)
            lds_barrier();                           // (2) done
            int cw = 0;
            for (int w2 = 0; w2 < NW - 1; w2++) cw += cntw[kQueueWaves + w2];
            cw = uni(cw);
            const int tn = uni(thT[2]);
            const int wc = cw < 64 ? cw : 64;
            wk = lane < wc ? winK[lane] : ~0ull;
            wn = lane < wc ? winN[lane] : INT_MAX;
            wcnt = uni(wc);
            if (cw > 64) { thK = uni64(((u64)(unsigned)th65[0] << 32) | (unsigned)th65[1]); thN = uni(th65[2]); }
            else { thK = uni64(((u64)(unsigned)thT[0] << 32) | (unsigned)thT[1]); thN = tn; }
            n_coop++;
        } else {                                     // (test knob: the exact selection of 65 minima on one helper wave)
            if (lane == 0) hcmd[0] = kQCmdExact;
            lds_barrier();
            lds_barrier();
            const int wc = uni(cntw[0]);
            wk = lane < wc ? winK[lane] : ~0ull;
            wn = lane < wc ? winN[lane] : INT_MAX;
            wcnt = wc;
            thK = uni64(((u64)(unsigned)th65[0] << 32) | (unsigned)th65[1]);
            thN = uni(th65[2]);
        }
        BLANCE_WAVE_SYNC();                          // (the scratch is rowTag's again after this)
        rb_wcnt = wcnt;
        n_rebuild++;
        (void)n_stripe;
#ifdef BLANCE_PHASE_PROF
        rb_cycles += clock64() - rb_t0;
#endif
    };
    rebuild();
    // remove node x from the window if it is there; insert (key, x) if below THETA
    auto win_remove = [&](int x) {
        const u64 hit = __ballot(wn == x);
        if (hit) {
            const int r = __ffsll((long long)hit) - 1;
            const int nh = wave_from_next((int)(unsigned)(wk >> 32), (int)kKeyNoneV);
            const int nl = wave_from_next((int)(unsigned)wk, (int)kKeyNoneV);
            const int nn = wave_from_next(wn, INT_MAX);
            if (lane >= r) { wk = ((u64)(unsigned)nh << 32) | (unsigned)nl; wn = nn; }
            wcnt = uni(wcnt - 1);
        }
    };
    auto win_insert = [&](u64 key, int x) {
        if (!qless(key, x, thK, thN)) return;        // stays outside: still >= THETA
        const u64 before = __ballot(lane < wcnt && qless(wk, wn, key, x));      // sorted: a prefix of the lanes
        const int p = __popcll(before);
        if (p >= 64) { thK = uni64(key); thN = uni(x); return; }   // full, and behind every entry: it is the bound now
        if (wcnt == 64) {                            // the largest entry falls off and becomes THETA
            const unsigned h = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(wk >> 32), 63);
            const unsigned l = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)wk, 63);
            thK = ((u64)h << 32) | l;
            thN = __builtin_amdgcn_readlane(wn, 63);
            wcnt = uni(wcnt - 1);
        }
        const int ph = wave_from_prev((int)(unsigned)(wk >> 32), (int)kKeyNoneV);
        const int pl = wave_from_prev((int)(unsigned)wk, (int)kKeyNoneV);
        const int pn = wave_from_prev(wn, INT_MAX);
        if (lane > p) { wk = ((u64)(unsigned)ph << 32) | (unsigned)pl; wn = pn; }
        if (lane == p) { wk = key; wn = x; }
        wcnt = uni(wcnt + 1);
    };

    PH_DECL;
    long long n_bulk = 0, n_moved = 0, n_exact = 0, n_dense = 0, n_bound = 0;
#ifndef BLANCE_SIMT_EMU
    // the assembly walk (k_queue_walk.h): k = 2, NumPartitions > 0, power-of-two node weights; q.spec & 32: test knob, never
    const bool walk_asm = lean_ok && (k == 1 || k == 2) && NP > 0 && KM == 2 && !(q.spec & 32) && NXp <= 4096;
    const int cfa = (int)((((unsigned)(size_t)cntL) >> 2) | ((((unsigned)(size_t)totL) >> 2) << 16));
    const int cfb = (int)((((unsigned)(size_t)shL) >> 2) | ((((unsigned)(size_t)ffT) >> 2) << 16));
    const int cfc = (int)((((unsigned)(size_t)bitsL) >> 2) | ((unsigned)(BW * 4) << 16));
    int cfd = (int)(((unsigned)(size_t)gB) >> 2);   // | B << 16, per batch
    const int cfe = (int)((((unsigned)(size_t)ntL) >> 2) | ((((unsigned)(size_t)lpT) >> 2) << 16));
    int mo1 = 0, mo2 = 0;                            // output nodes of the lanes the assembly walk moved
    const u64 lp_one = uni64((u64)__double_as_longlong(NP > 0 ? (double)1 / (double)NP : 0.0));      // = lpT[1], as bits
#endif
    int stop_pos = q.end, stop_why = kQStopNone;

    PH(11);
    constexpr int kRecPre = 16;                      // step records of up to 16 words are fetched a batch ahead
    int pre[kRecPre];
    {
        const int B0 = q.end - q.beg < 64 ? q.end - q.beg : 64;
#pragma unroll
        for (int r = 0; r < kRecPre; r++) {
            const int idx = r * 64 + lane;
            pre[r] = (r < RW && RW <= kRecPre && idx < B0 * RW) ? q.rec[(size_t)q.beg * RW + idx] : 0;
        }
    }
    if (RW <= kRecPre) {                             // the first batch's records into LDS
        const int B0 = q.end - q.beg < 64 ? q.end - q.beg : 64;
#pragma unroll
        for (int r = 0; r < kRecPre; r++) {
            const int idx = r * 64 + lane;
            if (r < RW && idx < B0 * RW) recS[idx] = pre[r];
        }
    }
    for (int oi = q.beg; ; oi += 64) {
        oi = uni(oi); wcnt = uni(wcnt); stop_why = uni(stop_why);
        if (oi >= q.end || stop_why != kQStopNone) break;
        const int B = q.end - oi < 64 ? q.end - oi : 64;
        // The batch's records are in LDS already (staged when the last batch ended, out of registers that were loaded while it
        // was walked): the decoding below reads LDS only and runs while the last batch's bumps of nodeToNodeCounts and its
        // bit maps are still on their way -- they are waited for where the first read of the matrix is issued.
        if (RW > kRecPre) {
            BLANCE_QWAIT_BUMPS();
            for (int r = 0; r < RW; r++) {
                const int idx = r * 64 + lane;
                if (idx < B * RW) recS[idx] = q.rec[(size_t)oi * RW + idx];
            }
        }
        BLANCE_WAVE_SYNC();

        PH(0);
        // ---- lane j looks at step oi + j
        const bool act = lane < B;
        const int* rj = recS + (act ? lane : 0) * RW;
        int row = NX;
        const int wj = rj[1];
        int ownv[KM], ntn_own[KM], hv[KH];
        u64 oK[KM];
#pragma unroll
        for (int j = 0; j < KM; j++) { ownv[j] = -1; ntn_own[j] = 0; oK[j] = ~0ull; }
#pragma unroll
        for (int j = 0; j < KH; j++) hv[j] = -1;
        const double vstick = __hiloint2double(rj[3], rj[2]);
        int nown = 0;
        // simple: at most k own nodes, all candidates, held once and in no other list; at most KH higher priority
        // nodes; no node in a lower priority list that could be promoted is ... (checked when it is taken)
        bool simple = act;
        bool has_other = false;                      // the partition holds nodes in lower priority states
        int ov0 = -1, ov1 = -1, n_o = 0;             // ... the first two of them (a taken one would be promoted: not the lean walk's business)
        {
            const int hT = rj[kRecHead + q.top_state * SW];
            if ((hT >> 16) != kListAbsent && (hT & 0xffff) > 0) row = rj[kRecHead + q.top_state * SW + 1];   // plan.go:134-138
            const int hs = rj[kRecHead + s * SW];
            nown = (hs >> 16) == kListAbsent ? 0 : (hs & 0xffff);
            if (nown > k) { simple = false; nown = 0; }
#pragma unroll
            for (int j = 0; j < KM; j++) {
                if (j < nown) {
                    const int o = rj[kRecHead + s * SW + 1 + j];
                    ownv[j] = o;
                    if (o >= N || !(flL[o < NXp ? o : 0] & 1)) simple = false;
#pragma unroll
                    for (int jj = 0; jj < KM; jj++) if (jj < j && ownv[jj] == o) simple = false;
                }
            }
            int n_h = 0;
            for (int t = 0; t < M; t++) {
                if (t == s) continue;
                const int h = rj[kRecHead + t * SW];
                if ((h >> 16) == kListAbsent) continue;
                const bool higher = (q.higher_mask >> t) & 1;
                for (int jj = 0; jj < (h & 0xffff); jj++) {
                    const int x = rj[kRecHead + t * SW + 1 + jj];
#pragma unroll
                    for (int j = 0; j < KM; j++) if (ownv[j] == x) simple = false;   // excluded or demoted: not a plain step
                    if (higher) {
                        if (n_h >= KH) simple = false;
#pragma unroll
                        for (int e = 0; e < KH; e++) if (e == n_h) hv[e] = x;
                        n_h++;
                    } else {
                        has_other = true;
                        if (n_o == 0) ov0 = x;
                        if (n_o == 1) ov1 = x;
                        n_o++;
                    }
                }
            }
            if (!simple) {
                nown = 0;
#pragma unroll
                for (int j = 0; j < KM; j++) ownv[j] = -1;
            }
        }
        PH(1);
        BLANCE_QWAIT_BUMPS();                        // earlier bumps of nodeToNodeCounts / its bit maps are done before the reads below
        if (RW <= kRecPre) {                         // the next batch's records: on their way while this one is walked
            const int on = oi + 64, Bn = q.end - on < 64 ? q.end - on : 64;
#pragma unroll
            for (int r = 0; r < kRecPre; r++) {
                const int idx = r * 64 + lane;
                pre[r] = (r < RW && on < q.end && idx < Bn * RW) ? q.rec[(size_t)on * RW + idx] : 0;
            }
        }
        // the own nodes' entries of the step's row: issued here so that they travel with the row bit maps below (one round
        // trip to the L2 for both; a lane whose row an earlier step of the batch bumps reads them again at its turn)
        if (NP > 0) {
#pragma unroll
            for (int j = 0; j < KM; j++)
                if (simple && j < nown) ntn_own[j] = BLANCE_QLD(q.ntn + (size_t)row * N + ownv[j]);
        }
        // ---- folded mode on / off
        {
            const bool want = NP > 0 && lean_ok && __ballot(act && !(simple && row == NX && nown == 0 && n_o <= 2 && hv[1] < 0)) == 0;
            if (want != fold) {
                bool can = true;
                if (want) {                          // row "" into LDS (16-bit entries; a larger one: no folding)
                    bool big = false;
                    for (int i = 0; i < G; i++) {
                        const int n = i * 64 + lane;
                        const int v = n < N ? BLANCE_QLD(q.ntn + (size_t)NX * N + n) : 0;
                        if (v >= 60000) big = true;
                        ntL[n] = (unsigned short)(v < 60000 ? v : 0);
                    }
                    can = __ballot(big) == 0;
                }
                if (can) {
                    fold = want;
                    BLANCE_WAVE_SYNC();
                    for (int i = 0; i < G; i++) gB[i * 64 + lane] = gkey(i * 64 + lane);
                    BLANCE_WAVE_SYNC();
                    rebuild();
                }
            }
        }
        // ---- the row bit maps of the batch's steps: lanes 0..31 / 32..63 copy one 4 BW-byte row each per round
        // (16-byte loads past the L1; nothing bumps the maps before the batch ends)
        bool bits_pending = false;
        if (NP > 0 && !fold) {
            const int BQ = BW >> 2;
            if (BQ <= 32 && NW == kQueueWaves && !(q.spec & 128)) {
                // the helper waves copy the rows (bits_part) while this wave goes on with what needs no bit: the dirty-row tags,
                // the own nodes' exact keys; the second barrier stands in front of the walk  (q.spec & 128: test knob, this wave copies)
                rowS[lane] = row;
                if (lane == 0) { hcmd[0] = kQCmdBits; hcmd[1] = B; }
                lds_barrier();                       // (1) posted
                bits_pending = true;
            } else if (BQ <= 32) {
                // (all 32 rounds' loads in flight at once -- 128 registers, the wave has the file to itself -- one round trip to the L2)
                qv4 v[4][8];
#pragma unroll
                for (int g8 = 0; g8 < 4; g8++) {
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int r2 = 16 * g8 + 2 * u;
                        const int row_r = __builtin_amdgcn_readlane(row, r2), row_r1 = __builtin_amdgcn_readlane(row, r2 + 1);
                        const int myrow = (lane >> 5) ? row_r1 : row_r;
                        const int c = lane & 31;
                        v[g8][u] = q_load_row16((const qv4*)q.ntn_bits + (size_t)((r2 + (lane >> 5) < B && c < BQ) ? myrow : 0) * BQ + (c < BQ ? c : 0));
                    }
                }
#pragma unroll
                for (int g8 = 0; g8 < 4; g8++) {
                    BLANCE_QROWS_ARRIVED(v[g8]);
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int rr = 16 * g8 + 2 * u + (lane >> 5), c = lane & 31;
                        if (rr < B && c < BQ) ((qv4*)bitsL)[rr * BQ + c] = v[g8][u];
                    }
                }
            } else {
                for (int r2 = 0; r2 < B; r2 += 2) {
                    const int rr = r2 + (lane >> 5);
                    const int row_r = __builtin_amdgcn_readlane(row, r2), row_r1 = __builtin_amdgcn_readlane(row, r2 + 1 < 64 ? r2 + 1 : r2);
                    const int myrow = (lane >> 5) ? row_r1 : row_r;
                    for (int c = (lane & 31); c < BW; c += 32)
                        if (rr < B) bitsL[rr * BW + c] = BLANCE_QLD(q.ntn_bits + (size_t)myrow * BW + c);
                }
            }
        }
        PH(2);
        // an earlier step of the batch with my row bumps entries I read: "dirty" (re-read at my turn)
        // (one of the lanes that share a row wins the tag; all the others -- and, conservatively, possibly the first of
        // them -- re-read at their turn)
        bool dirty = false;
        if (NP > 0 && act) rowTag[row] = (unsigned char)lane;
        BLANCE_WAVE_SYNC();
        if (NP > 0 && act) dirty = rowTag[row] != (unsigned char)lane;
        {   // the lane that won may be the LAST of its row: it is dirty too unless it is the first -- decide by a second round
            const u64 losers = __ballot(dirty);
            BLANCE_WAVE_SYNC();
            if (NP > 0 && act && dirty) rowTag[row] = 64;          // mark rows that more than one lane has
            BLANCE_WAVE_SYNC();
            if (NP > 0 && act && !dirty && rowTag[row] == 64) dirty = true;
            (void)losers;
        }
        if (fold) dirty = false;                     // (the shared row is in LDS, kept exactly by the walk itself)
        BLANCE_WAVE_SYNC();
        // exact keys of the own nodes, sorted: what a stay emits (keeping the same nodes in another order changes no counter)
        u64 lastK = 0;
        int lastN = -1;
        int sortv[KM];
        u64 sKv[KM];                                 // the own nodes' exact keys in the order of sortv
#pragma unroll
        for (int j = 0; j < KM; j++) { sortv[j] = -1; sKv[j] = ~0ull; }
        bool sfail = true;
        auto own_keys = [&]() {
#pragma unroll
            for (int j = 0; j < KM; j++) { sortv[j] = -1; oK[j] = ~0ull; }
            u64 sK[KM];
#pragma unroll
            for (int j = 0; j < KM; j++) sK[j] = ~0ull;
#pragma unroll
            for (int j = 0; j < KM; j++) {
                if (j < nown) {
                    const int o = ownv[j];
                    const u64 b = sortable_bits(queue_score(cntL[o], ntn_own[j], totL[o], (flL[o] >> 1) & 1, wL[o], NP,
                                                            vstick, q.booster_kind, lpT, ffT));
                    oK[j] = b;
                    u64 cb = b;
                    int cn = o;
#pragma unroll
                    for (int e = 0; e < KM; e++) {
                        if (e <= j) {
                            const bool first = e == j || qless(cb, cn, sK[e], sortv[e]);
                            if (first) {
                                const u64 tb = sK[e]; const int tn = sortv[e];
                                sK[e] = cb; sortv[e] = cn;
                                cb = tb; cn = tn;
                            }
                        }
                    }
                }
            }
            lastK = 0; lastN = -1;
#pragma unroll
            for (int j = 0; j < KM; j++) { if (j == k - 1) { lastK = sK[j]; lastN = sortv[j]; } sKv[j] = sK[j]; }
            sfail = !simple || nown != k;            // fewer nodes than constraints: never a stay
        };
        if (simple) own_keys();
        // what a stay emits sits in outS from the start (plan.go:299-301 leaves everything as it is); a step that moves
        // overwrites its slot
        auto emit_stay = [&]() {
            int* o = outS + lane * OWs;
            o[0] = k;
#pragma unroll
            for (int j = 0; j < KM; j++) if (j < k) o[1 + j] = sortv[j];
        };
        if (act && !sfail) emit_stay();
        u64 stalemask = 0;                           // lanes whose own nodes' counters an earlier step of the batch changed
        BLANCE_WAVE_SYNC();
        if (bits_pending) lds_barrier();             // (2) the batch's row bit maps are in LDS

        PH(3);
        int bumped_upto = 0;                         // steps [0, bumped_upto) of the batch have their rows bumped
        auto flush_bumps = [&](int upto) {
            if (NP > 0 && lane >= bumped_upto && lane < upto) {
                const int n = outS[lane * OWs] & 0xffff;
                for (int j = 0; j < n; j++) {
                    const int x = outS[lane * OWs + 1 + j];
                    if (x >= 0 && x < N) {
                        atomicAdd(q.ntn + (size_t)row * N + x, 1);
                        atomicOr((int*)q.ntn_bits + (size_t)row * BW + (x >> 5), (int)(1u << (x & 31)));
                    }
                }
            }
            bumped_upto = upto > bumped_upto ? upto : bumped_upto;
        };

        int cur = 0;
        // lanes that can only be resolved by the general code when they do not stay
        const u64 slowmask = __ballot(!simple || dirty || n_o > 2 || hv[1] >= 0) | (lean_ok ? 0ull : ~0ull);
        const u64 actmask = B >= 64 ? ~0ull : ((1ull << B) - 1);
        const u64 sfailmask = __ballot(sfail), dirtymask = __ballot(dirty);
        const int own_a = sortv[0], own_b = sortv[1];
        for (;;) {
            cur = uni(cur); wcnt = uni(wcnt); thK = uni64(thK); thN = uni(thN); stalemask = uni64(stalemask);
            if (cur >= B) break;
#ifdef BLANCE_QDEBUG3
            if (lane == 0) printf("[q3] outer cur %d B %d wcnt %d\n", cur, B, wcnt);
#endif
            int f = B;
            bool retried = false, back_to_asm = false, skip_lean = false;
#ifdef BLANCE_QDEBUG4
            int lean_why = 0;
#define BLANCE_QWHY(x) lean_why = (x)
#else
#define BLANCE_QWHY(x)
#endif
#ifndef BLANCE_SIMT_EMU
            // ================= the lean walk in assembly (k_queue_walk.h) for the plain case; it leaves at a step it does not take
            if (walk_asm) {
                QueueWalkState st;
                st.wk = wk; st.wn = wn; st.o1 = mo1; st.o2 = mo2; st.cur = cur; st.wcnt = wcnt; st.thK = thK; st.thN = thN;
                st.stale = stalemask; st.moved = 0; st.code = 0;
                queue_walk_k2(st, lastK, lastN, own_a, own_b, sKv[0], sKv[1], hv[0], wj, (ov0 & 0xffff) | (ov1 << 16), lane,
                              sfailmask | dirtymask, slowmask, actmask, cfa, cfb, cfc,
                              (cfd & 0xffff) | (B << 16) | ((k == 2 ? 1 : 0) << 24) | ((fold ? 1 : 0) << 25), cfe, lp_one);
                wk = st.wk; wn = st.wn; mo1 = st.o1; mo2 = st.o2;
                const int c1_ = uni(st.cur);
                wcnt = uni(st.wcnt); thK = uni64(st.thK); thN = uni(st.thN); stalemask = uni64(st.stale);
                const u64 mv = uni64(st.moved);
                if ((mv >> lane) & 1) {              // plan.go:299: the steps that moved (their rows are bumped with the batch's)
                    int* o = outS + lane * OWs;
                    o[0] = k; o[1] = mo1;
                    if (k == 2) o[2] = mo2;
                }
                const int nm = __popcll(mv);
                n_moved += nm;
                n_bulk += (c1_ - cur) - nm;
                cur = c1_;
                const int code = uni(st.code);
                if (code == 0 || cur >= B) break;
                // code 1: the step at cur needs the general code (its lane's data is out of date, or no window entry can be
                // taken as it stands) -- the C++ twin below would find the same and leave; code 2: the twin may still take it
                // (promotion, a counter beyond the tables, a window to rebuild).  The folded row has no general code.
                skip_lean = code == 1 && !fold;
            }
#endif
            if (skip_lean) f = cur;
            else
            // ================= the lean walk: steps in order as long as they stay or move the plain way =================
            // (a step whose two best candidates are the first eligible entries of the window, none of them with a
            // nodeToNodeCounts entry, all nodes' weights powers of two: one pass of straight-line code per step)
            for (;;) {
                cur = uni(cur); wcnt = uni(wcnt); thK = uni64(thK); thN = uni(thN); stalemask = uni64(stalemask);
                if (cur >= B) { f = B; break; }
                u64 frontK = thK;
                int frontN = thN;
                if (wcnt > 0) {
                    frontK = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(wk >> 32), 0) << 32) |
                             (unsigned)__builtin_amdgcn_readlane((int)(unsigned)wk, 0);
                    frontN = __builtin_amdgcn_readlane(wn, 0);
                }
                const u64 fm = (__ballot(!qless(lastK, lastN, frontK, frontN)) | sfailmask | dirtymask | stalemask) & actmask & (~0ull << cur);
                f = uni(fm ? __ffsll((long long)fm) - 1 : B);
#ifdef BLANCE_QDEBUG3
                if (lane == 0) printf("[q3]  lean cur %d f %d fm %llx slow %llx stale %llx\n", cur, f, fm, slowmask, stalemask);
#endif
                n_bulk += f - cur;
                if (f >= B) break;
                if (((slowmask | stalemask) >> f) & 1) { BLANCE_QWHY(((slowmask >> f) & 1) ? 1 : 2); break; }
                const int w = __builtin_amdgcn_readlane(wj, f);
                const int oa = __builtin_amdgcn_readlane(own_a, f), ob = __builtin_amdgcn_readlane(own_b, f);   // -1: none
                const int hh = __builtin_amdgcn_readlane(hv[0], f);
                const u64 ka = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(sKv[0] >> 32), f) << 32) |
                               (unsigned)__builtin_amdgcn_readlane((int)(unsigned)sKv[0], f);
                const u64 kb = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(sKv[1] >> 32), f) << 32) |
                               (unsigned)__builtin_amdgcn_readlane((int)(unsigned)sKv[1], f);
                const int na = oa < 0 ? INT_MAX : oa, nb = ob < 0 ? INT_MAX : ob;
                bool lean_done = true;
                u64 t1 = ~0ull, t2 = ~0ull;
                int n1 = INT_MAX, n2 = INT_MAX;
                {
                    const bool elig = lane < wcnt && wn != oa && wn != ob && wn != hh;
                    const bool bit = NP > 0 && !fold && elig && ((bitsL[f * BW + (wn >> 5)] >> (wn & 31)) & 1) != 0;
                    const u64 em = __ballot(elig), dm = __ballot(bit);
                    u64 cm = em & ~dm;
                    u64 ck = k == 2 ? (cm & (cm - 1)) : cm;
                    if (ck == 0) lean_done = false;                   // fewer than k clean candidates in the window
                    else {
                        const int cutoff = __ffsll((long long)ck) - 1;
                        const u64 upto = cutoff >= 63 ? ~0ull : ((2ull << cutoff) - 1);
                        const u64 dl = dm & upto;
                        if (dl) {
                            // entries with their bit set in front of the k-th clean one: scored with an entry of 1 -- a lower
                            // bound of their exact score (plan.go:638-644 is monotone in the entry) -- they are out of the
                            // race if even that lies above the k-th clean candidate; else the matrix has to be read
                            n_bound++;
                            const u64 ckey = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(wk >> 32), cutoff) << 32) |
                                             (unsigned)__builtin_amdgcn_readlane((int)(unsigned)wk, cutoff);
                            bool keep = false;
                            if ((dl >> lane) & 1) {
                                const int tt = totL[wn];
                                double r = (double)cntL[wn];
                                r = r + lpT[1];
                                r = r + ((unsigned)tt < (unsigned)kFfTab ? ffT[tt] : (0.001 * (double)tt) / (double)NP);
                                r = ldexp(r, -(int)shL[wn]);
                                r = r - 0.0;
                                keep = !(sortable_bits(r) > ckey);
                            }
                            if (__ballot(keep)) lean_done = false;
                        }
                        if (lean_done) {
                            const int c1 = __ffsll((long long)cm) - 1;
                            t1 = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(wk >> 32), c1) << 32) |
                                 (unsigned)__builtin_amdgcn_readlane((int)(unsigned)wk, c1);
                            n1 = __builtin_amdgcn_readlane(wn, c1);
                            if (k == 2) {
                                const int c2 = cutoff;
                                t2 = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(wk >> 32), c2) << 32) |
                                     (unsigned)__builtin_amdgcn_readlane((int)(unsigned)wk, c2);
                                n2 = __builtin_amdgcn_readlane(wn, c2);
                            }
                        }
                    }
                }
                if (!lean_done) {
                    // the window ran dry (few entries, none of them a candidate): rebuild it once and look again
                    if (wcnt < 32 && wcnt * 4 < rb_wcnt * 3 && thN != INT_MAX && !retried) { rebuild(); retried = true; n_bulk -= f - cur; continue; }
                    BLANCE_QWHY(3);
                    break;
                }
                // the k smallest of the sorted pairs (ka, kb) and (t1, t2): which nodes leave, which enter
                int r1n, r2n = INT_MAX, lv1 = -1, lv2 = -1, en1 = -1, en2 = -1;
                u64 rlast;
                if (qless(ka, na, t1, n1)) {                          // the best own node stays in front
                    r1n = na; rlast = ka;
                    if (k == 2) {
                        if (qless(kb, nb, t1, n1)) { r2n = nb; rlast = kb; }               // both stay (another order at most)
                        else { r2n = n1; rlast = t1; en1 = n1; lv1 = ob; }
                    }
                } else {
                    r1n = n1; rlast = t1; en1 = n1;
                    if (k == 2) {
                        if (qless(ka, na, t2, n2)) { r2n = na; rlast = ka; lv1 = ob; }
                        else { r2n = n2; rlast = t2; en2 = n2; lv1 = oa; lv2 = ob; }
                    } else lv1 = oa;
                }
                // a taken node that the partition holds in a lower priority state is promoted (plan.go:294-297): it leaves those
                // lists, i.e. that state's counter of the node drops and the node's total does not grow.  np1 / np2: how many
                // lower priority lists hold en1 / en2 (a node twice in one list counts once: misc.go:45); rare -- the record is
                // only decoded when a taken node is among the step's lower priority nodes
                int np1 = 0, np2 = 0;
                bool promoted = false;
                {
                    const int l0 = __builtin_amdgcn_readlane(ov0, f), l1 = __builtin_amdgcn_readlane(ov1, f);
                    promoted = (en1 >= 0 && (en1 == l0 || en1 == l1)) || (en2 >= 0 && (en2 == l0 || en2 == l1));
                }
                auto promotions = [&](bool apply) {
                    const int* rf = recS + f * RW;
                    for (int t = 0; t < M; t++) {
                        if (t == s || ((q.higher_mask >> t) & 1)) continue;
                        const int h = rf[kRecHead + t * SW];
                        if ((h >> 16) == kListAbsent) continue;
                        for (int jj = 0; jj < (h & 0xffff); jj++) {
                            const int x = rf[kRecHead + t * SW + 1 + jj];
                            bool first = true;
                            for (int j2 = 0; j2 < jj; j2++) if (rf[kRecHead + t * SW + 1 + j2] == x) first = false;
                            if (!first || x < 0 || (x != en1 && x != en2)) continue;
                            if (apply) { if (lane == 0) q.cnt[t * NX + x] -= w; }
                            else if (x == en1) np1++;
                            else np2++;
                        }
                    }
                };
                if (promoted) promotions(false);
                const int rlastn = k == 2 ? r2n : r1n;
                if (rlastn == INT_MAX || !qless(rlast, rlastn, thK, thN)) {                  // beyond the window's reach
                    if (wcnt < 56 && wcnt * 8 < rb_wcnt * 7 && thN != INT_MAX && !retried) { rebuild(); retried = true; n_bulk -= f - cur; continue; }
                    BLANCE_QWHY(5);
                    break;
                }
                // lanes 0 .. 3 settle one node each: leaving, leaving, entering, entering
                int hx = lane == 0 ? lv1 : lane == 1 ? lv2 : lane == 2 ? en1 : lane == 3 ? en2 : -1;
                if (lane >= 4) hx = -1;
                int c_new = 0, t_new = 0;
                u64 nk = ~0ull;
                if (hx >= 0) {
                    const int ds = lane < 2 ? -w : w;
                    c_new = cntL[hx] + ds;
                    t_new = totL[hx] + ds - w * (lane == 2 ? np1 : lane == 3 ? np2 : 0);
                }
                int e_new = 0;
                if (fold && hx >= 0) e_new = (int)ntL[hx] + (lane >= 2 ? 1 : 0);          // plan.go:238-245: the chosen nodes' entries of row ""
                if (NP > 0 && __ballot(hx >= 0 && ((unsigned)t_new >= (unsigned)kFfTab || e_new >= kLpTab))) { BLANCE_QWHY(6); break; }   // beyond the tables: the general code divides
                if (promoted) promotions(true);               // (past the last way out of this step)
                if (hx >= 0) {
                    double r = (double)c_new;                         // queue_score with no stickiness and a power-of-two weight
                    if (NP > 0) { r = r + lpT[e_new]; r = r + ffT[t_new]; }
                    if (fold) ntL[hx] = (unsigned short)e_new;
                    r = ldexp(r, -(int)shL[hx]);
                    r = r - 0.0;
                    nk = sortable_bits(r);
                    cntL[hx] = c_new; totL[hx] = t_new; gB[hx] = nk;
                }
                n_moved++;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int x = __builtin_amdgcn_readlane(hx, j);
                    if (x < 0) continue;
                    const u64 xk = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(nk >> 32), j) << 32) |
                                   (unsigned)__builtin_amdgcn_readlane((int)(unsigned)nk, j);
                    win_remove(x);
                    win_insert(xk, x);
                    stalemask |= __ballot(own_a == x || own_b == x);
                }
                if (lane == f) {
                    int* o = outS + f * OWs;
                    o[0] = k; o[1] = r1n;
                    if (k == 2) o[2] = r2n;
                }
                cur = f + 1;
                retried = false;
#ifndef BLANCE_SIMT_EMU
                if (walk_asm) { back_to_asm = true; break; }
#endif
            }
            PH(4);
            if (back_to_asm) continue;
            if (f >= B) break;
            cur = f;
#ifdef BLANCE_QDEBUG4
            if (fold && lane == 0) printf("[q4] folded walk left at step %d: reason %d (1 slow, 2 stale, 3 no candidate, 4 promotion, 5 beyond THETA, 6 tables) wcnt %d\n", oi + f, lean_why, wcnt);
#endif
            if (fold) { stop_pos = oi + f; stop_why = kQStopShape; break; }        // (the general code reads the matrix, not the folded row)
            // lane-level views of the masks, and the front of the window, for the general code below
            bool stale = (stalemask >> lane) & 1;
            u64 frontK = thK;
            int frontN = thN;
            if (wcnt > 0) {
                frontK = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(wk >> 32), 0) << 32) |
                         (unsigned)__builtin_amdgcn_readlane((int)(unsigned)wk, 0);
                frontN = __builtin_amdgcn_readlane(wn, 0);
            }

            // ================= step f does not certainly stay =================
            const bool simple_f = __builtin_amdgcn_readlane(simple ? 1 : 0, f) != 0;
            if (!simple_f) { stop_pos = oi + f; stop_why = kQStopShape; cur = f; break; }
            const bool dirty_f = __builtin_amdgcn_readlane(dirty ? 1 : 0, f) != 0;
            const bool stale_f = __builtin_amdgcn_readlane(stale ? 1 : 0, f) != 0;
            bool have_bits = NP > 0;
            if (dirty_f) {                           // this step reads a row an earlier step of the batch bumps
                flush_bumps(f);
                BLANCE_QFENCE();
                BLANCE_WAVE_SYNC();                  // (the emulated lanes are not in lockstep: bumps before the re-read)
                have_bits = false;
            }
#ifdef BLANCE_QDEBUG2
            if (lane == f) printf("[q2] step %d turn: dirty %d stale %d sfail %d simple %d row %d own %d last %llx/%d front %llx/%d\n", oi + f, (int)dirty, (int)stale, (int)sfail, (int)simple, row, ownv[0], lastK, lastN, frontK, frontN);
#endif
            if (dirty_f || stale_f) {               // lane f's keys are out of date: again, from what is true now
                if (lane == f) {
                    if (NP > 0) {
#pragma unroll
                        for (int j = 0; j < KM; j++)
                            if (j < nown) ntn_own[j] = BLANCE_QLD(q.ntn + (size_t)row * N + ownv[j]);
                    }
                    own_keys();
                    dirty = false; stale = false;
                }
                stalemask &= ~(1ull << f);
                const bool ok = !sfail && qless(lastK, lastN, frontK, frontN);
                if ((__ballot(ok) >> f) & 1) {       // a stay after all
                    if (lane == f) emit_stay();
                    n_bulk++;
                    cur = f + 1;
                    continue;
                }
            }
            const int w = __builtin_amdgcn_readlane(wj, f);
            const int rowf = __builtin_amdgcn_readlane(row, f);
            const int nown_f = __builtin_amdgcn_readlane(nown, f);
            int qown[KM], qh[KH];
            u64 qK[KM];
#pragma unroll
            for (int j = 0; j < KM; j++) {
                qown[j] = __builtin_amdgcn_readlane(ownv[j], f);
                qK[j] = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(oK[j] >> 32), f) << 32) |
                        (unsigned)__builtin_amdgcn_readlane((int)(unsigned)oK[j], f);
            }
#pragma unroll
            for (int j = 0; j < KH; j++) qh[j] = __builtin_amdgcn_readlane(hv[j], f);
            const bool other_f = __builtin_amdgcn_readlane(has_other ? 1 : 0, f) != 0;

            PH(5);
            // the k best (key, node) of the step; wave uniform, ascending
            u64 bB[KM];
            int bN[KM];
            auto insert = [&](u64 b, int n) {
#pragma unroll
                for (int j = KM - 1; j >= 0; j--) {
                    if (j >= k) continue;
                    const bool here = qless(b, n, bB[j], bN[j]);
                    const bool above = j > 0 && qless(b, n, bB[j - 1], bN[j - 1]);
                    if (here) {
                        if (above) { bB[j] = bB[j - 1]; bN[j] = bN[j - 1]; }
                        else { bB[j] = b; bN[j] = n; }
                    }
                }
            };
            int n_out = 0;
            // The window analysis in front of the dense scan (which candidates carry bits, can any pick be final below THETA)
            // costs a third of what the scan does; in the tie regime -- many nodes at one load, every entry of the window held
            // back by its matrix entry -- general step after general step ends in the scan.  After three in a row the steps
            // go there directly (the scan is exact whatever the window holds), every eighth one looks at the window again.
            bool went_dense = false;
            int attempt = (q.spec & 16) ? 2 : 0;          // (q.spec & 16: test knob, every general step scores every node)
            if (dense_streak >= 3 && ((++dense_probe) & 7) != 0) attempt = 2;
            for (;; attempt++) {
                attempt = uni(attempt);
#pragma unroll
                for (int j = 0; j < KM; j++) { bB[j] = ~0ull; bN[j] = INT_MAX; }
#pragma unroll
                for (int j = 0; j < KM; j++) if (j < nown_f) insert(qK[j], qown[j]);
                if (attempt == 2) {
                    // ---- the window cannot decide (its entries' exact scores lie above THETA: rows with many entries,
                    // many nodes of equal load): score every node exactly -- what the reference's sort sees
                    n_dense++;
#ifdef BLANCE_QDIAG
                    if ((n_dense & 511) == 1 && n_dense < 512 * 120) {      // developer aid: why could the window not decide?
                        bool el_ = lane < wcnt;
                        for (int j = 0; j < KM; j++) el_ = el_ && wn != qown[j];
                        const bool bt_ = el_ && NP > 0 && have_bits && ((bitsL[f * BW + (wn >> 5)] >> (wn & 31)) & 1);
                        const u64 fK_ = wcnt > 0 ? (((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(wk >> 32), 0) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)wk, 0)) : ~0ull;
                        int tied_ = 0, below_ = 0, rowbits_ = 0, tiedclean_ = 0, lvl_ = 0;
                        const int c0_ = wcnt > 0 ? cntL[__builtin_amdgcn_readlane(wn, 0)] : -1;
                        for (int i = 0; i < G; i++) {
                            const int n = i * 64 + lane;
                            const bool a_ = n < N && (flL[n] & 1);
                            const bool b_ = have_bits && ((bitsL[f * BW + (n >> 5)] >> (n & 31)) & 1);
                            if (a_ && gB[n] == fK_) { tied_++; if (!b_) tiedclean_++; }
                            if (a_ && gB[n] < thK) below_++;
                            if (a_ && b_) rowbits_++;
                            if (a_ && cntL[n] == c0_) lvl_++;
                        }
                        for (int o = 32; o; o >>= 1) { tied_ += __shfl_xor(tied_, o, 64); below_ += __shfl_xor(below_, o, 64); rowbits_ += __shfl_xor(rowbits_, o, 64); tiedclean_ += __shfl_xor(tiedclean_, o, 64); lvl_ += __shfl_xor(lvl_, o, 64); }
                        const int ne_ = __popcll(__ballot(el_)), nd_ = __popcll(__ballot(bt_));
                        if (lane == 0) printf("[qdiag] step %d dense#%lld wcnt %d elig %d dirty %d theta_inf %d front==theta %d tied %d (clean %d) below_theta %d rowbits %d level(cnt %d) %d row %d own %d ownkey-front %lld have_bits %d\n",
                                              oi + f, n_dense, wcnt, ne_, nd_, (int)(thN == INT_MAX), (int)(fK_ == thK), tied_, tiedclean_, below_, rowbits_, c0_, lvl_, rowf, qown[0], (long long)(qK[0] - fK_), (int)have_bits);
                    }
#endif
                    PHM(dense_begin);
                    PH(12);
                    // every wave of the workgroup takes a share of the nodes (dense_part above); the command:
                    if (lane == 0) {
                        hcmd[0] = kQCmdDense; hcmd[1] = f; hcmd[2] = k;
                        hcmd[3] = qown[0]; hcmd[4] = qown[1]; hcmd[5] = qh[0]; hcmd[6] = qh[1];
                        hcmd[7] = (NP > 0 && have_bits) ? 1 : 0; hcmd[8] = rowf;
                        if (NP > 0 && have_bits) {       // no candidates: they count as entries with their bit set (row f is this step's alone)
#pragma unroll
                            for (int j = 0; j < KM; j++) if (qown[j] >= 0 && qown[j] < NXp) bitsL[f * BW + (qown[j] >> 5)] |= 1u << (qown[j] & 31);
#pragma unroll
                            for (int j = 0; j < KH; j++) if (qh[j] >= 0 && qh[j] < NXp) bitsL[f * BW + (qh[j] >> 5)] |= 1u << (qh[j] & 31);
                        }
                        // the step's k-th pick is no worse than the k-th best of its own nodes
                        u64 U = ~0ull;
                        if (nown_f >= k) { U = qK[0]; if (k > 1 && qK[1] > U) U = qK[1]; }
                        hcmd[10] = (int)(unsigned)(U >> 32); hcmd[11] = (int)(unsigned)U;
                    }
                    BLANCE_QWAIT_BUMPS();            // (rows bumped by this wave are read by the others)
                    lds_barrier();                   // (1) posted
                    PH(13);
                    dense_part();                    // (this wave's share)
                    lds_barrier();                   // (2) every wave's k best are in
                    PH(15);
                    {
                        u64 ek = ~0ull;
                        int en = INT_MAX;
                        if (lane < NW * k) {
                            const int* r = hres + ((lane / k) * KM + (lane % k)) * kQResWords;
                            ek = ((u64)(unsigned)r[0] << 32) | (unsigned)r[1];
                            en = r[2];
                            if (en == INT_MAX) ek = ~0ull;
                        }
                        for (int j = 0; j < k; j++) {
                            const QMin m = wave_min_key_node(ek, en);
                            if (m.node == INT_MAX) break;
                            insert(((u64)m.hi << 32) | m.lo, m.node);
                            if (en == m.node) { ek = ~0ull; en = INT_MAX; }
                        }
                    }
                    PHM(dense_end);
                    PH(16);
                    n_out = 0;
#pragma unroll
                    for (int j = 0; j < KM; j++) if (j < k && bN[j] != INT_MAX) n_out++;
                    went_dense = true;
                    break;
                }
                // window entries that are candidates for this partition (plan.go:142-156), and whose entry may be non-zero
                bool elig = lane < wcnt;
#pragma unroll
                for (int j = 0; j < KM; j++) elig = elig && wn != qown[j];
#pragma unroll
                for (int j = 0; j < KH; j++) elig = elig && wn != qh[j];
                bool bit = false;
                if (NP > 0 && elig) bit = have_bits ? ((bitsL[f * BW + (wn >> 5)] >> (wn & 31)) & 1) != 0 : true;
                const u64 em = __ballot(elig), dm = __ballot(elig && bit);
                const u64 cm = em & ~dm;             // clean: exact score = window key
                // the k-th clean candidate bounds what has to be looked at: everything behind it scores >= its g
                u64 ck = cm;
                for (int j = 1; j < k; j++) ck &= ck - 1;
                const int cutoff = ck ? __ffsll((long long)ck) - 1 : 64;
                const u64 upto = cutoff >= 63 ? ~0ull : ((2ull << cutoff) - 1);
                const u64 dl = dm & upto;            // entries to read from the matrix
                bool hopeless = false;
                if (dl == 0) {
                    // ---- fast: the first k eligible entries, as they stand
                    u64 e2 = em & upto;
                    for (int j = 0; j < k && e2; j++) {
                        const int c = __ffsll((long long)e2) - 1;
                        e2 &= e2 - 1;
                        const u64 cb = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(wk >> 32), c) << 32) |
                                       (unsigned)__builtin_amdgcn_readlane((int)(unsigned)wk, c);
                        insert(cb, __builtin_amdgcn_readlane(wn, c));
                    }
                } else if ([&]() -> bool {
                    // Can the window decide at all?  A pick is only final if it lies below THETA (a node outside the window
                    // scores at least THETA).  An entry with its bit set scores at least what it would with an entry of 1
                    // (plan.go:638-644 is monotone in the entry): when fewer than k of (own nodes, clean entries, such lower
                    // bounds) lie below THETA, no reading of the matrix can make the window's answer final -- the regime of
                    // many nodes at one load, where every entry of the window is held back by its matrix entry.
                    if (thN == INT_MAX || !have_bits) return false;      // (without the bit map a set bit is no promise of an entry)
                    bool can = false;
                    if (((cm & upto) >> lane) & 1) can = qless(wk, wn, thK, thN);
                    else if ((dl >> lane) & 1) {
                        u64 lbk = wk;
                        if (shL[wn] != 255) {
                            const int tt = totL[wn];
                            double r = (double)cntL[wn];
                            r = r + lpT[1];
                            r = r + ((unsigned)tt < (unsigned)kFfTab ? ffT[tt] : (0.001 * (double)tt) / (double)NP);
                            r = ldexp(r, -(int)shL[wn]);
                            r = r - 0.0;
                            lbk = sortable_bits(r);
                        }
                        can = qless(lbk, wn, thK, thN);
                    }
                    int possible = __popcll(__ballot(can));
#pragma unroll
                    for (int j = 0; j < KM; j++) if (j < nown_f && qless(qK[j], qown[j], thK, thN)) possible++;
                    return possible < k;
                }()) {
                    hopeless = true;
                } else {
                    // ---- entries with their bit set in front of the k-th clean one: exact scores from the matrix
                    n_exact++;
                    u64 ek = ~0ull;
                    int en = INT_MAX;
                    if ((dl >> lane) & 1) {
                        const int nt = BLANCE_QLD(q.ntn + (size_t)rowf * N + wn);
                        ek = nt ? sortable_bits(queue_score(cntL[wn], nt, totL[wn], (flL[wn] >> 1) & 1, wL[wn], NP, 0.0,
                                                            q.booster_kind, lpT, ffT)) : wk;
                        en = wn;
                    } else if (((cm & upto) >> lane) & 1) { ek = wk; en = wn; }
                    for (int j = 0; j < k; j++) {
                        const QMin m = wave_min_key_node(ek, en);
                        if (m.node == INT_MAX) break;
                        insert(((u64)m.hi << 32) | m.lo, m.node);
                        if (en == m.node) { ek = ~0ull; en = INT_MAX; }
                    }
                }
                // valid if every taken node lies below THETA (a node outside the window scores >= THETA)
                n_out = 0;
                bool below = true;
#pragma unroll
                for (int j = 0; j < KM; j++) {
                    if (j < k && bN[j] != INT_MAX) {
                        n_out++;
                        if (!qless(bB[j], bN[j], thK, thN)) below = false;
                    }
                }
                if (((n_out == k && below) || thN == INT_MAX) && !hopeless) break;
                // the window does not reach far enough.  Run dry (few entries): rebuild it around the current keys and
                // look again; full, or rebuilt already: its entries are held back by their matrix entries -- every node then
                if (attempt == 0 && wcnt < 56 && wcnt * 8 < rb_wcnt * 7) rebuild();
                else attempt = 1;
            }
            PH(6);
            dense_streak = went_dense ? (dense_streak < 1000 ? dense_streak + 1 : dense_streak) : 0;
            if (n_out < k) { stop_pos = oi + f; stop_why = kQStopShort; }      // fewer candidates than constraints: warnings
            if (stop_why != kQStopNone) { cur = f; break; }
            // a taken node that the partition holds in lower priority states is promoted (plan.go:294-297): it leaves those
            // lists -- their counters drop (only lane 0 touches them), and so does the node's total
            int promo[KM];                           // lists of lower priority states that hold the e-th taken node
#pragma unroll
            for (int e = 0; e < KM; e++) promo[e] = 0;
            if (other_f) {
                const int* rf = recS + f * RW;
                for (int t = 0; t < M; t++) {
                    if (t == s || ((q.higher_mask >> t) & 1)) continue;
                    const int h = rf[kRecHead + t * SW];
                    if ((h >> 16) == kListAbsent) continue;
                    for (int jj = 0; jj < (h & 0xffff); jj++) {
                        const int x = rf[kRecHead + t * SW + 1 + jj];
                        bool first = true;           // (a node twice in one list counts once: misc.go:45)
                        for (int j2 = 0; j2 < jj; j2++) if (rf[kRecHead + t * SW + 1 + j2] == x) first = false;
#pragma unroll
                        for (int e = 0; e < KM; e++) {
                            if (e < k && bN[e] == x && first) {
                                promo[e]++;
                                if (lane == 0) q.cnt[t * NX + x] -= w;
                            }
                        }
                    }
                }
            }
            // ---- commit (plan.go:290-301): own nodes not taken leave, taken nodes that are not own enter
            n_moved++;
#ifdef BLANCE_QDEBUG
            if (lane == 0) {
                printf("[q] step %d w %d row %d nown %d own %d %d keys %llx %llx high %d %d -> %d %d (%llx %llx) wcnt %d theta %llx/%d bits %d\n", oi + f, w, rowf, nown_f, qown[0], qown[1],
                       qK[0], qK[1], qh[0], qh[1], bN[0], bN[1], bB[0], bB[1], wcnt, thK, thN, (int)have_bits);
            }
            for (int l_ = 0; l_ < wcnt; l_++) { const unsigned h_ = __builtin_amdgcn_readlane((int)(unsigned)(wk >> 32), l_), lo_ = __builtin_amdgcn_readlane((int)(unsigned)wk, l_); const int n_ = __builtin_amdgcn_readlane(wn, l_);
                if (lane == 0) printf("[q]   win %d: node %d key %08x%08x cnt %d tot %d\n", l_, n_, h_, lo_, cntL[n_], totL[n_]); }
#endif
            int chg[2 * KM];                         // nodes whose counters change, wave uniform (-1: none)
            int nchg = 0;
#pragma unroll
            for (int j = 0; j < KM; j++) {
                bool kept = false;
#pragma unroll
                for (int e = 0; e < KM; e++) kept = kept || (e < k && bN[e] == qown[j]);
                chg[j] = (j < nown_f && !kept) ? qown[j] : -1;
            }
#pragma unroll
            for (int e = 0; e < KM; e++) {
                bool mine = false;
#pragma unroll
                for (int j = 0; j < KM; j++) mine = mine || (j < nown_f && qown[j] == bN[e]);
                chg[KM + e] = (e < k && !mine) ? bN[e] : -1;
            }
            // lanes 0 .. 2 KM - 1 settle one node each
            int hx = -1;
            u64 nk = ~0ull;
#pragma unroll
            for (int j = 0; j < 2 * KM; j++) if (lane == j) hx = chg[j];
            if (hx >= 0) {
                const int ds = lane < KM ? -w : w;
                int np = 0;
#pragma unroll
                for (int e = 0; e < KM; e++) if (lane == KM + e) np = promo[e];
                cntL[hx] += ds;
                totL[hx] += ds - w * np;
                nk = gkey(hx);
                gB[hx] = nk;
            }
            PH(7);
            BLANCE_WAVE_SYNC();
#pragma unroll
            for (int j = 0; j < 2 * KM; j++) {
                const int x = chg[j];
                if (x < 0) continue;
                nchg++;
                const u64 xk = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(nk >> 32), j) << 32) |
                               (unsigned)__builtin_amdgcn_readlane((int)(unsigned)nk, j);
                win_remove(x);
                win_insert(xk, x);
                // later lanes of the batch that hold x were validated against its old counters
#pragma unroll
                for (int jj = 0; jj < KM; jj++) stalemask |= __ballot(lane > f && ownv[jj] == x);
            }
            (void)nchg;
            if (lane == f) {
                int* o = outS + f * OWs;             // (its row is bumped with the batch's, plan.go:238-245)
                o[0] = k;
#pragma unroll
                for (int j = 0; j < KM; j++) if (j < k) o[1 + j] = bN[j];
            }
            PH(8);
            cur = f + 1;
        }
        // ---- the batch's outputs (up to the step that stopped the launch), and the bumps still pending
        const int done = stop_why == kQStopNone ? B : cur;
        BLANCE_WAVE_SYNC();
        if (RW <= kRecPre && oi + 64 < q.end) {      // the next batch's records (arrived long ago) into LDS: nobody reads this batch's any more
            const int Bn = q.end - (oi + 64) < 64 ? q.end - (oi + 64) : 64;
#pragma unroll
            for (int r = 0; r < kRecPre; r++) {
                const int idx = r * 64 + lane;
                if (r < RW && idx < Bn * RW) recS[idx] = pre[r];
            }
        }
        flush_bumps(done);
        for (int idx = lane; idx < done * OWs; idx += 64) q.out[(size_t)oi * OWs + idx] = outS[idx];
        PH(9);
        BLANCE_WAVE_SYNC();
    }
#ifdef BLANCE_PHASE_PROF
    if (lane == 0) {
        printf("[queue] k %d steps [%d, %d) stop %d why %d: stays %lld moved %lld exact %lld dense %lld rebuilds %lld (by three waves %lld)\n", k, q.beg, q.end, stop_pos, stop_why, n_bulk, n_moved, n_exact, n_dense, n_rebuild, n_coop);
        for (int i_ = 0; i_ < 17; i_++) printf("[queue phase %d] %.0f kcycles\n", i_, (double)ph_acc[i_] / 1e3);
        printf("[queue rebuilds] %.0f kcycles, %lld column scans of lane 0 (first scans: %.0f kcycles)\n", (double)rb_cycles / 1e3, rb_scans, (double)rb_scan_cycles / 1e3);
    }
#endif
    if (lane == 0) hcmd[0] = kQCmdExit;              // the helper waves leave
    lds_barrier();
    if (lane == 0) {
        q.stop[0] = stop_pos;
        q.stop[1] = stop_why;
        if (q.spec_count) *q.spec_count += n_bulk;
        if (q.qstats) { q.qstats[0] += n_moved; q.qstats[1] += n_exact; q.qstats[2] += n_rebuild; q.qstats[3] += n_dense; }
    }
    BLANCE_WAVE_SYNC();
    for (int i = 0; i < G; i++) {
        const int n = i * 64 + lane;
        if (n < NX) q.cnt[s * NX + n] = cntL[n];
    }
}

// dynamic LDS of k_pass_queue for a pass
static inline size_t queue_lds_bytes(int NX, int RW) {
    const size_t NXp = (size_t)((NX + 63) / 64) * 64, BW = ((NXp >> 5) + 3) & ~(size_t)3;
    const size_t RT = NXp + 64 > (size_t)kQScratch ? NXp + 64 : (size_t)kQScratch;      // rowTag / the cooperative rebuild's scratch
    return NXp * (8 + 4 + 4 + 4 + 1 + 1 + 2) + RT + sizeof(double) * (kLpTab + kFfTab) + sizeof(int32_t) * (size_t)(64 * RW) +
           sizeof(int32_t) * 64 * 3 + sizeof(int32_t) * 64 * BW + 64 + sizeof(int32_t) * (kQCmdWords + kQueueWaves * 2 * kQResWords + 64);
}

}  // namespace blance
