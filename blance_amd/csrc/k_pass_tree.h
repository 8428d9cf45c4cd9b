// k_pass_tree: the exact sequential state pass of a state WITHOUT hierarchy rules on one wave64,
// with a bound-ordered candidate structure instead of a scan of every node per step.
// Part of blance_hip.hip (one translation unit); see DESIGN.md section 4.
#pragma once

namespace blance {

// ============================================================================
// assignStateToPartitions (plan.go:253-303) with findBestNodes (plan.go:98-248)
// for a state that has no hierarchy rule.  Facts used:
//
//  * For a node n that is not one of the partition's own nodes, the exact score
//    of plan.go:634-689 is >= g[n], its partition-independent score (the
//    nodeToNodeCounts term is >= 0, IEEE add / divide are monotone), and equal
//    to g[n] bit for bit when the partition's nodeToNodeCounts entry is 0.
//  * A step changes the counters -- hence g -- of at most (old + chosen) nodes.
//
// So g sits in an LDS tournament tree: 64 leaves per group in LDS (sortable
// integer images of the fp64 scores), the minimum of group i in registers of
// lane i, the root one wave minimum away.  A step resolves as: exact scores of
// the partition's own nodes; then candidates in (g, position) order -- each
// scored exactly -- until the k-th best exact score beats the next g.  The cost
// of a step does not depend on the number of nodes.
//
//  * Steps that keep their nodes change no counter.  Lane j of a batch of 64
//    steps validates step j on its own (own nodes, scored exactly, in list order
//    and strictly below the root of the tree); the validated runs between two
//    other steps are committed at once, the other steps run the general code in
//    order, after which the remaining lanes are re-tested against the new root.
//  * nodeToNodeCounts entries (plan.go:238-245; only read when NumPartitions > 0)
//    of a batch are fetched up front: lane j loads the entries of its row for its
//    own nodes and for the 64 group minima (the likely candidates).  A lane whose
//    row is bumped by an earlier step of the batch re-reads at its turn.
//
// The walk is bounded (kWalkCap candidates); beyond that the step is resolved by
// scoring every node (dense_pick) -- correctness never depends on the bound.
// ============================================================================
constexpr int kTreeMaxNodes = 4096;      // LDS budget: 25 bytes per node + tables
constexpr int kWalkCap = 24;
constexpr int kMvW = 17;                 // words per mover record in LDS (14 used; odd stride: no bank conflicts)

#ifndef BLANCE_SIMT_EMU
// nodeToNodeCounts is read and bumped (atomics, at L2) by this wave all along the pass: its loads bypass L1
#define BLANCE_LD_COHERENT(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define BLANCE_AGENT_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent")
#else
#define BLANCE_LD_COHERENT(p) (*(p))
#define BLANCE_AGENT_FENCE()
#endif

struct TreeMin { unsigned hi, lo; int lane; };

// Minimum of a 64-bit key (hi:lo) over the wave and the LOWEST lane that holds it: two
// v_min_u32 DPP chains and a ballot.  Callers lay keys out so that lane order is position order.
__device__ __forceinline__ TreeMin wave_min_u64_lane(unsigned hi, unsigned lo) {
    TreeMin r;
    r.hi = wave_min_u32_bcast(hi);
    const bool ok = hi == r.hi;
    r.lo = wave_min_u32_bcast(ok ? lo : kKeyNoneV);
    const unsigned long long b = __ballot(ok && lo == r.lo);
    r.lane = __ffsll((long long)b) - 1;
    return r;
}

// nodeSorter.Score (plan.go:634-689), the reference's operations in the reference's order, with
// the two NumPartitions quotients from LDS tables filled by the same expressions and the
// division by a power-of-two node weight as an exponent shift (x / 2^e and ldexp(x, -e) are
// the same correctly rounded value).
__device__ __forceinline__ double tree_score(int cnt, int nt, int tot, int hasw, int w, int NP, double cf,
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

__device__ __forceinline__ bool key_less(unsigned long long a, int an, unsigned long long b, int bn) {
    return a < b || (a == b && an < bn);              // nodeSorter.Less on sortable images, plan.go:617-628
}

// KM: capacity of the step's output list (k <= KM).
template <int KM>
__global__ __launch_bounds__(64) void k_pass_tree(PassParams q) {
    typedef unsigned long long u64;
    constexpr int KH = 2, KO = 4;                    // higher / other state nodes a lane keeps for the short general step
    BLANCE_DYN_LDS(lds);
    const int lane = threadIdx.x;
    const int N = q.N, NX = q.NX, M = q.M, L = q.L, NP = q.NP, s = q.s, k = q.k, RW = q.RW;
    const int SW = 1 + L;                            // words per state inside a record
    const int G = (NX + 63) >> 6, NXp = G << 6;
    const int walk_cap = (q.spec & 2) ? 0 : kWalkCap;   // test knob: every general step scores all nodes
    const bool no_short = (q.spec & 4) != 0;         // test knob: every general step decodes its record

    u64* gB = (u64*)lds;                             // [NXp] sortable image of g, ~0 for nodes that are no candidates
    int* cntL = (int*)(gB + NXp);                    // [NXp] stateNodeCounts[s]
    int* totL = cntL + NXp;                          // [NXp] nodePartitionCounts (plan.go:118-124)
    int* wL = totL + NXp;                            // [NXp] node weights
    int* recS = wL + NXp;                            // [64 * RW] step records of the batch
    double* lpT = (double*)(recS + 64 * RW);         // [kLpTab] c / NP
    double* ffT = lpT + kLpTab;                      // [kFfTab] (0.001 * t) / NP
    u64* stkB = (u64*)(ffT + kFfTab);                // [kWalkCap] leaves taken out of the tree during a walk
    int* stkN = (int*)(stkB + kWalkCap);             // [kWalkCap]
    int* ntL = stkN + kWalkCap;                      // [NXp] folded mode: the shared row of nodeToNodeCounts
    int* mvL = ntL + NXp;                            // [64][kMvW] what lane j learnt about step j (the lean general step reads it)
    int* outS = mvL + 64 * kMvW;                     // [64][OW] the batch's outputs: written out, and their rows bumped, per batch
    unsigned char* flL = (unsigned char*)(outS + 64 * (KM + 1));   // [NXp] 1: in nodesNext, 2: has a weight

    for (int i = lane; i < kLpTab; i += 64) lpT[i] = NP > 0 ? (double)i / (double)NP : 0.0;
    for (int i = lane; i < kFfTab; i += 64) ffT[i] = NP > 0 ? (0.001 * (double)i) / (double)NP : 0.0;
    BLANCE_WAVE_SYNC();
    for (int i = 0; i < G; i++) {
        const int n = i * 64 + lane;
        int c = 0, t = 0, w = 0, fl = 0;
        if (n < NX) {
            c = q.cnt[s * NX + n];
            for (int tt = 0; tt <= M; tt++) t += q.cnt[tt * NX + n];
            w = q.node_weight[n];
            fl = ((n < N && q.alive[n]) ? 1 : 0) | (q.node_has_weight[n] ? 2 : 0);
        }
        cntL[n] = c; totL[n] = t; wL[n] = w; flL[n] = (unsigned char)fl; ntL[n] = 0;
    }
    BLANCE_WAVE_SYNC();

    // Folded mode: while every step of a batch has the SAME nodeToNodeCounts row (partitions without a
    // top priority node: row ""), that row's term is part of the leaf keys -- a candidate's exact score
    // is its leaf and a walk never reads the matrix.  fold = the row, or -1.
    int fold = -1;
    auto leaf_key = [&](int n) -> u64 {
        return (flL[n] & 1) ? sortable_bits(tree_score(cntL[n], fold >= 0 ? ntL[n] : 0, totL[n], (flL[n] >> 1) & 1, wL[n], NP,
                                                      0.0, q.booster_kind, lpT, ffT)) : ~0ull;
    };
    // group minima: lane i keeps the smallest (g, node) of leaves [64 i, 64 i + 64)
    unsigned gm_hi = kKeyNoneV, gm_lo = kKeyNoneV;
    int gm_n = INT_MAX;
    auto scan_group = [&](int i, unsigned& rh, unsigned& rl, int& rn) {
        const u64 v = gB[i * 64 + lane];
        const TreeMin m = wave_min_u64_lane((unsigned)(v >> 32), (unsigned)v);
        rh = m.hi; rl = m.lo;
        rn = (m.hi & m.lo) == kKeyNoneV ? INT_MAX : i * 64 + m.lane;
    };
    auto rebuild_tree = [&]() {
        for (int i = 0; i < G; i++) gB[i * 64 + lane] = leaf_key(i * 64 + lane);
        BLANCE_WAVE_SYNC();
        for (int i = 0; i < G; i++) {
            unsigned rh, rl; int rn;
            scan_group(i, rh, rl, rn);
            if (lane == i) { gm_hi = rh; gm_lo = rl; gm_n = rn; }
        }
    };
    rebuild_tree();
    // the two smallest leaves (root = the smallest), recomputed after every change of the tree
    bool root_valid = false;
    u64 rootB = ~0ull, t2B = ~0ull;
    int root_n = INT_MAX, t2n = INT_MAX;
    auto compute_top2 = [&]() {
        const TreeMin m = wave_min_u64_lane(gm_hi, gm_lo);
        rootB = ((u64)m.hi << 32) | m.lo;
        root_n = __builtin_amdgcn_readlane(gm_n, m.lane);
        t2B = ~0ull; t2n = INT_MAX;
        if (root_n != INT_MAX) {                     // the runner-up: another group's minimum, or the rest of the root's group
            u64 v = gB[m.lane * 64 + lane];
            if (lane == (root_n & 63)) v = ~0ull;
            const TreeMin s2 = wave_min_u64_lane((unsigned)(v >> 32), (unsigned)v);
            unsigned wh = gm_hi, wl = gm_lo;
            if (lane == m.lane) { wh = s2.hi; wl = s2.lo; }
            const TreeMin m2 = wave_min_u64_lane(wh, wl);
            if ((m2.hi & m2.lo) != kKeyNoneV) {
                t2B = ((u64)m2.hi << 32) | m2.lo;
                t2n = m2.lane == m.lane ? m.lane * 64 + s2.lane : __builtin_amdgcn_readlane(gm_n, m2.lane);
            }
        }
        root_valid = true;
    };

    // which word of the record's state lists this lane looks at when a general step decodes its record
    const int slot_t = lane / SW, slot_ix = lane - slot_t * SW;
    const bool slot_ok = lane < M * SW;
    const bool slot_higher = slot_ok && ((q.higher_mask >> slot_t) & 1);

    long long n_bulk = 0;
    int pf_v = 0, pf_n = -1, pf_f = -1;              // lanes 2, 3: the nodeToNodeCounts entries of the two smallest leaves, fetched for step pf_f
    PH_DECL;
#ifdef BLANCE_PHASE_PROF
    long long pc_general = 0, pc_taken = 0, pc_miss = 0, pc_batches = 0, pc_scans = 0, pc_short = 0, pc_stay = 0, pc_reorder = 0, pc_half = 0, pc_lean = 0, pc_lean_try = 0, pc_hit1 = 0, pc_hit2 = 0, pc_need2 = 0;
#define PC(x) (x)++
#else
#define PC(x)
#endif

    for (int oi = q.beg; oi < q.end; oi += 64) {
        const int B = q.end - oi < 64 ? q.end - oi : 64;
        PH(11); PC(pc_batches);
        BLANCE_AGENT_FENCE();                        // earlier bumps of nodeToNodeCounts are visible to the loads below
        for (int r = 0; r < RW; r++) {
            const int idx = r * 64 + lane;
            if (idx < B * RW) recS[idx] = q.rec[(size_t)oi * RW + idx];
        }
        BLANCE_WAVE_SYNC();

        PH(0);
        // ---- lane j looks at step oi + j: can it be a certain stay?  What it learns about the step
        // (weight, row, own / higher / other nodes, exact scores of the own nodes) stays in its registers.
        const bool act = lane < B;
        const int* rj = recS + (act ? lane : 0) * RW;
        int row = NX;
        const int wj = rj[1];
        int ownv[KM], ntn_own[KM], hv[KH], ov[KO];
        unsigned oKh[KM], oKl[KM];
#pragma unroll
        for (int j = 0; j < KM; j++) { ownv[j] = -1; ntn_own[j] = 0; oKh[j] = kKeyNoneV; oKl[j] = kKeyNoneV; }
#pragma unroll
        for (int j = 0; j < KH; j++) hv[j] = -1;
#pragma unroll
        for (int j = 0; j < KO; j++) ov[j] = -1;
        bool pok = act;                              // own list of at most k nodes inside nodesAll: ntn_own is loaded
        int nown = 0;                                // its length (k: the step may be a stay)
        const double vstick = __hiloint2double(rj[3], rj[2]);
        {
            const int hT = rj[kRecHead + q.top_state * SW];
            if ((hT >> 16) != kListAbsent && (hT & 0xffff) > 0) row = rj[kRecHead + q.top_state * SW + 1];   // plan.go:134-138
            const int hs = rj[kRecHead + s * SW];
            nown = (hs >> 16) == kListAbsent ? 0 : (hs & 0xffff);
            if (nown > k) { pok = false; nown = 0; }
            if (pok) {
#pragma unroll
                for (int j = 0; j < KM; j++) {
                    if (j < nown) {
                        const int o = rj[kRecHead + s * SW + 1 + j];
                        ownv[j] = o;
                        if (o >= N) pok = false;
                    }
                }
            }
            if (!pok) {
                nown = 0;
#pragma unroll
                for (int j = 0; j < KM; j++) ownv[j] = -1;
            }
        }
        // simple: every own node is a candidate, held once, in no other list; few nodes in the other lists
        bool simple = pok;
        if (simple) {
#pragma unroll
            for (int j = 0; j < KM; j++) {
                if (j < nown) {
                    if (!(flL[ownv[j]] & 1)) simple = false;
#pragma unroll
                    for (int jj = 0; jj < KM; jj++) if (jj < j && ownv[jj] == ownv[j]) simple = false;
                }
            }
            int n_h = 0, n_o = 0;
            for (int t = 0; t < M; t++) {
                if (t == s) continue;
                const int h = rj[kRecHead + t * SW];
                if ((h >> 16) == kListAbsent) continue;
                const bool higher = (q.higher_mask >> t) & 1;
                for (int jj = 0; jj < (h & 0xffff); jj++) {
                    const int x = rj[kRecHead + t * SW + 1 + jj];
#pragma unroll
                    for (int j = 0; j < KM; j++) if (ownv[j] == x) simple = false;   // excluded or demoted: not a plain stay
                    if (higher) {
                        if (n_h >= KH) simple = false;
#pragma unroll
                        for (int e = 0; e < KH; e++) if (e == n_h) hv[e] = x;
                        n_h++;
                    } else {
                        if (n_o >= KO || x > 0xffff) simple = false;
#pragma unroll
                        for (int e = 0; e < KO; e++) if (e == n_o) ov[e] = x | (t << 16);
                        n_o++;
                    }
                }
            }
        }
        PH(1);
        bool dirty = false;                          // an earlier step of the batch bumps my row
        if (NP > 0) {
#pragma unroll
            for (int j = 0; j < KM; j++)
                if (pok && j < nown) ntn_own[j] = BLANCE_LD_COHERENT(q.ntn + (size_t)row * N + ownv[j]);
            for (int i = 0; i < B - 1; i++) {
                const int ri = __builtin_amdgcn_readlane(row, i);
                if (lane > i && row == ri) dirty = true;
            }
        }
        if (NP > 0) {
            // every step of the batch without a top priority node: fold the shared row "" into the leaves
            const int want = (__ballot(act && row != NX) == 0) ? NX : -1;
            if (want != fold) {
                fold = want;
#ifdef BLANCE_FOLD_TRACE
                if (lane == 0) printf("[tree] batch at step %d: fold -> %d\n", oi, fold);
#endif
                if (fold >= 0)
                    for (int i = 0; i < G; i++) {
                        const int n = i * 64 + lane;
                        ntL[n] = n < N ? BLANCE_LD_COHERENT(q.ntn + (size_t)fold * N + n) : 0;
                    }
                BLANCE_WAVE_SYNC();
                rebuild_tree();
                root_valid = false;
            }
        }
        PH(2);
        u64 lastB = 0;
        int lastN = -1;
        bool sfail = !simple || nown != k;           // fewer nodes than constraints: never a stay
        bool stale = false;                          // an earlier general step of the batch touched my own nodes
        int sortv[KM];                               // the own nodes in (score, position) order: what a stay emits
#pragma unroll
        for (int j = 0; j < KM; j++) sortv[j] = 0;
        if (simple) {
            u64 sK[KM];
#pragma unroll
            for (int j = 0; j < KM; j++) sK[j] = ~0ull;
#pragma unroll
            for (int j = 0; j < KM; j++) {
                if (j < nown) {
                    const int o = ownv[j];
                    const u64 b = sortable_bits(tree_score(cntL[o], ntn_own[j], totL[o], (flL[o] >> 1) & 1, wL[o], NP,
                                                           vstick, q.booster_kind, lpT, ffT));
                    oKh[j] = (unsigned)(b >> 32); oKl[j] = (unsigned)b;
                    // insertion into the sorted prefix: keeping the same nodes in another order changes no counter
                    u64 cb = b;
                    int cn = o;
#pragma unroll
                    for (int e = 0; e < KM; e++) {
                        if (e <= j) {
                            const bool first = e == j || key_less(cb, cn, sK[e], sortv[e]);
                            if (first) {
                                const u64 tb = sK[e]; const int tn = sortv[e];
                                sK[e] = cb; sortv[e] = cn;
                                cb = tb; cn = tn;
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < KM; j++) if (j == k - 1) { lastB = sK[j]; lastN = sortv[j]; }
        }
        if constexpr (KM == 2) {
            int* mv = mvL + lane * kMvW;
            mv[0] = wj; mv[1] = row;
            mv[2] = ownv[0]; mv[3] = ownv[1];
            mv[4] = (int)oKh[0]; mv[5] = (int)oKl[0]; mv[6] = (int)oKh[1]; mv[7] = (int)oKl[1];
            mv[8] = hv[0]; mv[9] = hv[1];
#pragma unroll
            for (int e = 0; e < KO; e++) mv[10 + e] = ov[e];
        }
        BLANCE_WAVE_SYNC();

        PH(3);
        // ---- the batch in order: validated runs at once, the other steps one by one
        pf_f = -1;                                   // prefetched entries are tagged with a lane of THIS batch
        // Inside a batch the steps only write LDS: their outputs are staged in outS, and the bumps of nodeToNodeCounts
        // (plan.go:238-245) wait there too -- no step reads an entry an earlier step of the batch bumps, except the
        // "dirty" ones (same row as an earlier lane), which get the pending bumps flushed first.  So the only global
        // memory operations between two steps are the prefetches, and waiting for one never waits for a store.
        int bumped_upto = 0;                         // steps [0, bumped_upto) of the batch have their rows bumped
        const int OWs = q.OW;
        auto flush_bumps = [&](int upto) {
            if (NP > 0 && lane >= bumped_upto && lane < upto) {
                const int n = outS[lane * OWs] & 0xffff;
                for (int j = 0; j < n; j++) {
                    const int x = outS[lane * OWs + 1 + j];
                    if (x >= 0 && x < N) atomicAdd(q.ntn + (size_t)row * N + x, 1);
                }
            }
            bumped_upto = upto > bumped_upto ? upto : bumped_upto;
        };
        int cur = 0;
        while (cur < B) {
            PH(11);
            if (!root_valid) compute_top2();
            const bool fail = sfail || dirty || fold >= 0 || !key_less(lastB, lastN, rootB, root_n);
            const u64 fm = __ballot(act && fail) & (~0ull << cur);
            const int f = fm ? __ffsll((long long)fm) - 1 : B;
            if (lane >= cur && lane < f) {          // certain stays: plan.go:299-301 leaves everything as it is
                int* o = outS + lane * OWs;
                o[0] = k;
#pragma unroll
                for (int j = 0; j < KM; j++) if (j < k) o[1 + j] = sortv[j];
            }
            n_bulk += f - cur;
            PH(4);
            if (f >= B) break;
            PC(pc_general);
            const int rowf = __builtin_amdgcn_readlane(row, f);
            // nodeToNodeCounts entries of the two smallest leaves.  Lanes 4 and 5 fetched, a step ago, the entries
            // of the lane that was expected to fail next for the runner-up and the root of then -- one of them is
            // the root of now unless a lowered node got in front; what is missing is fetched now (lanes 2, 3), and
            // the same is started for the lane after this one.
            int nt1 = 0, nt2 = 0;                     // root's / runner-up's entry when hit1 / hit2
            bool hit1 = false, hit2 = false;
            if (NP > 0 && fold < 0) {
                const int e4n = __builtin_amdgcn_readlane(pf_n, 4), e4f = __builtin_amdgcn_readlane(pf_f, 4);
                const int e5n = __builtin_amdgcn_readlane(pf_n, 5), e5f = __builtin_amdgcn_readlane(pf_f, 5);
                const int e4v = __builtin_amdgcn_readlane(pf_v, 4), e5v = __builtin_amdgcn_readlane(pf_v, 5);
                if (e4f == f && e4n == root_n) { hit1 = true; nt1 = e4v; }
                else if (e5f == f && e5n == root_n) { hit1 = true; nt1 = e5v; }
                if (e4f == f && e4n == t2n) { hit2 = true; nt2 = e4v; }
                else if (e5f == f && e5n == t2n) { hit2 = true; nt2 = e5v; }
                const u64 fm2 = fm & (fm - 1);                         // the lane expected to fail after this one
                const int f2 = fm2 ? __ffsll((long long)fm2) - 1 : -1;
                const int rowf2 = __builtin_amdgcn_readlane(row, f2 < 0 ? 0 : f2);
                int my_node = -1, my_row = 0, my_f = -1;
                if (lane == 2 && !hit1) { my_node = root_n; my_row = rowf; my_f = f; }
                if (lane == 3 && !hit2) { my_node = t2n; my_row = rowf; my_f = f; }
                if (lane == 4 && f2 >= 0) { my_node = t2n; my_row = rowf2; my_f = f2; }
                if (lane == 5 && f2 >= 0) { my_node = root_n; my_row = rowf2; my_f = f2; }
                if (my_node >= 0 && my_node < N) {
                    pf_v = BLANCE_LD_COHERENT(q.ntn + (size_t)my_row * N + my_node);
                    pf_n = my_node; pf_f = my_f;
                }
            }

            // ================= general step for lane f's record =================
            const int* rf = recS + f * RW;
            const int w = __builtin_amdgcn_readlane(wj, f);
            const bool dirty_f = __builtin_amdgcn_readlane(dirty ? 1 : 0, f) != 0;
            if (dirty_f && fold < 0 && NP > 0) {   // this step reads a row an earlier step of the batch bumps
                flush_bumps(f);
                BLANCE_AGENT_FENCE();
            }
            const bool quick = !no_short && !dirty_f && __builtin_amdgcn_readlane((simple && !stale) ? 1 : 0, f) != 0;
            // the first candidate's nodeToNodeCounts entry: in flight while the step is decoded
            const bool pre_ok = NP > 0 && fold < 0 && !dirty_f && root_n < N;   // the root's entry: fetched above
            const int pre_n = root_n;

            // what either form of the general step leaves behind: the chosen nodes, and per lane a node whose counters changed
            int bN[KM];
#pragma unroll
            for (int j = 0; j < KM; j++) bN[j] = INT_MAX;
            int n_out = 0;
            int hx = -1;
            bool changed = false;
            bool lean_done = false;
            if constexpr (KM == 2) {
                // ---- the lean general step (k <= 2, a "short" step): the four contenders -- the partition's own nodes with
                // the exact keys of its validating lane, and the two smallest leaves -- sit in lanes 0..3, are sorted by a
                // three-stage compare-exchange network inside the quad, and commit lane-parallel.  Nothing unexamined can
                // get in as long as the k-th taken is not after the runner-up in (g, position) order; otherwise, and for
                // promotions / demotions, the general code below takes the step.
                if (quick && fold < 0) {
                    PC(pc_lean_try);
                    const int* mv = mvL + f * kMvW;
                    const int own0 = mv[2], own1 = mv[3], h0 = mv[8], h1 = mv[9];
                    int s_n = INT_MAX, s_org = lane & 3;
                    u64 s_b = ~0ull;
                    if (lane < 2) {
                        const int n = mv[2 + lane];
                        if (n >= 0) { s_n = n; s_b = ((u64)(unsigned)mv[4 + 2 * lane] << 32) | (unsigned)mv[5 + 2 * lane]; }
                    } else if (lane < 4) {
                        const int c = lane == 2 ? root_n : t2n;
                        const u64 cB = lane == 2 ? rootB : t2B;
                        if (c != INT_MAX && c != own0 && c != own1 && c != h0 && c != h1) {     // plan.go:142-156
                            // the root with its exact score; the runner-up with its lower bound g for now: its entry
                            // is only waited for if that gets it taken
                            int nt = 0;
                            if (NP > 0 && lane == 2) nt = hit1 ? nt1 : pf_v;
                            if (NP > 0 && lane == 3 && hit2) nt = nt2;
                            s_n = c;
                            s_b = nt ? sortable_bits(tree_score(cntL[c], nt, totL[c], (flL[c] >> 1) & 1, wL[c], NP, 0.0,
                                                                q.booster_kind, lpT, ffT)) : cB;
                        }
                    }
#define BLANCE_CEX(CTRL, LOWER)                                                                                  \
                    {                                                                                            \
                        const unsigned ph_ = (unsigned)dpp_mov<CTRL>((int)(unsigned)(s_b >> 32));                \
                        const unsigned pl_ = (unsigned)dpp_mov<CTRL>((int)(unsigned)s_b);                        \
                        const int pn_ = dpp_mov<CTRL>(s_n), po_ = dpp_mov<CTRL>(s_org);                          \
                        const u64 pb_ = ((u64)ph_ << 32) | pl_;                                                  \
                        const bool mine_ = key_less(s_b, s_n, pb_, pn_), theirs_ = key_less(pb_, pn_, s_b, s_n); \
                        if ((LOWER) ? theirs_ : mine_) { s_b = pb_; s_n = pn_; s_org = po_; }                    \
                    }
#ifdef BLANCE_PHASE_PROF
                    if (hit1) pc_hit1++;
                    if (hit2) pc_hit2++;
#endif
                    BLANCE_CEX(0xB1, !(lane & 1))     // (0,1) (2,3): the lower lane keeps the smaller
                    BLANCE_CEX(0x4E, !(lane & 2))     // (0,2) (1,3)
                    BLANCE_CEX(0xD8, lane == 1)       // (1,2)
#undef BLANCE_CEX
                    const bool taken = lane < k && s_n != INT_MAX;
                    bool ok = __popcll(__ballot(taken)) == k;
                    const u64 kB = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(s_b >> 32), k - 1) << 32) |
                                   (unsigned)__builtin_amdgcn_readlane((int)(unsigned)s_b, k - 1);
                    const int kn = __builtin_amdgcn_readlane(s_n, k - 1);
                    if (t2n != INT_MAX && key_less(t2B, t2n, kB, kn)) ok = false;      // a leaf after the runner-up could get in
                    // taken from the tree but held by the partition in another state: promoted / demoted
                    bool prom = false;
#pragma unroll
                    for (int e = 0; e < KO; e++) prom = prom || (taken && s_org >= 2 && (mv[10 + e] & 0xffff) == s_n);
                    if (__ballot(prom)) ok = false;
#ifdef BLANCE_PHASE_PROF
                    if (__ballot(taken && s_org == 3)) pc_need2++;
#endif
                    if (ok && NP > 0 && !hit2 && __ballot(taken && s_org == 3)) {
                        // the runner-up was taken on its lower bound: exact only if its entry is 0 (else the general code)
                        if (__builtin_amdgcn_readlane(pf_v, 3) != 0) ok = false;
                    }
                    if (ok) {
                        PC(pc_lean);
                        const bool valid = lane < 4 && s_n != INT_MAX;
                        const bool enter = valid && lane < k && s_org >= 2, leave = valid && lane >= k && s_org < 2;
                        changed = enter || leave;
                        if (changed) {               // plan.go:290-301
                            const int ds = enter ? w : -w;
                            hx = s_n;
                            cntL[s_n] += ds;
                            totL[s_n] += ds;
                            gB[s_n] = leaf_key(s_n);
                        }
#pragma unroll
                        for (int j = 0; j < KM; j++) if (j < k) bN[j] = __builtin_amdgcn_readlane(s_n, j);
                        n_out = k;
                        lean_done = true;
                    }
                }
            }
            if (!lean_done) {
            // the k best (score, position) so far; wave uniform, ascending
            u64 bB[KM];
#pragma unroll
            for (int j = 0; j < KM; j++) { bB[j] = ~0ull; bN[j] = INT_MAX; }
            auto insert = [&](u64 b, int n) {
#pragma unroll
                for (int j = KM - 1; j >= 0; j--) {
                    const bool here = key_less(b, n, bB[j], bN[j]);
                    const bool above = j > 0 && key_less(b, n, bB[j - 1], bN[j - 1]);
                    if (here) {
                        if (above) { bB[j] = bB[j - 1]; bN[j] = bN[j - 1]; }
                        else { bB[j] = b; bN[j] = n; }
                    }
                }
            };

            // ---- what the step knows about its partition: from lane f's registers (quick), or from the record
            int qown[KM], qh[KH], qo[KO];           // quick: own / higher / other nodes, wave uniform
#pragma unroll
            for (int j = 0; j < KM; j++) qown[j] = -1;
#pragma unroll
            for (int j = 0; j < KH; j++) qh[j] = -1;
#pragma unroll
            for (int j = 0; j < KO; j++) qo[j] = -1;
            double stick = 0.0;
            int wd = -1;                             // record decode: my word of the record's lists, and what it is
            bool valid = false, own_first = false;
            u64 m_own = 0, m_high = 0, m_oth = 0;
            if (quick) {
                PC(pc_short);
#pragma unroll
                for (int j = 0; j < KM; j++) {
                    if (j < k) {
                        qown[j] = __builtin_amdgcn_readlane(ownv[j], f);     // -1 beyond the list's length
                        const unsigned bh = (unsigned)__builtin_amdgcn_readlane((int)oKh[j], f);
                        const unsigned bl = (unsigned)__builtin_amdgcn_readlane((int)oKl[j], f);
                        if (qown[j] >= 0) insert(((u64)bh << 32) | bl, qown[j]);
                    }
                }
#pragma unroll
                for (int j = 0; j < KH; j++) qh[j] = __builtin_amdgcn_readlane(hv[j], f);
#pragma unroll
                for (int j = 0; j < KO; j++) qo[j] = __builtin_amdgcn_readlane(ov[j], f);
            } else {
                stick = __hiloint2double(uni(rf[3]), uni(rf[2]));
                bool hdr_present = false;
                if (slot_ok) {
                    const int hdr = rf[kRecHead + slot_t * SW];
                    hdr_present = (hdr >> 16) != kListAbsent;
                    wd = rf[kRecHead + lane];
                    valid = slot_ix >= 1 && hdr_present && slot_ix - 1 < (hdr & 0xffff);
                }
                m_own = __ballot(valid && slot_t == s);
                m_high = __ballot(valid && slot_higher);
                m_oth = __ballot(valid && slot_t != s);
                // the partition's own nodes, scored exactly by the lanes that hold them
                const bool is_own = (m_own >> lane) & 1;
                own_first = is_own;                  // first occurrence of the node in the list
                bool own_elig = is_own && wd < N && (flL[is_own ? wd : 0] & 1);
                for (u64 mm = m_own; mm; mm &= mm - 1) {
                    const int h = __ffsll((long long)mm) - 1;
                    const int y = __builtin_amdgcn_readlane(wd, h);
                    if (is_own && lane > h && wd == y) own_first = false;
                }
                for (u64 mm = m_high; mm; mm &= mm - 1) {
                    const int h = __ffsll((long long)mm) - 1;
                    const int y = __builtin_amdgcn_readlane(wd, h);
                    if (is_own && wd == y) own_elig = false;          // plan.go:146-154
                }
                own_elig = own_elig && own_first;
                int own_nt = 0;
                if (NP > 0 && own_elig) own_nt = fold >= 0 ? ntL[wd] : BLANCE_LD_COHERENT(q.ntn + (size_t)rowf * N + wd);
                u64 own_b = ~0ull;
                if (own_elig)
                    own_b = sortable_bits(tree_score(cntL[wd], own_nt, totL[wd], (flL[wd] >> 1) & 1, wL[wd], NP, stick,
                                                     q.booster_kind, lpT, ffT));
                for (u64 mm = __ballot(own_elig); mm; mm &= mm - 1) {
                    const int h = __ffsll((long long)mm) - 1;
                    const unsigned bh = (unsigned)__builtin_amdgcn_readlane((int)(own_b >> 32), h);
                    const unsigned bl = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)own_b, h);
                    insert(((u64)bh << 32) | bl, __builtin_amdgcn_readlane(wd, h));
                }
            }
            // is node c (wave uniform) one of the partition's own or higher priority nodes?
            auto skip_node = [&](int c) -> bool {
                if (quick) {
                    bool hit = false;
#pragma unroll
                    for (int j = 0; j < KM; j++) hit = hit || (j < k && qown[j] == c);
#pragma unroll
                    for (int j = 0; j < KH; j++) hit = hit || qh[j] == c;
                    return hit;
                }
                return (__ballot(valid && wd == c) & (m_own | m_high)) != 0;
            };

            PH(5);
            PH(6);
            // ---- walk the candidates in (g, position) order
            unsigned wm_hi = gm_hi, wm_lo = gm_lo;
            int wm_n = gm_n;
            int n_taken = 0;
            bool dense = walk_cap == 0;
            u64 nextB = rootB;                       // the tree is as the cached root saw it
            int next_n = root_n, next_lane = root_n == INT_MAX ? 0 : root_n >> 6;
            while (!dense) {
                if (next_n == INT_MAX) break;                                   // no candidate left
                if (bN[k - 1] != INT_MAX && key_less(bB[k - 1], bN[k - 1], nextB, next_n)) break;   // nothing can get in
                if (n_taken >= walk_cap) { dense = true; break; }
                const int c = next_n, gl = next_lane;
                const u64 cB = nextB;
                if (lane == 0) { stkB[n_taken] = cB; stkN[n_taken] = c; }
                n_taken++;
                PC(pc_taken);
                if (!skip_node(c)) {                                            // own: scored already; higher: no candidate
                    int nt = 0;
                    if (NP > 0 && fold < 0) {                                  // folded: the leaf is the exact score
                        const bool pref = pre_ok && pre_n == c;
                        nt = pref ? (hit1 ? nt1 : __builtin_amdgcn_readlane(pf_v, 2))
                                  : uni(BLANCE_LD_COHERENT(q.ntn + (size_t)rowf * N + c));
#ifdef BLANCE_PHASE_PROF
                        if (!pref) pc_miss++;
#endif
                    }
                    u64 eB = cB;                                               // entry 0: the score IS g
                    if (nt != 0)
                        eB = sortable_bits(tree_score(uni(cntL[c]), nt, uni(totL[c]), (uni((int)flL[c]) >> 1) & 1,
                                                      uni(wL[c]), NP, 0.0, q.booster_kind, lpT, ffT));
                    insert(eB, c);
                    // full, and its last entry is not after c in walk order: every later node scores >= its g > c's
                    if (bN[k - 1] != INT_MAX && !key_less(cB, c, bB[k - 1], bN[k - 1])) break;
                }
                // take c out of the tree and find the next candidate
                if (lane == (c & 63)) gB[c] = ~0ull;
                BLANCE_WAVE_SYNC();
                unsigned rh, rl; int rn;
                scan_group(gl, rh, rl, rn);
                if (lane == gl) { wm_hi = rh; wm_lo = rl; wm_n = rn; }
                const TreeMin m = wave_min_u64_lane(wm_hi, wm_lo);
                nextB = ((u64)m.hi << 32) | m.lo;
                next_n = __builtin_amdgcn_readlane(wm_n, m.lane);
                next_lane = m.lane;
            }
            PH(7);
            // put the walked leaves back (the chosen ones get new values below)
            if (n_taken > 0) {
                BLANCE_WAVE_SYNC();
                if (lane < n_taken) gB[stkN[lane]] = stkB[lane];
                BLANCE_WAVE_SYNC();
            }
            if (dense) {
                // ---- rare: score every node (exactly what the reference's sort sees), k successive minima
                if (quick) stick = __hiloint2double(uni(rf[3]), uni(rf[2]));
#pragma unroll
                for (int j = 0; j < KM; j++) { bB[j] = ~0ull; bN[j] = INT_MAX; }
                for (int pick = 0; pick < k; pick++) {
                    u64 lb = ~0ull;
                    int ln = INT_MAX;
                    for (int i = 0; i < G; i++) {
                        const int n = i * 64 + lane;
                        bool el = n < N && (flL[n] & 1);
#pragma unroll
                        for (int j = 0; j < KM; j++) if (bN[j] == n) el = false;
                        bool own = false;
                        if (quick) {
#pragma unroll
                            for (int j = 0; j < KM; j++) if (j < k && qown[j] == n) own = true;
#pragma unroll
                            for (int j = 0; j < KH; j++) if (qh[j] == n) el = false;
                        } else {
                            for (u64 mm = m_high | m_own; mm; mm &= mm - 1) {
                                const int h = __ffsll((long long)mm) - 1;
                                const int y = __builtin_amdgcn_readlane(wd, h);
                                if (y == n) { if ((m_high >> h) & 1) el = false; else own = true; }
                            }
                        }
                        if (el) {
                            const int nt = NP <= 0 ? 0 : (fold >= 0 ? ntL[n] : BLANCE_LD_COHERENT(q.ntn + (size_t)rowf * N + n));
                            const u64 b = sortable_bits(tree_score(cntL[n], nt, totL[n], (flL[n] >> 1) & 1, wL[n], NP,
                                                                   own ? stick : 0.0, q.booster_kind, lpT, ffT));
                            if (key_less(b, n, lb, ln)) { lb = b; ln = n; }
                        }
                    }
                    const unsigned mh = wave_min_u32_bcast((unsigned)(lb >> 32));
                    const bool k2 = (unsigned)(lb >> 32) == mh;
                    const unsigned ml = wave_min_u32_bcast(k2 ? (unsigned)lb : kKeyNoneV);
                    const bool k3 = k2 && (unsigned)lb == ml;
                    const unsigned mn = wave_min_u32_bcast(k3 ? (unsigned)ln : kKeyNoneV);
                    if ((int)mn == INT_MAX || mn == kKeyNoneV) break;
#pragma unroll
                    for (int j = 0; j < KM; j++) if (j == pick) { bB[j] = ((u64)mh << 32) | ml; bN[j] = (int)mn; }
                }
            }
            n_out = 0;
#pragma unroll
            for (int j = 0; j < KM; j++) if (j < k && bN[j] != INT_MAX) n_out++;
            PH(8);
#ifdef BLANCE_PHASE_PROF
            if (quick) {
                bool same = n_out == k;
#pragma unroll
                for (int j = 0; j < KM; j++) if (j < k && bN[j] != qown[j]) same = false;
                if (same) pc_stay++;
                int kept = 0;
#pragma unroll
                for (int j = 0; j < KM; j++)
#pragma unroll
                    for (int jj = 0; jj < KM; jj++) if (j < k && jj < n_out && bN[jj] == qown[j]) kept++;
                if (!same && kept == k) pc_reorder++;
                if (kept == k - 1 && k > 1) pc_half++;
            }
#endif

            // ---- commit (plan.go:238-245, :290-301).  A node of this state's old list leaves it, a
            // chosen node enters it, and a node that is either also leaves every OTHER list of the
            // partition that holds it.  Own nodes are settled by lanes of their own (quick: lanes
            // 0..k-1, else the lanes that hold them), newly chosen ones by lanes 60..63 (never list
            // words: records are at most 64 words).
            hx = -1;                                 // the node this lane settles
            bool h_own = false, h_chosen = false;
            int n_oth = 0;
            if (quick) {
#pragma unroll
                for (int j = 0; j < KM; j++) if (j < k && lane == j && qown[j] >= 0) { hx = qown[j]; h_own = true; }
#pragma unroll
                for (int j = 0; j < KM; j++) {
                    if (j < n_out) {
                        bool inown = false;
#pragma unroll
                        for (int jj = 0; jj < KM; jj++) inown = inown || (jj < k && qown[jj] == bN[j]);
                        if (h_own && hx == bN[j]) h_chosen = true;
                        if (!inown && lane == 60 + j) { hx = bN[j]; h_chosen = true; }
                    }
                }
                if (lane >= 60 && hx >= 0) {         // a chosen node the partition holds in another state: promoted / demoted
#pragma unroll
                    for (int e = 0; e < KO; e++) {
                        if (qo[e] >= 0 && (qo[e] & 0xffff) == hx) {
                            n_oth++;
                            q.cnt[(qo[e] >> 16) * NX + hx] -= w;
                        }
                    }
                }
            } else {
                if (own_first) { hx = wd; h_own = true; }
#pragma unroll
                for (int j = 0; j < KM; j++) {
                    if (j < n_out) {
                        const bool inown = (__ballot(valid && wd == bN[j]) & m_own) != 0;
                        if (h_own && wd == bN[j]) h_chosen = true;
                        if (!inown && lane == 60 + j) { hx = bN[j]; h_chosen = true; }
                    }
                }
                for (u64 mm = m_oth; mm; mm &= mm - 1) {
                    const int h = __ffsll((long long)mm) - 1;
                    const int y = __builtin_amdgcn_readlane(wd, h);
                    const int ty = __builtin_amdgcn_readlane(slot_t, h);
                    if (hx >= 0 && hx == y) {
                        n_oth++;
                        q.cnt[ty * NX + y] -= w;     // only this lane touches that counter
                    }
                }
            }
            const int ds = (h_chosen ? w : 0) - (h_own ? w : 0);
            const bool bumped = fold >= 0 && h_chosen;      // folded: the chosen node's row entry is part of its leaf
            changed = hx >= 0 && (ds != 0 || n_oth > 0 || bumped);
            if (changed) {
                cntL[hx] += ds;
                totL[hx] += ds - w * n_oth;
                if (bumped) ntL[hx] += 1;
                gB[hx] = leaf_key(hx);
            }
            }   // !lean_done
            BLANCE_WAVE_SYNC();
            PH(9);
            // the tree: a smaller leaf replaces its group's minimum in place, a grown minimum needs a scan
            const u64 chm = __ballot(changed);
            for (u64 mm = chm; mm; mm &= mm - 1) {
                const int h = __ffsll((long long)mm) - 1;
                const int x = __builtin_amdgcn_readlane(hx, h);
                const int gl = x >> 6;
                const u64 nb = gB[x];                                          // uniform address
                const u64 ob = ((u64)(unsigned)__builtin_amdgcn_readlane((int)gm_hi, gl) << 32) |
                               (unsigned)__builtin_amdgcn_readlane((int)gm_lo, gl);
                const int on = __builtin_amdgcn_readlane(gm_n, gl);
                if (key_less(nb, x, ob, on)) {
                    if (lane == gl) { gm_hi = (unsigned)(nb >> 32); gm_lo = (unsigned)nb; gm_n = x; }
                } else if (on == x && nb != ob) {
                    PC(pc_scans);
                    unsigned rh, rl; int rn;
                    scan_group(gl, rh, rl, rn);
                    if (lane == gl) { gm_hi = rh; gm_lo = rl; gm_n = rn; }
                }
                // later lanes of the batch that hold x were validated against its old counters
#pragma unroll
                for (int j = 0; j < KM; j++) if (lane > f && j < k && ownv[j] == x) { sfail = true; stale = true; }
            }
            if (chm) root_valid = false;

            if (lane == 0) {
                int is_nil = 0;
                if (n_out == 0 && q.n_alive == 0 && !q.hier) {                   // plan.go:142: a nil slice stays nil
                    bool any_higher_key = false;
                    for (int t = 0; t < M; t++)
                        if (((q.higher_mask >> t) & 1) && (rf[kRecHead + t * SW] >> 16) != kListAbsent) any_higher_key = true;
                    is_nil = !any_higher_key;
                }
                int* o = outS + f * OWs;             // (its row is bumped with the batch's, plan.go:238-245)
                o[0] = n_out | (is_nil << 16);
#pragma unroll
                for (int j = 0; j < KM; j++) if (j < k) o[1 + j] = j < n_out ? bN[j] : -1;
                if (n_out < k) {                     // plan.go:230-235
                    const int wi = *q.warn_count;
                    q.warn_part[wi] = rf[0];
                    q.warn_state[wi] = s;
                    *q.warn_count = wi + 1;
                }
            }
            cur = f + 1;
            PH(10);
        }
        // ---- the batch's outputs, and the bumps still pending
        BLANCE_WAVE_SYNC();
        flush_bumps(B);
        for (int idx = lane; idx < B * OWs; idx += 64) q.out[(size_t)oi * OWs + idx] = outS[idx];
        BLANCE_WAVE_SYNC();
    }
#ifdef BLANCE_PHASE_PROF
    if (lane == 0) {
        printf("[tree] k %d steps %d batches %lld general %lld (lean %lld of %lld tried, root hit %lld, runner-up hit %lld needed %lld; short %lld: stays %lld reorders %lld one-kept %lld) walked %lld prefetch-miss %lld commit-scans %lld\n",
               k, q.end - q.beg, pc_batches, pc_general, pc_lean, pc_lean_try, pc_hit1, pc_hit2, pc_need2, pc_short, pc_stay, pc_reorder, pc_half, pc_taken, pc_miss, pc_scans);
        for (int i_ = 0; i_ < 12; i_++) printf("[tree phase %d] %.0f kcycles\n", i_, (double)ph_acc[i_] / 1e3);
    }
#endif
    if (lane == 0 && q.spec_count) *q.spec_count += n_bulk;
    BLANCE_WAVE_SYNC();
    for (int i = 0; i < G; i++) {
        const int n = i * 64 + lane;
        if (n < NX) q.cnt[s * NX + n] = cntL[n];
    }
}

// dynamic LDS of k_pass_tree for a pass
static inline size_t tree_lds_bytes(int NX, int RW) {
    const size_t NXp = (size_t)((NX + 63) / 64) * 64;
    return NXp * (8 + 4 + 4 + 4 + 4 + 1) + sizeof(int32_t) * (size_t)(64 * RW) +
           sizeof(double) * (kLpTab + kFfTab) + (size_t)kWalkCap * 12 + sizeof(int32_t) * 64 * (kMvW + 5) + 64;
}

}  // namespace blance
