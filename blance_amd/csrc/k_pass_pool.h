// k_pass_pool: the exact sequential state pass of a state WITHOUT hierarchy rules, k <= 2, on one wave64 whose
// lanes hold the POOL of the nodes with the smallest scores -- a general step is a few wave minima over registers.
// Part of libblance_hip.so (tu_pool.hip); see DESIGN.md section 4.3.
#pragma once

namespace blance {

// ============================================================================
// assignStateToPartitions (plan.go:253-303) with findBestNodes (plan.go:98-248).  Same facts as k_pass_tree (a
// node that is not the partition's own scores >= g, its partition-independent score; a step changes g of at most
// old + chosen nodes) and the same batches of 64 steps (lane j validates step j as a certain stay), but another
// candidate structure for the steps that do move a copy:
//
//  * The POOL: the <= 64 nodes with the smallest (g, node), one per lane, g kept in that lane's registers; THETA =
//    the smallest (g, node) outside the pool (exact when the pool is built from all nodes, lowered when a node
//    outside is lowered).  A general step scores the pool entries for its partition lane-parallel -- the entry is
//    its g unless the partition's nodeToNodeCounts entry for it is non-zero (those entries are fetched per batch,
//    one load per moving step for all 64 lanes at once) -- and takes the k best of (own nodes with stickiness,
//    pool entries) by two wave minima; whatever it takes is proved to be before THETA, so nothing outside the pool
//    had to be looked at (else the pool is rebuilt and the step done again).
//  * Taken entries stay in the pool with their new g (in the chains that dominate a weighted rebalance the node just
//    raised is often the next best again); an own node given up joins the pool if it is before THETA.  Entries at
//    or behind THETA are dropped: nobody can take them before the next rebuild.
//  * No tree, no sorted window, no per-step maintenance: a step is ~250 instructions, a rebuild ~2,000 every few
//    dozen steps.
//
// Everything this kernel does not do itself -- partitions holding a node in two states, more than two higher
// priority nodes, promotions / demotions, weights <= 0, unmet constraints -- ends the launch (*stop_at = that step):
// the host runs one batch of k_pass_tree there and launches this kernel again behind it.
// ============================================================================
constexpr int kPoolEStride = 65;         // words per row of E: lane-contiguous both ways

template <int KM>
__global__ __launch_bounds__(64) void k_pass_pool(PassParams q) {
    static_assert(KM == 2, "two copies at most");
    typedef unsigned long long u64;
    constexpr int KH = 2, KO = 4;
    BLANCE_DYN_LDS(lds);
    const int lane = threadIdx.x;
    const int N = q.N, NX = q.NX, M = q.M, L = q.L, NP = q.NP, s = q.s, k = q.k, RW = q.RW;
    const int SW = 1 + L;
    const int G = (NX + 63) >> 6, NXp = G << 6;
    const u64 below = lane ? (~0ull >> (64 - lane)) : 0ull;

    u64* gB = (u64*)lds;                             // [NXp] sortable image of g of every node; ~0: no candidate
    u64* tmpK = gB + NXp;                            // [64] scratch of the rebuild
    int* cntL = (int*)(tmpK + 64);                   // [NXp] stateNodeCounts[s]
    int* totL = cntL + NXp;                          // [NXp] nodePartitionCounts (plan.go:118-124)
    int* wL = totL + NXp;                            // [NXp] node weights
    int* recS = wL + NXp;                            // [64 * RW] step records of the batch
    double* lpT = (double*)(recS + 64 * RW);         // [kLpTab] c / NP
    double* ffT = lpT + kLpTab;                      // [kFfTab] (0.001 * t) / NP
    int* tmpN = (int*)(ffT + kFfTab);                // [64]
    int* E = tmpN + 64;                              // [64 steps][kPoolEStride] nodeToNodeCounts[row of the step][pool entry's node]
    int* outS = E + 64 * kPoolEStride;               // [64][OW] the batch's outputs
    unsigned char* flL = (unsigned char*)(outS + 64 * 4);    // [NXp] 1: in nodesNext, 2: has a weight
    unsigned char* slotOf = flL + NXp;               // [NXp] the pool lane that holds the node, 0xff: none
    unsigned char* rcOf = slotOf + NXp;              // [NXp] the node's column of R (an own node of a moving step of the batch), 0xff: none
    int* R = (int*)(rcOf + NXp);                     // [64 columns][64 steps] nodeToNodeCounts[row of the step][own node of an EARLIER moving
                                                     // step]: what E needs when that node joins the pool

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
        cntL[n] = c; totL[n] = t; wL[n] = w; flL[n] = (unsigned char)fl; slotOf[n] = 0xff; rcOf[n] = 0xff;
    }
    BLANCE_WAVE_SYNC();
    auto g_key = [&](int n) -> u64 {                 // plan.go:634-689 without the partition's own terms
        return (flL[n] & 1) ? sortable_bits(tree_score(cntL[n], 0, totL[n], (flL[n] >> 1) & 1, wL[n], NP, 0.0, q.booster_kind,
                                                      lpT, ffT)) : ~0ull;
    };
    for (int i = 0; i < G; i++) gB[i * 64 + lane] = g_key(i * 64 + lane);
    BLANCE_WAVE_SYNC();

    // ---- the pool: my entry (pn < 0: none) and the bound on everything outside
    int pn = -1;
    u64 pk = ~0ull;
    u64 thK = ~0ull;
    int thN = INT_MAX;
    bool pool_ok = false;
    PH_DECL;
#undef PC
#ifdef BLANCE_PAR_STATS
    long long pc_slowjoins = 0, pc_batches = 0, pc_steps = 0, pc_rebuilds = 0, pc_exact = 0, pc_joins = 0, pc_drops = 0, pc_eloads = 0, pc_stale = 0, pc_ties = 0;
#define PC(x) (x)++
#else
#define PC(x)
#endif
    // (key, node) minimum over the lanes' candidates; the lane that holds it (-1: none)
    auto pool_min = [&](u64 ck, int cn, u64& rk, int& rn) -> int {
        const TreeMin m = wave_min_u64_lane((unsigned)(ck >> 32), (unsigned)ck);
        rk = ((u64)m.hi << 32) | m.lo;
        if (rk == ~0ull) { rn = INT_MAX; return -1; }
        const u64 tied = __ballot(ck == rk);
        int l = m.lane;
        if (tied & (tied - 1)) {                     // equal scores: the position decides (plan.go:617-628)
            PC(pc_ties);
            const unsigned mn = wave_min_u32_bcast(ck == rk ? (unsigned)cn : kKeyNoneV);
            l = __ffsll((long long)__ballot(ck == rk && (unsigned)cn == mn)) - 1;
        }
        rn = __builtin_amdgcn_readlane(cn, l);
        return l;
    };
    auto rebuild = [&]() {
        PC(pc_rebuilds);
        if (pn >= 0) slotOf[pn] = 0xff;
        // the smallest (g, node) of my column of leaves {64 t + lane}
        u64 cmK = ~0ull;
        int cmN = INT_MAX;
        for (int t = 0; t < G; t++) {
            const int n = t * 64 + lane;
            const u64 v = gB[n];
            if (v < cmK) { cmK = v; cmN = n; }
        }
        tmpK[lane] = cmK; tmpN[lane] = cmN;
        BLANCE_WAVE_SYNC();
        int rk = 0;                                  // my column minimum's rank among the 64
        for (int i = 0; i < 64; i++) rk += key_less(tmpK[i], tmpN[i], cmK, cmN) ? 1 : 0;
        BLANCE_WAVE_SYNC();
        const int nonempty = __popcll(__ballot(cmN != INT_MAX));
        int qsel = 32, count = 0;
        u64 omK = ~0ull;                             // smallest leaf of my column outside the pool
        int omN = INT_MAX;
        while (nonempty > 0) {
            // members: every leaf not after the column minimum of rank qsel - 1 (more than 48: a lower threshold; the
            // other lanes stay free for nodes that join later)
            if (qsel > nonempty) qsel = nonempty;
            const u64 who = __ballot(cmN != INT_MAX && rk == qsel - 1);
            const int wl = __ffsll((long long)who) - 1;
            const u64 TK = readlane_u64(cmK, wl);
            const int TN = __builtin_amdgcn_readlane(cmN, wl);
            count = 0;
            omK = ~0ull; omN = INT_MAX;
            for (int t = 0; t < G; t++) {
                const int n = t * 64 + lane;
                const u64 v = gB[n];
                const bool member = v != ~0ull && !key_less(TK, TN, v, n);
                const u64 mb = __ballot(member);
                if (member) {
                    const int pos = count + __popcll(mb & below);
                    if (pos < 64) { tmpK[pos] = v; tmpN[pos] = n; }
                } else if (v < omK) { omK = v; omN = n; }
                count += __popcll(mb);
            }
            if (count <= 48 || qsel == 1) break;
            qsel >>= 1;
        }
        BLANCE_WAVE_SYNC();
        {
            const TreeMin m = wave_min_u64_lane((unsigned)(omK >> 32), (unsigned)omK);
            thK = ((u64)m.hi << 32) | m.lo;
            const unsigned mn = wave_min_u32_bcast(omK == thK ? (unsigned)omN : kKeyNoneV);
            thN = thK == ~0ull ? INT_MAX : (int)mn;
        }
        if (count > 64) count = 64;                  // (one member: qsel == 1)
        pn = lane < count ? tmpN[lane] : -1;
        pk = lane < count ? tmpK[lane] : ~0ull;
        if (pn >= 0) slotOf[pn] = (unsigned char)lane;
        BLANCE_WAVE_SYNC();
    };

    long long n_bulk = 0;
    int stopped = -1, why = 0;                       // the step this launch could not do (1: not a plain step, 2: the pool is
                                                     // not enough, 3: promotion / demotion)
    for (int oi = q.beg; oi < q.end && stopped < 0; oi += 64) {
        const int B = q.end - oi < 64 ? q.end - oi : 64;
        PC(pc_batches);
        PH(11);
        BLANCE_AGENT_FENCE();                        // earlier bumps of nodeToNodeCounts are visible to the loads below
        for (int r = 0; r < RW; r++) {
            const int idx = r * 64 + lane;
            if (idx < B * RW) recS[idx] = q.rec[(size_t)oi * RW + idx];
        }
        BLANCE_WAVE_SYNC();
        if (!pool_ok) { rebuild(); pool_ok = true; }

        // ---- lane j reads step oi + j's record (as in k_pass_tree)
        const bool act = lane < B;
        const int* rj = recS + (act ? lane : 0) * RW;
        int row = NX;
        const int wj = rj[1];
        int ownv[KM], ntn_own[KM], hv[KH], ov[KO];
        u64 oK[KM];
#pragma unroll
        for (int j = 0; j < KM; j++) { ownv[j] = -1; ntn_own[j] = 0; oK[j] = ~0ull; }
#pragma unroll
        for (int j = 0; j < KH; j++) hv[j] = -1;
#pragma unroll
        for (int j = 0; j < KO; j++) ov[j] = -1;
        bool simple = act;                           // the step is one this kernel does
        int nown = 0;
        const double vstick = __hiloint2double(rj[3], rj[2]);
        {
            const int hT = rj[kRecHead + q.top_state * SW];
            if ((hT >> 16) != kListAbsent && (hT & 0xffff) > 0) row = rj[kRecHead + q.top_state * SW + 1];   // plan.go:134-138
            const int hs = rj[kRecHead + s * SW];
            nown = (hs >> 16) == kListAbsent ? 0 : (hs & 0xffff);
            if (nown > k) { simple = false; nown = 0; }
#pragma unroll
            for (int j = 0; j < KM; j++) {
                if (simple && j < nown) {
                    const int o = rj[kRecHead + s * SW + 1 + j];
                    if (o >= N || !(flL[o < NXp ? o : 0] & 1)) simple = false;
                    else ownv[j] = o;
                }
            }
            if (simple && nown == 2 && ownv[0] == ownv[1]) simple = false;
            if (!simple) {
                nown = 0;
#pragma unroll
                for (int j = 0; j < KM; j++) ownv[j] = -1;
            }
            if (wj <= 0) simple = false;             // (a weight <= 0 would raise the node it leaves)
        }
        if (simple) {
            int n_h = 0, n_o = 0;
            for (int t = 0; t < M; t++) {
                if (t == s) continue;
                const int h = rj[kRecHead + t * SW];
                if ((h >> 16) == kListAbsent) continue;
                const bool higher = (q.higher_mask >> t) & 1;
                for (int jj = 0; jj < (h & 0xffff); jj++) {
                    const int x = rj[kRecHead + t * SW + 1 + jj];
#pragma unroll
                    for (int j = 0; j < KM; j++) if (ownv[j] == x) simple = false;   // excluded or demoted: not for this kernel
                    if (higher) {
                        if (n_h >= KH) simple = false;
#pragma unroll
                        for (int e = 0; e < KH; e++) if (e == n_h) hv[e] = x;
                        n_h++;
                    } else {
                        if (n_o >= KO) simple = false;
#pragma unroll
                        for (int e = 0; e < KO; e++) if (e == n_o) ov[e] = x;
                        n_o++;
                    }
                }
            }
        }
        // Every step bumps the entries of its row for the nodes it ends up with (plan.go:238-245).  A later step of the
        // batch with the SAME row sees them: its copies of those entries (E, ntn_own) are bumped when the earlier step
        // is done.  sharer: a later step of the batch has my row.
        bool sharer = false, dirty = false;          // dirty: an earlier step of the batch has my row
        if (NP > 0) {
#pragma unroll
            for (int j = 0; j < KM; j++)
                if (simple && j < nown) ntn_own[j] = BLANCE_LD_COHERENT(q.ntn + (size_t)row * N + ownv[j]);
            for (int i = 0; i < B - 1; i++) {
                const int ri = __builtin_amdgcn_readlane(row, i);
                const u64 same = __ballot(act && lane > i && row == ri);
                if (lane == i && same) sharer = true;
                if (lane > i && row == ri) dirty = true;
            }
        }
        // own nodes: exact scores, in (score, position) order (what a stay emits)
        u64 lastB = 0;
        int lastN = -1;
        auto own_keys = [&]() {
#pragma unroll
            for (int j = 0; j < KM; j++) {
                if (j < nown) {
                    const int o = ownv[j];
                    oK[j] = sortable_bits(tree_score(cntL[o], ntn_own[j], totL[o], (flL[o] >> 1) & 1, wL[o], NP, vstick,
                                                     q.booster_kind, lpT, ffT));
                }
            }
            if (nown == 2 && key_less(oK[1], ownv[1], oK[0], ownv[0])) {
                const u64 tk = oK[0]; oK[0] = oK[1]; oK[1] = tk;
                const int tn = ownv[0]; ownv[0] = ownv[1]; ownv[1] = tn;
                const int tt = ntn_own[0]; ntn_own[0] = ntn_own[1]; ntn_own[1] = tt;
            }
            lastB = nown == k ? oK[k - 1] : 0;
            lastN = nown == k ? ownv[k - 1] : -1;
        };
        if (simple) own_keys();

        PH(0);
        // ---- the smallest (g, node) anybody could be offered: a step whose own nodes are all before it keeps them
        u64 rootK;
        int rootN;
        {
            pool_min(pk, pn < 0 ? INT_MAX : pn, rootK, rootN);
            if (key_less(thK, thN, rootK, rootN)) { rootK = thK; rootN = thN; }
        }
        // (a step whose row an earlier step bumps is looked at when that one is done: its own nodes' entries may change)
        bool fail = act && !(simple && !dirty && nown == k && key_less(lastB, lastN, rootK, rootN));
        bool stale = dirty;                          // an earlier step of the batch changed my own nodes (or their entries)
        bool hasE = false;                           // row E[lane] holds my row's entries for the pool
        // ---- nodeToNodeCounts entries of the moving steps' rows for the pool: one load per step, all lanes at once
        auto load_E = [&](u64 steps) {
            while (steps) {
                int f4[4], v4[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    f4[t] = steps ? __ffsll((long long)steps) - 1 : -1;
                    if (steps) steps &= steps - 1;
                    v4[t] = 0;
                    if (f4[t] >= 0) {
                        const int rf = __builtin_amdgcn_readlane(row, f4[t]);
                        if (pn >= 0) v4[t] = BLANCE_LD_COHERENT(q.ntn + (size_t)rf * N + pn);
                        PC(pc_eloads);
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; t++) if (f4[t] >= 0) E[f4[t] * kPoolEStride + lane] = v4[t];
            }
            BLANCE_WAVE_SYNC();
        };
        bool hasR = false;                           // my row's entries for the own nodes of the earlier moving steps are in R
        int n_rc = 0;                                // columns of R in use
        if (NP > 0) {
            const u64 fm0 = __ballot(fail && simple);
            load_E(fm0);
            if (fail && simple) { hasE = true; hasR = true; }
            PH(1);
            // the own nodes of the moving steps: whoever is given up may join the pool while later steps still move
            for (u64 m = fm0; m && n_rc < 64;) {
                int x4[4], v4[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    x4[t] = -1; v4[t] = 0;
                    if (m && n_rc < 64) {
                        const int i = __ffsll((long long)m) - 1;
                        const int x = __builtin_amdgcn_readlane(ownv[t & 1], i);
                        if (t & 1) m &= m - 1;
                        if (x >= 0 && rcOf[x] == 0xff) {
                            x4[t] = x;
                            if (hasR) v4[t] = BLANCE_LD_COHERENT(q.ntn + (size_t)row * N + x);   // (a step before i may give it up too)
                        }
                    }
                }
                BLANCE_WAVE_SYNC();
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    if (x4[t] >= 0 && n_rc < 64 && rcOf[x4[t]] == 0xff) {           // (a node two steps own: one column)
                        BLANCE_WAVE_SYNC();
                        if (lane == 0) rcOf[x4[t]] = (unsigned char)n_rc;
                        R[n_rc * 64 + lane] = v4[t];
                        n_rc++;
                        BLANCE_WAVE_SYNC();
                    }
                }
            }
        }

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
        // a step with my row is done with nodes y0, y1: my copies of the entries it bumps
        auto same_row_bumps = [&](int f, int y0, int y1) {
            const int rf = __builtin_amdgcn_readlane(row, f);
            const int q0 = y0 >= 0 && y0 < NXp ? (int)slotOf[y0] : 0xff, q1 = y1 >= 0 && y1 < NXp ? (int)slotOf[y1] : 0xff;
            const int c0 = y0 >= 0 && y0 < NXp ? (int)rcOf[y0] : 0xff, c1 = y1 >= 0 && y1 < NXp ? (int)rcOf[y1] : 0xff;
            if (lane > f && act && row == rf) {
                if (hasE && q0 != 0xff) E[lane * kPoolEStride + q0] += 1;
                if (hasE && q1 != 0xff) E[lane * kPoolEStride + q1] += 1;
                if (hasR && c0 != 0xff) R[c0 * 64 + lane] += 1;
                if (hasR && c1 != 0xff) R[c1 * 64 + lane] += 1;
#pragma unroll
                for (int jj = 0; jj < KM; jj++)
                    if (ownv[jj] >= 0 && (ownv[jj] == y0 || ownv[jj] == y1)) { ntn_own[jj] += 1; stale = true; fail = true; }
            }
        };

        PH(2);
        int cur = 0;
        while (cur < B) {
            const u64 fm = __ballot(fail) & (~0ull << cur);
            const int f = fm ? __ffsll((long long)fm) - 1 : B;
            if (lane >= cur && lane < f) {          // certain stays: plan.go:299-301 leaves everything as it is
                int* o = outS + lane * OWs;
                o[0] = k;
#pragma unroll
                for (int j = 0; j < KM; j++) if (j < k) o[1 + j] = ownv[j];
            }
            n_bulk += f - cur;
            // the stays bump their rows too
            for (u64 m = __ballot(lane >= cur && lane < f && sharer); m; m &= m - 1) {
                const int i = __ffsll((long long)m) - 1;
                same_row_bumps(i, __builtin_amdgcn_readlane(ownv[0], i), __builtin_amdgcn_readlane(ownv[1], i));
            }
            cur = f;
            if (f >= B) break;

            PH(3);
            // ================= general step for lane f's record =================
            PC(pc_steps);
            if (!__builtin_amdgcn_readlane(simple ? 1 : 0, f)) { stopped = oi + f; why = 1; break; }
            if (__builtin_amdgcn_readlane(stale ? 1 : 0, f)) {
                PC(pc_stale);
                if (lane == f) { own_keys(); stale = false; }
            }
            if (NP > 0 && !__builtin_amdgcn_readlane(hasE ? 1 : 0, f)) {
                flush_bumps(f);                      // what the steps done so far bumped is part of what is read
                BLANCE_AGENT_FENCE();
                BLANCE_WAVE_SYNC();
                load_E(1ull << f);
                if (lane == f) hasE = true;
            }
            const int w = __builtin_amdgcn_readlane(wj, f);
            const int own0 = __builtin_amdgcn_readlane(ownv[0], f), own1 = __builtin_amdgcn_readlane(ownv[1], f);
            const int h0 = __builtin_amdgcn_readlane(hv[0], f), h1 = __builtin_amdgcn_readlane(hv[1], f);
            const u64 oK0 = readlane_u64(oK[0], f), oK1 = readlane_u64(oK[1], f);
            PH(4);
            int bN[KM], bL[KM];                      // the step's nodes in order; bL: pool lane, or -1 - index of an own node
            u64 bB[KM];
            int n_out = 0;
            bool retried = false;
            for (;;) {
                // my pool entry as a candidate of this partition: its g, made exact where its entry is not 0
                const bool cand = pn >= 0 && pn != own0 && pn != own1 && pn != h0 && pn != h1;     // plan.go:142-156
                const int e = (cand && NP > 0) ? E[f * kPoolEStride + lane] : 0;
                u64 ck = cand ? pk : ~0ull;
                bool exact = e == 0;
                u64 k1, k2;
                int n1, n2, l1, l2;
                for (;;) {
                    l1 = pool_min(ck, pn, k1, n1);
                    l2 = pool_min(lane == l1 ? ~0ull : ck, pn, k2, n2);
                    // a taken entry with a non-zero count: its exact score (>= g) instead, and again
                    const bool fix = (lane == l1 || lane == l2) && !exact;
                    if (!__ballot(fix)) break;
                    PC(pc_exact);
                    if (fix) {
                        ck = sortable_bits(tree_score(cntL[pn], e, totL[pn], (flL[pn] >> 1) & 1, wL[pn], NP, 0.0, q.booster_kind,
                                                      lpT, ffT));
                        exact = true;
                    }
                }
                PH(5);
                // the k best of (own nodes, the two best pool entries); everything wave uniform
#pragma unroll
                for (int j = 0; j < KM; j++) { bB[j] = ~0ull; bN[j] = INT_MAX; bL[j] = -1; }
                auto insert = [&](u64 b, int n, int l) {
                    if (n == INT_MAX || n < 0) return;
#pragma unroll
                    for (int j = KM - 1; j >= 0; j--) {
                        const bool here = j < k && key_less(b, n, bB[j], bN[j]);
                        const bool above = j > 0 && key_less(b, n, bB[j - 1], bN[j - 1]);
                        if (here) {
                            if (above) { bB[j] = bB[j - 1]; bN[j] = bN[j - 1]; bL[j] = bL[j - 1]; }
                            else { bB[j] = b; bN[j] = n; bL[j] = l; }
                        }
                    }
                };
                insert(oK0, own0, -1);
                insert(oK1, own1, -2);
                if (l1 >= 0) insert(k1, n1, l1);
                if (l2 >= 0) insert(k2, n2, l2);
                n_out = 0;
#pragma unroll
                for (int j = 0; j < KM; j++) if (j < k && bN[j] != INT_MAX) n_out++;
                // nothing outside the pool can get in: the k-th taken is before THETA
                const bool ok = n_out == k && key_less(bB[k - 1], bN[k - 1], thK, thN);
                if (ok) break;
                if (retried) { n_out = -1; break; }  // a fresh pool is not enough (or the constraint cannot be met): the general code
                flush_bumps(f);
                if (NP > 0) BLANCE_AGENT_FENCE();
                BLANCE_WAVE_SYNC();
                rebuild();
                retried = true;
                hasE = false;                        // the pool changed under every row of E
                if (NP > 0) {
                    const u64 fm2 = __ballot(fail && simple && lane >= f);
                    load_E(fm2);
                    if (fail && simple && lane >= f) hasE = true;
                }
            }
            if (n_out < 0) { stopped = oi + f; why = 2; break; }
#ifdef BLANCE_SIMT_EMU
            {
                const int rowf_ = __builtin_amdgcn_readlane(row, f);
                if (getenv("BLANCE_PAR_TRACE") && lane < 8)
                    fprintf(stderr, "[pool] state %d step %d lane %d: pn %d pk %llx E %d | row %d own %d %d (%llx %llx) -> %d %d (%llx %llx) theta %llx %d\n", s, oi + f, lane, pn, pk,
                            E[f * kPoolEStride + lane], rowf_, own0, own1, oK0, oK1, bN[0], bN[1], bB[0], bB[1], thK, thN);
            }
#endif
            // taken from the pool but held by the partition in another state: promoted / demoted -- the general code
            {
                bool prom = false;
#pragma unroll
                for (int e2 = 0; e2 < KO; e2++) {
                    const int x = __builtin_amdgcn_readlane(ov[e2], f);
#pragma unroll
                    for (int j = 0; j < KM; j++) prom = prom || (j < k && bL[j] >= 0 && x >= 0 && bN[j] == x);
                }
                if (prom) { stopped = oi + f; why = 3; break; }
            }

            PH(6);
            // ---- commit (plan.go:290-301): taken entries enter, own nodes that were not taken leave
            const bool kept0 = own0 >= 0 && (bN[0] == own0 || (k > 1 && bN[1] == own0));
            const bool kept1 = own1 >= 0 && (bN[0] == own1 || (k > 1 && bN[1] == own1));
            const int rel0 = own0 >= 0 && !kept0 ? own0 : -1, rel1 = own1 >= 0 && !kept1 ? own1 : -1;
            const int rs0 = rel0 >= 0 ? (int)slotOf[rel0] : 0xff, rs1 = rel1 >= 0 ? (int)slotOf[rel1] : 0xff;
            // which node this lane settles: its pool entry (taken, or a released node that is a pool entry), or -- lanes
            // 62, 63 for released nodes outside the pool (a pool never has more than 62 entries)
            int hx = -1, hd = 0;
            bool in_pool = false;
#pragma unroll
            for (int j = 0; j < KM; j++) if (j < k && bL[j] == lane) { hx = pn; hd = w; in_pool = true; }
            if (rs0 != 0xff && lane == rs0) { hx = rel0; hd = -w; in_pool = true; }
            if (rs1 != 0xff && lane == rs1) { hx = rel1; hd = -w; in_pool = true; }
            if (rel0 >= 0 && rs0 == 0xff && lane == 62) { hx = rel0; hd = -w; }
            if (rel1 >= 0 && rs1 == 0xff && lane == 63) { hx = rel1; hd = -w; }
            u64 nk = ~0ull;
            if (hx >= 0) {
                cntL[hx] += hd;
                totL[hx] += hd;
                nk = g_key(hx);
                gB[hx] = nk;
                if (in_pool) {
                    pk = nk;
                    if (!key_less(nk, hx, thK, thN)) {   // at or behind THETA: nobody can take it before a rebuild
                        PC(pc_drops);
                        slotOf[hx] = 0xff;
                        pn = -1; pk = ~0ull;
                    }
                }
            }
            PH(7);
            // a released node outside the pool that is before THETA joins the pool (a free lane), or lowers THETA
            for (int t = 0; t < 2; t++) {
                const int x = t == 0 ? rel0 : rel1, rs = t == 0 ? rs0 : rs1, hl = 62 + t;
                if (x < 0 || rs != 0xff) continue;
                const u64 xk = readlane_u64(nk, hl);
                if (!key_less(xk, x, thK, thN)) continue;
                const u64 freem = __ballot(pn < 0) & ((1ull << 62) - 1);
                if (freem) {
                    PC(pc_joins);
                    const int fl2 = __ffsll((long long)freem) - 1;
                    if (lane == fl2) { pn = x; pk = xk; slotOf[x] = (unsigned char)fl2; }
                    // its entries of the rows of the steps still to move: fetched with the batch (R), else read now
                    // (with everything done so far written out)
                    if (NP > 0) {
                        const int rc = rcOf[x];
                        const bool need = lane > f && act && hasE;
                        if (__ballot(need && (!hasR || rc == 0xff))) {
                            PC(pc_slowjoins);
                            flush_bumps(f);
                            BLANCE_AGENT_FENCE();
                            BLANCE_WAVE_SYNC();
                            if (need) E[lane * kPoolEStride + fl2] = BLANCE_LD_COHERENT(q.ntn + (size_t)row * N + x);
                        } else if (need) {
                            E[lane * kPoolEStride + fl2] = R[rc * 64 + lane];
                        }
                    }
                } else {
                    thK = xk; thN = x;
                }
            }
            BLANCE_WAVE_SYNC();
            PH(8);
            // later steps of the batch that hold a changed node were validated against its old counters
            {
                const int c0 = bL[0] >= 0 ? bN[0] : -1, c1 = (k > 1 && bL[1] >= 0) ? bN[1] : -1;
#pragma unroll
                for (int j = 0; j < KM; j++)
                    if (lane > f && ownv[j] >= 0 && (ownv[j] == c0 || ownv[j] == c1 || ownv[j] == rel0 || ownv[j] == rel1)) { fail = true; stale = true; }
                // a lowered node may now be before the own nodes of a step validated as a stay
                for (int t = 0; t < 2; t++) {
                    const int x = t == 0 ? rel0 : rel1;
                    if (x < 0) continue;
                    const int rs = t == 0 ? rs0 : rs1;
                    const u64 xk = readlane_u64(nk, rs != 0xff ? rs : 62 + t);
                    if (lane > f && act && !fail && !key_less(lastB, lastN, xk, x)) fail = true;
                }
            }
            if (lane == 0) {
                int* o = outS + f * OWs;             // (its row is bumped with the batch's, plan.go:238-245)
                o[0] = k;
#pragma unroll
                for (int j = 0; j < KM; j++) if (j < k) o[1 + j] = bN[j];
            }
            if (NP > 0 && __builtin_amdgcn_readlane(sharer ? 1 : 0, f)) same_row_bumps(f, bN[0], k > 1 ? bN[1] : -1);
            PH(9);
            cur = f + 1;
        }
        // ---- the batch's outputs (the steps done), and their bumps
        BLANCE_WAVE_SYNC();
        if (NP > 0 && act) {
#pragma unroll
            for (int j = 0; j < KM; j++) if (ownv[j] >= 0) rcOf[ownv[j]] = 0xff;
        }
        flush_bumps(cur);
        for (int idx = lane; idx < cur * OWs; idx += 64) q.out[(size_t)oi * OWs + idx] = outS[idx];
        BLANCE_WAVE_SYNC();
        PH(10);
    }
#ifdef BLANCE_PHASE_PROF
    if (lane == 0 && q.end - q.beg > 100000)
        for (int i_ = 0; i_ < 12; i_++) printf("[pool phase %d] %.0f kcycles\n", i_, (double)ph_acc[i_] / 1e3);
#endif
#ifdef BLANCE_PAR_STATS
    if (lane == 0)
        printf("[pool] k %d steps %d batches %lld general steps %lld (exact rescoring %lld, stale %lld, ties %lld) rebuilds %lld joins %lld (%lld with loads) drops %lld E rows %lld stopped %d (%d)\n",
               k, q.end - q.beg, pc_batches, pc_steps, pc_exact, pc_stale, pc_ties, pc_rebuilds, pc_joins, pc_slowjoins, pc_drops, pc_eloads, stopped, why);
#endif
    if (lane == 0) {
        *q.stop_at = stopped < 0 ? q.end : stopped;
        if (q.spec_count) *q.spec_count += n_bulk;
    }
    BLANCE_WAVE_SYNC();
    for (int i = 0; i < G; i++) {
        const int n = i * 64 + lane;
        if (n < NX) q.cnt[s * NX + n] = cntL[n];
    }
}

static inline size_t pool_lds_bytes(int NX, int RW) {
    const size_t NXp = (size_t)((NX + 63) / 64) * 64;
    return NXp * (8 + 4 + 4 + 4 + 1 + 1 + 1) + 64 * 64 * 4 + 64 * 8 + sizeof(int32_t) * (size_t)(64 * RW) + sizeof(double) * (kLpTab + kFfTab) +
           sizeof(int32_t) * (64 + 64 * kPoolEStride + 64 * 4) + 64;
}

}  // namespace blance
