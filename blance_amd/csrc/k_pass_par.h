// k_pass_par: the exact sequential state pass of a state WITHOUT hierarchy rules, k <= 2, with the steps of a
// batch resolved by their own lanes in parallel and committed as far as they provably equal the sequential result.
// Part of libblance_hip.so (tu_par.hip); see DESIGN.md section 4.4.
#pragma once

namespace blance {

// ============================================================================
// assignStateToPartitions (plan.go:253-303) with findBestNodes (plan.go:98-248).  k_pass_tree resolves the steps
// that move a copy one after the other on one wave (about 1,200 instructions each).  Here lane j of a batch of 64
// steps resolves step j ITSELF, against the state the earlier steps of the batch are expected to leave behind,
// and a prefix of the batch is committed when the expectation is proved right:
//
//  * The POOL: the <= 64 nodes with the smallest partition-independent scores g, sorted by (g, position), in LDS.
//    Every node outside it scores >= THETA (the smallest g outside the pool, kept up to date when a node outside
//    changes).  Bit p of a 64-bit mask stands for pool position p.  A pool node whose counters change stays where
//    it is with its new g and joins the few STALE entries (every step looks at all of those, then at the ordered
//    ones in order); the pool is sorted again when they get many, and rebuilt from all nodes when a step needs
//    more than it holds below THETA.
//  * Step j sees the pool minus what the steps before it took.  Lane j is handed that as a mask (taken_in), inserts
//    the partition's own nodes (exact scores, with stickiness) and then the available pool entries in order, each
//    scored exactly with the partition's nodeToNodeCounts entry (prefetched for the whole pool: E), until its
//    k-th best is not after the next entry -- plan.go's sort restricted to the nodes that can matter.
//  * The result is the sequential one if (1) taken_in is what the earlier lanes REALLY took (checked with a prefix
//    OR over the lanes' results: by induction over the lanes the first lane's input is right, hence its result,
//    hence the second lane's input ...), (2) its k-th best is before THETA and before every new score an earlier
//    lane of this round gives a node (prefix minimum), (3) no earlier lane of the round touches its own nodes.
//    The longest prefix of lanes that pass is committed at once; the others are resolved again in the next round
//    with the corrected input.  Steps that keep their nodes are just lanes that take nothing.
//  * HAND-ME-DOWNS.  In a weighted rebalance the node one moving step gives up is what the next one takes -- the
//    moving steps form a chain.  So every lane publishes what it expects to give up (node, counters' change, g
//    afterwards; first from its own scores against the pool's front, then from its previous result), and a lane scores
//    the expected releases of its NEAREST EARLIER MOVING lane p exactly, like pool entries (their nodeToNodeCounts
//    entries for every lane's row are fetched with the batch: R).  Its result then stands if p does release exactly
//    that, no lane between p and it moves anything, and its k-th best is before everything that changed before p
//    (prefix minimum AT p) and before what p's picks became -- p's releases themselves need no bound any more.
//
// What this kernel does not do itself -- partitions holding a node in two states, more than two higher priority
// nodes, promotions / demotions, weights <= 0, unmet constraints -- ends the launch: *stop_at says where, the host
// runs one batch of k_pass_tree there and launches this kernel again behind it.  The kernel also keeps count of
// what its rounds and pool rebuilds cost against what k_pass_tree would spend on the same steps (one general step
// per step that changes something); where the steps depend on each other so much that it loses, it stops with
// *stop_at = -1 - step and k_pass_tree finishes the pass.
// ============================================================================
constexpr int kParWalk = 10;             // ordered pool entries a lane looks at per round at most
constexpr int kParStale = 6;             // stale entries beyond which the pool is sorted again
// rough instruction counts behind the give-up rule
constexpr int kParCostRound = 650, kParCostRefill = 2400, kParCostResort = 600, kParCostTreeStep = 1200, kParCostTreeBatch = 700;
constexpr int kParWindow = 32;           // batches between two looks at the balance

constexpr int kParLpTab = 128, kParFfTab = 1024;      // LDS tables of this kernel (smaller than k_pass_tree's: room for R)
// nodeSorter.Score (plan.go:634-689) exactly as tree_score, with this kernel's table sizes
__device__ __forceinline__ double par_score(int cnt, int nt, int tot, int hasw, int w, int NP, double cf, int booster,
                                            const double* lpT, const double* ffT) {
    double r = (double)cnt;                           // plan.go:664-670
    if (NP > 0) {
        const double lp = (unsigned)nt < (unsigned)kParLpTab ? lpT[nt] : (double)nt / (double)NP;      // :638-644
        const double ff = (unsigned)tot < (unsigned)kParFfTab ? ffT[tot] : (0.001 * (double)tot) / (double)NP;   // :647-652
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

__device__ __forceinline__ unsigned long long shfl_up_u64(unsigned long long v, int d) {
    const unsigned hi = (unsigned)__shfl_up((int)(unsigned)(v >> 32), d), lo = (unsigned)__shfl_up((int)(unsigned)v, d);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int l) {
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
    return ((unsigned long long)hi << 32) | lo;
}

template <int KM>
__global__ __launch_bounds__(64) void k_pass_par(PassParams q) {
    static_assert(KM == 2, "two copies at most");
    typedef unsigned long long u64;
    constexpr int KH = 2, KO = 4;
    BLANCE_DYN_LDS(lds);
    const int lane = threadIdx.x;
    const int N = q.N, NX = q.NX, M = q.M, L = q.L, NP = q.NP, s = q.s, k = q.k, RW = q.RW;
    const int SW = 1 + L;
    const int G = (NX + 63) >> 6, NXp = G << 6;
    const u64 below = lane ? (~0ull >> (64 - lane)) : 0ull;      // lanes before mine

    u64* gB = (u64*)lds;                             // [NXp] sortable image of g of every node; ~0: no candidate
    u64* poolK = gB + NXp;                           // [64] the pool, ascending in (g, node)
    u64* tmpK = poolK + 64;                          // [64] scratch of the refill
    int* cntL = (int*)(tmpK + 64);                   // [NXp] stateNodeCounts[s]
    int* totL = cntL + NXp;                          // [NXp] nodePartitionCounts (plan.go:118-124)
    int* wL = totL + NXp;                            // [NXp] node weights
    unsigned* touch = (unsigned*)(wL + NXp);         // [NXp] (round << 6 | 63 - lane) of the last lane that changed the node
    int* recS = (int*)(touch + NXp);                 // [64 * RW] step records of the batch
    double* lpT = (double*)(recS + 64 * RW);         // [kParLpTab] c / NP
    double* ffT = lpT + kParLpTab;                   // [kParFfTab] (0.001 * t) / NP
    int* poolN = (int*)(ffT + kParFfTab);            // [64] node | (column of E) << 16
    int* tmpN = poolN + 64;                          // [64]
    int* E = tmpN + 64;                              // [64 positions][64 lanes] nodeToNodeCounts[row of lane][pool node]
    int* outS = E + 64 * 64;                         // [64][OW] the batch's outputs
    u64* redS = (u64*)(outS + 64 * 4);               // [2] scratch: masks OR-ed over the committed lanes
    unsigned char* flL = (unsigned char*)(redS + 2);              // [NXp] 1: in nodesNext, 2: has a weight
    unsigned char* slotOf = flL + NXp;               // [NXp] the node's pool position, 0xff: not in the pool
    unsigned char* rcOf = slotOf + NXp;              // [NXp] the node's column of R (an own node of a moving step of the batch), 0xff: none
    int* R = (int*)(rcOf + NXp);                     // [64 columns][64 lanes] nodeToNodeCounts[row of lane][own node of a moving step]
    u64* vK = (u64*)(R + 64 * 64);                   // [64 lanes][2] what a step is expected to give up: the node's g afterwards,
    int* vN = (int*)(vK + 128);                      // ... the node (-1: nothing),
    int* vD = vN + 128;                              // ... the change of its counters,
    int* vT = vD + 128;                              // ... and the later lane that takes it in this round (-1: nobody)

    for (int i = lane; i < kParLpTab; i += 64) lpT[i] = NP > 0 ? (double)i / (double)NP : 0.0;
    for (int i = lane; i < kParFfTab; i += 64) ffT[i] = NP > 0 ? (0.001 * (double)i) / (double)NP : 0.0;
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
        cntL[n] = c; totL[n] = t; wL[n] = w; flL[n] = (unsigned char)fl; touch[n] = 0; slotOf[n] = 0xff; rcOf[n] = 0xff;
    }
    BLANCE_WAVE_SYNC();
    // g of node n with its counters moved by d (plan.go:634-689 without the partition's own terms)
    auto g_key = [&](int n, int d) -> u64 {
        return (flL[n] & 1) ? sortable_bits(par_score(cntL[n] + d, 0, totL[n] + d, (flL[n] >> 1) & 1, wL[n], NP, 0.0,
                                                      q.booster_kind, lpT, ffT)) : ~0ull;
    };
    for (int i = 0; i < G; i++) gB[i * 64 + lane] = g_key(i * 64 + lane, 0);
    BLANCE_WAVE_SYNC();

    // ---- the pool (wave uniform): Pn entries, A: positions nobody took yet, theta: every node outside scores >= it
    int Pn = 0;
    u64 A = 0, S = 0, thK = ~0ull;                   // A: ordered entries nobody took; S: stale entries (examined by every step)
    int thN = INT_MAX;
    bool pool_ok = false;
    bool chain_mode = false;                         // the moving steps of the last batches took what their predecessors gave up
    PH_DECL;
#if defined(BLANCE_PHASE_PROF) || defined(BLANCE_PAR_STATS)
#define BLANCE_PAR_COUNT 1
#endif
#undef PC
#ifdef BLANCE_PAR_COUNT
    long long pc_resorts = 0, pc_batches = 0, pc_rounds = 0, pc_refills = 0, pc_eloads = 0, pc_in = 0, pc_pm = 0, pc_th = 0, pc_stale = 0,
              pc_unres = 0, pc_moved = 0, pc_owns = 0, pc_joins = 0, pc_handed = 0;
#define PC(x) (x)++
#else
#define PC(x)
#endif
    auto refill = [&]() {
        PC(pc_refills);
        if (lane < Pn) slotOf[poolN[lane] & 0xffff] = 0xff;
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
            // members: every leaf not after the column minimum of rank qsel - 1 (more than 64: a lower threshold)
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
            if (count <= 40 || qsel == 1) break;        // (room for the nodes that join later)
            qsel >>= 1;
        }
        BLANCE_WAVE_SYNC();
        {
            const TreeMin m = wave_min_u64_lane((unsigned)(omK >> 32), (unsigned)omK);
            thK = ((u64)m.hi << 32) | m.lo;
            const unsigned mn = wave_min_u32_bcast(omK == thK ? (unsigned)omN : kKeyNoneV);
            thN = thK == ~0ull ? INT_MAX : (int)mn;
        }
        // sort the members: rank by counting
        const u64 myK = lane < count ? tmpK[lane] : ~0ull;
        const int myN = lane < count ? tmpN[lane] : INT_MAX;
        int r2 = 0;
        for (int i = 0; i < count; i++) r2 += key_less(tmpK[i], tmpN[i], myK, myN) ? 1 : 0;
        BLANCE_WAVE_SYNC();
        if (lane < count) { poolK[r2] = myK; poolN[r2] = myN | (r2 << 16); slotOf[myN] = (unsigned char)r2; }
        BLANCE_WAVE_SYNC();
        Pn = count;
        A = count >= 64 ? ~0ull : ((1ull << count) - 1);
        S = 0;
    };
    // the pool's live entries in order again (their columns of E stay where they are)
    auto resort = [&]() {
        PC(pc_resorts);
        const u64 live = A | S;
        const bool mine_live = lane < Pn && ((live >> lane) & 1);
        const u64 myK = mine_live ? poolK[lane] : ~0ull;
        const int myE = lane < Pn ? poolN[lane] : INT_MAX;      // dead entries go behind the live ones, in position order
        const int myN = mine_live ? (myE & 0xffff) : (0x10000 | lane);
        tmpK[lane] = myK; tmpN[lane] = myN;
        BLANCE_WAVE_SYNC();
        int r2 = 0;
        for (int i = 0; i < Pn; i++) r2 += key_less(tmpK[i], tmpN[i], myK, myN) ? 1 : 0;
        BLANCE_WAVE_SYNC();
        if (lane < Pn) {
            poolK[r2] = myK; poolN[r2] = myE;
            slotOf[myE & 0xffff] = mine_live ? (unsigned char)r2 : 0xff;
        }
        BLANCE_WAVE_SYNC();
        const int nl = __popcll(live);
        A = nl >= 64 ? ~0ull : ((1ull << nl) - 1);
        S = 0;
    };

    long long n_bulk = 0;
    int stopped = -1;                                // the step this launch could not do
    bool gave_up = false;                            // ... because k_pass_tree is the cheaper way through this pass
    int w_batches = 0;                               // since the last look: batches, and the balance of instructions
    long long w_cost = 0, w_tree = 0;
    unsigned gen = 0;                                // round counter
    for (int oi = q.beg; oi < q.end && stopped < 0; oi += 64) {
        const int B = q.end - oi < 64 ? q.end - oi : 64;
        PC(pc_batches);
        if (w_batches >= kParWindow) {
            if (w_cost > w_tree + w_tree / 4 && !(q.spec & 8)) { stopped = oi; gave_up = true; break; }
            w_batches = 0; w_cost = 0; w_tree = 0;
        }
        w_batches++;
        w_tree += kParCostTreeBatch;
        BLANCE_AGENT_FENCE();                        // earlier bumps of nodeToNodeCounts are visible to the loads below
        for (int r = 0; r < RW; r++) {
            const int idx = r * 64 + lane;
            if (idx < B * RW) recS[idx] = q.rec[(size_t)oi * RW + idx];
        }
        BLANCE_WAVE_SYNC();

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
        bool simple = act;                           // the step is one this kernel resolves
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
        // is committed.  sharer: a later step of the batch has my row.
        bool sharer = false;
        if (NP > 0) {
#pragma unroll
            for (int j = 0; j < KM; j++)
                if (simple && j < nown) ntn_own[j] = BLANCE_LD_COHERENT(q.ntn + (size_t)row * N + ownv[j]);
            for (int i = 0; i < B - 1; i++) {
                const int ri = __builtin_amdgcn_readlane(row, i);
                const u64 same = __ballot(act && lane > i && row == ri);
                if (lane == i && same) sharer = true;
            }
        }
        PH(0);

        // ---- rounds: resolve every step not yet done, commit the proven prefix
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
        unsigned own_gen = 0;                        // the round my own nodes' scores were computed in (0: never)
        bool hasR = false;                           // my row's entries for the own nodes of the steps expected to move are in R
        int relN[KM];                                // what I am expected to give up: node (-1: nothing), its g afterwards
        u64 relK[KM];
#pragma unroll
        for (int j = 0; j < KM; j++) { relN[j] = -1; relK[j] = ~0ull; }
        u64 movers_prev = 0;                         // the lanes expected to change something
        int n_rc = 0;                                // columns of R in use
        bool hasE = false, askE = false;             // my row's entries for the pool are in E / were missed last round
        bool tried_refill = false;                   // the head of the round already got a fresh pool
        bool guess = true;                           // no result of a previous round to start from
        u64 taken_in = 0;
        int idle = 0;                                // rounds in a row that committed nothing
        while (cur < B) {
            gen++;
            PC(pc_rounds);
            w_cost += kParCostRound;
            const bool mine = act && lane >= cur;
            if (!pool_ok) {
                flush_bumps(cur);
                if (NP > 0) BLANCE_AGENT_FENCE();
                BLANCE_WAVE_SYNC();
                refill();
                w_cost += kParCostRefill;
                pool_ok = true;
                hasE = false; askE = false; guess = true;
            } else if (__popcll(S) > kParStale) {
                resort();
                w_cost += kParCostResort;
                guess = true;
            }
            // ---- own nodes: exact scores, recomputed after a change
            {
                bool need = false;
#pragma unroll
                for (int j = 0; j < KM; j++)
                    if (mine && simple && j < nown) need = need || own_gen == 0 || (touch[ownv[j]] >> 6) >= own_gen;
                if (__ballot(need)) {
                    PC(pc_owns);
                    if (need) {
#pragma unroll
                        for (int j = 0; j < KM; j++) {
                            if (j < nown) {
                                const int o = ownv[j];
                                oK[j] = sortable_bits(par_score(cntL[o], ntn_own[j], totL[o], (flL[o] >> 1) & 1, wL[o], NP,
                                                                 vstick, q.booster_kind, lpT, ffT));
                            }
                        }
                        own_gen = gen;
                    }
                }
            }
            // the first pool entry nobody took
            const int fpos = A ? __ffsll((long long)A) - 1 : 64;
            const u64 frontK = A ? poolK[fpos] : ~0ull;
            const int frontN = A ? (poolN[fpos] & 0xffff) : INT_MAX;
            // how many entries step j is expected to take: the copies it lacks, and its own nodes not before the front
            int mg = 0;
            if (mine && simple) {
                mg = k - nown;
#pragma unroll
                for (int j = 0; j < KM; j++) if (j < nown && !key_less(oK[j], ownv[j], frontK, frontN)) mg++;
                if (mg > k) mg = k;
            }
            // ---- nodeToNodeCounts entries of my row for the pool: lanes that may take something
            if (NP > 0) {
                const bool want = mine && simple && !hasE && (askE || mg > 0);
                if (__ballot(want)) {
                    PC(pc_eloads);
                    if (bumped_upto < cur) {         // the bumps of the steps done are part of what is read
                        flush_bumps(cur);
                        BLANCE_AGENT_FENCE();
                        BLANCE_WAVE_SYNC();
                    }
                    for (int p = 0; p < Pn; p++) {
                        const int pe = poolN[p];
                        if (want) E[(pe >> 16) * 64 + lane] = BLANCE_LD_COHERENT(q.ntn + (size_t)row * N + (pe & 0xffff));
                    }
                    if (want) { hasE = true; askE = false; }
                }
            }
            // ---- the entries of every moving step's row for the own nodes of the moving steps (once per batch)
            if (NP > 0 && n_rc == 0 && __ballot(mg > 0)) {
                const u64 mv0 = __ballot(mg > 0);
                if (bumped_upto < cur) {
                    flush_bumps(cur);
                    BLANCE_AGENT_FENCE();
                    BLANCE_WAVE_SYNC();
                }
                if (mg > 0) hasR = true;
                for (u64 m = mv0; m && n_rc < 64;) {
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
                                if (hasR) v4[t] = BLANCE_LD_COHERENT(q.ntn + (size_t)row * N + x);
                            }
                        }
                    }
                    BLANCE_WAVE_SYNC();
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        if (x4[t] >= 0 && n_rc < 64 && rcOf[x4[t]] == 0xff) {     // (a node two steps own: one column)
                            BLANCE_WAVE_SYNC();
                            if (lane == 0) rcOf[x4[t]] = (unsigned char)n_rc;
                            R[n_rc * 64 + lane] = v4[t];
                            n_rc++;
                            BLANCE_WAVE_SYNC();
                        }
                    }
                }
                if (n_rc == 0) n_rc = -1;            // (nothing to fetch: do not come back)
            }
            if (guess) {
                // what a lane is expected to give up: its own nodes that are not before the pool's front
#pragma unroll
                for (int j = 0; j < KM; j++) {
                    relN[j] = -1; relK[j] = ~0ull;
                    if (mine && simple && j < nown && !key_less(oK[j], ownv[j], frontK, frontN)) {
                        relN[j] = ownv[j];
                        relK[j] = g_key(ownv[j], -wj);
                    }
                }
                movers_prev = __ballot(mg > 0);
                // in a chain only the first moving step takes from the pool's front; else every one does
                const u64 c1 = __ballot(mg >= 1) & below, c2 = __ballot(mg >= 2) & below;
                int upto = fpos + __popcll(c1) + __popcll(c2);
                if (chain_mode && movers_prev) {
                    const int fm1 = __ffsll((long long)movers_prev) - 1;
                    const int mg1 = __builtin_amdgcn_readlane(mg, fm1);
                    upto = fpos + (lane > fm1 ? mg1 : 0);
                }
                taken_in = upto >= 64 ? ~0ull : ((1ull << upto) - 1);
                guess = false;
            }
            // publish; my nearest earlier moving lane
#pragma unroll
            for (int j = 0; j < KM; j++) { vN[lane * 2 + j] = relN[j]; vK[lane * 2 + j] = relK[j]; vD[lane * 2 + j] = -wj; vT[lane * 2 + j] = 64; }
            BLANCE_WAVE_SYNC();
            const u64 before_me = movers_prev & below & ~((1ull << cur) - 1);
            const int pl = before_me ? 63 - __builtin_clzll(before_me) : -1;
            PH(1);

            // ---- lane j resolves step j
            u64 bB[KM];
            int bN[KM], bP[KM];                      // the k best so far, ascending; bP: pool position, or -1 - index of an own node
#pragma unroll
            for (int j = 0; j < KM; j++) { bB[j] = ~0ull; bN[j] = INT_MAX; bP[j] = -1; }
            auto insert = [&](u64 b, int n, int p) {
#pragma unroll
                for (int j = KM - 1; j >= 0; j--) {
                    const bool here = j < k && key_less(b, n, bB[j], bN[j]);
                    const bool above = j > 0 && key_less(b, n, bB[j - 1], bN[j - 1]);
                    if (here) {
                        if (above) { bB[j] = bB[j - 1]; bN[j] = bN[j - 1]; bP[j] = bP[j - 1]; }
                        else { bB[j] = b; bN[j] = n; bP[j] = p; }
                    }
                }
            };
            const bool run = mine && simple;
            if (run) {
#pragma unroll
                for (int j = 0; j < KM; j++) if (j < nown) insert(oK[j], ownv[j], -1 - j);
            }
            u64 avail = run ? (A & ~taken_in) : 0ull;
            bool done = !run, unres = false;         // unres: the step needs more than this round could give it
            bool blind = false;                      // a release of lane pl I could not score: it stays behind the bound
            int hdD[KM];                             // counters' change already in the score of a hand-me-down I look at
#pragma unroll
            for (int j = 0; j < KM; j++) hdD[j] = 0;
            if (run && pl >= 0) {
#pragma unroll
                for (int t = 0; t < KM; t++) {
                    const int x = vN[pl * 2 + t];
                    if (x >= 0 && x != ownv[0] && x != ownv[1] && x != hv[0] && x != hv[1]) {      // plan.go:142-156
                        const u64 gk = vK[pl * 2 + t];
                        const int d = vD[pl * 2 + t];
                        hdD[t] = d;
                        int e = 0;
                        bool can = slotOf[x] == 0xff;          // (a pool entry: lane pl's mask hides it, the bound keeps it out)
                        if (NP > 0 && can) {
                            const int rc = rcOf[x];
                            if (hasR && rc != 0xff) e = R[rc * 64 + lane];
                            else can = false;
                        }
                        if (!can) blind = true;
                        else {
                            u64 ek = gk;
                            if (e != 0)
                                ek = sortable_bits(par_score(cntL[x] + d, e, totL[x] + d, (flL[x] >> 1) & 1, wL[x], NP, 0.0,
                                                             q.booster_kind, lpT, ffT));
                            insert(ek, x, -10 - t);
                        }
                    }
                }
            }
            // one pool entry, scored exactly for my partition
            auto examine = [&](int p, int pe, u64 gk) {
                const int c = pe & 0xffff;
                if (c != ownv[0] && c != ownv[1] && c != hv[0] && c != hv[1]) {      // plan.go:142-156
                    int e = 0;
                    if (NP > 0) {
                        if (hasE) e = E[(pe >> 16) * 64 + lane];
                        else { askE = true; unres = true; done = true; }
                    }
                    if (!unres) {
                        u64 ek = gk;                 // entry 0: the score IS g
                        if (e != 0)
                            ek = sortable_bits(par_score(cntL[c], e, totL[c], (flL[c] >> 1) & 1, wL[c], NP, 0.0,
                                                          q.booster_kind, lpT, ffT));
                        insert(ek, c, p);
                    }
                }
            };
            for (u64 m = S; m; m &= m - 1) {         // the stale entries, every one of them
                const int p = __ffsll((long long)m) - 1;
                const int pe = poolN[p];
                const u64 gk = poolK[p];
                if (!done && !((taken_in >> p) & 1)) examine(p, pe, gk);
            }
            for (int it = 0; it < kParWalk; it++) {
                if (!__ballot(!done)) break;
                if (!done) {
                    if (avail == 0) {
                        done = true;                 // the whole pool seen: THETA bounds the rest
                    } else {
                        const int p = __ffsll((long long)avail) - 1;
                        avail &= avail - 1;
                        const int pe = poolN[p];
                        const u64 gk = poolK[p];
                        const bool full = bN[k - 1] != INT_MAX;
                        if (full && !key_less(gk, pe & 0xffff, bB[k - 1], bN[k - 1])) done = true;   // nothing from here on can get in
                        else examine(p, pe, gk);
                    }
                }
            }
            if (!done) unres = true;
            PH(2);
            // ---- what the step does: entries taken, own nodes given up; the new g of every node it changes
            u64 pm = 0;                              // pool positions taken
            u64 cmask = 0, smask = 0;                // pool positions whose node I change / that stay in the pool with a new g
            int n_out = 0;
            int chN[2 * KM];
            u64 chK[2 * KM];
            int chD[2 * KM], chB[2 * KM];            // chB: what an earlier lane of the round already did to the node's counters
            bool took_h = false;                     // I take something lane pl gives up
#pragma unroll
            for (int j = 0; j < 2 * KM; j++) { chN[j] = -1; chK[j] = ~0ull; chD[j] = 0; chB[j] = 0; }
            bool prom = false;
            if (run && !unres) {
#pragma unroll
                for (int j = 0; j < KM; j++) {
                    if (j < k && bN[j] != INT_MAX) {
                        n_out++;
                        if (bP[j] >= 0 || bP[j] <= -10) {
                            if (bP[j] >= 0) pm |= 1ull << bP[j];
                            chN[j] = bN[j]; chD[j] = wj;
#pragma unroll
                            for (int t = 0; t < KM; t++) if (bP[j] == -10 - t) { chB[j] = hdD[t]; took_h = true; }
#pragma unroll
                            for (int e = 0; e < KO; e++) prom = prom || ov[e] == bN[j];     // held in another state: promoted / demoted
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < KM; j++) {
                    if (j < nown) {
                        bool kept = false;
#pragma unroll
                        for (int jj = 0; jj < KM; jj++) kept = kept || (jj < k && bN[jj] == ownv[j]);
                        if (!kept) { chN[KM + j] = ownv[j]; chD[KM + j] = -wj; }
                    }
                }
            }
            // a release of lane pl that I take: lane pl need not account for it
            if (run && !unres && took_h) {
#pragma unroll
                for (int j = 0; j < KM; j++)
#pragma unroll
                    for (int t = 0; t < KM; t++) if (j < k && bP[j] == -10 - t) atomicMin(vT + pl * 2 + t, lane);
            }
            BLANCE_WAVE_SYNC();
            int takenBy[KM];                         // the lane that takes what I give up (64: nobody)
#pragma unroll
            for (int j = 0; j < KM; j++) takenBy[j] = vT[lane * 2 + j];
            u64 cminK = ~0ull;                       // smallest new (g, node) of the nodes I change (what a later lane takes over: its business)
            int cminN = INT_MAX;
            u64 raisedK = ~0ull;                     // ... of the nodes I take
            int raisedN = INT_MAX;
            int chP[2 * KM];                         // the changed node's pool position, or -1
#pragma unroll
            for (int j = 0; j < 2 * KM; j++) chP[j] = -1;
            if (__ballot(chN[0] >= 0 || chN[1] >= 0 || chN[KM] >= 0 || chN[KM + 1] >= 0)) {
#pragma unroll
                for (int j = 0; j < 2 * KM; j++) {
                    if (chN[j] >= 0) {
                        chK[j] = g_key(chN[j], chB[j] + chD[j]);
                        const bool handed = j >= KM && takenBy[j - KM] < 64;
                        if (!handed && key_less(chK[j], chN[j], cminK, cminN)) { cminK = chK[j]; cminN = chN[j]; }
                        if (j < KM && key_less(chK[j], chN[j], raisedK, raisedN)) { raisedK = chK[j]; raisedN = chN[j]; }
                        int p = j < KM ? (bP[j] >= 0 ? bP[j] : 0xff) : (int)slotOf[chN[j]];
                        if (p == 0xff || !(((A | S) >> p) & 1)) p = -1;
                        chP[j] = p;
                        const bool stays = p >= 0 && key_less(chK[j], chN[j], thK, thN);   // (at or behind THETA nobody can take it before a rebuild)
                        if (p >= 0) cmask |= 1ull << p;
                        if (stays) smask |= 1ull << p;
                    }
                }
            }
            // does what I give up now equal what I published?
            bool stable = true;
#pragma unroll
            for (int j = 0; j < KM; j++) {
                const bool now = chN[KM + j] >= 0;
                if (now != (relN[j] >= 0) || (now && chK[KM + j] != relK[j])) stable = false;
                relN[j] = now ? chN[KM + j] : -1;
                relK[j] = now ? chK[KM + j] : ~0ull;
            }
            const u64 movers_now = __ballot(run && !unres && (chN[0] >= 0 || chN[1] >= 0 || chN[KM] >= 0 || chN[KM + 1] >= 0));
            // the columns of E of the nodes I end up with (pool members), for the later steps with my row
            int oc[KM];
            bool hot = false;                        // I keep a node that is a pool entry: a later step with my row scores it
#pragma unroll
            for (int j = 0; j < KM; j++) {
                oc[j] = -1;
                if (run && sharer && !unres && j < k && bN[j] != INT_MAX) {
                    int p = bP[j] >= 0 ? bP[j] : (int)slotOf[bN[j]];
                    if (p != 0xff && (((A | S) >> p) & 1)) {
                        oc[j] = poolN[p] >> 16;
                        if (bP[j] < 0) hot = true;
                    }
                }
            }
            // ---- which lanes' results are the sequential ones
            // (1) the input: what the lanes before me took
            u64 tex = run ? cmask : 0ull;            // inclusive prefix OR, then shifted by one lane
            u64 pmin = run ? cminK : ~0ull;          // inclusive prefix minimum of the new (g, node)
            int pminN = run ? cminN : INT_MAX;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const u64 t = shfl_up_u64(tex, d), m2 = shfl_up_u64(pmin, d);
                const int n2 = __shfl_up(pminN, d);
                if (lane >= d) { tex |= t; if (key_less(m2, n2, pmin, pminN)) { pmin = m2; pminN = n2; } }
            }
            const u64 tin = tex;                      // inclusive
            tex = shfl_up_u64(tex, 1); pmin = shfl_up_u64(pmin, 1); pminN = __shfl_up(pminN, 1);
            if (lane == 0) { tex = 0; pmin = ~0ull; pminN = INT_MAX; }
            const bool ok_in = ((A | S) & ~tex) == ((A | S) & ~taken_in);
            // (2) my k-th best against what nobody looked at: nodes outside the pool, nodes earlier lanes changed
            const u64 kthK = bB[k - 1];
            const int kthN = bN[k - 1];
            const bool ok_th = n_out == k && key_less(kthK, kthN, thK, thN);
            bool ok_pm = key_less(kthK, kthN, pmin, pminN);
            {
                // with lane pl's releases scored exactly: everything that changed BEFORE lane pl, and what its picks became
                const int src = pl >= 0 ? pl : lane;
                const u64 aK = ((u64)(unsigned)__shfl((int)(unsigned)(pmin >> 32), src) << 32) | (unsigned)__shfl((int)(unsigned)pmin, src);
                const int aN = __shfl(pminN, src);
                const u64 bK = ((u64)(unsigned)__shfl((int)(unsigned)(raisedK >> 32), src) << 32) | (unsigned)__shfl((int)(unsigned)raisedK, src);
                const int bNn = __shfl(raisedN, src);
                const int st = __shfl(stable ? 1 : 0, src);
                if (pl >= 0 && !blind) {
                    const u64 between = movers_now & below & ~((2ull << pl) - 1);
                    ok_pm = st != 0 && between == 0 && key_less(kthK, kthN, aK, aN) && key_less(kthK, kthN, bK, bNn);
                } else if (pl >= 0) {
                    // (a release I could not score may be taken over by a LATER lane and is then in nobody's minimum)
                    ok_pm = ok_pm && st != 0;
#pragma unroll
                    for (int t = 0; t < KM; t++)
                        if (vN[pl * 2 + t] >= 0 && !key_less(kthK, kthN, vK[pl * 2 + t], vN[pl * 2 + t])) ok_pm = false;
                }
            }
            // (3) my own nodes, changed by an earlier lane of this round
#pragma unroll
            for (int j = 0; j < 2 * KM; j++) if (chN[j] >= 0) atomicMax(touch + chN[j], (gen << 6) | (unsigned)(63 - lane));
            if (run && sharer && !unres) {           // the nodes I keep: their entries of my row are bumped all the same
#pragma unroll
                for (int j = 0; j < KM; j++)
                    if (j < k && bN[j] != INT_MAX && bP[j] < 0) atomicMax(touch + bN[j], (gen << 6) | (unsigned)(63 - lane));
            }
            BLANCE_WAVE_SYNC();
            bool stale = false;
#pragma unroll
            for (int j = 0; j < KM; j++) {
                if (run && j < nown) {
                    const unsigned t = touch[ownv[j]];
                    if ((t >> 6) == gen && (int)(63 - (t & 63)) < lane) stale = true;
                }
            }
            const u64 hotm = __ballot(hot) & ~((1ull << cur) - 1);      // (cur < 64)
            const bool good = run && !unres && !prom && ok_in && ok_th && ok_pm && !stale && !(hotm & below);
            const u64 badm = __ballot(mine && !good);
            const int first = badm ? __ffsll((long long)badm) - 1 : B;     // lanes [cur, first) are proven
            PH(3);
            // ---- commit them
            if (lane >= cur && lane < first) {
#pragma unroll
                for (int j = 0; j < 2 * KM; j++) {
                    if (chN[j] >= 0) {                // plan.go:290-301
                        atomicAdd(cntL + chN[j], chD[j]);     // (what I give up may be taken by a later lane of the same commit)
                        atomicAdd(totL + chN[j], chD[j]);
                        const bool handed = j >= KM && takenBy[j - KM] < first;     // that lane writes the node's g
                        if (!handed) {
                            gB[chN[j]] = chK[j];
                            if (chP[j] >= 0) poolK[chP[j]] = chK[j];
                        }
                    }
                }
                int* o = outS + lane * OWs;
                o[0] = k;
#pragma unroll
                for (int j = 0; j < KM; j++) if (j < k) o[1 + j] = bN[j];
#ifdef BLANCE_SIMT_EMU
                if (getenv("BLANCE_PAR_TRACE"))
                    fprintf(stderr, "[par] state %d step %d part %d row %d own %d %d (ntn %d %d keys %llx %llx) -> %d %d (%llx %llx) pm %llx round %u dirty %d A %llx in %llx theta %llx %d Pn %d watch %llx new g %llx %llx | %llx %llx S %llx\n", s, oi + lane,
                            rj[0], row, ownv[0], ownv[1], ntn_own[0], ntn_own[1], oK[0], oK[1], bN[0], bN[1], bB[0], bB[1], pm, gen, (int)sharer,
                            A, taken_in, thK, thN, Pn, getenv("BLANCE_PAR_WATCH") ? gB[atoi(getenv("BLANCE_PAR_WATCH"))] : 0ull, chK[0], chK[1], chK[2], chK[3], S);
#endif
            }
            if (first > cur) {
                // positions changed by the committed lanes leave the ordered part; those still below THETA are stale entries
                const bool com = lane >= cur && lane < first;
                if (lane < 2) redS[lane] = 0;
                BLANCE_WAVE_SYNC();
                if (com && smask) atomicOr(redS, smask);
                const u64 changed = readlane_u64(tin, first - 1);
                BLANCE_WAVE_SYNC();
                const u64 stay = redS[0];
                A &= ~changed;
                S = (S & ~changed) | stay;
                // nodes outside the pool that ended up before THETA: into the pool (stale entries), or THETA comes down
#pragma unroll
                for (int j = 0; j < 2 * KM; j++) {
                    const bool handed = j >= KM && takenBy[j - KM] < first;
                    u64 jm = __ballot(com && chN[j] >= 0 && chP[j] < 0 && !handed && key_less(chK[j], chN[j], thK, thN));
                    while (jm) {
                        const int i = __ffsll((long long)jm) - 1;
                        jm &= jm - 1;
                        const int x = __builtin_amdgcn_readlane(chN[j], i);
                        const u64 xk = readlane_u64(chK[j], i);
                        if (!key_less(xk, x, thK, thN)) continue;
                        if (Pn < 64) {
                            PC(pc_joins);
                            const int slot = Pn;
                            if (lane == 0) { poolK[slot] = xk; poolN[slot] = x | (slot << 16); slotOf[x] = (unsigned char)slot; }
                            if (NP > 0) {
                                const int rc = rcOf[x];
                                const bool need = lane >= first && act && hasE;
                                if (__ballot(need && !(hasR && rc != 0xff))) {
                                    flush_bumps(first);
                                    BLANCE_AGENT_FENCE();
                                    BLANCE_WAVE_SYNC();
                                    if (need) E[slot * 64 + lane] = BLANCE_LD_COHERENT(q.ntn + (size_t)row * N + x);
                                } else if (need) {
                                    E[slot * 64 + lane] = R[rc * 64 + lane];
                                }
                            }
                            S |= 1ull << slot;
                            Pn++;
                            BLANCE_WAVE_SYNC();
                        } else {
                            thK = xk; thN = x;
                        }
                    }
                }
                {
                    const int nt = __popcll(__ballot(com && took_h)), nm = __popcll(__ballot(com && (chN[0] >= 0 || chN[1] >= 0)));
                    if (nm >= 2) chain_mode = 2 * nt >= nm;
                }
                // the later steps with a committed step's row: their copies of the entries it bumps
                for (u64 m = __ballot(com && sharer); m; m &= m - 1) {
                    const int i = __ffsll((long long)m) - 1;
                    const int ri = __builtin_amdgcn_readlane(row, i);
#pragma unroll
                    for (int j = 0; j < KM; j++) {
                        const int y = __builtin_amdgcn_readlane(bN[j], i), col = __builtin_amdgcn_readlane(oc[j], i);
                        const int rcy = (j < k && y != INT_MAX && y >= 0) ? (int)rcOf[y] : 0xff;
                        if (lane >= first && row == ri && j < k && y != INT_MAX) {
                            if (col >= 0 && hasE) E[col * 64 + lane] += 1;
                            if (hasR && rcy != 0xff) R[rcy * 64 + lane] += 1;
#pragma unroll
                            for (int jj = 0; jj < KM; jj++) if (ownv[jj] == y) { ntn_own[jj] += 1; own_gen = 0; }
                        }
                    }
                }
#ifdef BLANCE_PAR_COUNT
                pc_moved += __popcll(__ballot(lane >= cur && lane < first && (chN[0] >= 0 || chN[1] >= 0)));
                pc_handed += __popcll(__ballot(lane >= cur && lane < first && took_h));
#endif
                const int n_same = __popcll(__ballot(com && pm == 0 && chN[KM] < 0 && chN[KM + 1] < 0));
                n_bulk += n_same;
                w_tree += (long long)(first - cur - n_same) * kParCostTreeStep;
                idle = 0;
                tried_refill = false;
            } else {
                idle++;
            }
            BLANCE_WAVE_SYNC();
            taken_in = tex;                          // (bits of committed lanes are gone from A)
            movers_prev = movers_now;
            cur = first;
            if (cur >= B) break;
            // ---- why did lane `first` not pass?
            {
                const bool f_simple = __builtin_amdgcn_readlane(simple ? 1 : 0, first) != 0;
                const bool f_unres = __builtin_amdgcn_readlane(unres ? 1 : 0, first) != 0;
                const bool f_ask = __builtin_amdgcn_readlane(askE ? 1 : 0, first) != 0;
                const bool f_prom = __builtin_amdgcn_readlane((prom && !unres && ok_in && ok_pm && ok_th && !stale) ? 1 : 0, first) != 0;
                const bool f_th = __builtin_amdgcn_readlane(ok_th ? 1 : 0, first) != 0;
                const bool f_short = __builtin_amdgcn_readlane(n_out, first) < k;
#ifdef BLANCE_PAR_COUNT
                if (!__builtin_amdgcn_readlane(ok_in ? 1 : 0, first)) pc_in++;
                else if (!__builtin_amdgcn_readlane(ok_pm ? 1 : 0, first)) pc_pm++;
                else if (__builtin_amdgcn_readlane(stale ? 1 : 0, first)) pc_stale++;
                else if (f_unres) pc_unres++;
                else if (!f_th) pc_th++;
#endif
#ifdef BLANCE_SIMT_EMU
                if (getenv("BLANCE_PAR_TRACE") && idle > 0 && lane == first)
                    fprintf(stderr, "[par] head %d fails: unres %d ask %d th %d (kth %llx %d theta %llx %d) n_out %d A %llx S %llx in %llx Pn %d own %d %d keys %llx %llx\n", oi + first, (int)unres,
                            (int)askE, (int)ok_th, kthK, kthN, thK, thN, n_out, A, S, taken_in, Pn, ownv[0], ownv[1], oK[0], oK[1]);
#endif
                if (!f_simple || f_prom) { stopped = oi + cur; break; }
                // at the head of a round nothing earlier can be blamed: the pool is what falls short
                if (idle > 0 && !f_ask) {
                    if (f_unres || !f_th || f_short) {
                        if (tried_refill) { stopped = oi + cur; break; }   // a fresh pool did not help: the general code
                        pool_ok = false;
                        tried_refill = true;
                    }
                }
                if (idle >= 4) { stopped = oi + cur; break; }
            }
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
    }
#ifdef BLANCE_PAR_COUNT
    if (lane == 0) {
        printf("[par] k %d steps %d batches %lld rounds %lld resorts %lld refills %lld E loads %lld own recomputes %lld; movers %lld (%lld took hand-me-downs) joins %lld; cuts: input %lld changed-score %lld stale %lld unresolved %lld theta %lld\n",
               k, q.end - q.beg, pc_batches, pc_rounds, pc_resorts, pc_refills, pc_eloads, pc_owns, pc_moved, pc_handed, pc_joins, pc_in, pc_pm, pc_stale, pc_unres, pc_th);
#ifdef BLANCE_PHASE_PROF
        for (int i_ = 0; i_ < 4; i_++) printf("[par phase %d] %.0f kcycles\n", i_, (double)ph_acc[i_] / 1e3);
#endif
    }
#endif
    if (lane == 0) {
        *q.stop_at = stopped < 0 ? q.end : (gave_up ? -1 - stopped : stopped);
        if (q.spec_count) *q.spec_count += n_bulk;
    }
    BLANCE_WAVE_SYNC();
    for (int i = 0; i < G; i++) {
        const int n = i * 64 + lane;
        if (n < NX) q.cnt[s * NX + n] = cntL[n];
    }
}

static inline size_t par_lds_bytes(int NX, int RW) {
    const size_t NXp = (size_t)((NX + 63) / 64) * 64;
    return NXp * (8 + 4 + 4 + 4 + 4 + 1 + 1 + 1) + 2 * 64 * 8 + sizeof(int32_t) * (size_t)(64 * RW) + sizeof(double) * (kParLpTab + kParFfTab) +
           64 * 64 * 4 + 128 * (8 + 4 + 4 + 4) +
           sizeof(int32_t) * (64 + 64 + 64 * 64 + 64 * 4) + 16 + 64;
}

}  // namespace blance
