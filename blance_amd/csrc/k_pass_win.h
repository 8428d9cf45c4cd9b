// k_pass_win: the lean form of k_pass_tree for the passes that dominate a weighted rebalance -- a flat state with
// k <= 2 copies, NumPartitions > 0 -- built around a sorted WINDOW of the smallest leaves.
// Part of libblance_hip.so (tu_tree.hip); see DESIGN.md section 4.3.
#pragma once

namespace blance {

// ============================================================================
// Same facts and the same batch structure as k_pass_tree (validation of 64 steps by 64 lanes, validated runs
// committed at once, outputs and nodeToNodeCounts bumps staged per batch).  What differs is the general step:
//
//  * The kWinT smallest leaves by (g, position) are kept OUT of the tournament tree, sorted, one per lane in
//    lanes 0..kWinT-1 (one DPP row).  Invariant: every leaf still in the tree is after the window's last entry.
//    So the candidates of the next steps are known before those steps run, and their nodeToNodeCounts entries
//    are fetched right after a step's decision for the lane expected to fail next -- a whole step ahead.
//  * A step scores all window entries exactly, lane-parallel (the entry is 0 most of the time: the score IS g),
//    puts the partition's own nodes (exact keys from the validating lane) into lanes 8 and 9, and takes its k
//    nodes by k minima over that one DPP row.  Nothing unexamined can get in as long as the k-th taken is not
//    after the window's last entry.
//  * Taken entries leave the window; changed nodes go where their new key belongs (window or tree, the window's
//    last entry falling back into the tree when it is full); the window is refilled from the tree's root, one
//    group rescan and one root minimum per entry -- the only tree maintenance there is.
//
// Everything this kernel does not do itself -- a step whose validating lane's data is stale or not "simple"
// (more than two higher nodes, nodes held in two states, ...), a dirty row, promotions, a window that is too
// short for the step, unmet constraints -- ends the launch: the steps done so far are written out, *stop_at
// says where, the host runs one batch of k_pass_tree there and launches this kernel again behind it.
// ============================================================================
constexpr int kWinT = 8;

template <int KM>
__global__ __launch_bounds__(64) void k_pass_win(PassParams q) {
    static_assert(KM == 2, "two copies at most");
    typedef unsigned long long u64;
    constexpr int KH = 2, KO = 4;
    BLANCE_DYN_LDS(lds);
    const int lane = threadIdx.x;
    const int N = q.N, NX = q.NX, M = q.M, L = q.L, NP = q.NP, s = q.s, k = q.k, RW = q.RW;
    const int SW = 1 + L;
    const int G = (NX + 63) >> 6, NXp = G << 6;

    u64* gB = (u64*)lds;                             // [NXp] leaves still in the tree: sortable image of g; ~0 otherwise
    int* cntL = (int*)(gB + NXp);                    // [NXp] stateNodeCounts[s]
    int* totL = cntL + NXp;                          // [NXp] nodePartitionCounts (plan.go:118-124)
    int* wL = totL + NXp;                            // [NXp] node weights
    int* recS = wL + NXp;                            // [64 * RW] step records of the batch
    double* lpT = (double*)(recS + 64 * RW);         // [kLpTab] c / NP
    double* ffT = lpT + kLpTab;                      // [kFfTab] (0.001 * t) / NP
    int* mvL = (int*)(ffT + kFfTab);                 // [64][kMvW] what lane j learnt about step j
    int* outS = mvL + 64 * kMvW;                     // [64][OW] the batch's outputs
    unsigned char* flL = (unsigned char*)(outS + 64 * (KM + 1));   // [NXp] 1: in nodesNext, 2: has a weight

    for (int i = lane; i < kLpTab; i += 64) lpT[i] = NP > 0 ? (double)i / (double)NP : 0.0;
    for (int i = lane; i < kFfTab; i += 64) ffT[i] = NP > 0 ? (0.001 * (double)i) / (double)NP : 0.0;
    BLANCE_WAVE_SYNC();
    auto leaf_key = [&](int n) -> u64 {
        return (flL[n] & 1) ? sortable_bits(tree_score(cntL[n], 0, totL[n], (flL[n] >> 1) & 1, wL[n], NP, 0.0, q.booster_kind,
                                                      lpT, ffT)) : ~0ull;
    };
    int tree_count = 0;                              // leaves in the tree
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
        tree_count += __popcll(__ballot(fl & 1));
    }
    BLANCE_WAVE_SYNC();
    for (int i = 0; i < G; i++) gB[i * 64 + lane] = leaf_key(i * 64 + lane);
    BLANCE_WAVE_SYNC();

    // ---- the tree: lane i keeps the smallest (g, node) of leaves [64 i, 64 i + 64)
    unsigned gm_hi = kKeyNoneV, gm_lo = kKeyNoneV;
    int gm_n = INT_MAX;
    auto scan_group = [&](int i) {
        const u64 v = gB[i * 64 + lane];
        const TreeMin m = wave_min_u64_lane((unsigned)(v >> 32), (unsigned)v);
        if (lane == i) { gm_hi = m.hi; gm_lo = m.lo; gm_n = (m.hi & m.lo) == kKeyNoneV ? INT_MAX : i * 64 + m.lane; }
    };
    for (int i = 0; i < G; i++) scan_group(i);
    auto tree_put = [&](int n, u64 b, bool is_new) {  // a leaf (new, or with a smaller key): its group's minimum in place
        if (lane == (n & 63)) gB[n] = b;
        if (lane == (n >> 6) && key_less(b, n, ((u64)gm_hi << 32) | gm_lo, gm_n)) { gm_hi = (unsigned)(b >> 32); gm_lo = (unsigned)b; gm_n = n; }
        if (is_new) tree_count++;
        BLANCE_WAVE_SYNC();
    };
    auto tree_take = [&](int n) {                     // a leaf leaves the tree
        if (lane == (n & 63)) gB[n] = ~0ull;
        BLANCE_WAVE_SYNC();
        if (__builtin_amdgcn_readlane(gm_n, n >> 6) == n) scan_group(n >> 6);
        tree_count--;
    };

    // ---- the window: lanes 0..Wc-1, ascending; Pv / Pf: the entry's nodeToNodeCounts value fetched for lane Pf's step
    unsigned Wh = kKeyNoneV, Wl = kKeyNoneV;
    int Wn = INT_MAX, Pv = 0, Pf = -1, Wc = 0;
    auto win_remove = [&](int pos) {                  // lanes behind pos move down by one
        const unsigned a = (unsigned)dpp_mov<0x101>((int)Wh), b = (unsigned)dpp_mov<0x101>((int)Wl);
        const int c = dpp_mov<0x101>(Wn), d = dpp_mov<0x101>(Pv), e = dpp_mov<0x101>(Pf);
        if (lane >= pos && lane < Wc - 1) { Wh = a; Wl = b; Wn = c; Pv = d; Pf = e; }
        if (lane == Wc - 1) { Wh = kKeyNoneV; Wl = kKeyNoneV; Wn = INT_MAX; Pf = -1; }
        Wc--;
    };
    auto win_last = [&](u64& b, int& n) {
        b = ((u64)(unsigned)__builtin_amdgcn_readlane((int)Wh, Wc - 1) << 32) | (unsigned)__builtin_amdgcn_readlane((int)Wl, Wc - 1);
        n = __builtin_amdgcn_readlane(Wn, Wc - 1);
    };
    auto win_insert = [&](int n, u64 b) {             // sorted insertion; a full window drops its last entry into the tree
        if (Wc == kWinT) {
            u64 lb; int ln;
            win_last(lb, ln);
            if (lane == kWinT - 1) { Wh = kKeyNoneV; Wl = kKeyNoneV; Wn = INT_MAX; Pf = -1; }
            Wc--;
            tree_put(ln, lb, true);
        }
        const int pos = __popcll(__ballot(lane < Wc && key_less(((u64)Wh << 32) | Wl, Wn, b, n)));
        const unsigned a = (unsigned)dpp_mov<0x111>((int)Wh), c = (unsigned)dpp_mov<0x111>((int)Wl);
        const int d = dpp_mov<0x111>(Wn), e = dpp_mov<0x111>(Pv), g = dpp_mov<0x111>(Pf);
        if (lane > pos && lane <= Wc && lane < kWinT) { Wh = a; Wl = c; Wn = d; Pv = e; Pf = g; }
        if (lane == pos) { Wh = (unsigned)(b >> 32); Wl = (unsigned)b; Wn = n; Pf = -1; }
        Wc++;
    };
    auto refill = [&]() {                            // from the tree's root while there is room
        while (Wc < kWinT && tree_count > 0) {
            const TreeMin m = wave_min_u64_lane(gm_hi, gm_lo);
            const int n = __builtin_amdgcn_readlane(gm_n, m.lane);
            if (lane == Wc) { Wh = m.hi; Wl = m.lo; Wn = n; Pf = -1; }
            Wc++;
            tree_take(n);
        }
    };
    // a node with a new key goes where it belongs (in_tree: its leaf is still there with the old key)
    auto place = [&](int n, u64 b, bool in_tree) {
        bool to_win = false;
        if (Wc > 0) {
            u64 lb; int ln;
            win_last(lb, ln);
            to_win = key_less(b, n, lb, ln);
        }
        if (to_win) {
            if (in_tree) tree_take(n);
            win_insert(n, b);
        } else {
            tree_put(n, b, !in_tree);
        }
    };
    refill();

    long long n_bulk = 0;
    int stopped = -1;                                // the step this launch could not do
    for (int oi = q.beg; oi < q.end && stopped < 0; oi += 64) {
        const int B = q.end - oi < 64 ? q.end - oi : 64;
        BLANCE_AGENT_FENCE();                        // earlier bumps of nodeToNodeCounts are visible to the loads below
        for (int r = 0; r < RW; r++) {
            const int idx = r * 64 + lane;
            if (idx < B * RW) recS[idx] = q.rec[(size_t)oi * RW + idx];
        }
        BLANCE_WAVE_SYNC();

        // ---- lane j looks at step oi + j (as in k_pass_tree)
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
        bool pok = act;
        int nown = 0;
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
        bool simple = pok;
        bool has_other = false;                      // holds nodes in a lower priority state: a taken node may be one of them
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
                    for (int j = 0; j < KM; j++) if (ownv[j] == x) simple = false;
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
                        has_other = true;
                    }
                }
            }
        }
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
        u64 lastB = 0;
        int lastN = -1;
        bool sfail = !simple || nown != k;           // fewer nodes than constraints: never a stay
        bool stale = false;                          // an earlier general step of the batch touched my own nodes
        int sortv[KM];
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
        {
            int* mv = mvL + lane * kMvW;
            mv[0] = wj; mv[1] = row;
            mv[2] = ownv[0]; mv[3] = ownv[1];
            mv[4] = (int)oKh[0]; mv[5] = (int)oKl[0]; mv[6] = (int)oKh[1]; mv[7] = (int)oKl[1];
            mv[8] = hv[0]; mv[9] = hv[1];
#pragma unroll
            for (int e = 0; e < KO; e++) mv[10 + e] = ov[e];
        }
        BLANCE_WAVE_SYNC();

        // ---- the batch in order
        int bumped_upto = 0;
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
        Pf = -1;                                     // tags name lanes of THIS batch
        int cur = 0, done = 0;                       // steps [0, done) of the batch are done
        while (cur < B) {
            // the smallest leaf of all: the window's first entry
            const u64 rootB = Wc > 0 ? (((u64)(unsigned)__builtin_amdgcn_readlane((int)Wh, 0) << 32) |
                                        (unsigned)__builtin_amdgcn_readlane((int)Wl, 0)) : ~0ull;
            const int root_n = Wc > 0 ? __builtin_amdgcn_readlane(Wn, 0) : INT_MAX;
            const bool fail = sfail || dirty || !key_less(lastB, lastN, rootB, root_n);
            const u64 fm = __ballot(act && fail) & (~0ull << cur);
            const int f = fm ? __ffsll((long long)fm) - 1 : B;
            if (lane >= cur && lane < f) {          // certain stays
                int* o = outS + lane * OWs;
                o[0] = k;
#pragma unroll
                for (int j = 0; j < KM; j++) if (j < k) o[1 + j] = sortv[j];
            }
            n_bulk += f - cur;
            done = f;
            if (f >= B) break;

            // ================= general step for lane f =================
            const bool can = __builtin_amdgcn_readlane((simple && !stale && !dirty) ? 1 : 0, f) != 0;
            const int* mv = mvL + f * kMvW;
            const int w = mv[0], rowf = mv[1];
            if (!can || w <= 0) { stopped = oi + f; break; }      // (a weight <= 0 would raise the node it leaves)
            const int own0 = mv[2], own1 = mv[3], h0 = mv[8], h1 = mv[9];
            // the contenders of DPP row 0: window entries with their exact scores, own nodes in lanes 8, 9
            unsigned eh = kKeyNoneV, el = kKeyNoneV;
            int en = INT_MAX;
            if (lane < Wc) {
                const int c = Wn;
                if (c != own0 && c != own1 && c != h0 && c != h1) {             // plan.go:142-156
                    int nt = 0;
                    if (NP > 0) nt = Pf == f ? Pv : BLANCE_LD_COHERENT(q.ntn + (size_t)rowf * N + c);
                    en = c;
                    if (nt) {
                        const u64 b = sortable_bits(tree_score(cntL[c], nt, totL[c], (flL[c] >> 1) & 1, wL[c], NP, 0.0,
                                                               q.booster_kind, lpT, ffT));
                        eh = (unsigned)(b >> 32); el = (unsigned)b;
                    } else {
                        eh = Wh; el = Wl;            // entry 0: the score IS g
                    }
                }
            } else if (lane >= 8 && lane < 10) {
                const int o = mv[2 + (lane - 8)];
                if (o >= 0) { en = o; eh = (unsigned)mv[4 + 2 * (lane - 8)]; el = (unsigned)mv[5 + 2 * (lane - 8)]; }
            }
            // k minima over the row; (score, position) order
            int pl[KM], pn[KM];                      // winner lanes / nodes
            u64 pb[KM];
#pragma unroll
            for (int j = 0; j < KM; j++) { pl[j] = -1; pn[j] = INT_MAX; pb[j] = ~0ull; }
#pragma unroll
            for (int j = 0; j < KM; j++) {
                if (j < k) {
                    const unsigned mh = row_min_u32<16>(eh);
                    const bool k2 = eh == mh;
                    const unsigned ml = row_min_u32<16>(k2 ? el : kKeyNoneV);
                    const bool k3 = k2 && el == ml;
                    const unsigned mn = row_min_u32<16>(k3 ? (unsigned)en : kKeyNoneV);
                    const int wn = __builtin_amdgcn_readlane((int)mn, 0);
                    if (wn != INT_MAX && wn != (int)kKeyNoneV) {
                        const u64 who = __ballot(lane < 16 && k3 && en == wn);
                        pl[j] = __ffsll((long long)who) - 1;
                        pn[j] = wn;
                        pb[j] = ((u64)(unsigned)__builtin_amdgcn_readlane((int)mh, 0) << 32) | (unsigned)__builtin_amdgcn_readlane((int)ml, 0);
                        if (lane == pl[j]) { eh = kKeyNoneV; el = kKeyNoneV; en = INT_MAX; }
                    }
                }
            }
            int n_out = 0;
#pragma unroll
            for (int j = 0; j < KM; j++) if (j < k && pl[j] >= 0) n_out++;
            bool ok = n_out == k;
            if (ok && tree_count > 0) {              // a leaf of the tree is after the window's last entry: can it get in?
                u64 lb; int ln;
                win_last(lb, ln);
#pragma unroll
                for (int j = 0; j < KM; j++) if (j == k - 1 && key_less(lb, ln, pb[j], pn[j])) ok = false;
            }
            if (ok && __builtin_amdgcn_readlane(has_other ? 1 : 0, f)) {       // promoted / demoted: the general code
#pragma unroll
                for (int j = 0; j < KM; j++)
#pragma unroll
                    for (int e = 0; e < KO; e++) if (j < k && pl[j] < 8 && mv[10 + e] == pn[j]) ok = false;
            }
            if (!ok) { stopped = oi + f; break; }

            // the entries of the window for the lane expected to fail next: in flight from here on
            if (NP > 0) {
                const u64 fm2 = fm & (fm - 1);
                const int f2 = fm2 ? __ffsll((long long)fm2) - 1 : -1;
                const int rowf2 = __builtin_amdgcn_readlane(row, f2 < 0 ? 0 : f2);
                if (f2 >= 0 && lane < Wc && Wn < N) { Pv = BLANCE_LD_COHERENT(q.ntn + (size_t)rowf2 * N + Wn); Pf = f2; }
            }

            // ---- commit (plan.go:290-301): taken window entries enter, own nodes that were not taken leave
            bool enter = false, leave = false;
#pragma unroll
            for (int j = 0; j < KM; j++) if (j < k && lane == pl[j] && lane < 8) enter = true;
            if (lane >= 8 && lane < 10 && mv[2 + (lane - 8)] >= 0) {
                leave = true;
#pragma unroll
                for (int j = 0; j < KM; j++) if (j < k && pl[j] == lane) leave = false;
            }
            const int hx = enter ? Wn : (leave ? mv[2 + (lane - 8)] : -1);
            u64 nb = ~0ull;
            if (enter || leave) {
                const int ds = enter ? w : -w;
                cntL[hx] += ds;
                totL[hx] += ds;
                nb = leaf_key(hx);
            }
            {
                int* o = outS + f * OWs;
                if (lane == 0) o[0] = k;
#pragma unroll
                for (int j = 0; j < KM; j++) if (j < k && lane == 0) o[1 + j] = pn[j];
            }
            // ---- window and tree: taken entries out (the higher lane first), changed nodes to their places, refill
            const u64 entm = __ballot(enter), levm = __ballot(leave);
            int cx[4];
            u64 cb[4];
            bool cnew[4];
            int nc = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) { cx[t] = -1; cb[t] = ~0ull; cnew[t] = false; }
            for (u64 mm = entm | levm; mm; mm &= mm - 1) {
                const int h = __ffsll((long long)mm) - 1;
                const int x = __builtin_amdgcn_readlane(hx, h);
                const u64 b = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(nb >> 32), h) << 32) |
                              (unsigned)__builtin_amdgcn_readlane((int)(unsigned)nb, h);
#pragma unroll
                for (int t = 0; t < 4; t++) if (t == nc) { cx[t] = x; cb[t] = b; cnew[t] = h < 8; }
                nc++;
                // later lanes of the batch that hold x were validated against its old counters
#pragma unroll
                for (int j = 0; j < KM; j++) if (lane > f && ownv[j] == x) { sfail = true; stale = true; }
            }
            for (int p = 7; p >= 0; p--) if ((entm >> p) & 1) win_remove(p);
#pragma unroll
            for (int t = 0; t < 4; t++) {
                if (t < nc) {
                    bool in_tree = !cnew[t];
                    if (in_tree) {                   // a node that left: its leaf may sit in the window instead of the tree
                        const u64 inw = __ballot(lane < Wc && Wn == cx[t]);
                        if (inw) { win_remove(__ffsll((long long)inw) - 1); in_tree = false; }
                    }
                    place(cx[t], cb[t], in_tree);
                }
            }
            refill();
            cur = f + 1;
            done = cur;
        }
        // ---- the batch's outputs (the steps done), and their bumps
        BLANCE_WAVE_SYNC();
        flush_bumps(done);
        for (int idx = lane; idx < done * OWs; idx += 64) q.out[(size_t)oi * OWs + idx] = outS[idx];
        BLANCE_WAVE_SYNC();
    }
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

static inline size_t win_lds_bytes(int NX, int RW) {
    const size_t NXp = (size_t)((NX + 63) / 64) * 64;
    return NXp * (8 + 4 + 4 + 4 + 1) + sizeof(int32_t) * (size_t)(64 * RW) + sizeof(double) * (kLpTab + kFfTab) +
           sizeof(int32_t) * 64 * (kMvW + 3) + 64;
}

}  // namespace blance
