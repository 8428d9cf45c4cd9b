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
constexpr int kTreeMaxNodes = 4096;      // LDS budget: 21 bytes per node + tables
constexpr int kWalkCap = 24;

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
    BLANCE_DYN_LDS(lds);
    const int lane = threadIdx.x;
    const int N = q.N, NX = q.NX, M = q.M, L = q.L, NP = q.NP, s = q.s, k = q.k, RW = q.RW;
    const int SW = 1 + L;                            // words per state inside a record
    const int G = (NX + 63) >> 6, NXp = G << 6;
    const int walk_cap = (q.spec & 2) ? 0 : kWalkCap;   // test knob: every general step scores all nodes

    u64* gB = (u64*)lds;                             // [NXp] sortable image of g, ~0 for nodes that are no candidates
    int* cntL = (int*)(gB + NXp);                    // [NXp] stateNodeCounts[s]
    int* totL = cntL + NXp;                          // [NXp] nodePartitionCounts (plan.go:118-124)
    int* wL = totL + NXp;                            // [NXp] node weights
    int* recS = wL + NXp;                            // [64 * RW] step records of the batch
    int* sn = recS + 64 * RW;                        // [64 groups][64 steps] prefetched nodeToNodeCounts entries
    double* lpT = (double*)(sn + 64 * 64);           // [kLpTab] c / NP
    double* ffT = lpT + kLpTab;                      // [kFfTab] (0.001 * t) / NP
    u64* stkB = (u64*)(ffT + kFfTab);                // [kWalkCap] leaves taken out of the tree during a walk
    int* stkN = (int*)(stkB + kWalkCap);             // [kWalkCap]
    unsigned char* flL = (unsigned char*)(stkN + kWalkCap);   // [NXp] 1: in nodesNext, 2: has a weight

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
        cntL[n] = c; totL[n] = t; wL[n] = w; flL[n] = (unsigned char)fl;
        gB[n] = (fl & 1) ? sortable_bits(tree_score(c, 0, t, fl >> 1, w, NP, 0.0, q.booster_kind, lpT, ffT)) : ~0ull;
    }
    BLANCE_WAVE_SYNC();

    // group minima: lane i keeps the smallest (g, node) of leaves [64 i, 64 i + 64)
    unsigned gm_hi = kKeyNoneV, gm_lo = kKeyNoneV;
    int gm_n = INT_MAX;
    auto scan_group = [&](int i, unsigned& rh, unsigned& rl, int& rn) {
        const u64 v = gB[i * 64 + lane];
        const TreeMin m = wave_min_u64_lane((unsigned)(v >> 32), (unsigned)v);
        rh = m.hi; rl = m.lo;
        rn = (m.hi & m.lo) == kKeyNoneV ? INT_MAX : i * 64 + m.lane;
    };
    for (int i = 0; i < G; i++) {
        unsigned rh, rl; int rn;
        scan_group(i, rh, rl, rn);
        if (lane == i) { gm_hi = rh; gm_lo = rl; gm_n = rn; }
    }
    bool root_valid = false;
    u64 rootB = ~0ull;
    int root_n = INT_MAX;

    // which word of the record's state lists this lane looks at in a general step
    const int slot_t = lane / SW, slot_ix = lane - slot_t * SW;
    const bool slot_ok = lane < M * SW;
    const bool slot_higher = slot_ok && ((q.higher_mask >> slot_t) & 1);

    long long n_bulk = 0;
    int snn = INT_MAX;                               // the node my group's prefetched entries belong to

    for (int oi = q.beg; oi < q.end; oi += 64) {
        const int B = q.end - oi < 64 ? q.end - oi : 64;
        BLANCE_AGENT_FENCE();                        // earlier bumps of nodeToNodeCounts are visible to the loads below
        for (int r = 0; r < RW; r++) {
            const int idx = r * 64 + lane;
            if (idx < B * RW) recS[idx] = q.rec[(size_t)oi * RW + idx];
        }
        BLANCE_WAVE_SYNC();

        // ---- lane j looks at step oi + j: can it be a certain stay?
        const bool act = lane < B;
        const int* rj = recS + (act ? lane : 0) * RW;
        int row = NX;
        int ownv[KM], ntn_own[KM];
#pragma unroll
        for (int j = 0; j < KM; j++) { ownv[j] = 0; ntn_own[j] = 0; }
        bool pok = act;                              // own list complete and inside nodesAll: ntn_own is loaded
        bool sfail = !act;
        const double vstick = __hiloint2double(rj[3], rj[2]);
        {
            const int hT = rj[kRecHead + q.top_state * SW];
            if ((hT >> 16) != kListAbsent && (hT & 0xffff) > 0) row = rj[kRecHead + q.top_state * SW + 1];   // plan.go:134-138
            const int hs = rj[kRecHead + s * SW];
            if ((hs >> 16) == kListAbsent || (hs & 0xffff) != k) pok = false;
            if (pok) {
#pragma unroll
                for (int j = 0; j < KM; j++) {
                    if (j < k) {
                        const int o = rj[kRecHead + s * SW + 1 + j];
                        ownv[j] = o;
                        if (o >= N) pok = false;
                    }
                }
            }
            if (!pok) sfail = true;
            if (!sfail) {
#pragma unroll
                for (int j = 0; j < KM; j++) {
                    if (j < k) {
                        if (!(flL[ownv[j]] & 1)) sfail = true;
#pragma unroll
                        for (int jj = 0; jj < KM; jj++) if (jj < j && ownv[jj] == ownv[j]) sfail = true;
                    }
                }
                // held in another state as well: excluded (higher) or demoted (lower) -- not a plain stay
                for (int t = 0; t < M; t++) {
                    if (t == s) continue;
                    const int h = rj[kRecHead + t * SW];
                    if ((h >> 16) == kListAbsent) continue;
                    for (int jj = 0; jj < (h & 0xffff); jj++) {
                        const int x = rj[kRecHead + t * SW + 1 + jj];
#pragma unroll
                        for (int j = 0; j < KM; j++) if (j < k && ownv[j] == x) sfail = true;
                    }
                }
            }
        }
        bool dirty = false;                          // an earlier step of the batch bumps my row
        if (NP > 0) {
#pragma unroll
            for (int j = 0; j < KM; j++)
                if (pok && j < k) ntn_own[j] = BLANCE_LD_COHERENT(q.ntn + (size_t)row * N + ownv[j]);
            for (int i = 0; i < G; i++) {
                const int gn = __builtin_amdgcn_readlane(gm_n, i);
                int v = 0;
                if (act && gn < N) v = BLANCE_LD_COHERENT(q.ntn + (size_t)row * N + gn);
                sn[i * 64 + lane] = v;
            }
            snn = gm_n;
            for (int i = 0; i < B - 1; i++) {
                const int ri = __builtin_amdgcn_readlane(row, i);
                if (lane > i && row == ri) dirty = true;
            }
        }
        u64 lastB = 0;
        int lastN = -1;
        if (!sfail) {
            u64 prevB = 0;
            int prevN = -1;
#pragma unroll
            for (int j = 0; j < KM; j++) {
                if (j < k) {
                    const int o = ownv[j];
                    const u64 b = sortable_bits(tree_score(cntL[o], ntn_own[j], totL[o], (flL[o] >> 1) & 1, wL[o], NP,
                                                           vstick, q.booster_kind, lpT, ffT));
                    if (j > 0 && !key_less(prevB, prevN, b, o)) sfail = true;     // the list order is the score order
                    prevB = b; prevN = o;
                }
            }
            lastB = prevB; lastN = prevN;
        }
        BLANCE_WAVE_SYNC();

        // ---- the batch in order: validated runs at once, the other steps one by one
        int cur = 0;
        while (cur < B) {
            if (!root_valid) {
                const TreeMin m = wave_min_u64_lane(gm_hi, gm_lo);
                rootB = ((u64)m.hi << 32) | m.lo;
                root_n = __builtin_amdgcn_readlane(gm_n, m.lane);
                root_valid = true;
            }
            const bool fail = sfail || dirty || !key_less(lastB, lastN, rootB, root_n);
            const u64 fm = __ballot(act && fail) & (~0ull << cur);
            const int f = fm ? __ffsll((long long)fm) - 1 : B;
            if (lane >= cur && lane < f) {          // certain stays: plan.go:299-301 leaves everything as it is
                int* o = q.out + (size_t)(oi + lane) * q.OW;
                o[0] = k;
#pragma unroll
                for (int j = 0; j < KM; j++) {
                    if (j < k) {
                        o[1 + j] = ownv[j];
                        if (NP > 0) atomicAdd(q.ntn + (size_t)row * N + ownv[j], 1);   // plan.go:238-245
                    }
                }
            }
            n_bulk += f - cur;
            if (f >= B) break;

            // ================= general step for lane f's record =================
            const int* rf = recS + f * RW;
            const int p = uni(rf[0]), w = uni(rf[1]);
            const double stick = __hiloint2double(uni(rf[3]), uni(rf[2]));
            int top = -1;
            {
                const int hT = uni(rf[kRecHead + q.top_state * SW]);
                if ((hT >> 16) != kListAbsent && (hT & 0xffff) > 0) top = uni(rf[kRecHead + q.top_state * SW + 1]);
            }
            const int rowf = top < 0 ? NX : top;
            const bool pv_ok = NP > 0 && __builtin_amdgcn_readlane(pok ? 1 : 0, f) != 0 &&
                               __builtin_amdgcn_readlane(dirty ? 1 : 0, f) == 0;
            const bool sn_ok = NP > 0 && __builtin_amdgcn_readlane(dirty ? 1 : 0, f) == 0;

            // my word of the record's lists, and what it is
            int wd = -1;
            bool valid = false;
            bool hdr_present = false;
            if (slot_ok) {
                const int hdr = rf[kRecHead + slot_t * SW];
                hdr_present = (hdr >> 16) != kListAbsent;
                wd = rf[kRecHead + lane];
                valid = slot_ix >= 1 && hdr_present && slot_ix - 1 < (hdr & 0xffff);
            }
            const u64 m_own = __ballot(valid && slot_t == s);
            const u64 m_high = __ballot(valid && slot_higher);
            const u64 m_oth = __ballot(valid && slot_t != s);
            const bool any_higher_key = __ballot(slot_ok && slot_ix == 0 && slot_higher && hdr_present) != 0;
            auto in_list = [&](int x, u64 mask) -> bool { return (__ballot(valid && wd == x) & mask) != 0; };

            // the partition's own nodes, scored exactly by the lanes that hold them
            const bool is_own = (m_own >> lane) & 1;
            bool own_first = is_own;                 // first occurrence of the node in the list
            bool own_elig = is_own && wd < N && (flL[wd < NXp && wd >= 0 ? wd : 0] & 1);
            for (u64 mm = m_own; mm; mm &= mm - 1) {
                const int h = __ffsll((long long)mm) - 1;
                const int y = __builtin_amdgcn_readlane(wd, h);
                if (is_own && lane > h && wd == y) own_first = false;
            }
            for (u64 mm = m_high; mm; mm &= mm - 1) {
                const int h = __ffsll((long long)mm) - 1;
                const int y = __builtin_amdgcn_readlane(wd, h);
                if (is_own && wd == y) own_elig = false;              // plan.go:146-154
            }
            own_elig = own_elig && own_first;
            int own_nt = 0;
            if (NP > 0) {
                bool have = false;
#pragma unroll
                for (int j = 0; j < KM; j++) {
                    const int v = __builtin_amdgcn_readlane(ntn_own[j], f);
                    if (pv_ok && is_own && slot_ix - 1 == j) { own_nt = v; have = true; }
                }
                if (own_elig && !have) own_nt = BLANCE_LD_COHERENT(q.ntn + (size_t)rowf * N + wd);
            }
            u64 own_b = ~0ull;
            if (own_elig)
                own_b = sortable_bits(tree_score(cntL[wd], own_nt, totL[wd], (flL[wd] >> 1) & 1, wL[wd], NP, stick,
                                                 q.booster_kind, lpT, ffT));

            // the k best (score, position) so far; wave uniform, ascending
            u64 bB[KM];
            int bN[KM];
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
            for (u64 mm = __ballot(own_elig); mm; mm &= mm - 1) {
                const int h = __ffsll((long long)mm) - 1;
                const unsigned bh = (unsigned)__builtin_amdgcn_readlane((int)(own_b >> 32), h);
                const unsigned bl = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)own_b, h);
                insert(((u64)bh << 32) | bl, __builtin_amdgcn_readlane(wd, h));
            }

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
                const bool skip = in_list(c, m_own | m_high);                   // own: scored above; higher: no candidate
                if (!skip) {
                    int nt = 0;
                    if (NP > 0) {
                        const bool pref = sn_ok && __builtin_amdgcn_readlane(snn, gl) == c;
                        nt = pref ? uni(sn[gl * 64 + f]) : uni(BLANCE_LD_COHERENT(q.ntn + (size_t)rowf * N + c));
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
            // put the walked leaves back (the chosen ones get new values below)
            if (n_taken > 0) {
                BLANCE_WAVE_SYNC();
                if (lane < n_taken) gB[stkN[lane]] = stkB[lane];
                BLANCE_WAVE_SYNC();
            }
            if (dense) {
                // ---- rare: score every node (exactly what the reference's sort sees), k successive minima
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
                        for (u64 mm = m_high | m_own; mm; mm &= mm - 1) {
                            const int h = __ffsll((long long)mm) - 1;
                            const int y = __builtin_amdgcn_readlane(wd, h);
                            if (y == n) { if ((m_high >> h) & 1) el = false; else own = true; }
                        }
                        if (el) {
                            const int nt = NP > 0 ? BLANCE_LD_COHERENT(q.ntn + (size_t)rowf * N + n) : 0;
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
            int n_out = 0;
#pragma unroll
            for (int j = 0; j < KM; j++) if (j < k && bN[j] != INT_MAX) n_out++;

            // ---- commit (plan.go:238-245, :290-301).  A node of this state's old list leaves it, a
            // chosen node enters it, and a node that is either also leaves every OTHER list of the
            // partition that holds it.  Own nodes are handled by the lanes that hold them, newly
            // chosen ones by lanes 60..63 (never list words: records are at most 64 words).
            int hx = -1;                             // the node this lane settles
            bool h_own = false, h_chosen = false;
            if (own_first) { hx = wd; h_own = true; }
#pragma unroll
            for (int j = 0; j < KM; j++) {
                if (j < n_out) {
                    const bool inown = in_list(bN[j], m_own);
                    if (h_own && wd == bN[j]) h_chosen = true;
                    if (!inown && lane == 60 + j) { hx = bN[j]; h_chosen = true; }
                }
            }
            int n_oth = 0;
            for (u64 mm = m_oth; mm; mm &= mm - 1) {
                const int h = __ffsll((long long)mm) - 1;
                const int y = __builtin_amdgcn_readlane(wd, h);
                const int ty = __builtin_amdgcn_readlane(slot_t, h);
                if (hx >= 0 && hx == y) {
                    n_oth++;
                    q.cnt[ty * NX + y] -= w;         // only this lane touches that counter
                }
            }
            const int ds = (h_chosen ? w : 0) - (h_own ? w : 0);
            const bool changed = hx >= 0 && (ds != 0 || n_oth > 0);
            if (changed) {
                const int c2 = cntL[hx] + ds, t2 = totL[hx] + ds - w * n_oth;
                cntL[hx] = c2; totL[hx] = t2;
                gB[hx] = (flL[hx] & 1) ? sortable_bits(tree_score(c2, 0, t2, (flL[hx] >> 1) & 1, wL[hx], NP, 0.0,
                                                                  q.booster_kind, lpT, ffT)) : ~0ull;
            }
            if (NP > 0 && lane < n_out) {
                int cn = INT_MAX;
#pragma unroll
                for (int j = 0; j < KM; j++) if (j == lane) cn = bN[j];
                if (cn < N) atomicAdd(q.ntn + (size_t)rowf * N + cn, 1);         // plan.go:238-245
            }
            BLANCE_WAVE_SYNC();
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
                    unsigned rh, rl; int rn;
                    scan_group(gl, rh, rl, rn);
                    if (lane == gl) { gm_hi = rh; gm_lo = rl; gm_n = rn; }
                }
                // later lanes of the batch that hold x were validated against its old counters
#pragma unroll
                for (int j = 0; j < KM; j++) if (lane > f && j < k && ownv[j] == x) sfail = true;
            }
            if (chm) root_valid = false;

            if (lane == 0) {
                const int is_nil = (n_out == 0 && q.n_alive == 0 && !any_higher_key && !q.hier);
                int* o = q.out + (size_t)(oi + f) * q.OW;
                o[0] = n_out | (is_nil << 16);
#pragma unroll
                for (int j = 0; j < KM; j++) if (j < k) o[1 + j] = j < n_out ? bN[j] : -1;
                if (n_out < k) {                     // plan.go:230-235
                    const int wi = *q.warn_count;
                    q.warn_part[wi] = p;
                    q.warn_state[wi] = s;
                    *q.warn_count = wi + 1;
                }
            }
            cur = f + 1;
        }
    }
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
    return NXp * (8 + 4 + 4 + 4 + 1) + sizeof(int32_t) * (size_t)(64 * RW + 64 * 64) +
           sizeof(double) * (kLpTab + kFfTab) + (size_t)kWalkCap * 12 + 64;
}

}  // namespace blance
