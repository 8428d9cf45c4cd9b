// k_pass_seq: the exact sequential state pass (always applicable).
// Part of blance_hip.hip (one translation unit); see DESIGN.md section 4.
#pragma once

namespace blance {

// ============================================================================
// The sequential state pass: assignStateToPartitions (plan.go:253-303) with
// findBestNodes (plan.go:98-248) inlined.  ONE workgroup walks the partitions
// in pass order; thread t owns nodes t, t+T, ... and keeps their load counts,
// total counts, weights and partition-independent scores in registers, so a
// step costs one barrier per argmin and no table traffic.
// ============================================================================
// nodeSorter.Score (plan.go:632-688) with its two NP quotients precomputed: the
// nodeToNodeCounts one from an LDS table, the fill-factor one cached per node.
// Same operations in the same order as node_score().
__device__ __forceinline__ double seq_score(int cnt, int ntn, double ff, int hasw, int w, int NP, double cf,
                                            int booster, const double* lpT) {
    double lp = 0.0;
    if (NP > 0) lp = (unsigned)ntn < (unsigned)kLpTab ? lpT[ntn] : (double)ntn / (double)NP;
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

// HIER: the state has hierarchy rules (plan.go:174-226); the variant without carries none of that
// code or its uniform state (it more than halves the kernel's scalar-register spills).
// KM: capacity of the step's output list (k <= KM); the k <= 2 variant keeps the bookkeeping short.
template <int T, int NPT, bool HIER, int KM>
__global__ __launch_bounds__(T) void k_pass_seq(PassParams q) {
    BLANCE_DYN_LDS(lds);
    RedSlot* red = (RedSlot*)lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int N = q.N, NX = q.NX, M = q.M, L = q.L, NP = q.NP, s = q.s, k = q.k;
    const int SW = 1 + L;                        // words per state inside a record
    int round = 0;

    int cntv[NPT], totv[NPT], wv[NPT], lpos[NPT];
    unsigned alive_m = 0, hasw_m = 0;
    double g[NPT], ffv[NPT];                     // ffv: the fill-factor term (0.001 * total) / NP, plan.go:647-652
#pragma unroll
    for (int i = 0; i < NPT; i++) {
        int n = tid + i * T;
        cntv[i] = 0; totv[i] = 0; wv[i] = 0; lpos[i] = -1; g[i] = 0.0; ffv[i] = 0.0;
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
            if (NP > 0) ffv[i] = (0.001 * (double)totv[i]) / (double)NP;
        }
    }
    // quotient table of the nodeToNodeCounts term (plan.go:638-644), filled by the same expression
    double* lpT = (double*)(red + 2 * (T / 64));   // [kLpTab]
    if (NP > 0) {
        for (int i = tid; i < kLpTab; i += T) lpT[i] = (double)i / (double)NP;
        __syncthreads();
    }

    // ---- verified-stay speculation (flat passes; DESIGN.md "Verified stays"): mirrors of the
    // per-node counters that any thread may read, a first-use table of top priority nodes
    const bool spec_ok = q.spec && q.rule_begin == q.rule_end && k >= 1 && k <= KM;
    int* cntL = (int*)(lpT + kLpTab);              // [NX]
    int* totL = cntL + NX;                         // [NX]
    int* markL = totL + NX;                        // [NX + 1] lowest lane of the batch that uses the row
    int* shI = markL + NX + 1;                     // [0] first failing lane
    double* shD = (double*)(shI + 2);              // [0] score of the bound node
    if (spec_ok) {
#pragma unroll
        for (int i = 0; i < NPT; i++) {
            int n = tid + i * T;
            if (n < NX) { cntL[n] = cntv[i]; totL[n] = totv[i]; markL[n] = INT_MAX; }
        }
        if (tid == 0) markL[NX] = INT_MAX;
        __syncthreads();
    }
    bool try_spec = spec_ok;
    long long spec_steps = 0;

    // step record of the current partition: lane j of every wave holds word j
    int recw = 0, recw_next = 0;
    bool have_next = false;                        // recw_next holds the record of step oi
    // nodeToNodeCounts row of the NEXT step, fetched one step ahead (NP > 0)
    int ntn_pre[NPT];
    int pre_row = -1;
#pragma unroll
    for (int i = 0; i < NPT; i++) ntn_pre[i] = 0;

    PH_DECL;
    int oi = q.beg;
    while (oi < q.end) {
        PH(0);
        // ---- Speculate that the next steps keep their nodes: a stay changes no counter, so
        // thread a can check step oi + a against the state as it is now.  The partition's own
        // nodes, scored exactly, must come out in list order below the smallest
        // partition-independent score of the cluster (a lower bound of every other
        // candidate: the omitted terms are >= 0 and IEEE add / divide / subtract are
        // monotone).  The verified prefix is committed; the first other step runs below.
        if (spec_ok && try_spec) {
            double ms = pos_inf();
            int mn = INT_MAX;
#pragma unroll
            for (int u = 0; u < NPT; u++) {
                if (((alive_m >> u) & 1) && better(g[u], tid + u * T, ms, mn)) { ms = g[u]; mn = tid + u * T; }
            }
            const int gmin_n = uni(block_argmin<T>(ms, mn, red, round));
#pragma unroll
            for (int u = 0; u < NPT; u++) if (tid + u * T == gmin_n) shD[0] = g[u];
            __syncthreads();
            const double gmin_s = gmin_n == INT_MAX ? pos_inf() : shD[0];
            for (;;) {
                const int B = q.end - oi < T ? q.end - oi : T;
                const bool active = tid < B;
                bool fail = false;
                int own[KM];
#pragma unroll
                for (int j = 0; j < KM; j++) own[j] = 0;
                int vrow = NX;
                if (active) {
                    const int32_t* r = q.rec + (size_t)(oi + tid) * q.RW;
                    const double vstick = __hiloint2double(r[3], r[2]);
                    const int hT = r[kRecHead + q.top_state * SW];
                    if ((hT >> 16) != kListAbsent && (hT & 0xffff) > 0) vrow = r[kRecHead + q.top_state * SW + 1];
                    const int hs = r[kRecHead + s * SW];
                    if ((hs >> 16) == kListAbsent || (hs & 0xffff) != k) fail = true;
                    if (!fail) {
                        double prev_s = 0.0;
                        int prev_n = -1;
#pragma unroll
                        for (int j = 0; j < KM; j++) {
                            if (j >= k || fail) break;
                            const int o = r[kRecHead + s * SW + 1 + j];
                            own[j] = o;
                            if (o >= N || !q.alive[o]) { fail = true; break; }
                            // held in another state as well: excluded (higher) or demoted (lower)
                            for (int t = 0; t < M; t++) {
                                if (t == s) continue;
                                const int h = r[kRecHead + t * SW];
                                if ((h >> 16) == kListAbsent) continue;
                                for (int jj = 0; jj < (h & 0xffff); jj++)
                                    if (r[kRecHead + t * SW + 1 + jj] == o) fail = true;
                            }
                            const int nt = NP > 0 ? q.ntn[(size_t)vrow * N + o] : 0;
                            const double so = node_score(cntL[o], nt, totL[o], q.node_has_weight[o], q.node_weight[o],
                                                         NP, vstick, q.booster_kind);
                            if (j > 0 && !better(prev_s, prev_n, so, o)) fail = true;
                            if (!better(so, o, gmin_s, gmin_n)) fail = true;
                            prev_s = so; prev_n = o;
                        }
                    }
                }
                if (tid == 0) shI[0] = B;
                // an earlier step of the batch with the same top priority node bumps my row
                if (NP > 0 && active) atomicMin(&markL[vrow], tid);
                __syncthreads();
                if (NP > 0 && active && markL[vrow] < tid) fail = true;
                if (active && fail) atomicMin(&shI[0], tid);
                __syncthreads();
                const int nok = shI[0];
                if (NP > 0 && active) markL[vrow] = INT_MAX;
                if (tid < nok) {
                    int* o = q.out + (size_t)(oi + tid) * q.OW;
                    o[0] = k;
#pragma unroll
                    for (int j = 0; j < KM; j++) {
                        if (j < k) {
                            o[1 + j] = own[j];
                            if (NP > 0) q.ntn[(size_t)vrow * N + own[j]] += 1;    // plan.go:238-245
                        }
                    }
                }
                __syncthreads();
                oi += nok;
                spec_steps += nok;
                if (nok < B || oi >= q.end) break;
            }
            if (oi >= q.end) break;
            have_next = false;
            pre_row = -1;
        }

        PH(1);
        if (have_next) recw = recw_next;
        else recw = lane < q.RW ? q.rec[(size_t)oi * q.RW + lane] : 0;
        have_next = oi + 1 < q.end;
        if (have_next && lane < q.RW) recw_next = q.rec[(size_t)(oi + 1) * q.RW + lane];
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
        if (NP > 0 && pre_row == row) {
#pragma unroll
            for (int i = 0; i < NPT; i++) ntnv[i] = ntn_pre[i];
        } else {
#pragma unroll
            for (int i = 0; i < NPT; i++) {
                int n = tid + i * T;
                ntnv[i] = (NP > 0 && n < N) ? q.ntn[(size_t)row * N + n] : 0;
            }
        }
        // the next step's row, in flight during this step; entries this step bumps are patched at commit
        int next_row = -1;
        if (NP > 0 && have_next) {
            int hdr = __builtin_amdgcn_readlane(recw_next, kRecHead + q.top_state * SW);
            next_row = NX;
            if ((hdr >> 16) != kListAbsent && (hdr & 0xffff) > 0)
                next_row = __builtin_amdgcn_readlane(recw_next, kRecHead + q.top_state * SW + 1);
#pragma unroll
            for (int i = 0; i < NPT; i++) {
                int n = tid + i * T;
                ntn_pre[i] = n < N ? q.ntn[(size_t)next_row * N + n] : 0;
            }
        }
        pre_row = next_row;

        PH(2);
        // membership of my nodes in the higher-priority lists (plan.go:146-154)
        // and in this state's current list (plan.go:654-662)
        unsigned inh_m = 0, own_m = 0, oth_m = 0;    // oth_m: held in another state of this partition
        int any_higher_key = 0;
        for (int t = 0; t < M; t++) {
            int hdr = REC(kRecHead + t * SW);
            if ((hdr >> 16) == kListAbsent) continue;
            int len = hdr & 0xffff;
            bool higher = (q.higher_mask >> t) & 1;
            if (higher) any_higher_key = 1;
            for (int j = 0; j < len; j++) {
                int x = REC(kRecHead + t * SW + 1 + j);
#pragma unroll
                for (int i = 0; i < NPT; i++) {
                    const unsigned hit = (x == tid + i * T ? 1u : 0u) << i;
                    inh_m |= higher ? hit : 0u;
                    own_m |= t == s ? hit : 0u;
                    oth_m |= t != s ? hit : 0u;
                }
            }
        }
        const unsigned elig_m = alive_m & ~inh_m;

        PH(3);
        double sc[NPT];
#pragma unroll
        for (int i = 0; i < NPT; i++) {
            bool own = (own_m >> i) & 1;
            if (own || ntnv[i] != 0)
                sc[i] = seq_score(cntv[i], ntnv[i], ffv[i], (hasw_m >> i) & 1, wv[i], NP,
                                  own ? stick : 0.0, q.booster_kind, lpT);
            else
                sc[i] = g[i];
        }

        PH(4);
        int chosen[KM];
#pragma unroll
        for (int j = 0; j < KM; j++) chosen[j] = -1;
        int n_out = 0;
        unsigned emitted_m = 0;                    // my nodes already in the output list

        if (HIER && q.hier) {                      // plan.go:174-226
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
                    for (int c = 0; c < KM; c++) if (c < n_out && chosen[c] == x) dup = true;
                    if (!dup) {
#pragma unroll
                        for (int c = 0; c < KM; c++) if (c == n_out) chosen[c] = x;
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
            for (int c = 0; c < KM; c++) if (c == n_out) chosen[c] = best;
            n_out++;
#pragma unroll
            for (int u = 0; u < NPT; u++) if (best == tid + u * T) emitted_m |= 1u << u;
        }

        PH(5);
        // ---- commit (plan.go:238-245, :290-301); every thread updates the nodes it owns:
        // a node of this state's old list leaves it (:290-293), a chosen node enters it (:299-301),
        // and a node that is either of those also leaves every OTHER state list of the partition
        // that holds it (:290-297 -- the reference walks all states' lists)
        unsigned changed_m = 0;
        {
            const unsigned moved_m = (own_m | emitted_m) & oth_m;      // rare: promotions / demotions
            if (__ballot(moved_m != 0)) {
                for (int t = 0; t < M; t++) {
                    if (t == s) continue;
                    int hdr = REC(kRecHead + t * SW);
                    if ((hdr >> 16) == kListAbsent) continue;
                    int len = hdr & 0xffff;
                    for (int j = 0; j < len; j++) {
                        int x = REC(kRecHead + t * SW + 1 + j);
#pragma unroll
                        for (int u = 0; u < NPT; u++) {
                            if (((moved_m >> u) & 1) && x == tid + u * T) {
                                totv[u] -= w;
                                q.cnt[t * NX + x] -= w;             // only this thread touches the node's counters
                                changed_m |= 1u << u;
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < NPT; u++) {
                const int leave = ((own_m >> u) & 1) ? w : 0, enter = ((emitted_m >> u) & 1) ? w : 0;
                cntv[u] += enter - leave;
                totv[u] += enter - leave;
                if (NP > 0 && ((emitted_m >> u) & 1)) {                   // plan.go:238-245
                    q.ntn[(size_t)row * N + tid + u * T] = ntnv[u] + 1;
                    if (next_row == row) ntn_pre[u] = ntnv[u] + 1;
                }
            }
            changed_m |= own_m | emitted_m;
        }
        PH(6);
        if (changed_m) {
#pragma unroll
            for (int u = 0; u < NPT; u++)
                if ((changed_m >> u) & 1) {
                    if (NP > 0) ffv[u] = (0.001 * (double)totv[u]) / (double)NP;
                    g[u] = seq_score(cntv[u], 0, ffv[u], (hasw_m >> u) & 1, wv[u], NP, 0.0, q.booster_kind, lpT);
                    if (spec_ok) { cntL[tid + u * T] = cntv[u]; totL[tid + u * T] = totv[u]; }
                }
        }
        if (spec_ok) {                             // speculate again after a step that kept its nodes
            int hs = REC(kRecHead + s * SW);
            bool stay = (hs >> 16) != kListAbsent && (hs & 0xffff) == k && n_out == k;
#pragma unroll
            for (int c = 0; c < KM; c++)
                if (stay && c < k && chosen[c] != REC(kRecHead + s * SW + 1 + c)) stay = false;
            try_spec = stay;
        }
        PH(7);
        if (tid == 0) {
            int is_nil = (n_out == 0 && q.n_alive == 0 && !any_higher_key && !q.hier);
            int* o = q.out + (size_t)oi * q.OW;
            o[0] = n_out | (is_nil << 16);
#pragma unroll
            for (int c = 0; c < KM; c++) if (c < k) o[1 + c] = chosen[c];
            if (n_out < k) {                       // plan.go:230-235
                int wi = *q.warn_count;
                q.warn_part[wi] = p;
                q.warn_state[wi] = s;
                *q.warn_count = wi + 1;
            }
        }
#undef REC
        oi++;
        PH(8);
    }
    PH_DUMP(q.end - q.beg);
    if (spec_ok && tid == 0 && q.spec_count) *q.spec_count += spec_steps;

#pragma unroll
    for (int i = 0; i < NPT; i++) {
        int n = tid + i * T;
        if (n < NX) q.cnt[s * NX + n] = cntv[i];
    }
}


}  // namespace blance
