// k_stay_by_top: a region-chain pass in which EVERY step keeps its nodes, verified in parallel.
// Part of tu_chain.hip; see DESIGN.md section 4.
#pragma once

namespace blance {

// A converged sweep is a sweep of stays.  A stay changes no load counter, so the only state that
// flows from step to step in such a pass is nodeToNodeCounts (plan.go:238-245) -- and row `top` of it
// is read and bumped only by steps whose top priority node is `top` (plan.go:134-138, :238-245).
// Under the hypothesis "every step stays" the steps of different top priority nodes are therefore
// independent: one THREAD per top priority node walks that node's steps in pass order with its row
// of nodeToNodeCounts in LDS, validates each step exactly as k_pass_chain's stay test does (own
// nodes scored with the exact row entries, sorted by (score, position), exclude classes consistent
// with that order, all below the smallest partition-independent score of the region -- a lower
// bound of every other candidate) and writes what the step emits.  k_pass_chain runs the same test
// 64 steps at a time on ONE wave per region because it cannot know in advance that nothing moves;
// here 4,096 threads (config 3) run at once.  A single step that fails the test raises `flag` and
// the host runs the pass with k_pass_chain from the same (untouched) state.
// Work list: steps grouped by the GLOBAL leaf index of their top priority node, pass order inside a
// group (stable counting sort by the driver): top_off[leaf] .. top_off[leaf + 1] into top_order.
template <int KM>
__global__ __launch_bounds__(64) void k_stay_by_top(StayParams q) {
    BLANCE_DYN_LDS(lds);
    const int lane = threadIdx.x;
    const int rg = q.wg_region[blockIdx.x];
    const int lo = q.reg_lo[rg], hi = q.reg_hi[rg], size = hi - lo;
    const int NX = q.NX, NP = q.NP, k = q.k;
    // region tables (as in k_pass_chain), then this workgroup's rows of nodeToNodeCounts: row of thread t at rowL[leaf * 64 + t]
    int* cntL = (int*)lds;                           // [size]
    int* totL = cntL + size;
    int* nidL = totL + size;
    int* wgtL = nidL + size;
    int* flgL = wgtL + size;                         // bit 0 alive (in nodesNext), bit 1 has weight
    int* clsL = flgL + size;
    int* cszL = clsL + size;
    int* rowL = cszL + size;                         // [size][64]
    // the two NumPartitions quotients of the score from tables, by k_pass_chain's own chain_score (plan.go:638-652)
    double* lp_tab = (double*)(rowL + size * 64 + ((size * 71) & 1));     // 8-byte aligned: 71 ints per leaf before it
    double* ff_tab = lp_tab + kLpTab;
    for (int i = lane; i < kLpTab; i += 64) lp_tab[i] = NP > 0 ? (double)i / (double)NP : 0.0;
    for (int i = lane; i < kFfTab; i += 64) ff_tab[i] = NP > 0 ? (0.001 * (double)i) / (double)NP : 0.0;
    __syncthreads();
    double ms = pos_inf();
    int mn = INT_MAX;
    for (int i = lane; i < size; i += 64) {
        const int pos = lo + i, n = q.leaf_node[pos];
        int c = 0, t = 0, w = 0, fl = 0, cl = -1;
        if (n >= 0) {
            c = q.cnt[q.s * NX + n];
            for (int tt = 0; tt <= q.M; tt++) t += q.cnt[tt * NX + n];
            w = q.node_weight[n];
            fl = ((n < q.N && q.alive[n]) ? 1 : 0) | (q.node_has_weight[n] ? 2 : 0);
            cl = q.leaf_cls[pos];
            if (fl & 1) {                            // partition-independent score of a candidate
                const double g = chain_score(c, 0, t, (fl >> 1) & 1, w, NP, 0.0, q.booster_kind, lp_tab, ff_tab);
                if (better(g, n, ms, mn)) { ms = g; mn = n; }
            }
        }
        cntL[i] = c; totL[i] = t; nidL[i] = n >= 0 ? n : -2; wgtL[i] = w; flgL[i] = fl; clsL[i] = cl;
        cszL[i] = q.cls_size[pos];
    }
    for (int i = lane; i < size * 64; i += 64) rowL[i] = 0;
    // smallest (g, node) over the region's candidates: the bound every kept node has to beat
    const int gmin_n = wave_argmin(ms, mn);
    double gmin_s = pos_inf();
    {
        const unsigned long long bm = __ballot(mn == gmin_n && gmin_n != INT_MAX);
        if (bm) {
            const int wl = __ffsll((long long)bm) - 1;
            gmin_s = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ms), wl), __builtin_amdgcn_readlane(__double2loint(ms), wl));
        }
    }
    __syncthreads();
    const int my_leaf = 64 * q.wg_chunk[blockIdx.x] + lane;          // local to the region
    if (my_leaf >= size) return;
    const int gl = lo + my_leaf;
    const int beg = q.top_off[gl], end = q.top_off[gl + 1];
    bool bad = false;
    // the record words the stay test reads (stick lo / hi, top leaf, counts word, top's exclude class, own leaves) of the
    // next step travel while this one is tested; the chain indices are read three steps ahead
    constexpr int kW = 5 + KM;
    int nx[kW];
    auto fetch = [&](int ci) {
        const int32_t* rp = q.crec + (size_t)ci * kCW;
        nx[0] = rp[2]; nx[1] = rp[3]; nx[2] = rp[4]; nx[3] = rp[5]; nx[4] = rp[6];
#pragma unroll
        for (int e = 0; e < KM; e++) nx[5 + e] = rp[kCOwn + e];
    };
    int ci0 = beg < end ? q.top_order[beg] : 0, ci1 = beg + 1 < end ? q.top_order[beg + 1] : 0, ci2 = beg + 2 < end ? q.top_order[beg + 2] : 0;
    if (beg < end) fetch(ci0);
    for (int pos = beg; pos < end && !bad; pos++) {
        int cur[kW];
#pragma unroll
        for (int e = 0; e < kW; e++) cur[e] = nx[e];
        const int ci = ci0;
        ci0 = ci1; ci1 = ci2;
        ci2 = pos + 3 < end ? q.top_order[pos + 3] : 0;
        if (pos + 1 < end) fetch(ci0);
        const double vstick = __hiloint2double(cur[1], cur[0]);
        const int vtl = cur[2], cn = cur[3];
        int own[KM];
#pragma unroll
        for (int j = 0; j < KM; j++) own[j] = cur[5 + j];
        int oc[KM + 1];
        oc[0] = cur[4];
        if (vtl != my_leaf) bad = true;                                                   // (grouping went wrong: never)
        if (!((cn >> 24) & 1) || (cn & 0xff) != k || ((cn >> 25) & 1)) bad = true;        // exactly k nodes, all here
        int oi[KM], on[KM];
        double so[KM];
#pragma unroll
        for (int j = 0; j < KM; j++) {
            on[j] = -3; so[j] = 0.0; oi[j] = 0; oc[j + 1] = -1;
            if (j < k) {
                int li = own[j];
                if (li < 0 || li >= size) { bad = true; li = 0; }
                oi[j] = li;
                on[j] = nidL[li];
                oc[j + 1] = clsL[li];
                if (!(flgL[li] & 1)) bad = true;
                so[j] = chain_score(cntL[li], rowL[li * 64 + lane], totL[li], (flgL[li] >> 1) & 1, wgtL[li], NP, vstick, q.booster_kind,
                                    lp_tab, ff_tab);
            }
        }
        // what a stay emits: its nodes in (score, position) order (plan.go:185-226)
#pragma unroll
        for (int j = 1; j < KM; j++) {
#pragma unroll
            for (int e = j; e > 0; e--) {
                if (e < k && better(so[e], on[e], so[e - 1], on[e - 1])) {
                    const double ts = so[e]; so[e] = so[e - 1]; so[e - 1] = ts;
                    const int tn = on[e]; on[e] = on[e - 1]; on[e - 1] = tn;
                    const int ti = oi[e]; oi[e] = oi[e - 1]; oi[e - 1] = ti;
                    const int tc = oc[e + 1]; oc[e + 1] = oc[e]; oc[e] = tc;
                }
            }
        }
        // anchors top, own_0 .. own_{k-2}: their exclude classes must leave candidates, and own_j must not sit in a
        // class excluded before its slot
        {
            int cov = 0;
#pragma unroll
            for (int j = 0; j < KM; j++) {
                if (j < k) {
                    if (oc[j] < 0) bad = true;
                    bool dup = false;
#pragma unroll
                    for (int e = 0; e < KM; e++) if (e < j && oc[e] == oc[j]) dup = true;
                    if (!dup && oc[j] >= 0) cov += cszL[oc[j]];
                    if (cov >= size) bad = true;
#pragma unroll
                    for (int e = 0; e < KM; e++) if (e <= j && oc[e] >= 0 && oc[e] == oc[j + 1]) bad = true;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < KM; j++)
            if (j < k && !better(so[j], on[j], gmin_s, gmin_n)) bad = true;
        if (bad) break;
        int32_t* op = q.out + (size_t)ci * q.OW;
        op[0] = k;
#pragma unroll
        for (int j = 0; j < KM; j++) {
            if (j < k) {
                op[1 + j] = on[j];
                rowL[oi[j] * 64 + lane] += 1;                                             // plan.go:238-245
            }
        }
    }
    if (bad) *q.flag = 1;
}

}  // namespace blance
