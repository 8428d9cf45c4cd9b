// k_stay_by_top: a region-chain pass in which EVERY step keeps its nodes, verified in parallel.
// Part of tu_chain.hip; see DESIGN.md section 4.
#pragma once

namespace blance {

// A converged sweep is a sweep of stays.  A stay changes no load counter, so the only state that
// flows from step to step in such a pass is nodeToNodeCounts (plan.go:238-245) -- and row `top` of it
// is read and bumped only by steps whose top priority node is `top` (plan.go:134-138, :238-245).
// Under the hypothesis "every step stays" that row is known WITHOUT walking: a stay emits the nodes it
// holds, so entry x of row `top`, as step j of that node's steps (pass order) reads it, is the number of
// steps before j that hold x -- a rank among equals.  Every step is then validated on its own, exactly
// as k_pass_chain's stay test does (own nodes scored with those row entries, sorted by (score,
// position), exclude classes consistent with that order, all below the smallest partition-independent
// score of the region -- a lower bound of every other candidate), and writes what the step emits.
// One WAVE per top priority node, a step per lane, 64 steps a round: the ranks inside a round come from
// a loop over the round's lanes (v_readlane), the ranks of earlier rounds from a row of counters in
// LDS the round's lanes bump when they are through.  (Round 5 had one THREAD per top priority node
// walk its steps with the row in LDS: 100 dependent steps a thread at config 3, 0.34 ms for what is
// 100,000 independent checks.)  A single step that fails the test raises `flag` and the host runs the
// pass with k_pass_chain from the same (untouched) state.
// Work list: steps grouped by the GLOBAL leaf index of their top priority node, pass order inside a
// group (stable counting sort by the driver): top_off[leaf] .. top_off[leaf + 1] into top_order.
constexpr int kStayWaves = 4;                        // top priority nodes of one workgroup, a wave each
constexpr int kStaySplit = 64 / kStayWaves;          // workgroups per entry of the (region, 64 leaves) work table

// nodeSorter.Score (plan.go:638-679) with both NumPartitions quotients divided out in place: the expressions
// chain_score's LDS tables are filled with, so the same bits
__device__ __forceinline__ double stay_score(int cnt, int ntn, int tot, int hasw, int w, int NP, double cf, int booster) {
    double lp = 0.0, ff = 0.0;
    if (NP > 0) {
        lp = (double)ntn / (double)NP;
        ff = (0.001 * (double)tot) / (double)NP;
    }
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

template <int KM>
__global__ __launch_bounds__(64 * kStayWaves) void k_stay_by_top(StayParams q) {
    BLANCE_DYN_LDS(lds);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wgi = blockIdx.x / kStaySplit, sub = blockIdx.x % kStaySplit;
    const int rg = q.wg_region[wgi];
    const int lo = q.reg_lo[rg], hi = q.reg_hi[rg], size = hi - lo;
    const int NX = q.NX, NP = q.NP, k = q.k;
    const int leaf0 = 64 * q.wg_chunk[wgi] + sub * kStayWaves;       // this workgroup's top priority nodes: leaves leaf0 .. leaf0 + 3 of the region
    if (leaf0 >= size) return;
    {
        const int l1 = leaf0 + kStayWaves < size ? leaf0 + kStayWaves : size;
        if (q.top_off[lo + leaf0] == q.top_off[lo + l1]) return;     // no step has one of them on top
    }
    // region tables (as in k_pass_chain), then a row of nodeToNodeCounts per wave: what the rounds so far have emitted
    int* cntL = (int*)lds;                           // [size]
    int* totL = cntL + size;
    int* nidL = totL + size;
    int* wgtL = nidL + size;
    int* flgL = wgtL + size;                         // bit 0 alive (in nodesNext), bit 1 has weight
    int* clsL = flgL + size;
    int* cszL = clsL + size;
    int* rowT = cszL + size;                         // [kStayWaves][size]
    int* redN = rowT + kStayWaves * size;            // [kStayWaves]
    double* redS = (double*)(redN + kStayWaves + ((size * (7 + kStayWaves) + kStayWaves) & 1));     // [kStayWaves], 8-byte aligned
    double ms = pos_inf();
    int mn = INT_MAX;
    for (int i = tid; i < size; i += 64 * kStayWaves) {
        const int pos = lo + i, n = q.leaf_node[pos];
        int c = 0, t = 0, w = 0, fl = 0, cl = -1;
        if (n >= 0) {
            c = q.cnt[q.s * NX + n];
            for (int tt = 0; tt <= q.M; tt++) t += q.cnt[tt * NX + n];
            w = q.node_weight[n];
            fl = ((n < q.N && q.alive[n]) ? 1 : 0) | (q.node_has_weight[n] ? 2 : 0);
            cl = q.leaf_cls[pos];
            if (fl & 1) {                            // partition-independent score of a candidate
                const double g = stay_score(c, 0, t, (fl >> 1) & 1, w, NP, 0.0, q.booster_kind);
                if (better(g, n, ms, mn)) { ms = g; mn = n; }
            }
        }
        cntL[i] = c; totL[i] = t; nidL[i] = n >= 0 ? n : -2; wgtL[i] = w; flgL[i] = fl; clsL[i] = cl;
        cszL[i] = q.cls_size[pos];
    }
    for (int i = tid; i < size * kStayWaves; i += 64 * kStayWaves) rowT[i] = 0;
    // smallest (g, node) over the region's candidates: the bound every kept node has to beat
    {
        const int wn = wave_argmin(ms, mn);
        double wsc = pos_inf();
        const unsigned long long bm = __ballot(mn == wn && wn != INT_MAX);
        if (bm) {
            const int wl = __ffsll((long long)bm) - 1;
            wsc = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ms), wl), __builtin_amdgcn_readlane(__double2loint(ms), wl));
        }
        if (lane == 0) { redN[wave] = wn; redS[wave] = wsc; }
    }
    __syncthreads();
    double gmin_s = pos_inf();
    int gmin_n = INT_MAX;
#pragma unroll
    for (int w2 = 0; w2 < kStayWaves; w2++)
        if (redN[w2] != INT_MAX && better(redS[w2], redN[w2], gmin_s, gmin_n)) { gmin_s = redS[w2]; gmin_n = redN[w2]; }
    // ---- from here on the waves go their own ways (no block barrier below)
    const int my_leaf = leaf0 + wave;                // local to the region
    if (my_leaf >= size) return;
    const int gl = lo + my_leaf;
    const int beg = q.top_off[gl], end = q.top_off[gl + 1];
    int* row = rowT + wave * size;
    bool bad = false;
    // the record words the stay test reads (stick lo / hi, top leaf, counts word, top's exclude class, own leaves) of the
    // next round travel while this one is tested; the chain indices are read two rounds ahead
    constexpr int kW = 5 + KM;
    int nx[kW];
    auto fetch = [&](int ci) {
        const int32_t* rp = q.crec + (size_t)ci * kCW;
        nx[0] = rp[2]; nx[1] = rp[3]; nx[2] = rp[4]; nx[3] = rp[5]; nx[4] = rp[6];
#pragma unroll
        for (int e = 0; e < KM; e++) nx[5 + e] = rp[kCOwn + e];
    };
#pragma unroll
    for (int e = 0; e < kW; e++) nx[e] = 0;
    int ci0 = beg + lane < end ? q.top_order[beg + lane] : 0, ci1 = beg + 64 + lane < end ? q.top_order[beg + 64 + lane] : 0;
    if (beg + lane < end) fetch(ci0);
    for (int base = beg; base < end; base += 64) {
        const int nv = end - base < 64 ? end - base : 64;                                 // steps of this round (uniform)
        const bool valid = lane < nv;
        int cur[kW];
#pragma unroll
        for (int e = 0; e < kW; e++) cur[e] = nx[e];
        const int ci = ci0;
        ci0 = ci1;
        ci1 = base + 128 + lane < end ? q.top_order[base + 128 + lane] : 0;
        if (base + 64 + lane < end) fetch(ci0);
        const double vstick = __hiloint2double(cur[1], cur[0]);
        const int vtl = cur[2], cn = cur[3];
        int oc[KM + 1];
        oc[0] = cur[4];
        if (valid && vtl != my_leaf) bad = true;                                          // (grouping went wrong: never)
        if (valid && (!((cn >> 24) & 1) || (cn & 0xff) != k || ((cn >> 25) & 1))) bad = true;   // exactly k nodes, all here
        int oi[KM], on[KM], rk[KM];
#pragma unroll
        for (int j = 0; j < KM; j++) {
            int li = (valid && j < k) ? cur[5 + j] : -1;
            if (valid && j < k && (li < 0 || li >= size)) { bad = true; li = -1; }
            oi[j] = li;                                                                   // (-1: no such node, equal to nothing below)
            rk[j] = 0;
        }
        // how many earlier steps of this round hold my nodes (the round's lanes in pass order)
        for (int e = 0; e + 1 < nv; e++) {
            const bool before = e < lane;
#pragma unroll
            for (int j2 = 0; j2 < KM; j2++) {
                const int xe = __builtin_amdgcn_readlane(oi[j2], e);
                if (xe < 0) continue;                                                     // (uniform)
#pragma unroll
                for (int j = 0; j < KM; j++) rk[j] += (before && xe == oi[j]) ? 1 : 0;
            }
        }
        double so[KM];
#pragma unroll
        for (int j = 0; j < KM; j++) {
            on[j] = -3; so[j] = 0.0; oc[j + 1] = -1;
            if (j < k) {
                const int li = oi[j] >= 0 ? oi[j] : 0;
                oi[j] = li;
                on[j] = nidL[li];
                oc[j + 1] = clsL[li];
                if (valid && !(flgL[li] & 1)) bad = true;
                so[j] = stay_score(cntL[li], row[li] + rk[j], totL[li], (flgL[li] >> 1) & 1, wgtL[li], NP, vstick, q.booster_kind);
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
        if (valid) {
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
#pragma unroll
            for (int j = 0; j < KM; j++)
                if (j < k && !better(so[j], on[j], gmin_s, gmin_n)) bad = true;
        }
        if (__ballot(bad)) break;                                                         // (uniform: the pass does not stand)
        BLANCE_WAVE_SYNC();                                                               // (every lane has read the row of the rounds before)
        if (valid) {
            int32_t* op = q.out + (size_t)ci * q.OW;
            op[0] = k;
#pragma unroll
            for (int j = 0; j < KM; j++) {
                if (j < k) {
                    op[1 + j] = on[j];
                    atomicAdd(&row[oi[j]], 1);                                            // plan.go:238-245
                }
            }
        }
        BLANCE_WAVE_SYNC();                                                               // (one wave: LDS is in order)
    }
    if (bad) *q.flag = 1;
}

}  // namespace blance
