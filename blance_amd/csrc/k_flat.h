// Flat bulk engine: certain-stay runs, fresh-identical runs (threshold search + stable radix sort).
// Part of blance_hip.hip (one translation unit); see DESIGN.md section 4.
#pragma once

namespace blance {

// ============================================================================
// Flat bulk engine: passes of a state without hierarchy rules.  Two kinds of
// steps are resolved without walking them one by one, both exactly:
//
//  * certain stays (k = 1): the partition keeps its node if even the largest
//    score it could have (nodeToNodeCounts term bounded from above) beats the
//    smallest partition-independent score of every other candidate, which is a
//    lower bound of that candidate's true score (the skipped terms are >= 0
//    and IEEE add / divide / subtract are monotone).  Stays change no counter,
//    so a whole run of them is validated by independent threads.
//  * fresh identical partitions (k = 1): partitions without nodes, exclusions
//    or own rows all see the same candidate scores; each node's score grows
//    with every pick it receives, so the greedy's picks are the R smallest
//    elements of the merged per-node score sequences, in sorted order
//    ((score, position) ascending).  A threshold search finds how many picks
//    each node gets, a stable radix sort orders them.
//
// Everything else goes through k_pass_seq on a sub-range of the pass.
// ============================================================================

__device__ __forceinline__ unsigned long long sortable_key(double v) {
    if (v == 0.0) v = 0.0;                         // -0.0 and +0.0 compare equal
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

// tot[n], g[n] and the kTopList smallest (g, n) over nodesNext.  One workgroup.
__global__ __launch_bounds__(1024) void k_flat_prepare(FlatParams q, int32_t* tot, double* g, double* top_g,
                                                       int32_t* top_n) {
    BLANCE_DYN_LDS(lds);
    RedSlot* red = (RedSlot*)lds;
    int round = 0;
    const int tid = threadIdx.x;
    for (int n = tid; n < q.NX; n += 1024) {
        int tsum = 0;
        for (int t = 0; t <= q.M; t++) tsum += q.cnt[t * q.NX + n];
        tot[n] = tsum;
        g[n] = node_score(q.cnt[q.s * q.NX + n], 0, tsum, q.node_has_weight[n], q.node_weight[n], q.NP, 0.0,
                          q.booster_kind);
    }
    __syncthreads();
    double last_s = 0.0;
    int last_n = -1;
    for (int r = 0; r < kTopList; r++) {
        double bs = pos_inf();
        int bn = INT_MAX;
        for (int n = tid; n < q.N; n += 1024) {
            if (!q.alive[n]) continue;
            double v = g[n];
            if (r > 0 && !better(last_s, last_n, v, n)) continue;     // already listed
            if (better(v, n, bs, bn)) { bs = v; bn = n; }
        }
        int best = block_argmin<1024>(bs, bn, red, round);
        if (best == INT_MAX) {
            if (tid == 0) { top_g[r] = pos_inf(); top_n[r] = INT_MAX; }
            last_s = pos_inf(); last_n = INT_MAX;
        } else {
            last_s = g[best]; last_n = best;
            if (tid == 0) { top_g[r] = last_s; top_n[r] = best; }
        }
    }
}

// steps of the pass per nodeToNodeCounts row: an upper bound of any entry of that row
__global__ void k_flat_row_count(FlatParams q, int32_t* row_count) {
    int oi = blockIdx.x * blockDim.x + threadIdx.x;
    if (oi >= q.P) oi = -1;
    const int32_t* r = q.rec + (size_t)(oi < 0 ? 0 : oi) * q.RW;
    int hdr = r[kRecHead + q.top_state * (1 + q.L)];
    int top = ((hdr >> 16) != kListAbsent && (hdr & 0xffff) > 0) ? r[kRecHead + q.top_state * (1 + q.L) + 1] : -1;
    if (oi >= 0 && top >= 0) atomicAdd(&row_count[top], 1);
    unsigned long long none = __ballot(oi >= 0 && top < 0);    // the "" row: one atomic per wave
    if (none && (int)(threadIdx.x & 63) == __ffsll((long long)none) - 1) atomicAdd(&row_count[q.NX], __popcll(none));
}

// Classify the steps [beg, end): record the first one that is not a certain
// stay and the first one that is not "fresh identical" to step beg.
// The first step of the range for which a flag holds: every WAVE leaves the smallest such step of its 64 in
// part[flag kind][wave] (INT_MAX if none) -- no atomics: when every step is flagged (a fresh pass: nothing is a stay),
// thousands of waves would otherwise hit one word at once -- and k_flat_scan_min reduces the two rows.
__device__ __forceinline__ void scan_note_first(bool flag, int oi, int32_t* part_row) {
    const unsigned long long m = __ballot(flag);
    const int first = m ? __ffsll((long long)m) - 1 : -1;
    if ((int)(threadIdx.x & 63) == (first < 0 ? 0 : first))
        part_row[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = first < 0 ? INT_MAX : oi;
}

__global__ __launch_bounds__(1024) void k_flat_scan_min(int n_waves, const int32_t* part /* [2][n_waves] */, int32_t* scan) {
    BLANCE_DYN_LDS(lds);
    int* red = (int*)lds;                            // [2][16]
    const int tid = threadIdx.x;
    int m0 = INT_MAX, m1 = INT_MAX;
    for (int i = tid; i < n_waves; i += 1024) {
        const int a = part[i], b = part[n_waves + i];
        m0 = a < m0 ? a : m0;
        m1 = b < m1 ? b : m1;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const int a = __shfl_xor(m0, off, 64), b = __shfl_xor(m1, off, 64);
        m0 = a < m0 ? a : m0;
        m1 = b < m1 ? b : m1;
    }
    if ((tid & 63) == 0) { red[tid >> 6] = m0; red[16 + (tid >> 6)] = m1; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 0; w < 16; w++) { m0 = red[w] < m0 ? red[w] : m0; m1 = red[16 + w] < m1 ? red[16 + w] : m1; }
        scan[0] = m0;                                // first step that is not a certain stay (INT_MAX: none)
        scan[1] = m1;                                // first step that is not fresh-identical to the range's first
    }
}

__global__ void k_flat_scan(FlatParams q, int beg, int end) {
    int oi = beg + blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = oi < end;
    if (!in_range) oi = end - 1;                     // keep the wave whole for the ballots below
    const int SW = 1 + q.L;
    const int32_t* r = q.rec + (size_t)oi * q.RW;
    const int w = r[1];
    const double stick = __hiloint2double(r[3], r[2]);
    int hT = r[kRecHead + q.top_state * SW];
    int top = ((hT >> 16) != kListAbsent && (hT & 0xffff) > 0) ? r[kRecHead + q.top_state * SW + 1] : -1;
    int hs = r[kRecHead + q.s * SW];
    int own_len = (hs >> 16) == kListAbsent ? 0 : (hs & 0xffff);
    int all_len = 0, high_len = 0;                     // nodes the partition holds in any / in higher priority states
    for (int t = 0; t < q.M; t++) {
        int h = r[kRecHead + t * SW];
        if ((h >> 16) == kListAbsent) continue;
        all_len += h & 0xffff;
        if ((q.higher_mask >> t) & 1) high_len += h & 0xffff;
    }
    // ---- fresh identical to step beg?
    {
        const int32_t* r0 = q.rec + (size_t)beg * q.RW;
        // no node anywhere: nothing to exclude, demote or promote (plan.go:290-297).  With
        // NumPartitions == 0 no score term depends on the partition (plan.go:638,:647), so one node
        // in a higher priority state -- just not a candidate, plan.go:146-154 -- is allowed too
        // (k_fresh_excl), and the top priority node does not matter.
        bool fresh = w > 0 && w == r0[1] &&
                     (q.NP == 0 ? (q.k <= 2 && all_len == high_len && high_len <= 1) : (q.k == 1 && all_len == 0 && top < 0));
        scan_note_first(in_range && !fresh, oi, q.scan_part + q.scan_waves);
    }
    // ---- certain stay?
    bool stay = false;
    if (q.k == 1 && own_len == 1) {
        int o = r[kRecHead + q.s * SW + 1];
        bool ok = o < q.N && q.alive[o];
        for (int t = 0; t < q.M && ok; t++) {           // o in another list of the partition: promoted / excluded
            if (t == q.s) continue;
            int h = r[kRecHead + t * SW];
            if ((h >> 16) == kListAbsent) continue;
            for (int j = 0; j < (h & 0xffff); j++) if (r[kRecHead + t * SW + 1 + j] == o) ok = false;
        }
        if (ok) {
            int ub = q.NP > 0 ? q.row_count[top < 0 ? q.NX : top] : 0;
            double s_hi = node_score(q.cnt[q.s * q.NX + o], ub, q.tot[o], q.node_has_weight[o], q.node_weight[o],
                                     q.NP, stick, q.booster_kind);
            // smallest other candidate: first listed node that is neither o nor excluded
            bool found = false;
            for (int e = 0; e < kTopList && !found; e++) {
                int n = q.top_n[e];
                if (n == INT_MAX) { found = true; stay = true; break; }        // no other candidate at all
                bool excl = n == o;
                for (int t = 0; t < q.M && !excl; t++) {
                    int h = r[kRecHead + t * SW];
                    if ((h >> 16) == kListAbsent || !((q.higher_mask >> t) & 1)) continue;
                    for (int j = 0; j < (h & 0xffff); j++) if (r[kRecHead + t * SW + 1 + j] == n) excl = true;
                }
                if (excl) continue;
                found = true;
                stay = better(s_hi, o, q.top_g[e], n);
            }
        }
    }
    scan_note_first(in_range && !stay, oi, q.scan_part);
}

// commit a run of certain stays: the lists do not change; nodeToNodeCounts does (plan.go:238-245)
// (bump = 0: the whole pass is this one run of stays -- nothing reads nodeToNodeCounts afterwards, plan.go:266 drops it)
__global__ void k_flat_commit_stay(FlatParams q, int beg, int end, int bump) {
    int oi = beg + blockIdx.x * blockDim.x + threadIdx.x;
    if (oi >= end) return;
    const int SW = 1 + q.L;
    const int32_t* r = q.rec + (size_t)oi * q.RW;
    int o = r[kRecHead + q.s * SW + 1];
    int* out = q.out + (size_t)oi * q.OW;
    out[0] = 1;
    out[1] = o;
    if (q.NP > 0 && bump) {
        int hT = r[kRecHead + q.top_state * SW];
        int top = ((hT >> 16) != kListAbsent && (hT & 0xffff) > 0) ? r[kRecHead + q.top_state * SW + 1] : -1;
        atomicAdd(&q.ntn[(size_t)(top < 0 ? q.NX : top) * q.N + o], 1);
    }
}

// ---- fresh identical run: how many of the R picks each node receives --------
// (score of node n after c picks of weight w; hasw / nw: the node's weight, loaded by the caller)
__device__ __forceinline__ unsigned long long fresh_key_w(const FlatParams& q, int hasw, int nw, int cnt0, int tot0, int ntn0,
                                                          int w, int c) {
    return sortable_key(node_score(cnt0 + c * w, ntn0 + c, tot0 + c * w, hasw, nw, q.NP, 0.0, q.booster_kind));
}
__device__ __forceinline__ unsigned long long fresh_key(const FlatParams& q, int n, int cnt0, int tot0, int ntn0,
                                                        int w, int c) {
    return fresh_key_w(q, q.node_has_weight[n], q.node_weight[n], cnt0, tot0, ntn0, w, c);
}

// #{c in [0, R): key(n, c) <= tau}; the keys grow with c
__device__ __forceinline__ int fresh_count_le(const FlatParams& q, int n, int cnt0, int tot0, int ntn0, int w, int R,
                                              unsigned long long tau) {
    int lo = 0, hi = R;                              // first c in [0, R] with key > tau
    while (lo < hi) {
        int mid = lo + (hi - lo) / 2;
        if (fresh_key(q, n, cnt0, tot0, ntn0, w, mid) <= tau) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(1024) void k_fresh_threshold(FlatParams q, int beg, int R, int32_t* m_out,
                                                          int32_t* m_off) {
    BLANCE_DYN_LDS(lds);
    long long* part = (long long*)lds;               // [1024] partial sums, then [1024] scan scratch
    const int tid = threadIdx.x;
    const int w = q.rec[(size_t)beg * q.RW + 1];
    unsigned long long lo = 0, hi = ~0ull;           // smallest tau with total(tau) >= R
    // per node: count_le(lo - 1) and count_le(hi) bracket count_le(mid) of every later probe, so
    // the inner searches shrink with the outer one (a node's count is monotone in tau)
    constexpr int kPer = 8;                          // N <= 8192 = 8 * 1024
    int c_lo[kPer], c_hi[kPer], c_mid[kPer];
    // this thread's nodes: their counters stay in registers over the ~60 probes of the search
    int n_c0[kPer], n_t0[kPer], n_nt[kPer], n_hw[kPer], n_nw[kPer];
    bool n_on[kPer];
#pragma unroll
    for (int i = 0; i < kPer; i++) {
        c_lo[i] = 0; c_hi[i] = R; c_mid[i] = 0;
        const int n = tid + i * 1024;
        n_on[i] = n < q.N && q.alive[n];
        n_c0[i] = n_on[i] ? q.cnt[q.s * q.NX + n] : 0;
        n_t0[i] = n_on[i] ? q.tot[n] : 0;
        n_nt[i] = (n_on[i] && q.NP > 0) ? q.ntn[(size_t)q.NX * q.N + n] : 0;
        n_hw[i] = n_on[i] ? q.node_has_weight[n] : 0;
        n_nw[i] = n_on[i] ? q.node_weight[n] : 0;
    }
    int probe = 0;
    {
        // A bracket before the bisection.  With A nodes in the race, the first ceil(R / A) elements of every node's sequence
        // are >= R elements, so the R-th smallest is <= the largest of the nodes' elements number ceil(R / A) - 1; and a tau
        // below every node's element number floor((R - 1) / A) has at most A floor((R - 1) / A) < R elements at or under
        // it.  Nodes of equal load (a fresh plan) make the two bounds ONE key and the search below has nothing to do;
        // otherwise it starts from a few thousand ulps instead of the whole 64-bit key space.
        int mine = 0;
#pragma unroll
        for (int i = 0; i < kPer; i++) mine += n_on[i] ? 1 : 0;
        long long cntA = mine;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) cntA += __shfl_xor((int)cntA, off, 64);
        long long* row = part + 32;
        if ((tid & 63) == 0) row[tid >> 6] = cntA;
        __syncthreads();
        long long A = 0;
#pragma unroll
        for (int wv = 0; wv < 16; wv++) A += row[wv];
        __syncthreads();
        if (A > 0 && R > 0) {
            const int q_hi = (int)(((long long)R + A - 1) / A) - 1, q_lo = (int)(((long long)R - 1) / A);
            unsigned long long kmin = ~0ull, kmax = 0;
#pragma unroll
            for (int i = 0; i < kPer; i++) {
                if (!n_on[i]) continue;
                const unsigned long long a = fresh_key_w(q, n_hw[i], n_nw[i], n_c0[i], n_t0[i], n_nt[i], w, q_lo);
                const unsigned long long b = fresh_key_w(q, n_hw[i], n_nw[i], n_c0[i], n_t0[i], n_nt[i], w, q_hi < R ? q_hi : R - 1);
                kmin = a < kmin ? a : kmin;
                kmax = b > kmax ? b : kmax;
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const unsigned long long a = ((unsigned long long)(unsigned)__shfl_xor((int)(unsigned)(kmin >> 32), off, 64) << 32) |
                                             (unsigned)__shfl_xor((int)(unsigned)kmin, off, 64);
                const unsigned long long b = ((unsigned long long)(unsigned)__shfl_xor((int)(unsigned)(kmax >> 32), off, 64) << 32) |
                                             (unsigned)__shfl_xor((int)(unsigned)kmax, off, 64);
                kmin = a < kmin ? a : kmin;
                kmax = b > kmax ? b : kmax;
            }
            unsigned long long* urow = (unsigned long long*)(part + 64);
            if ((tid & 63) == 0) { urow[tid >> 6] = kmin; urow[16 + (tid >> 6)] = kmax; }
            __syncthreads();
#pragma unroll
            for (int wv = 0; wv < 16; wv++) {
                kmin = urow[wv] < kmin ? urow[wv] : kmin;
                kmax = urow[16 + wv] > kmax ? urow[16 + wv] : kmax;
            }
            __syncthreads();
            if (kmin <= kmax) { lo = kmin; hi = kmax; }
        }
    }
    while (lo < hi) {
        unsigned long long mid = lo + (hi - lo) / 2;
        long long sum = 0;
#pragma unroll
        for (int i = 0; i < kPer; i++) {
            if (!n_on[i]) continue;
            const int ntn0 = n_nt[i], c0 = n_c0[i], t0 = n_t0[i];
            int a = c_lo[i], b = c_hi[i];            // first c in [a, b] with key > mid
            while (a < b) {
                int m = a + (b - a) / 2;
                if (fresh_key_w(q, n_hw[i], n_nw[i], c0, t0, ntn0, w, m) <= mid) a = m + 1; else b = m;
            }
            c_mid[i] = a;
            sum += a;
        }
        // 64 outer probes: the block sum costs two barriers (wave sums by lane shuffles, 16 partials in LDS)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const int lo32 = __shfl_xor((int)(unsigned)sum, off, 64), hi32 = __shfl_xor((int)(sum >> 32), off, 64);
            sum += (long long)(((unsigned long long)(unsigned)hi32 << 32) | (unsigned)lo32);
        }
        // (partials alternate between two LDS rows: one barrier per probe is enough)
        long long* row = part + (probe & 1) * 16;
        probe++;
        if ((tid & 63) == 0) row[tid >> 6] = sum;
        __syncthreads();
        long long total = 0;
#pragma unroll
        for (int wv = 0; wv < 16; wv++) total += row[wv];
        if (total >= R) {
            hi = mid;
#pragma unroll
            for (int i = 0; i < kPer; i++) c_hi[i] = c_mid[i];
        } else {
            lo = mid + 1;
#pragma unroll
            for (int i = 0; i < kPer; i++) c_lo[i] = c_mid[i];
        }
    }
    __syncthreads();                                 // (the last probe's partials are read; part is reused below)
    const unsigned long long tau = lo;
    // picks strictly below tau, then the ties at tau in node order
    const int per = (q.N + 1023) / 1024;
    const int nb = tid * per, ne = nb + per < q.N ? nb + per : q.N;
    long long below = 0, ties = 0;
    for (int n = nb; n < ne; n++) {
        int lt = 0, le = 0;
        if (q.alive[n]) {
            int ntn0 = q.NP > 0 ? q.ntn[(size_t)q.NX * q.N + n] : 0;
            int c0 = q.cnt[q.s * q.NX + n], t0 = q.tot[n];
            lt = tau == 0 ? 0 : fresh_count_le(q, n, c0, t0, ntn0, w, R, tau - 1);
            le = fresh_count_le(q, n, c0, t0, ntn0, w, R, tau);
        }
        m_out[n] = lt;
        m_off[n] = le - lt;                          // ties of node n, consumed below
        below += lt;
        ties += le - lt;
    }
    // exclusive scans over threads (contiguous node slices keep node order)
    long long* sb = part;
    long long* st = part + 1024;
    sb[tid] = below; st[tid] = ties;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        long long a = tid >= off ? sb[tid - off] : 0, b = tid >= off ? st[tid - off] : 0;
        __syncthreads();
        sb[tid] += a; st[tid] += b;
        __syncthreads();
    }
    long long total_below = sb[1023];
    long long rem = (long long)R - total_below;      // ties to hand out, in node order
    long long ties_before = st[tid] - ties;
    long long elems_before = sb[tid] - below;        // picks below tau of earlier nodes
    for (int n = nb; n < ne; n++) {
        long long avail = m_off[n];
        long long left = rem - ties_before;
        long long take = left <= 0 ? 0 : (avail < left ? avail : left);
        int m = m_out[n] + (int)take;
        ties_before += avail;
        m_out[n] = m;
    }
    __syncthreads();
    // element offsets: exclusive scan of the final counts
    long long mine = 0;
    for (int n = nb; n < ne; n++) mine += m_out[n];
    sb[tid] = mine;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        long long a = tid >= off ? sb[tid - off] : 0;
        __syncthreads();
        sb[tid] += a;
        __syncthreads();
    }
    long long acc = sb[tid] - mine;
    for (int n = nb; n < ne; n++) { m_off[n] = (int)acc; acc += m_out[n]; }
    if (tid == 1023) m_off[q.N] = (int)sb[1023];
    (void)elems_before;
}

// the R picked (score, node) elements in node-major order
__global__ void k_fresh_emit(FlatParams q, int beg, int R, const int32_t* m_off, unsigned long long* keys,
                             int32_t* vals) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= R) return;
    int lo = 0, hi = q.N;                            // last n with m_off[n] <= e
    while (hi - lo > 1) {
        int mid = (lo + hi) / 2;
        if (m_off[mid] <= e) lo = mid; else hi = mid;
    }
    int n = lo, c = e - m_off[n];
    int w = q.rec[(size_t)beg * q.RW + 1];
    int ntn0 = q.NP > 0 ? q.ntn[(size_t)q.NX * q.N + n] : 0;
    // (int_keys: node_score is (double)(cnt + c w) + 0.0 + 0.0 - 0.0 then, plan.go:664-686 with NumPartitions == 0 and no weight)
    keys[e] = q.int_keys ? (unsigned long long)((long long)q.cnt[q.s * q.NX + n] + (long long)c * w + (1ll << 32))
                         : fresh_key(q, n, q.cnt[q.s * q.NX + n], q.tot[n], ntn0, w, c);
    vals[e] = n;
}

// The sorted sequence of a fresh run when every counter starts at zero and the keys are the integers count + c w (no
// NumPartitions, no node weights, one weight): (c, n) in (key, node) order is the nodes of nodesNext by id, again and
// again; node n's share of the first RS elements follows from its place in that list.
__global__ void k_fresh_cycle(int RS, int A, int N, const int32_t* alive_ids, const int32_t* alive_rank, int32_t* vals, int32_t* m) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < RS) vals[e] = alive_ids[e % A];
    if (e < N) {
        const int r = alive_rank[e];
        m[e] = r < 0 ? 0 : RS / A + (r < RS % A ? 1 : 0);
    }
}

__global__ void k_fresh_commit_steps(FlatParams q, int beg, int R, const int32_t* picks /* [k R], k per step */) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= R) return;
    int* out = q.out + (size_t)(beg + j) * q.OW;
    out[0] = q.k;
    for (int c = 0; c < q.k; c++) out[1 + c] = picks[(size_t)j * q.k + c];
}

__global__ void k_fresh_commit_nodes(FlatParams q, int beg, const int32_t* m, int32_t* cnt) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= q.N) return;
    int w = q.rec[(size_t)beg * q.RW + 1];
    if (m[n] == 0) return;
    cnt[q.s * q.NX + n] += m[n] * w;                 // plan.go:299-301
    if (q.NP > 0) q.ntn[(size_t)q.NX * q.N + n] += m[n];
}

// ---- fresh run, NumPartitions == 0, every step with at most ONE node that is no candidate (its
// higher priority node, plan.go:146-154).  S = the sequence of picks the run would make without
// the exclusions (k_fresh_threshold / emit / sort, one element more than steps).  A step whose
// turn falls on its own excluded node h takes the next element instead and leaves h at the head
// of the queue: "pending".  A pending node is taken by the first later step that does not exclude
// it too.  So the state between steps is one bit (is a node pending -- it can only be the previous
// step's excluded node), the step function b' = b ? A_t : B_t with A_t = (e_t == e_{t-1}),
// B_t = (S[t] == e_t) is known per step, and a scan over function composition resolves the run:
//   pick_t = b_t ? (A_t ? S[t + 1] : e_{t-1}) : (B_t ? S[t + 1] : S[t]).
// Exact as long as the pending node does not come up again in S while it waits (its next pick
// would then be due before anybody may take it): the first step where it does is reported and
// the run ends there.
__device__ __forceinline__ int fresh_excluded(const FlatParams& q, int oi) {
    const int32_t* r = q.rec + (size_t)oi * q.RW;
    const int SW = 1 + q.L;
    for (int t = 0; t < q.M; t++) {
        if (!((q.higher_mask >> t) & 1)) continue;
        const int h = r[kRecHead + t * SW];
        if ((h >> 16) != kListAbsent && (h & 0xffff) > 0) return r[kRecHead + t * SW + 1];
    }
    return -1;
}

// k = 1 or 2 picks per step.  With two, a step takes the two smallest of the CURRENT scores (plan.go:171-172, :228-229:
// no commit between its picks), which are the next two elements of S as long as those are two different nodes; a
// pending node x goes first (its score is the smallest there is), then the rest from S.  Step function, with
// B_t = (e_t among the k elements the step would take from S), C_t = (e_t among the k - 1 it would take after x):
//   b' = b ? (A_t ? 1 : C_t) : B_t.
struct FreshStep { unsigned A, B, C; };
__device__ __forceinline__ FreshStep fresh_step(int k, const int32_t* S, int t, int e, int eprev) {
    FreshStep r;
    const int base = k * t;
    r.A = (e >= 0 && e == eprev) ? 1u : 0u;
    r.B = (e >= 0 && (S[base] == e || (k == 2 && S[base + 1] == e))) ? 1u : 0u;
    r.C = (e >= 0 && k == 2 && S[base + 1] == e) ? 1u : 0u;
    return r;
}

// Two launches over G workgroups of 1024 threads, thread g owning steps [g per, g per + per) of the run (round 6; one
// workgroup walked 64 steps per thread twice, 300 us for 65,536 steps):
//   k_fresh_excl_scan  every thread composes its steps' functions; an inclusive scan under composition over the workgroup
//                      leaves each thread's prefix in comp[] and the workgroup's whole function in blk[];
//   k_fresh_excl_apply the state that enters a workgroup is the earlier workgroups' functions applied in turn to "nothing
//                      pending"; a thread's entering state follows from its predecessor's prefix; the steps are replayed.
// A function {0, 1} -> {0, 1} is two bits: f(x) = (f >> x) & 1.
__device__ __forceinline__ int fresh_excl_per(int R, int G) { return (R + G * 1024 - 1) / (G * 1024); }

__global__ __launch_bounds__(1024) void k_fresh_excl_scan(FlatParams q, int beg, int R, const int32_t* S /* [k R + k] */,
                                                          unsigned char* comp_out /* [G * 1024] */, unsigned char* blk /* [G] */) {
    BLANCE_DYN_LDS(lds);
    unsigned char* comp = (unsigned char*)lds;       // [1024] composite step function of a thread's slice: bit x = f(x)
    unsigned char* tmp = comp + 1024;
    const int tid = threadIdx.x, k = q.k;
    const int gid = blockIdx.x * 1024 + tid;
    const int per = fresh_excl_per(R, (int)gridDim.x);
    const long long s0 = (long long)gid * per;
    const int t0 = s0 < R ? (int)s0 : R, t1 = t0 + per < R ? t0 + per : R;
    unsigned f = 2;                                  // identity: f(0) = 0, f(1) = 1
    int eprev = t0 > 0 && t0 < R ? fresh_excluded(q, beg + t0 - 1) : -1;
    for (int t = t0; t < t1; t++) {
        const int e = fresh_excluded(q, beg + t);
        const FreshStep st = fresh_step(k, S, t, e, eprev);
        const unsigned g0 = st.B, g1 = st.A ? 1u : st.C;                 // the step's function: g(0), g(1)
        const unsigned f0 = (f & 1) ? g1 : g0, f1 = (f & 2) ? g1 : g0;
        f = f0 | (f1 << 1);
        eprev = e;
    }
    comp[tid] = (unsigned char)f;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {       // inclusive scan under composition (earlier function first)
        unsigned mine = comp[tid];
        if (tid >= off) {
            const unsigned early = comp[tid - off];
            const unsigned g0 = (mine >> (early & 1)) & 1, g1 = (mine >> ((early >> 1) & 1)) & 1;
            mine = g0 | (g1 << 1);
        }
        tmp[tid] = (unsigned char)mine;
        __syncthreads();
        comp[tid] = tmp[tid];
        __syncthreads();
    }
    comp_out[gid] = comp[tid];
    if (tid == 1023) blk[blockIdx.x] = comp[tid];
}

__global__ __launch_bounds__(1024) void k_fresh_excl_apply(FlatParams q, int beg, int R, const int32_t* S /* [k R + k] */,
                                                           const unsigned char* comp /* [G * 1024] */, const unsigned char* blk,
                                                           int32_t* picks /* [k R] */, int32_t* first_bad) {
    const int tid = threadIdx.x, k = q.k;
    const int gid = blockIdx.x * 1024 + tid;
    const int per = fresh_excl_per(R, (int)gridDim.x);
    const long long s0 = (long long)gid * per;
    const int t0 = s0 < R ? (int)s0 : R, t1 = t0 + per < R ? t0 + per : R;
    unsigned b = 0;                                  // the run starts with nothing pending
    for (int j = 0; j < (int)blockIdx.x; j++) b = ((unsigned)blk[j] >> b) & 1u;
    if (tid > 0) b = ((unsigned)comp[gid - 1] >> b) & 1u;
    int eprev = t0 > 0 && t0 < R ? fresh_excluded(q, beg + t0 - 1) : -1;
    for (int t = t0; t < t1; t++) {
        const int e = fresh_excluded(q, beg + t);
        const FreshStep st = fresh_step(k, S, t, e, eprev);
        const int base = k * t;
        int p0, p1 = -1;
        if (!b) {                                    // the next k elements of S, skipping e
            int i = base;
            if (S[i] == e && e >= 0) i++;
            p0 = S[i++];
            if (k == 2) { if (S[i] == e && e >= 0) i++; p1 = S[i]; }
        } else if (!st.A) {                          // the pending node, then k - 1 elements of S, skipping e
            p0 = eprev;
            if (k == 2) { int i = base + 1; if (S[i] == e && e >= 0) i++; p1 = S[i]; }
        } else {                                     // still excluded: k elements of S behind it
            p0 = S[base + 1];
            if (k == 2) p1 = S[base + 2];
        }
        const unsigned nb = b ? (st.A ? 1u : st.C) : st.B;               // e pending after this step
        picks[base] = p0;
        if (k == 2) picks[base + 1] = p1;
        // not what the reference does if the pending node comes up again, or a node would be taken twice in a step
        bool bad = e >= 0 && (p0 == e || p1 == e);
        if (k == 2 && p0 == p1) bad = true;
        if (nb && S[base + k] == e) bad = true;
        if (bad) atomicMin(first_bad, t);
        b = nb;
        eprev = e;
    }
}

__global__ void k_fresh_hist(int n, const int32_t* picks, int32_t* m) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) atomicAdd(&m[picks[j]], 1);
}

// ---- stable LSD radix sort of (64-bit key, 32-bit value) pairs, 8 bits per pass.
// One wave64 per tile of kSortTile elements; ranks inside a wave come from ballots.
constexpr int kSortTile = 2048;

__device__ __forceinline__ unsigned long long same_digit_lanes(int digit, bool valid) {
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        unsigned long long m = __ballot((digit >> b) & 1);
        peers &= ((digit >> b) & 1) ? m : ~m;
    }
    return peers;
}

// bits in which the keys differ from keys[0]: byte positions that are equal everywhere need no pass
constexpr int kVarbitsPer = 16;          // keys per thread of k_sort_varbits
__global__ void k_sort_varbits(int n, const unsigned long long* keys, unsigned long long* out) {
    const int i0 = (blockIdx.x * blockDim.x + threadIdx.x) * kVarbitsPer;
    const unsigned long long k0 = keys[0];
    unsigned long long v = 0ull;
#pragma unroll
    for (int j = 0; j < kVarbitsPer; j++) v |= i0 + j < n ? (keys[i0 + j] ^ k0) : 0ull;
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lo |= (unsigned)__shfl_xor((int)lo, off, 64);
        hi |= (unsigned)__shfl_xor((int)hi, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        const unsigned long long bits = ((unsigned long long)hi << 32) | lo;
        // most waves bring nothing new: look before paying for a contended atomic
        if (bits & ~*(volatile unsigned long long*)out) atomicOr(out, bits);
    }
}

__global__ __launch_bounds__(64) void k_sort_hist(int n, int shift, const unsigned long long* keys, int n_tiles,
                                                  int32_t* hist /* [256][n_tiles] */) {
    BLANCE_DYN_LDS(lds);
    int* cnt = (int*)lds;                            // [256]
    const int lane = threadIdx.x, tile = blockIdx.x;
    for (int i = lane; i < 256; i += 64) cnt[i] = 0;
    __syncthreads();
    int beg = tile * kSortTile, end = beg + kSortTile < n ? beg + kSortTile : n;
    for (int base = beg; base < end; base += 64) {
        int i = base + lane;
        bool valid = i < end;
        int digit = valid ? (int)((keys[i] >> shift) & 0xff) : 0;
        unsigned long long peers = same_digit_lanes(digit, valid);
        if (valid && (peers & ((1ull << lane) - 1)) == 0) cnt[digit] += __popcll(peers);   // lowest peer adds
        __syncthreads();
    }
    for (int i = lane; i < 256; i += 64) hist[(size_t)i * n_tiles + tile] = cnt[i];
}

__global__ __launch_bounds__(64) void k_sort_scatter(int n, int shift, const unsigned long long* keys_in,
                                                     const int32_t* vals_in, unsigned long long* keys_out,
                                                     int32_t* vals_out, int n_tiles, const int32_t* offsets) {
    BLANCE_DYN_LDS(lds);
    int* pos = (int*)lds;                            // [256] next output slot per digit
    const int lane = threadIdx.x, tile = blockIdx.x;
    for (int i = lane; i < 256; i += 64) pos[i] = offsets[(size_t)i * n_tiles + tile];
    __syncthreads();
    int beg = tile * kSortTile, end = beg + kSortTile < n ? beg + kSortTile : n;
    for (int base = beg; base < end; base += 64) {
        int i = base + lane;
        bool valid = i < end;
        unsigned long long key = valid ? keys_in[i] : 0;
        int digit = valid ? (int)((key >> shift) & 0xff) : 0;
        unsigned long long peers = same_digit_lanes(digit, valid);
        unsigned long long lower = peers & ((1ull << lane) - 1);
        int dst = 0;
        if (valid) dst = pos[digit] + __popcll(lower);
        __syncthreads();
        if (valid && lower == 0) pos[digit] += __popcll(peers);
        __syncthreads();
        if (valid) { keys_out[dst] = key; vals_out[dst] = vals_in[i]; }
    }
}


}  // namespace blance
