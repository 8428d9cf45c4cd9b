// Data-parallel kernels of a sweep: live lists, counts, categories, stable partitions, records, convergence.
// Part of blance_hip.hip (one translation unit); see DESIGN.md section 4.
#pragma once

namespace blance {

// ============================================================================
// Data-parallel kernels around the pass
// ============================================================================

struct DevProblem {   // device pointers + sizes shared by the elementwise kernels
    int32_t N, NX, M, L, P;
    int32_t weights_nil;
    const uint8_t* node_removed;   // view of this sweep (all zero after sweep 1)
    const uint8_t* node_added;
    const int32_t* part_weight;
    const uint8_t* part_has_weight;
    int32_t* live; int32_t* live_len; uint8_t* live_kind;
    int32_t* prv;  int32_t* prv_len;  uint8_t* prv_kind;
    uint8_t* in_prev; uint8_t* never_equal;
};

// ---- blance_upload: the O(P) part of blance_validate and the sizes the host needs, computed where the arrays have just
// landed (round 5: the host's loops over two million offsets and a million partitions were half of an upload).
// Result block: [0] error bits, [1] longest list (L), [2] partitions not in prevMap, [3] INT_MAX - (4 i + check) of the first
// list i whose shape is wrong (0: none); 64-bit words from byte 16:
// [2] result capacity (sum of max(k, len)), [3] sum of |weights|, [4] |weight| * prev entries.
constexpr int kVErrMonotone = 1, kVErrKind = 2, kVErrLong = 4, kVErrOrder = 8, kVErrAssignId = 16, kVErrPrevId = 32;
struct ValidateParams {
    int32_t P, M, weights_nil;
    int32_t k[kMaxStates];
    const int32_t* a_off; const uint8_t* a_kind;
    const int32_t* p_off; const uint8_t* p_kind;
    const int32_t* part_order; const int32_t* part_weight; const uint8_t* part_has_weight; const uint8_t* part_in_prev;
    uint32_t* seen;                // [(P + 31) / 32] zeroed: part_order as a permutation
    int32_t* res;
};
// (a grid of at most 1,024 workgroups strides over the lists; a workgroup folds its four waves in LDS and makes one atomic
// per word -- one atomic per WAVE was 80 K atomics on five addresses, 0.8 ms of a 1.7 ms upload)
__global__ __launch_bounds__(256) void k_validate_parts(ValidateParams v) {
    BLANCE_DYN_LDS(lds);
    unsigned long long (*red)[8] = (unsigned long long (*)[8])lds;      // [4 waves][8]
    const long long PM = (long long)v.P * v.M;
    const long long n = PM > v.P ? PM : v.P, stride = (long long)gridDim.x * blockDim.x;
    int err = 0, L = 0, fresh = 0, first = 0;
    unsigned long long cap = 0, sumw = 0, aprev = 0;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
        if (idx < PM) {
            const int a = v.a_off[idx + 1] - v.a_off[idx], b = v.p_off[idx + 1] - v.p_off[idx];
            // the host's loop reports the first (list, check) that fails: INT_MAX - (4 idx + check), largest wins
            int f = 0;
            if (a > 0xffff || b > 0xffff) { err |= kVErrLong; f = INT_MAX - (int)(4 * idx + 2); }
            if (v.a_kind[idx] > kListSet || v.p_kind[idx] > kListSet) { err |= kVErrKind; f = INT_MAX - (int)(4 * idx + 1); }
            if (a < 0 || b < 0) { err |= kVErrMonotone; f = INT_MAX - (int)(4 * idx); }
            first = f > first ? f : first;
            const int l = a > b ? a : b;
            L = l > L ? l : L;
            const int k = v.k[idx % v.M];
            if (a >= 0) cap += (unsigned long long)(a > k ? a : k);
        }
        if (idx < v.P) {
            const int p = (int)idx;
            const int o = v.part_order[p];
            if (o < 0 || o >= v.P) err |= kVErrOrder;
            else if (atomicOr((int*)v.seen + (o >> 5), (int)(1u << (o & 31))) & (int)(1u << (o & 31))) err |= kVErrOrder;
            long long w = (!v.weights_nil && v.part_has_weight[p]) ? (long long)v.part_weight[p] : 1;
            if (w < 0) w = -w;
            sumw += (unsigned long long)w;
            if (v.part_in_prev[p]) {
                long long cnt = (long long)v.p_off[(long long)(p + 1) * v.M] - v.p_off[(long long)p * v.M];
                if (cnt < 0) cnt = 0;                   // (reported as not monotone)
                aprev += (unsigned long long)(w * cnt);
            } else fresh++;
        }
    }
    // wave totals, the workgroup's four waves through LDS, then one atomic per word
    for (int o = 32; o; o >>= 1) {
        err |= __shfl_xor(err, o, 64);
        const int f2 = __shfl_xor(first, o, 64);
        first = f2 > first ? f2 : first;
        const int l2 = __shfl_xor(L, o, 64);
        L = l2 > L ? l2 : L;
        fresh += __shfl_xor(fresh, o, 64);
        cap += ((unsigned long long)(unsigned)__shfl_xor((int)(unsigned)(cap >> 32), o, 64) << 32) | (unsigned)__shfl_xor((int)(unsigned)cap, o, 64);
        sumw += ((unsigned long long)(unsigned)__shfl_xor((int)(unsigned)(sumw >> 32), o, 64) << 32) | (unsigned)__shfl_xor((int)(unsigned)sumw, o, 64);
        aprev += ((unsigned long long)(unsigned)__shfl_xor((int)(unsigned)(aprev >> 32), o, 64) << 32) | (unsigned)__shfl_xor((int)(unsigned)aprev, o, 64);
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[wv][0] = (unsigned long long)(unsigned)err; red[wv][1] = (unsigned long long)(unsigned)first; red[wv][2] = (unsigned long long)(unsigned)L;
        red[wv][3] = (unsigned long long)(unsigned)fresh; red[wv][4] = cap; red[wv][5] = sumw; red[wv][6] = aprev;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (int)(blockDim.x >> 6);
        for (int w2 = 1; w2 < nw; w2++) {
            err |= (int)red[w2][0];
            first = (int)red[w2][1] > first ? (int)red[w2][1] : first;
            L = (int)red[w2][2] > L ? (int)red[w2][2] : L;
            fresh += (int)red[w2][3];
            cap += red[w2][4]; sumw += red[w2][5]; aprev += red[w2][6];
        }
        unsigned long long* r64 = (unsigned long long*)(v.res + 4);
        if (err) atomicOr(v.res, err);
        if (first) atomicMax(v.res + 3, first);
        if (L) atomicMax(v.res + 1, L);
        if (fresh) atomicAdd(v.res + 2, fresh);
        if (cap) atomicAdd(r64, cap);
        if (sumw) atomicAdd(r64 + 1, sumw);
        if (aprev) atomicAdd(r64 + 2, aprev);
    }
}
// node ids of a CSR's payload in [0, NX)
__global__ void k_validate_ids(long long n, int NX, const int32_t* ids, int bit, int32_t* res) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool bad = i < n && (ids[i] < 0 || ids[i] >= NX);
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr(res, bit);
}

// nextPartitions = copy of partitionsToAssign minus nodesToRemove (plan.go:83-88)
__global__ void k_live_init(DevProblem d, const int32_t* a_off, const int32_t* a_nodes,
                            const uint8_t* a_kind, const int32_t* p_off, const int32_t* p_nodes,
                            const uint8_t* p_kind) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= d.P * d.M) return;
    int len = 0;
    for (int i = a_off[idx]; i < a_off[idx + 1]; i++) {
        int n = a_nodes[i];
        if (!d.node_removed[n]) d.live[(size_t)idx * d.L + len++] = n;
    }
    d.live_len[idx] = len;
    d.live_kind[idx] = a_kind[idx] == kListAbsent ? kListAbsent : kListSet;
    len = 0;
    for (int i = p_off[idx]; i < p_off[idx + 1]; i++) d.prv[(size_t)idx * d.L + len++] = p_nodes[i];
    d.prv_len[idx] = len;
    d.prv_kind[idx] = p_kind[idx];
}

// the two per-partition byte flags a plan starts from (in prevMap / never equal to its prevMap entry)
__global__ void k_flags_init(int P, const uint8_t* in_prev0, const uint8_t* never0, uint8_t* in_prev, uint8_t* never) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < P) { in_prev[p] = in_prev0[p]; never[p] = never0[p]; }
}

// sweeps >= 2: every present key is a non-nil slice again (plan.go:418)
__global__ void k_live_refresh(DevProblem d) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= d.P * d.M) return;
    if (d.live_kind[idx] != kListAbsent) d.live_kind[idx] = kListSet;
}

// countStateNodes (plan.go:374-399): extra loads ...
__global__ void k_count_loads(int n_loads, int NX, int later_sweep, const int32_t* st, const int32_t* nd,
                              const int32_t* wt, const uint8_t* first_only, int32_t* cnt) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_loads) return;
    if (later_sweep && first_only[i]) return;
    atomicAdd(&cnt[st[i] * NX + nd[i]], wt[i]);
}

// ... and the prevMap view of the partitions being assigned
__global__ void k_count_prev(DevProblem d, int32_t* cnt) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= d.P * d.M) return;
    int p = idx / d.M, m = idx % d.M;
    if (!d.in_prev[p]) return;
    int w = (!d.weights_nil && d.part_has_weight[p]) ? d.part_weight[p] : 1;
    for (int i = 0; i < d.prv_len[idx]; i++) atomicAdd(&cnt[m * d.NX + d.prv[(size_t)idx * d.L + i]], w);
}

// partitionSorter category (plan.go:542-561)
__global__ void k_category(DevProblem d, int m, int any_removed, int add_nil, uint8_t* cat) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= d.P) return;
    int cv = 2;
    bool is0 = false;
    if (any_removed && d.in_prev[p]) {
        int idx = p * d.M + m;
        if (d.prv_kind[idx] == kListSet)
            for (int i = 0; i < d.prv_len[idx]; i++)
                if (d.node_removed[d.prv[(size_t)idx * d.L + i]]) { is0 = true; break; }
    }
    if (is0) cv = 0;
    else if (!add_nil) {
        bool hit = false;
        for (int t = 0; t < d.M && !hit; t++) {
            int idx = p * d.M + t;
            if (d.live_kind[idx] == kListAbsent) continue;
            for (int i = 0; i < d.live_len[idx]; i++)
                if (d.node_added[d.live[(size_t)idx * d.L + i]]) { hit = true; break; }
        }
        if (!hit) cv = 1;
    }
    cat[p] = (uint8_t)cv;
}

// Stable partition of a sequence by a small key (the per-pass category of
// partitionSorter, plan.go:519-562; the region of a step): per-chunk bucket
// counts -> exclusive scan (bucket major) -> stable scatter.  One wave64 per
// chunk of kPartChunk elements; ranks inside a round of 64 come from ballots.
constexpr int kPartChunk = 1024;

__device__ __forceinline__ int part_key(const int32_t* key32, const uint8_t* key8, const int32_t* index, int i) {
    int j = index ? index[i] : i;
    return key8 ? (int)key8[j] : key32[j];
}

__global__ __launch_bounds__(64) void k_part_count(int n, const int32_t* key32, const uint8_t* key8,
                                                   const int32_t* index, int n_chunks, int B,
                                                   int32_t* counts /* [B][n_chunks] */) {
    BLANCE_DYN_LDS(lds);
    int* hist = (int*)lds;                           // [B]
    const int lane = threadIdx.x, chunk = blockIdx.x;
    for (int i = lane; i < B; i += 64) hist[i] = 0;
    __syncthreads();
    int beg = chunk * kPartChunk, end = beg + kPartChunk < n ? beg + kPartChunk : n;
    for (int base = beg; base < end; base += 64) {
        int i = base + lane;
        if (i < end) atomicAdd(&hist[part_key(key32, key8, index, i)], 1);
    }
    __syncthreads();
    for (int i = lane; i < B; i += 64) counts[(size_t)i * n_chunks + chunk] = hist[i];
}

__global__ __launch_bounds__(64) void k_part_scatter(int n, const int32_t* key32, const uint8_t* key8,
                                                     const int32_t* index, const int32_t* values, int n_chunks,
                                                     int B, int nbits, const int32_t* offsets, int32_t* out,
                                                     int32_t* out_index) {
    BLANCE_DYN_LDS(lds);
    int* pos = (int*)lds;                            // [B] next output slot per bucket
    const int lane = threadIdx.x, chunk = blockIdx.x;
    for (int i = lane; i < B; i += 64) pos[i] = offsets[(size_t)i * n_chunks + chunk];
    __syncthreads();
    int beg = chunk * kPartChunk, end = beg + kPartChunk < n ? beg + kPartChunk : n;
    for (int base = beg; base < end; base += 64) {
        int i = base + lane;
        bool valid = i < end;
        int key = valid ? part_key(key32, key8, index, i) : 0;
        unsigned long long peers = __ballot(valid);
        for (int bit = 0; bit < nbits; bit++) {
            unsigned long long m = __ballot((key >> bit) & 1);
            peers &= ((key >> bit) & 1) ? m : ~m;
        }
        unsigned long long lower = peers & ((1ull << lane) - 1);
        int dst = valid ? pos[key] + __popcll(lower) : 0;
        __syncthreads();
        if (valid && lower == 0) pos[key] += __popcll(peers);
        __syncthreads();
        if (valid) {
            out[dst] = values ? values[i] : i;
            if (out_index) out_index[dst] = i;
        }
    }
}

// Region of every step of the pass (the region of its top priority node), or
// flags[0] if a step has none.  Nodes the partition holds in this state outside
// that region leave the state whatever the step decides (plan.go:290-293): they
// become events for the chains that own them (n_ev counts them per step; flags[7]
// says there are any; flags[6] is raised if some lie in no region at all --
// k_chain_orphans un-counts those).
__global__ void k_chain_classify(DevProblem d, int m, int top_state, const int32_t* order,
                                 const int32_t* node_region, int32_t* regid, int32_t* n_ev, int32_t* flags) {
    int oi = blockIdx.x * blockDim.x + threadIdx.x;
    if (oi > d.P) return;
    if (oi == d.P) { n_ev[oi] = 0; return; }
    int p = order[oi];
    int idxT = p * d.M + top_state;
    int top = (d.live_kind[idxT] != kListAbsent && d.live_len[idxT] > 0) ? d.live[(size_t)idxT * d.L] : -1;
    int rg = top >= 0 ? node_region[top] : -1;
    int ne = 0;
    if (rg >= 0) {
        int idx = p * d.M + m;
        if (d.live_kind[idx] != kListAbsent)
            for (int i = 0; i < d.live_len[idx]; i++) {
                int r2 = node_region[d.live[(size_t)idx * d.L + i]];
                if (r2 == rg) continue;
                if (r2 >= 0) ne++; else flags[6] = 1;
                flags[7] = 1;                      // the pass has nodes outside their partition's region
            }
    }
    if (rg < 0) { flags[0] = 1; rg = 0; }
    regid[oi] = rg;
    n_ev[oi] = ne;
}

// the events of every step, in pass order, at the slots an exclusive scan of n_ev gave
__global__ void k_chain_ev_fill(DevProblem d, int m, int top_state, const int32_t* order, const int32_t* node_region,
                                const int32_t* reg_lo, const int32_t* node_leaf_pos, const int32_t* regid,
                                const int32_t* ev_pos, int32_t* ev_key, int32_t* ev_oi, int32_t* ev_leaf, int32_t* ev_w) {
    int oi = blockIdx.x * blockDim.x + threadIdx.x;
    if (oi >= d.P) return;
    if (ev_pos[oi + 1] == ev_pos[oi]) return;
    int p = order[oi];
    int w = (!d.weights_nil && d.part_has_weight[p]) ? d.part_weight[p] : 1;
    int idx = p * d.M + m, rg = regid[oi], e = ev_pos[oi];
    for (int i = 0; i < d.live_len[idx]; i++) {
        int x = d.live[(size_t)idx * d.L + i];
        int r2 = node_region[x];
        if (r2 == rg || r2 < 0) continue;
        ev_key[e] = r2; ev_oi[e] = oi; ev_leaf[e] = node_leaf_pos[x] - reg_lo[r2]; ev_w[e] = w;
        e++;
    }
}

// nodes of this state that lie in no region: nobody reads their counters during the pass
__global__ void k_chain_orphans(DevProblem d, int m, int top_state, const int32_t* order, const int32_t* node_region,
                                int32_t* cnt) {
    int oi = blockIdx.x * blockDim.x + threadIdx.x;
    if (oi >= d.P) return;
    int p = order[oi];
    int w = (!d.weights_nil && d.part_has_weight[p]) ? d.part_weight[p] : 1;
    int idx = p * d.M + m;
    if (d.live_kind[idx] == kListAbsent) return;
    for (int i = 0; i < d.live_len[idx]; i++) {
        int x = d.live[(size_t)idx * d.L + i];
        if (node_region[x] < 0) atomicAdd(&cnt[m * d.NX + x], -w);
    }
}

// Compact chain records (layout: blance_kernels.h): the step's nodes as leaf
// indices local to its region.  Steps the chain kernel cannot represent raise flags[0].
// one step's compact record (24 words at r); returns the global leaf index of the step's top priority node (0 if it has
// none inside a region: flags[0] is raised for such a step)
__device__ __forceinline__ int gather_chain_record(const DevProblem& d, int m, int top_state, int higher_mask, int p, int oi,
                                                    const int32_t* state_stickiness, const uint8_t* state_has_stickiness,
                                                    const int32_t* node_leaf_pos, const int32_t* node_region,
                                                    const int32_t* reg_lo, const int32_t* leaf_cls, const int32_t* cls_size,
                                                    int flat, int32_t* r, int32_t* flags) {
    for (int j = 0; j < kCW; j++) r[j] = -1;
    int w = 1;
    double stick = 1.5;
    if (!d.weights_nil) {
        if (d.part_has_weight[p]) { w = d.part_weight[p]; stick = (double)w; }
        else if (state_has_stickiness[m]) stick = (double)state_stickiness[m];
    }
    r[0] = oi; r[1] = w; r[2] = __double2loint(stick); r[3] = __double2hiint(stick);
    int idxT = p * d.M + top_state;
    int top = (d.live_kind[idxT] != kListAbsent && d.live_len[idxT] > 0) ? d.live[(size_t)idxT * d.L] : -1;
    int rg = flat ? 0 : (top >= 0 ? node_region[top] : -1);
    if (rg < 0) { flags[0] = 1; r[4] = 0; r[5] = 0; r[6] = -1; return 0; }
    const int lo = reg_lo[rg];
    if (flat) {
        r[4] = top >= 0 ? top : d.NX;              // the "" row when there is no top priority node
        r[6] = -1;                                 // no anchor: nothing is excluded in the first slot
        r[23] = 0;
    } else {
        r[4] = node_leaf_pos[top] - lo;
        r[6] = leaf_cls[node_leaf_pos[top]];
        r[23] = r[6] >= 0 ? cls_size[lo + r[6]] : 0;
    }
    bool bad = false;
    int own_nodes[kChainOwn];
    int n_own = 0, n_all = 0, n_h = 0, n_low = 0, present = 0, remote = 0;
    int idx = p * d.M + m;
    if (d.live_kind[idx] != kListAbsent) {
        present = 1;
        for (int j = 0; j < d.live_len[idx]; j++) {
            int x = d.live[(size_t)idx * d.L + j];
            if (n_all >= kChainOwn) { bad = true; break; }
            own_nodes[n_all++] = x;
            if (node_region[x] != rg) { remote = 1; continue; }     // leaves by an event (or as an orphan)
            r[kCOwn + n_own++] = node_leaf_pos[x] - lo;
        }
    }
    for (int t = 0; t < d.M && !bad; t++) {
        if (t == m) continue;
        int ix = p * d.M + t;
        if (d.live_kind[ix] == kListAbsent) continue;
        const bool higher = (higher_mask >> t) & 1;
        for (int j = 0; j < d.live_len[ix]; j++) {
            int x = d.live[(size_t)ix * d.L + j];
            for (int e = 0; e < n_all; e++) if (own_nodes[e] == x) bad = true;   // a node held in two states
            if (node_region[x] != rg) continue;      // never a candidate of this region's chain
            int loc = node_leaf_pos[x] - lo;
            if (higher) {
                // the top priority node's exclude class is excluded in every slot anyway
                if (r[6] >= 0 && leaf_cls[node_leaf_pos[x]] == r[6]) continue;
                if (n_h >= kChainHigh) { bad = true; break; }
                r[kCHigh + n_h++] = loc;
            } else {
                if (n_low >= kChainLow) { bad = true; break; }
                r[kCLow + n_low] = loc;
                r[kCLowState + n_low++] = t;
            }
        }
    }
    r[5] = n_own | (n_h << 8) | (n_low << 16) | (present << 24) | (remote << 25);
    if (bad) flags[0] = 1;
    return flat ? 0 : lo + r[4];
}

__global__ void k_gather_chain(DevProblem d, int m, int top_state, int higher_mask, const int32_t* chain_order,
                               const int32_t* chain_oi,
                               const int32_t* state_stickiness, const uint8_t* state_has_stickiness,
                               const int32_t* node_leaf_pos, const int32_t* node_region, const int32_t* reg_lo,
                               const int32_t* leaf_cls, const int32_t* cls_size, int flat, int32_t* crec,
                               int32_t* flags, int32_t* topkey /* or null: global leaf of the step's top priority node */) {
    // a thread builds its record in LDS (row stride kCW + 1: no bank conflicts); the workgroup then writes its
    // 256 records as one contiguous block -- per-thread 24-word rows written straight to HBM cost 3.7 times
    // their bytes in write traffic (rocprofv3 WRITE_SIZE, round 2)
    BLANCE_DYN_LDS(lds);
    const int tid = threadIdx.x;
    const int i = blockIdx.x * blockDim.x + tid;
    int32_t* r = (int32_t*)lds + tid * (kCW + 1);
    if (i < d.P) {
        const int tl = gather_chain_record(d, m, top_state, higher_mask, chain_order[i], chain_oi ? chain_oi[i] : i, state_stickiness,
                                           state_has_stickiness, node_leaf_pos, node_region, reg_lo, leaf_cls, cls_size, flat, r, flags);
        if (topkey) topkey[i] = tl;
    }
    __syncthreads();
    const int first = blockIdx.x * blockDim.x;
    const int n_here = d.P - first < (int)blockDim.x ? d.P - first : (int)blockDim.x;
    for (int j = tid; j < n_here * kCW; j += blockDim.x)
        crec[(size_t)first * kCW + j] = ((const int32_t*)lds)[(j / kCW) * (kCW + 1) + j % kCW];
}


// Exclusive scan of n ints.  One tile = 8192 elements of one workgroup of 1024 threads, 8 contiguous
// per thread (coalesced).  Short arrays: k_scan_excl, one workgroup carrying across its tiles.  Long
// arrays, three launches that fill the chip: k_scan_tile_sums (a total per tile) -> k_scan_excl over
// the totals -> k_scan_apply (every tile scans itself from its carry).
constexpr int kScanTile = 8192;

// scans tile [base, base + 8192) in place from `carry` (write = true) and returns the tile's total
__device__ __forceinline__ int scan_tile(int n, int32_t* data, int base, int carry, bool write, int* wsum) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int v[8];
    int sum = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        int i = base + tid * 8 + j;
        v[j] = i < n ? data[i] : 0;
        sum += v[j];
    }
    int incl = sum;                          // inclusive scan of the thread sums inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) { int x = wsum[w]; if (w < wave) wbase += x; total += x; }
    if (write) {
        int acc = carry + wbase + incl - sum;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            int i = base + tid * 8 + j;
            if (i < n) data[i] = acc;
            acc += v[j];
        }
    }
    __syncthreads();
    return total;
}

__global__ __launch_bounds__(1024) void k_scan_excl(int n, int32_t* data) {
    BLANCE_DYN_LDS(lds);
    int carry = 0;
    for (int base = 0; base < n; base += kScanTile) carry += scan_tile(n, data, base, carry, true, (int*)lds);
}

__global__ __launch_bounds__(1024) void k_scan_tile_sums(int n, int32_t* data, int32_t* sums) {
    BLANCE_DYN_LDS(lds);
    const int total = scan_tile(n, data, blockIdx.x * kScanTile, 0, false, (int*)lds);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void k_scan_apply(int n, int32_t* data, const int32_t* sums) {
    BLANCE_DYN_LDS(lds);
    (void)scan_tile(n, data, blockIdx.x * kScanTile, sums[blockIdx.x], true, (int*)lds);
}

__global__ void k_region_offsets(int B, int n_chunks, int P, const int32_t* offsets, int32_t* reg_off) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > B) return;
    reg_off[b] = b == B ? P : offsets[(size_t)b * n_chunks];
}

// element-wise difference / sum of int32 vectors (the load change a rank of a sharded plan contributes)
__global__ void k_vec_sub(int n, const int32_t* a, const int32_t* b, int32_t* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] - b[i];
}
__global__ void k_vec_add(int n, const int32_t* a, const int32_t* b, int32_t* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

// Several small fills and one copy in ONE launch: the driver used to enqueue each as a hipMemsetAsync / hipMemcpyAsync of its
// own (a fill kernel of the runtime per call, 25 of them per PlanNextMap at config 3).  All int32 words.
struct FillCopy {
    int32_t* z[4];          // zero z[i][0 .. zn[i])
    int32_t zn[4];
    int32_t* cd;            // cd[0 .. cn) = cs[0 .. cn)
    const int32_t* cs;
    int32_t cn;
};
__global__ void k_fill_copy(FillCopy a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; j++) if (i < a.zn[j]) a.z[j][i] = 0;
    if (i < a.cn) a.cd[i] = a.cs[i];
}

// ---- result as CSR on the device (blance_download): list lengths -> exclusive scan -> gather
__global__ void k_result_len(DevProblem d, int32_t* len_out /* [P*M + 1] */) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int PM = d.P * d.M;
    if (idx > PM) return;
    len_out[idx] = (idx < PM && d.live_kind[idx] != kListAbsent) ? d.live_len[idx] : 0;
}
__global__ void k_result_gather(DevProblem d, const int32_t* off, int32_t* nodes_out) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= d.P * d.M) return;
    const int n = off[idx + 1] - off[idx];
    for (int i = 0; i < n; i++) nodes_out[off[idx] + i] = d.live[(size_t)idx * d.L + i];
}

// Step records in pass order: what findBestNodes needs to know about its partition.  A thread builds its record in
// LDS (row stride RW | 1: no bank conflicts), the workgroup writes its 256 records as one contiguous block (per-thread
// rows written straight to HBM cost 3.6 times their bytes in write traffic, rocprofv3 WRITE_SIZE).
// (row_count, or null: the flat bulk driver's bound of a nodeToNodeCounts row -- the steps of the pass per top priority node,
// k_flat.h: k_flat_row_count -- counted here, where the record is made anyway; row NX is the "" row)
__global__ void k_gather(DevProblem d, int m, int top_state, int RW, const int32_t* order,
                         const int32_t* state_stickiness, const uint8_t* state_has_stickiness,
                         int32_t* rec, int32_t* row_count, int NX) {
    BLANCE_DYN_LDS(lds);
    const int tid = threadIdx.x, ST = RW | 1;
    const int oi = blockIdx.x * blockDim.x + tid;
    int32_t* r = (int32_t*)lds + tid * ST;
    if (row_count) {
        int top = -1;
        if (oi < d.P) {
            const int idxT = order[oi] * d.M + top_state;
            if (d.live_kind[idxT] != kListAbsent && d.live_len[idxT] > 0) top = d.live[(size_t)idxT * d.L];
            if (top >= 0) atomicAdd(&row_count[top], 1);
        }
        const unsigned long long none = __ballot(oi < d.P && top < 0);      // the "" row: one atomic per wave
        if (none && (int)(tid & 63) == __ffsll((long long)none) - 1) atomicAdd(&row_count[NX], __popcll(none));
    }
    if (oi < d.P) {
        int p = order[oi];
        int w = 1;                                         // plan.go:269-275
        double stick = 1.5;                                // plan.go:104-115
        if (!d.weights_nil) {
            if (d.part_has_weight[p]) { w = d.part_weight[p]; stick = (double)w; }
            else if (state_has_stickiness[m]) stick = (double)state_stickiness[m];
        }
        r[0] = p; r[1] = w;
        r[2] = __double2loint(stick); r[3] = __double2hiint(stick);
        for (int t = 0; t < d.M; t++) {
            int idx = p * d.M + t;
            int32_t* rs = r + kRecHead + t * (1 + d.L);
            int len = d.live_kind[idx] == kListAbsent ? 0 : d.live_len[idx];
            rs[0] = len | ((int)d.live_kind[idx] << 16);
            for (int i = 0; i < d.L; i++) rs[1 + i] = i < len ? d.live[(size_t)idx * d.L + i] : -1;
        }
    }
    __syncthreads();
    const int first = blockIdx.x * blockDim.x;
    const int n_here = d.P - first < (int)blockDim.x ? d.P - first : (int)blockDim.x;
    for (int j = tid; j < n_here * RW; j += blockDim.x)
        rec[(size_t)first * RW + j] = ((const int32_t*)lds)[(j / RW) * ST + j % RW];
}

// k_pass_queue's bit maps ("is nodeToNodeCounts[row][n] not zero", BW words per row) rebuilt from the matrix: kernels
// other than k_pass_queue bump the matrix only, so the host runs this before k_pass_queue follows one of them in a pass.
// One wave64 per row (four to a workgroup): a load instruction reads 64 consecutive entries (256 B, coalesced), one ballot
// packs them into 64 bits, lane i keeps the ballot of the row's i-th group of 64 entries and writes it as two words --
// 16 loads in flight per lane.  (Round 4 read 32 consecutive entries per THREAD: every load of a wave touched 64 cache
// lines, 10.8 x the matrix fetched per launch by the counters.)
__global__ __launch_bounds__(256) void k_ntn_bits(int N, int rows, int BW, const int32_t* ntn, uint32_t* bits) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;                           // (a whole wave leaves: the ballots below see full waves only)
    const int32_t* r = ntn + (size_t)row * N;
    uint32_t* o = bits + (size_t)row * BW;
    for (int g0 = 0; g0 < BW; g0 += 128) {             // 64 groups of 64 entries = 128 words per round
        unsigned long long mine = 0;
        for (int i0 = 0; i0 < 64; i0 += 16) {
            int v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const long long n = ((long long)(g0 >> 1) + i0 + u) * 64 + lane;
                v[u] = n < N ? r[n] : 0;
            }
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const unsigned long long b = __ballot(v[u] != 0);
                if (lane == i0 + u) mine = b;
            }
        }
        const int w = g0 + 2 * lane;
        if (w < BW) o[w] = (uint32_t)mine;
        if (w + 1 < BW) o[w + 1] = (uint32_t)(mine >> 32);
    }
}

// Apply the pass's choices to the live lists (plan.go:290-299); list edits only
// touch the step's own partition, so this runs in parallel after the pass.
__device__ __forceinline__ bool gate_closed(const Gate& g) {
    uint32_t mk = g.mask;
    bool closed = false;
    while (mk) {
        const int b = __ffsll((long long)mk) - 1;
        mk &= mk - 1;
        closed |= g.flags[b] != 0;
    }
    return closed;
}

__global__ void k_scatter(DevProblem d, int m, int OW, const int32_t* order, const int32_t* out, Gate gate) {
    int oi = blockIdx.x * blockDim.x + threadIdx.x;
    if (oi >= d.P) return;
    if (gate_closed(gate)) return;                   // (the pass did not stand: the host runs it again and scatters then)
    int p = order[oi];
    const int32_t* o = out + (size_t)oi * OW;
    int n_out = o[0] & 0xffff, is_nil = o[0] >> 16;
    // the partition's list in this state as the pass saw it: the pass itself only wrote `out`
    const int idx_m = p * d.M + m;
    // (that list is only rewritten at the end of this function)
    const int n_old = d.live_kind[idx_m] == kListAbsent ? 0 : d.live_len[idx_m];
    const int32_t* old_l = d.live + (size_t)idx_m * d.L;
    for (int t = 0; t < d.M; t++) {
        int idx = p * d.M + t;
        if (t == m) continue;
        if (d.live_kind[idx] == kListAbsent) continue;
        int len = d.live_len[idx], w = 0;
        int32_t* lst = d.live + (size_t)idx * d.L;
        for (int i = 0; i < len; i++) {
            int x = lst[i];
            bool rm = false;
            for (int j = 0; j < n_old; j++) rm |= old_l[j] == x;
            for (int j = 0; j < n_out; j++) rm |= o[1 + j] == x;
            if (!rm) lst[w++] = x;
        }
        d.live_len[idx] = w;
        d.live_kind[idx] = kListSet;
    }
    int idx = p * d.M + m;
    for (int j = 0; j < n_out; j++) d.live[(size_t)idx * d.L + j] = o[1 + j];
    d.live_len[idx] = n_out;
    d.live_kind[idx] = is_nil ? kListNil : kListSet;
}

// Convergence test (plan.go:36-45) fused with the write-back prevMap[name] =
// partitionsToAssign[name] = nextMap[name] (plan.go:49-52).
__global__ void k_converge(DevProblem d, int32_t* not_match, Gate gate) {
    if (gate_closed(gate)) return;                   // (uniform: the sweep's last pass did not stand, the host comes back)
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = p < d.P;
    if (!in_range) p = d.P - 1;                      // keep the wave whole for the ballot below
    bool diff = !d.in_prev[p] || d.never_equal[p];
    for (int m = 0; m < d.M && in_range; m++) {
        int idx = p * d.M + m;
        int len = d.live_len[idx];
        if (d.live_kind[idx] != d.prv_kind[idx] || len != d.prv_len[idx]) diff = true;
        for (int i = 0; i < len; i++) {
            int x = d.live[(size_t)idx * d.L + i];
            if (!diff && d.prv[(size_t)idx * d.L + i] != x) diff = true;
            d.prv[(size_t)idx * d.L + i] = x;
        }
        d.prv_len[idx] = len;
        d.prv_kind[idx] = d.live_kind[idx];
    }
    if (in_range) {
        d.in_prev[p] = 1;
        d.never_equal[p] = 0;
    }
    // one word for the whole sweep: one atomic per wave, and none once it is set (in a first sweep
    // every partition differs)
    const unsigned long long dm = __ballot(in_range && diff);
    if (dm && (int)(threadIdx.x & 63) == __ffsll((long long)dm) - 1 && *(volatile int32_t*)not_match == 0) atomicOr(not_match, 1);
}


// ---------------------------------------------------------------------------
// CalcPartitionMoves (moves.go:41-136) for every partition: one thread per
// partition, lists are a handful of node ids.  Moves are written to the
// partition's slice of the output (capacity = its list entries), n_moves[p] says
// how many; the host compacts.
// ---------------------------------------------------------------------------
struct MovesParams {
    int32_t P, M, favor_min_nodes;
    const int32_t* beg_off; const int32_t* beg_nodes;
    const int32_t* end_off; const int32_t* end_nodes;
    int32_t* op_node; int32_t* op_state; int32_t* op_kind; int32_t* n_moves;
};

__device__ __forceinline__ bool moves_in(const int32_t* off, const int32_t* nodes, int idx, int x) {
    for (int i = off[idx]; i < off[idx + 1]; i++) if (nodes[i] == x) return true;
    return false;
}

__global__ void k_calc_moves(MovesParams q) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= q.P) return;
    const int M = q.M, S = M + 1, base = p * S;
    const int out0 = q.beg_off[base] + q.end_off[base];
    int n = 0;
    auto add_move = [&](int node, int state, int kind) {         // addMoves + seen, moves.go:49-58
        for (int i = 0; i < n; i++) if (q.op_node[out0 + i] == node) return;
        q.op_node[out0 + n] = node; q.op_state[out0 + n] = state; q.op_kind[out0 + n] = kind;
        n++;
    };
    auto in_any = [&](const int32_t* off, const int32_t* nodes, int x) {   // x in flattenNodesByState(...)
        for (int t = 0; t < S; t++) if (moves_in(off, nodes, base + t, x)) return true;
        return false;
    };
    auto state_changes = [&](int si, int lo, int hi, int kind) {           // findStateChanges, moves.go:121-136
        for (int e = q.end_off[base + si]; e < q.end_off[base + si + 1]; e++) {
            int node = q.end_nodes[e];
            for (int i = lo; i < hi; i++)
                if (moves_in(q.beg_off, q.beg_nodes, base + i, node)) add_move(node, si, kind);
        }
    };
    auto clean_adds = [&](int si) {                                       // moves.go:77-82
        for (int e = q.end_off[base + si]; e < q.end_off[base + si + 1]; e++) {
            int x = q.end_nodes[e];
            if (!moves_in(q.beg_off, q.beg_nodes, base + si, x) && !in_any(q.beg_off, q.beg_nodes, x))
                add_move(x, si, BLANCE_OP_ADD);
        }
    };
    auto clean_dels = [&](int si) {                                       // moves.go:84-89
        for (int e = q.beg_off[base + si]; e < q.beg_off[base + si + 1]; e++) {
            int x = q.beg_nodes[e];
            if (!moves_in(q.end_off, q.end_nodes, base + si, x) && !in_any(q.end_off, q.end_nodes, x))
                add_move(x, -1, BLANCE_OP_DEL);
        }
    };
    if (!q.favor_min_nodes) {
        for (int si = 0; si < M; si++) {
            state_changes(si, si + 1, M, BLANCE_OP_PROMOTE);
            state_changes(si, 0, si, BLANCE_OP_DEMOTE);
            clean_adds(si);
            clean_dels(si);
        }
    } else {
        for (int si = M - 1; si >= 0; si--) {
            clean_dels(si);
            state_changes(si, 0, si, BLANCE_OP_DEMOTE);
            state_changes(si, si + 1, M, BLANCE_OP_PROMOTE);
            clean_adds(si);
        }
    }
    q.n_moves[p] = n;
}

// the moves of every partition packed back to back (op_off = exclusive scan of n_moves)
__global__ void k_moves_compact(MovesParams q, const int32_t* op_off, int32_t* node_out, int32_t* state_out, int32_t* kind_out) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= q.P) return;
    const int src = q.beg_off[p * (q.M + 1)] + q.end_off[p * (q.M + 1)], dst = op_off[p], n = op_off[p + 1] - dst;
    for (int i = 0; i < n; i++) {
        node_out[dst + i] = q.op_node[src + i]; state_out[dst + i] = q.op_state[src + i]; kind_out[dst + i] = q.op_kind[src + i];
    }
}

// ---- plan quality (blance_plan_stats_get): countStateNodes (plan.go:374-399) over the RESULT map
__global__ void k_stats_load(DevProblem d, int32_t* load /* [M][NX] */, const int32_t* constraints,
                             unsigned long long* unmet /* [M] */) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= d.P * d.M) return;
    const int p = idx / d.M, m = idx - p * d.M;
    const int len = d.live_kind[idx] == kListAbsent ? 0 : d.live_len[idx];
    const int w = (!d.weights_nil && d.part_has_weight[p]) ? d.part_weight[p] : 1;     // plan.go:387-394
    for (int i = 0; i < len; i++) atomicAdd(&load[(size_t)m * d.NX + d.live[(size_t)idx * d.L + i]], w);
    const int k = constraints[m] > 0 ? constraints[m] : 0;
    if (len < k) atomicAdd(&unmet[m], (unsigned long long)(k - len));
}

// (partition, slot) pairs whose node breaks a hierarchy rule of its state against the partition's top priority node or an
// earlier node of the same list: outside the include interval or inside the exclude interval of such an anchor
// (includeExcludeNodes, plan.go:723-734; the anchors of plan.go:185-212).  One thread per (partition, state).
__global__ void k_stats_rules(DevProblem d, int top_state, const int32_t* rule_off /* [M + 1] */, const AnchorSet* anchors /* [R][NX + 1] */,
                              const int32_t* node_leaf_pos, unsigned long long* viol /* [M] */) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= d.P * d.M) return;
    const int p = idx / d.M, m = idx - p * d.M;
    const int r0 = rule_off[m], r1 = rule_off[m + 1];
    if (r1 <= r0) return;
    const int len = d.live_kind[idx] == kListAbsent ? 0 : d.live_len[idx];
    const int ti = p * d.M + top_state;
    const int top = (d.live_kind[ti] != kListAbsent && d.live_len[ti] > 0) ? d.live[(size_t)ti * d.L] : d.NX;     // NX: the anchor ""
    int bad = 0;
    for (int i = 0; i < len; i++) {
        const int c = d.live[(size_t)idx * d.L + i];
        const int lp = node_leaf_pos[c];
        bool v = false;
        for (int r = r0; r < r1 && !v; r++)
            for (int j = -1; j < i && !v; j++) {
                const AnchorSet a = anchors[(size_t)r * (d.NX + 1) + (j < 0 ? top : d.live[(size_t)idx * d.L + j])];
                v = lp < a.alo || lp >= a.ahi || (lp >= a.blo && lp < a.bhi);
            }
        bad += v;
    }
    if (bad) atomicAdd(&viol[m], (unsigned long long)bad);
}

// one workgroup per state: min / max / sum / sum of squares / nodes in use over nodesNext
__global__ __launch_bounds__(256) void k_stats_reduce(int N, int NX, const uint8_t* alive, const int32_t* load,
                                                      long long* out /* [M][5] */) {
    BLANCE_DYN_LDS(lds);
    long long (*sh)[256] = (long long (*)[256])lds;      // [5][256]
    const int m = blockIdx.x, tid = threadIdx.x;
    long long mn = LLONG_MAX, mx = LLONG_MIN, sum = 0, sq = 0, used = 0;
    for (int n = tid; n < N; n += 256) {
        if (!alive[n]) continue;
        const long long v = load[(size_t)m * NX + n];
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
        sum += v;
        sq += v * v;
        used += v > 0;
    }
    sh[0][tid] = mn; sh[1][tid] = mx; sh[2][tid] = sum; sh[3][tid] = sq; sh[4][tid] = used;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (tid < off) {
            sh[0][tid] = sh[0][tid + off] < sh[0][tid] ? sh[0][tid + off] : sh[0][tid];
            sh[1][tid] = sh[1][tid + off] > sh[1][tid] ? sh[1][tid + off] : sh[1][tid];
            sh[2][tid] += sh[2][tid + off];
            sh[3][tid] += sh[3][tid + off];
            sh[4][tid] += sh[4][tid + off];
        }
        __syncthreads();
    }
    if (tid < 5) out[m * 5 + tid] = sh[tid][0];
}

}  // namespace blance
