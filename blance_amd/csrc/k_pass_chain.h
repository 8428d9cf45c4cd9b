// k_pass_chain: region chains (one wave64 per hierarchy region), verified-stay speculation, integer-key mode.
// Part of blance_hip.hip (one translation unit); see DESIGN.md section 4.
#pragma once

namespace blance {

// ============================================================================
// Region chains.  When the state's hierarchy rule cuts the cluster into regions
// (every node's include set is the same leaf interval as its neighbours'), a
// step whose top priority node and current nodes all live in one region reads
// and writes only that region's counters.  Steps of different regions commute,
// so each region's steps run as an independent in-order chain on one wave64:
// lanes own the region's leaves, the region's slice of nodeToNodeCounts sits
// in LDS, and the argmin is a DPP reduction -- no barrier, no global traffic
// on the critical path.  A chain that would have to look outside its region
// (fallback to candidateNodes[0], unmet constraints) raises flags[1] and the
// host redoes the whole pass with k_pass_seq.
// ============================================================================
template <int CTRL>
__device__ __forceinline__ void argmin_stage(double& s, int& n) {
    int lo2 = dpp_mov<CTRL>(__double2loint(s));
    int hi2 = dpp_mov<CTRL>(__double2hiint(s));
    int n2 = dpp_mov<CTRL>(n);
    double s2 = __hiloint2double(hi2, lo2);
    if (better(s2, n2, s, n)) { s = s2; n = n2; }
}

// (score, position) argmin over one wave64; result is wave-uniform.
__device__ __forceinline__ int wave_argmin(double s, int n) {
    argmin_stage<0xB1>(s, n);     // quad_perm [1,0,3,2]
    argmin_stage<0x4E>(s, n);     // quad_perm [2,3,0,1]
    argmin_stage<0x141>(s, n);    // row_half_mirror
    argmin_stage<0x140>(s, n);    // row_mirror: every row of 16 now agrees
    double bs = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s), 0),
                                 __builtin_amdgcn_readlane(__double2loint(s), 0));
    int bn = __builtin_amdgcn_readlane(n, 0);
#pragma unroll
    for (int r = 16; r < 64; r += 16) {
        double s2 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s), r),
                                     __builtin_amdgcn_readlane(__double2loint(s), r));
        int n2 = __builtin_amdgcn_readlane(n, r);
        if (better(s2, n2, bs, bn)) { bs = s2; bn = n2; }
    }
    return bn;
}

// nodeSorter.Score with the two quotients that do not depend on the node taken
// from LDS tables filled by the same expressions (bit-identical by construction).
__device__ __forceinline__ double chain_score(int cnt, int ntn, int tot, int hasw, int w, int NP, double cf,
                                              int booster, const double* lp_tab, const double* ff_tab) {
    double lp = 0.0, ff = 0.0;
    if (NP > 0) {
        bool lin = (unsigned)ntn < (unsigned)kLpTab, fin = (unsigned)tot < (unsigned)kFfTab;
        lp = lp_tab[lin ? ntn : 0];
        ff = ff_tab[fin ? tot : 0];
        if (!lin) lp = (double)ntn / (double)NP;
        if (!fin) ff = (0.001 * (double)tot) / (double)NP;
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

// 32-bit minimum over one wave64 (wave-uniform result)
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    unsigned t;
    t = (unsigned)dpp_mov<0xB1>((int)v);  v = t < v ? t : v;
    t = (unsigned)dpp_mov<0x4E>((int)v);  v = t < v ? t : v;
    t = (unsigned)dpp_mov<0x141>((int)v); v = t < v ? t : v;
    t = (unsigned)dpp_mov<0x140>((int)v); v = t < v ? t : v;
    unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)v, 0), r1 = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    unsigned r2 = (unsigned)__builtin_amdgcn_readlane((int)v, 32), r3 = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    r0 = r1 < r0 ? r1 : r0;
    r2 = r3 < r2 ? r3 : r2;
    return r2 < r0 ? r2 : r0;
}

// (score, position) argmin of eligible lanes over one wave64 through three 32-bit
// minima: the high and the low word of the order-preserving integer image of the
// double, then the position.  Same total order as better() for non-NaN scores.
__device__ __forceinline__ int wave_argmin3(double s, int n, bool ok) {
    if (s == 0.0) s = 0.0;                                   // -0.0 == +0.0
    unsigned long long b = (unsigned long long)__double_as_longlong(s);
    b = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
    const unsigned hi = (unsigned)(b >> 32), lo = (unsigned)b;
    const unsigned mh = wave_min_u32_bcast(ok ? hi : kKeyNoneV);
    const bool ok2 = ok && hi == mh;
    const unsigned ml = wave_min_u32_bcast(ok2 ? lo : kKeyNoneV);
    const bool ok3 = ok2 && lo == ml;
    const unsigned mn = wave_min_u32_bcast(ok3 ? (unsigned)n : kKeyNoneV);
    return __ballot(ok) ? (int)mn : INT_MAX;
}

// FAST = the pass has NP == 0 and no node weights: nodeSorter.Score is then
// double(count) - currentFactor with currentFactor in {1.5, integers}, so
// 2 * score is an exact small integer and (score, position) packs into one
// 32-bit key: [ 2*count - 2*currentFactor + 2^17 | node id (13 bits) ].  Lanes
// whose counters leave the representable range make the chain escape.
constexpr int kKeyBias = 1 << 17;
constexpr unsigned kKeyNone = 0xffffffffu;
constexpr int kCompactMax = 8000;    // |count| bound of the compact keys of the blank-run loop

constexpr int kChainWaves = 8;                     // most waves of a region's workgroup (k_pass_chain below; launched with 4 or 8)
constexpr int kStageVec = kCW / 4;                 // 16-byte words of a stage per lane of one wave: the records of its 64 steps
static_assert(kCW % 4 == 0 && kStageVec == 6, "BLANCE_STAGE_EACH lists 6 words per lane");
// A stage = 64 steps per wave of the workgroup (256 steps on four waves, 512 on eight).  Its records (96 bytes a step) travel
// as 16-byte loads, each wave of the workgroup its share (the records of the 64 steps it tests in a round that starts with
// the stage): 6 per lane, ALL in flight at once and a stage ahead.  The words as named
// values: as an array the compiler keeps them in scratch memory, and a store to scratch waits for the load it stores.
#define BLANCE_STAGE_EACH(X) X(0) X(1) X(2) X(3) X(4) X(5)
#define BLANCE_STAGE_DECL(t) int4 pre##t = {0, 0, 0, 0};
#define BLANCE_STAGE_FETCH(t) { const int i_ = q0_ + lane + 64 * t; pre##t = src_[i_ < n4_ ? i_ : 0]; }
#define BLANCE_STAGE_COMMIT(t) dst_[q0_ + lane + 64 * t] = pre##t;      /* the whole quarter: words past a short stage are never read */
// (words past a short stage re-read its first one: nothing outside the stage's records is touched; a record is 96 bytes,
// 16-byte aligned)
#define BLANCE_STAGE_FETCH_ALL(crec, base_, cend_)                                                               \
    {                                                                                                             \
        const int n4_ = ((cend_) - (base_) < stage ? (cend_) - (base_) : stage) * (kCW / 4);                     \
        const int q0_ = wave * (64 * kStageVec);                                                                  \
        const int4* src_ = (const int4*)((crec) + (size_t)(base_) * kCW);                                         \
        BLANCE_STAGE_EACH(BLANCE_STAGE_FETCH)                                                                     \
    }

// HELPER WAVES (round 6).  A pass of (mostly) stays is bound by the instruction count of the stay test, one wave per region;
// the test reads only LDS (the mirrors of the per-leaf registers, the staged records, the region's nodeToNodeCounts rows).
// The kernel therefore runs as a workgroup of 4 or 8 waves (one or two per SIMD of the CU; eight when the LDS they need is
// there: a stage of 512 steps): in a speculation round wave w
// tests steps [b + 64 w, b + 64 w + 64), all at the same time and under the same hypothesis -- "every step of the round
// stays": no counter changes, and a step's row of nodeToNodeCounts is bumped only by the steps with its top priority node
// (plan.go:238-245).  What an EARLIER step of the round with my top priority node would have bumped is added in by the test
// itself: every wave enters its steps' top priority nodes in its own table (the first lane per node; a second lane of the
// same wave with that node fails, as before), and a lane of wave w looks its node up in the tables of waves 0 .. w - 1 -- a
// hit names the one step of that wave whose row bumps it has to count (that step keeps its nodes if it is committed at all:
// the round commits a prefix).  Wave 0 combines the verdicts, commits the prefix's row bumps from the records (the waves
// have staged their steps' outputs already -- a slot that is not committed is written again by whatever commits it).  Every
// wave bumps the rows of ITS committed steps (LDS atomics; a round that is not committed whole ends with one more barrier so
// that wave 0 sees them).  The waves also share a stage's housekeeping (kQCmd 2, stage_service): each commits its quarter of
// the stage's records from its own prefetch registers and writes its quarter of the finished stage's outputs to HBM.
// Everything else (events, blank runs, general steps) is wave 0's alone, the helpers parked on the barrier.
constexpr int kChainCtl = 32;                       // words of the command block between the waves

// (regions of more than 256 leaves: eight leaves a lane need the registers of a wave that has its SIMD to itself -- four waves)
template <int NPTC>
constexpr int chain_waves_max() { return NPTC >= 8 ? 4 : kChainWaves; }

template <int NPTC, int KM, bool FAST>
__global__ __launch_bounds__(64 * chain_waves_max<NPTC>()) void k_pass_chain(ChainParams q) {
    BLANCE_DYN_LDS(lds);
    if (q.flags[0]) return;
    const int lane = threadIdx.x & 63;
    const int wave = uni((int)(threadIdx.x >> 6)), NWv = uni((int)(blockDim.x >> 6));
    const int rg = q.region_base + blockIdx.x;
    const int lo = q.reg_lo[rg], hi = q.reg_hi[rg], size = hi - lo;
    const int cbeg = q.reg_off[rg], cend = q.reg_off[rg + 1];
    if (cbeg >= cend && !(q.ev_off && q.ev_off[rg] != q.ev_off[rg + 1])) return;   // no step, no event
    const int N = q.N, NX = q.NX, M = q.M, NP = q.NP, s = q.s, k = q.k;
    const int stage = 64 * NWv;                      // steps whose records / outputs are in LDS at a time
    // LDS: quotient tables, mirrors of the per-leaf registers (read by the stay
    // validators), a kChainStage-step staging area for records and outputs (no global memory
    // operation inside the step loop), the region's nodeToNodeCounts rows
    double* lp_tab = (double*)lds;                   // [kLpTab]
    double* ff_tab = lp_tab + kLpTab;                // [kFfTab]
    double* gL = ff_tab + kFfTab;                    // [size]
    int* cntL = (int*)(gL + size);                   // [size]
    int* totL = cntL + size;
    int* nidL = totL + size;
    int* wgtL = nidL + size;
    int* flgL = wgtL + size;                         // bit 0 alive (in nodesNext), bit 1 has weight
    int* clsL = flgL + size;                         // exclude class of the leaf's node, -1 if none
    int* cszL = clsL + size;                         // leaves covered by class c
    // [stage][kCW], 16-byte aligned (inside the launch's slack; by offset, so that the pointer stays an LDS pointer)
    int* recbuf = cszL + size + ((4 - (int)(((unsigned char*)(cszL + size) - lds) >> 2)) & 3);
    int* outbuf = recbuf + stage * kCW;              // [stage][OW]
    const int MS = size + 1;
    int* markL = outbuf + stage * q.OW;              // [NWv][size + 1] per wave: first lane of its 64 steps per top priority node
    int* ctl = markL + NWv * MS;                     // [kChainCtl] wave 0's command, the waves' verdicts
    int* ntn_l = ctl + kChainCtl;                    // [size][ST] nodeToNodeCounts rows, padded stride
    const int ST = size + 1;
    // ---- the stay test of one step (lane a of wave w tests step sb of the round that starts at b0): plan.go:98-248 under the
    // hypothesis that the step keeps its nodes.  stay_mark first (all waves), then -- behind a barrier when several waves
    // take part -- stay_test, which leaves the step's nodes in emission order in on[].
    auto stay_mark = [&](const int sb, const bool active, const int b0, const int a, const int w) {
        if (NP > 0) {
            const int vtl = recbuf[(active ? sb : b0) * kCW + 4];
            if (active) atomicMin(&markL[w * MS + ((vtl >= 0 && vtl <= size) ? vtl : size)], a);
        }
    };
    auto stay_unmark = [&](const int sb, const bool active, const int b0, const int w) {
        if (NP > 0 && active) {
            const int vtl = recbuf[sb * kCW + 4];
            markL[w * MS + ((vtl >= 0 && vtl <= size) ? vtl : size)] = INT_MAX;
        }
        (void)b0;
    };
    auto stay_test = [&](const int sb, const bool active, const int b0, const int a, const int w, const int next_ev,
                         const double bound_s, const int bound_n, int (&on)[KM], int (&oi)[KM], int& vtl_out) -> bool {
        const int* rp = recbuf + (active ? sb : b0) * kCW;
        bool fail = false;
        const double vstick = __hiloint2double(rp[3], rp[2]);
        const int vtl = rp[4];
        vtl_out = vtl;
        const int mt = (vtl >= 0 && vtl <= size) ? vtl : size;
        const int cn = rp[5];
        if (!((cn >> 24) & 1) || (cn & 0xff) != k || ((cn >> 25) & 1)) fail = true;   // exactly k nodes, all here
        if (rp[0] > next_ev) fail = true;                             // an event comes first
        int oc[KM + 1], adj[KM];
        double so[KM];
        oc[0] = rp[6];
#pragma unroll
        for (int j = 0; j < KM; j++) {
            on[j] = -3; so[j] = 0.0; oi[j] = 0; oc[j + 1] = -1; adj[j] = 0;
            if (j < k) {
                int li = rp[kCOwn + j];
                if (li < 0 || li >= size) { fail = true; li = 0; }
                oi[j] = li;
            }
        }
        if (!FAST && NP > 0) {
            // an earlier step of the round with my top priority node -- one per earlier wave at most among the steps that can be
            // committed -- has bumped my row at ITS nodes (it keeps them, or the prefix ends before me): plan.go:238-245
            for (int w2 = 0; w2 < w; w2++) {
                const int pl = markL[w2 * MS + mt];
                if (pl != INT_MAX) {
                    const int* pr = recbuf + (b0 + 64 * w2 + pl) * kCW;
#pragma unroll
                    for (int j2 = 0; j2 < KM; j2++) {
                        if (j2 < k) {
                            const int x = pr[kCOwn + j2];
#pragma unroll
                            for (int j = 0; j < KM; j++) if (j < k && oi[j] == x) adj[j]++;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < KM; j++) {
            if (j < k) {
                const int li = oi[j];
                on[j] = nidL[li];
                oc[j + 1] = clsL[li];
                // the partition's own nodes: candidates, scored exactly
                if (!(flgL[li] & 1)) fail = true;
                const int nt = (!FAST && NP > 0) ? ntn_l[vtl * ST + li] + adj[j] : 0;
                so[j] = chain_score(cntL[li], nt, totL[li], (flgL[li] >> 1) & 1, wgtL[li], NP, vstick,
                                    q.booster_kind, lp_tab, ff_tab);
            }
        }
        // A step that keeps its nodes emits them in (score, position) order -- slot j takes the best
        // node left (plan.go:185-226) -- which need not be the list order; no counter changes
        // either way.  Sort them (insertion sort, k <= 4) and check the slots in that order.
#pragma unroll
        for (int j = 1; j < KM; j++) {
#pragma unroll
            for (int e = j; e > 0; e--) {
                if (e < k && better(so[e], on[e], so[e - 1], on[e - 1])) {
                    const double ts = so[e]; so[e] = so[e - 1]; so[e - 1] = ts;
                    const int tn = on[e]; on[e] = on[e - 1]; on[e - 1] = tn;
                    const int tc = oc[e + 1]; oc[e + 1] = oc[e]; oc[e] = tc;
                }
            }
        }
        // anchors top, own_0 .. own_{k-2}: their exclude classes must leave candidates,
        // and own_j must not sit in a class excluded before its slot
        {
            int cov = 0;
#pragma unroll
            for (int j = 0; j < KM; j++) {
                if (j < k) {
                    if (oc[j] < 0 && !(q.flat && j == 0)) fail = true;
                    bool dup = false;
#pragma unroll
                    for (int e = 0; e < KM; e++) if (e < j && oc[e] == oc[j]) dup = true;
                    if (!dup && oc[j] >= 0) cov += cszL[oc[j]];
                    if (cov >= size) fail = true;
#pragma unroll
                    for (int e = 0; e < KM; e++) if (e <= j && oc[e] >= 0 && oc[e] == oc[j + 1]) fail = true;
                }
            }
        }
        // every one of them below the bound
#pragma unroll
        for (int j = 0; j < KM; j++)
            if (j < k && !better(so[j], on[j], bound_s, bound_n)) fail = true;
        // an own node also listed in a higher priority state is no candidate (the
        // record keeps such leaves under "higher"; gather refuses nodes held twice)
        // an earlier step of MY wave's 64 with the same top priority node would have bumped my row
        if (NP > 0 && active && markL[w * MS + mt] < a) fail = true;
        return fail || !active;
    };
    // what a stay emits, staged (a slot that is not committed is written again by whatever commits it)
    auto stay_stage = [&](const int sb, const bool active, const int (&on)[KM]) {
        if (active) {
            int* op = outbuf + sb * q.OW;
            op[0] = k;
#pragma unroll
            for (int j = 0; j < KM; j++) if (j < k) op[1 + j] = on[j];
        }
    };
    // the prefix's row bumps, every wave its own steps (plan.go:238-245): lanes below `mine` of this wave are committed
    auto stay_bump = [&](const int a, const int mine, const int vtl, const int (&oi)[KM]) {
        if (!FAST && NP > 0 && a < mine) {
#pragma unroll
            for (int j = 0; j < KM; j++) if (j < k) atomicAdd(&ntn_l[vtl * ST + oi[j]], 1);
        }
    };
    // a round's committed prefix from the waves' verdicts (bit a of wave w: step b + 64 w + a is NOT a certain stay)
    auto round_prefix = [&]() -> int {
        int nok = 0;
        for (int w2 = 0; w2 < NWv && nok == 64 * w2; w2++) {
            const unsigned long long hf = ((unsigned long long)(unsigned)ctl[9 + 2 * w2] << 32) | (unsigned)ctl[8 + 2 * w2];
            nok += hf ? __ffsll((long long)hf) - 1 : 64;
        }
        return uni(nok);
    };
    // ---- a stage's housekeeping, every wave a quarter (command 2; ctl[7]: 1 = this stage's records into LDS, 2 = the finished
    // stage's outputs to HBM).  Quarter w = the 64 steps wave w tests in a round that starts with the stage.
    BLANCE_STAGE_EACH(BLANCE_STAGE_DECL)
    if (cbeg < cend) BLANCE_STAGE_FETCH_ALL(q.crec, cbeg, cend)
    auto stage_service = [&](const int fl) {
        // (the records first: their loads were issued a stage ago, and behind the output stores below the wait for them
        // would be a wait for the stores -- loads and stores share the counter)
        if (fl & 1) {
            int4* dst_ = (int4*)recbuf;
            const int q0_ = wave * (64 * kStageVec);
            BLANCE_STAGE_EACH(BLANCE_STAGE_COMMIT)
            BLANCE_WAVE_SYNC();                      // (a lane reads records other lanes of its wave wrote; LDS is in order within a wave)
        }
        if (fl & 2) {
            const int pbase = uni(ctl[26]), nd = uni(ctl[27]);
            const int hi = nd < 64 * wave + 64 ? nd : 64 * wave + 64;
            for (int i = 64 * wave * q.OW + lane; i < hi * q.OW; i += 64) q.out[(size_t)pbase * q.OW + i] = outbuf[i];
        }
        if (fl & 1) {
            const int sbase = uni(ctl[24]);
            if (sbase + stage < cend) BLANCE_STAGE_FETCH_ALL(q.crec, sbase + stage, cend)
        }
    };
    if (wave != 0) {
        // ---- a helper wave: parked on the barrier until wave 0 posts a command
        for (;;) {
            lds_barrier();                           // (A) posted; the tables are as the round sees them
            const int cmd = uni(ctl[0]);
            if (cmd == 0) break;
            if (cmd == 2) {
                stage_service(uni(ctl[7]));
                lds_barrier();                       // (S) the stage's records are in LDS, the last stage's outputs on their way
                continue;
            }
            const int b0 = uni(ctl[1]), nbh = uni(ctl[2]);
            if (uni(ctl[7])) stage_service(uni(ctl[7]));     // (a stage's first round carries its housekeeping: my steps' records are mine)
            const int sb = b0 + 64 * wave + lane;
            const bool active = sb < nbh;
            stay_mark(sb, active, b0, lane, wave);
            lds_barrier();                           // (M) every wave's table is complete; so are the stage's records
            int on[KM], oi[KM], vtl = 0;
            const bool fail = stay_test(sb, active, b0, lane, wave, uni(ctl[3]), __hiloint2double(uni(ctl[5]), uni(ctl[4])), uni(ctl[6]),
                                        on, oi, vtl);
            stay_stage(sb, active, on);
            const unsigned long long fm = __ballot(fail);
            if (lane == 0) { ctl[8 + 2 * wave] = (int)(unsigned)fm; ctl[9 + 2 * wave] = (int)(unsigned)(fm >> 32); }
            lds_barrier();                           // (B) the verdicts are in
            const int nok = round_prefix();
            stay_bump(lane, nok - 64 * wave, vtl, oi);
            stay_unmark(sb, active, b0, wave);
            if (b0 + nok < nbh) lds_barrier();       // (C) wave 0 goes on alone inside the stage: it has to see the bumps
        }
        return;
    }
    if (!FAST && NP > 0) {
        for (int i = lane; i < kLpTab; i += 64) lp_tab[i] = (double)i / (double)NP;
        for (int i = lane; i < kFfTab; i += 64) ff_tab[i] = (0.001 * (double)i) / (double)NP;
        if (q.ntn_in_lds)
            for (int i = lane; i < (size + 1) * ST; i += 64) ntn_l[i] = 0;    // last row: "" (flat mode)
    }
    for (int i = lane; i < size; i += 64) cszL[i] = q.cls_size[lo + i];
    for (int i = lane; i < NWv * MS; i += 64) markL[i] = INT_MAX;
    BLANCE_WAVE_SYNC();                              // (one wave: LDS is in order; the helper meets the tables behind barrier A)

    // lane l owns leaves lo + l + 64 u
    int nid[NPTC], cntv[NPTC], totv[NPTC], wv[NPTC], cls[NPTC], mycsz[NPTC];
    unsigned alive_m = 0, hasw_m = 0;
    double g[NPTC];
    bool range_bad = false;
#pragma unroll
    for (int u = 0; u < NPTC; u++) mycsz[u] = 0;
#pragma unroll
    for (int u = 0; u < NPTC; u++) {
        const int pos = lo + lane + 64 * u;
        nid[u] = -2; cntv[u] = 0; totv[u] = 0; wv[u] = 0; cls[u] = -1; g[u] = 0.0;
        if (pos < hi) {
            int n = q.leaf_node[pos];
            if (n >= 0) {
                nid[u] = n;
                cntv[u] = q.cnt[s * NX + n];
                int tsum = 0;
                for (int t = 0; t <= M; t++) tsum += q.cnt[t * NX + n];
                totv[u] = tsum;
                wv[u] = q.node_weight[n];
                if (q.node_has_weight[n]) hasw_m |= 1u << u;
                if (n < N && q.alive[n]) alive_m |= 1u << u;
                g[u] = chain_score(cntv[u], 0, totv[u], (hasw_m >> u) & 1, wv[u], NP, 0.0, q.booster_kind, lp_tab, ff_tab);
                cls[u] = q.leaf_cls[pos];
                mycsz[u] = cls[u] >= 0 ? q.cls_size[lo + cls[u]] : 0;
                if (FAST && (cntv[u] >= (1 << 15) || cntv[u] <= -(1 << 15))) range_bad = true;
            }
            const int i = lane + 64 * u;
            gL[i] = g[u]; cntL[i] = cntv[u]; totL[i] = totv[u]; nidL[i] = nid[u]; wgtL[i] = wv[u];
            flgL[i] = ((alive_m >> u) & 1) | (((hasw_m >> u) & 1) << 1);
            clsL[i] = cls[u];
        }
    }
    // compact-key loop preconditions: node ids rise with the leaf index, and k exclude
    // classes can never cover the region (then "empty set -> reset", plan.go:746, cannot occur)
    bool compact_ok = false;
    if (FAST) {
        BLANCE_WAVE_SYNC();
        int prev = -1, mx = 0;
        bool mono = true;
        if (lane == 0) {
            for (int i = 0; i < size; i++) {
                const int n = nidL[i];
                if (n >= 0) { if (n <= prev) mono = false; prev = n; }
                if (cszL[i] > mx) mx = cszL[i];
            }
        }
        const int mono_u = __builtin_amdgcn_readlane(mono ? 1 : 0, 0), mx_u = __builtin_amdgcn_readlane(mx, 0);
        compact_ok = mono_u && (long long)mx_u * (k + 1) < (long long)size && size <= 256;
    }
    bool escaped = __ballot(range_bad) != 0;
    int stop_at = cbeg;                            // flat mode: first step this launch did not do
    bool stop_range = escaped;
    // stay speculation: tried again whenever the last general step turned out to be a stay
    const bool spec_ok = q.ntn_in_lds || NP == 0;
    bool try_spec = true, gmin_dirty = true;
    double gmin_s = 0.0;
    int gmin_n = INT_MAX;
    int spec_steps = 0, spec_batches = 0;          // statistics (lane 0)
    // events: nodes of other regions' partitions that leave this region's counters (see ChainParams)
    int ev_cur = q.ev_off ? q.ev_off[rg] : 0;
    const int ev_end = q.ev_off ? q.ev_off[rg + 1] : 0;
    int next_ev_oi = ev_cur < ev_end ? q.ev_oi[q.ev_perm[ev_cur]] : INT_MAX;
    auto apply_event = [&]() {
        const int e = q.ev_perm[ev_cur];
        const int el = q.ev_leaf[e], ew = q.ev_w[e];
#pragma unroll
        for (int u = 0; u < NPTC; u++) {
            if (el == lane + 64 * u) {
                cntv[u] -= ew;
                totv[u] -= ew;
                g[u] = FAST ? (double)cntv[u]
                            : chain_score(cntv[u], 0, totv[u], (hasw_m >> u) & 1, wv[u], NP, 0.0, q.booster_kind, lp_tab, ff_tab);
                gL[el] = g[u]; cntL[el] = cntv[u]; totL[el] = totv[u];
                if (FAST && cntv[u] <= -(1 << 15)) range_bad = true;
            }
        }
        BLANCE_WAVE_SYNC();
        gmin_dirty = true;
        ev_cur++;
        next_ev_oi = ev_cur < ev_end ? q.ev_oi[q.ev_perm[ev_cur]] : INT_MAX;
        // nothing of an event is in flight when the steps go on: otherwise the compiler guards every later write to a
        // register one of these loads used with a wait for ALL loads -- the next stage's records among them
        BLANCE_WAIT_VMEM();
    };
    PH_DECL;

    // A stage's records (24 KB) travel as 24 16-byte loads per lane, ALL in flight at once, and a stage ahead: they are
    // issued when the previous stage starts and land in LDS when it ends -- one wave per CU has nothing else to hide the
    // HBM round trip behind (a copy loop of dword loads, 16 in flight, paid six round trips per stage: more than the
    // stage's steps).
    int prev_base = 0, prev_done = 0;              // the finished stage whose outputs are still in LDS
    auto post_service = [&](const int fl, const int sbase) {
        if (lane == 0) { ctl[0] = 2; ctl[7] = fl; ctl[24] = sbase; ctl[26] = prev_base; ctl[27] = prev_done; }
        lds_barrier();                               // (A)
        stage_service(fl);
        lds_barrier();                               // (S)
        prev_done = 0;
    };
    for (int base = cbeg; base < cend && !escaped; base += stage) {
      const int nb = cend - base < stage ? cend - base : stage;
      PH(0);
      // The stage's housekeeping rides on its first round when that round is certain to come: no event can be due (the test
      // for one reads the stage's first record), the last step was a stay.  Else it is a command of its own.
      int pending_fl = 1 | (prev_done > 0 ? 2 : 0);
      if (!(spec_ok && try_spec && next_ev_oi == INT_MAX && nb > 64)) { post_service(pending_fl, base); pending_fl = 0; }
      int b = 0;
      while (b < nb) {
        while (!pending_fl && next_ev_oi < recbuf[b * kCW]) apply_event();      // due before this step (pass order)
        // ---- Speculate that the next (up to 64) steps keep their nodes.  A stay
        // changes no counter, so under that hypothesis every step sees the state as
        // it is now and lane a can check step b + a on its own: the partition's
        // nodes, scored exactly (stickiness, nodeToNodeCounts row), must beat a lower
        // bound of every other candidate -- the smallest partition-independent score
        // of the region (the terms it leaves out are >= 0 and IEEE add / divide /
        // subtract are monotone).  The verified prefix is committed; the first step
        // that is not a certain stay takes the general step below.
        if (spec_ok && try_spec) {
            if (gmin_dirty) {                      // smallest (g, node) over the region's live leaves
                double ms = pos_inf();
                int mn = INT_MAX;
#pragma unroll
                for (int u = 0; u < NPTC; u++) {
                    const bool ok = (alive_m >> u) & 1;
                    const bool take = ok && better(g[u], nid[u], ms, mn);
                    ms = take ? g[u] : ms;
                    mn = take ? nid[u] : mn;
                }
                gmin_n = wave_argmin(ms, mn);
                gmin_s = pos_inf();
#pragma unroll
                for (int u = 0; u < NPTC; u++) {
                    unsigned long long bm = __ballot(nid[u] == gmin_n);
                    if (bm) {
                        int wl = __ffsll((long long)bm) - 1;
                        gmin_s = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(g[u]), wl),
                                                  __builtin_amdgcn_readlane(__double2loint(g[u]), wl));
                    }
                }
                gmin_dirty = false;
            }
            const int a = lane;
            const int sb = b + a;
            const bool active = sb < nb;
            const bool multi = nb - b > 64;            // the helper waves test the steps behind my 64
            PH(12);
            if (multi) {
                if (lane == 0) {
                    ctl[0] = 1; ctl[1] = b; ctl[2] = nb; ctl[3] = next_ev_oi;
                    ctl[4] = __double2loint(gmin_s); ctl[5] = __double2hiint(gmin_s); ctl[6] = gmin_n;
                    ctl[7] = pending_fl; ctl[24] = base; ctl[26] = prev_base; ctl[27] = prev_done;
                }
                lds_barrier();                         // (A)
                if (pending_fl) { stage_service(pending_fl); pending_fl = 0; prev_done = 0; }
            }
            PH(13);
            stay_mark(sb, active, b, a, 0);
            if (multi) lds_barrier();                  // (M)
            else BLANCE_WAVE_SYNC();
            PH(14);
            int on[KM], oi[KM], vtl = 0;
            const bool fail = stay_test(sb, active, b, a, 0, next_ev_oi, gmin_s, gmin_n, on, oi, vtl);
            stay_stage(sb, active, on);
            const unsigned long long fm = __ballot(fail);
            int nok = fm ? __ffsll((long long)fm) - 1 : 64;
            PH(15);
            if (multi) {
                if (lane == 0) { ctl[8] = (int)(unsigned)fm; ctl[9] = (int)(unsigned)(fm >> 32); }
                lds_barrier();                         // (B) every wave's verdicts
                nok = round_prefix();
            }
            PH(16);
            stay_bump(a, nok, vtl, oi);                // (my own steps; the helpers bump theirs)
            BLANCE_WAVE_SYNC();
            stay_unmark(sb, active, b, 0);
            if (multi && b + nok < nb) lds_barrier();  // (C) the helpers' bumps before this wave goes on alone
            PH(17);
            if (lane == 0) { spec_steps += nok; spec_batches++; }
            b += nok;
            if (b >= nb) break;
            if (nok > 0 && (nok & 63) == 0) continue;   // whole waves' worth of stays: try the next round from there
            if (next_ev_oi < recbuf[b * kCW]) continue;          // an event is due before the step that failed
        }
        // ---- FAST mode, runs of blank steps (a partition that holds no node of this or
        // a lower priority state; its higher priority nodes are simply masked): nothing
        // to match, demote or un-count, so a step is k masked minima plus k counter
        // bumps.  Lane a pre-scans step b + a; the run is walked with everything in
        // registers, on compact keys [ 2*count + 2^14 : 15 | leaf : 8 | exclude class : 9 ]
        // that also carry what the next slot needs to know about the winner.  Needs node
        // ids rising with the leaf index (ties go to the lower node id, plan.go:617-628),
        // exclude classes too small to ever empty the candidate set, small counters;
        // otherwise the general step below does the work.
        if (FAST && compact_ok) {
            const int sb = b + lane;
            const bool active = sb < nb;
            const int* rp = recbuf + (active ? sb : b) * kCW;
            const int w0 = recbuf[b * kCW + 1];
            const int tcv = rp[6];
            const bool blank = active && (rp[5] & 0xff00ff) == 0 && rp[1] == w0 && (tcv >= 0 || q.flat) &&
                               rp[0] < next_ev_oi;
            int hv[kChainHigh];
#pragma unroll
            for (int j = 0; j < kChainHigh; j++) hv[j] = rp[kCHigh + j];
            const bool any_high = __ballot(blank && (rp[5] & 0xff00) != 0) != 0;
            const unsigned long long nm = __ballot(!blank);
            int run = nm ? __ffsll((long long)nm) - 1 : 64;
            if (run > nb - b) run = nb - b;
            bool small = true;                     // counters inside the compact key's range?
#pragma unroll
            for (int u = 0; u < NPTC; u++) if (cntv[u] <= -kCompactMax || cntv[u] >= kCompactMax) small = false;
            if (run > 0 && w0 > 0 && w0 < 64 && !__ballot(!small)) {
                unsigned key[NPTC];
#pragma unroll
                for (int u = 0; u < NPTC; u++) {
                    const unsigned c9 = cls[u] < 0 ? 511u : (unsigned)cls[u];
                    key[u] = ((alive_m >> u) & 1)
                                 ? (((unsigned)(2 * cntv[u] + (1 << 14)) << 17) | ((unsigned)(lane + 64 * u) << 9) | c9)
                                 : kKeyNone;
                }
                const unsigned bump = (unsigned)(2 * w0) << 17;
                bool esc = false;
                int r = 0;
                // two instances of the loop: the common one carries no code for higher priority nodes
                auto walk = [&](auto with_high) {
                    for (; r < run; r++) {
                        int acls = __builtin_amdgcn_readlane(tcv, r);
                        unsigned excl_m = 0;
                        if (decltype(with_high)::value) {   // plan.go:146-154
#pragma unroll
                            for (int j = 0; j < kChainHigh; j++) {
                                const int hj = __builtin_amdgcn_readlane(hv[j], r);
#pragma unroll
                                for (int u = 0; u < NPTC; u++) excl_m |= (hj == lane + 64 * u ? 1u : 0u) << u;
                            }
                        }
                        int chosen_l[KM];
#pragma unroll
                        for (int j = 0; j < KM; j++) chosen_l[j] = 0;
                        unsigned picked_m = 0;
#pragma unroll
                        for (int slot = 0; slot < KM; slot++) {
                            if (slot < k) {
                                if (acls < 0 && !(q.flat && slot == 0)) esc = true;   // anchor without an exclude class
#pragma unroll
                                for (int u = 0; u < NPTC; u++) excl_m |= (cls[u] == acls && acls >= 0 ? 1u : 0u) << u;
                                unsigned km = kKeyNone;
#pragma unroll
                                for (int u = 0; u < NPTC; u++) {
                                    const unsigned kv = ((excl_m >> u) & 1) ? kKeyNone : key[u];
                                    km = kv < km ? kv : km;
                                }
                                const unsigned kb = wave_min_u32_bcast(km);
                                if (kb == kKeyNone) esc = true;          // would fall back to candidateNodes[0]
#pragma unroll
                                for (int u = 0; u < NPTC; u++) picked_m |= (key[u] == kb && kb != kKeyNone ? 1u : 0u) << u;
                                chosen_l[slot] = (int)((kb >> 9) & 0xff);
                                const int wc = (int)(kb & 0x1ff);
                                acls = wc == 511 ? -1 : wc;              // the winner's class is excluded next
                            }
                        }
                        if (__ballot(esc)) break;
                        bool big = false;
#pragma unroll
                        for (int u = 0; u < NPTC; u++) {
                            if ((picked_m >> u) & 1) {
                                key[u] += bump;
                                cntv[u] += w0;
                                totv[u] += w0;
                                if (cntv[u] >= kCompactMax) big = true;
                            }
                        }
                        if (lane == 0) {
                            int* o = outbuf + (b + r) * q.OW;
                            o[0] = k;
#pragma unroll
                            for (int c = 0; c < KM; c++) if (c < k) o[1 + c] = nidL[chosen_l[c]];
                        }
                        if (__ballot(big)) { r++; break; }               // leave the compact range: general step next
                    }
                };
                if (any_high) walk(std::true_type{}); else walk(std::false_type{});
                // refresh the mirrors and the partition-independent scores of my leaves
#pragma unroll
                for (int u = 0; u < NPTC; u++) {
                    const int i = lane + 64 * u;
                    if (i < size) {
                        g[u] = (double)cntv[u];
                        gL[i] = g[u]; cntL[i] = cntv[u]; totL[i] = totv[u];
                        if (cntv[u] >= (1 << 15)) range_bad = true;
                    }
                }
                BLANCE_WAVE_SYNC();
                gmin_dirty = true;
                if (r > 0) try_spec = false;          // the run's steps were moves
                b += r;
                if (__ballot(range_bad)) { escaped = true; stop_range = true; break; }
                if (b >= nb) break;
                if (r > 0 && !__ballot(esc)) continue;
                // an escape inside the run: let the general step decide (it escapes the same way)
            }
        }
        // ---- general step: findBestNodes (plan.go:98-248) + commit (plan.go:290-301)
        PH(1);
        const int recw = lane < kCW ? recbuf[b * kCW + lane] : 0;
#define REC(i) __builtin_amdgcn_readlane(recw, (i))
        const int w = REC(1);
        const double stick = __hiloint2double(REC(3), REC(2));
        const int tl = REC(4);
        const int cn = REC(5);
        const int n_low = (cn >> 16) & 0xff;
        bool esc = false;
        int ntnv[NPTC];
#pragma unroll
        for (int u = 0; u < NPTC; u++) {
            ntnv[u] = 0;
            if (!FAST && NP > 0) {
                if (q.ntn_in_lds) { if (lane + 64 * u < size) ntnv[u] = ntn_l[tl * ST + lane + 64 * u]; }
                else if (nid[u] >= 0 && nid[u] < N) ntnv[u] = q.ntn[(size_t)(tl < size ? nidL[tl] : NX) * N + nid[u]];
            }
        }
        PH(2);
        unsigned inh_m = 0, own_m = 0;
        if (cn & 0xff) {
#pragma unroll
            for (int j = 0; j < kChainOwn; j++) {
                const int oj = REC(kCOwn + j);
#pragma unroll
                for (int u = 0; u < NPTC; u++) own_m |= (oj == lane + 64 * u ? 1u : 0u) << u;
            }
        }
        if (cn & 0xff00) {
#pragma unroll
            for (int j = 0; j < kChainHigh; j++) {
                const int hj = REC(kCHigh + j);
#pragma unroll
                for (int u = 0; u < NPTC; u++) inh_m |= (hj == lane + 64 * u ? 1u : 0u) << u;
            }
        }
        PH(3);
        const unsigned elig_m = alive_m & ~inh_m;
        double sc[NPTC];
        unsigned key[NPTC];
        if (FAST) {
            // 2 * stickiness: 3 or an even integer; out of range -> let the sequential pass do it
            const double s2 = stick + stick;
            const int stick2 = (s2 >= 0.0 && s2 < 32768.0) ? (int)s2 : 0;
            if (!(s2 >= 0.0 && s2 < 32768.0) || (double)stick2 != s2) esc = true;
#pragma unroll
            for (int u = 0; u < NPTC; u++) {
                const int v = 2 * cntv[u] - (((own_m >> u) & 1) ? stick2 : 0) + kKeyBias;
                key[u] = ((elig_m >> u) & 1) ? (((unsigned)v << 13) | (unsigned)nid[u]) : kKeyNone;
                sc[u] = 0.0;
            }
        } else {
            unsigned need_m = own_m;
#pragma unroll
            for (int u = 0; u < NPTC; u++) { if (ntnv[u] != 0) need_m |= 1u << u; key[u] = 0; }
            if (__ballot(need_m != 0)) {
#pragma unroll
                for (int u = 0; u < NPTC; u++) {
                    double full = chain_score(cntv[u], ntnv[u], totv[u], (hasw_m >> u) & 1, wv[u], NP,
                                              ((own_m >> u) & 1) ? stick : 0.0, q.booster_kind, lp_tab, ff_tab);
                    sc[u] = ((need_m >> u) & 1) ? full : g[u];
                }
            } else {
#pragma unroll
                for (int u = 0; u < NPTC; u++) sc[u] = g[u];
            }
        }
        PH(4);
        // The rule's k picks (plan.go:177-223).  Every anchor's include set is this
        // region, so the running set is the region minus the anchors' exclude classes;
        // an empty running set (plan.go:746 would reset it), an anchor without a
        // proper class, a fallback to candidateNodes[0] or a duplicate pick escape.
        int ec[KM];
#pragma unroll
        for (int j = 0; j < KM; j++) ec[j] = -2;
        int covered = 0;
        unsigned excl_m = 0;
        int chosen[KM], chosen_l[KM];
#pragma unroll
        for (int j = 0; j < KM; j++) { chosen[j] = -1; chosen_l[j] = -1; }
        int n_out = 0;
        int acls = REC(6);                           // exclude class of the current anchor ...
        int acsz = REC(23);                          // ... and the leaves it covers
        PH(5);
#pragma unroll
        for (int slot = 0; slot < KM; slot++) {
            if (slot < k) {
                bool dup = false;
#pragma unroll
                for (int j = 0; j < KM; j++) if (j < slot && ec[j] == acls) dup = true;
                if (acls < 0 && !(q.flat && slot == 0)) esc = true;
                if (!dup && acls >= 0) {
                    ec[slot] = acls;
                    covered += acsz;
#pragma unroll
                    for (int u = 0; u < NPTC; u++) excl_m |= (cls[u] == acls ? 1u : 0u) << u;
                }
                if (covered >= size) esc = true;
                int best;
                if (FAST) {
                    unsigned km = kKeyNone;
#pragma unroll
                    for (int u = 0; u < NPTC; u++) {
                        const unsigned kv = ((excl_m >> u) & 1) ? kKeyNone : key[u];
                        km = kv < km ? kv : km;
                    }
                    PH(6);
                    const unsigned kb = wave_min_u32(km);
                    best = kb == kKeyNone ? INT_MAX : (int)(kb & 0x1fff);
                } else {
                    double bs = pos_inf();
                    int bn = INT_MAX;
                    bool any = false;
#pragma unroll
                    for (int u = 0; u < NPTC; u++) {
                        const bool ok = ((elig_m & ~excl_m) >> u) & 1;
                        const bool take = ok && (!any || better(sc[u], nid[u], bs, bn));
                        bs = take ? sc[u] : bs;
                        bn = take ? nid[u] : bn;
                        any = any || ok;
                    }
                    PH(6);
                    best = wave_argmin3(bs, bn, any);
                }
                PH(7);
                if (best == INT_MAX) esc = true;
                int wcls = -1, wloc = -1, wcsz = 0;
#pragma unroll
                for (int u = 0; u < NPTC; u++) {
                    unsigned long long bm = __ballot(nid[u] == best);
                    if (bm) {
                        int wl = __ffsll((long long)bm) - 1;
                        wcls = __builtin_amdgcn_readlane(cls[u], wl);
                        wcsz = __builtin_amdgcn_readlane(mycsz[u], wl);
                        wloc = wl + 64 * u;
                    }
                }
#pragma unroll
                for (int c = 0; c < KM; c++) if (c < slot && chosen[c] == best) esc = true;   // duplicate pick
                chosen[slot] = best;
                chosen_l[slot] = wloc;
                n_out = slot + 1;
                acls = wcls;
                acsz = wcsz;
                PH(8);
            }
        }
        if (__ballot(esc)) { escaped = true; break; }

        // ---- commit: the owner lane of a leaf updates it
        int dc[NPTC], dt[NPTC];
#pragma unroll
        for (int u = 0; u < NPTC; u++) { dc[u] = 0; dt[u] = 0; }
#pragma unroll
        for (int u = 0; u < NPTC; u++) {             // old nodes of this state leave it (plan.go:290-293)
            const int d = ((own_m >> u) & 1) ? w : 0;
            dc[u] -= d; dt[u] -= d;
        }
#pragma unroll
        for (int c = 0; c < KM; c++) {               // chosen nodes enter it (plan.go:299-301)
            if (c < k) {
#pragma unroll
                for (int u = 0; u < NPTC; u++) {
                    const bool mine = chosen_l[c] == lane + 64 * u;
                    dc[u] += mine ? w : 0;
                    dt[u] += mine ? w : 0;
                    if (!FAST && NP > 0 && mine) {
                        if (q.ntn_in_lds) ntn_l[tl * ST + lane + 64 * u] = ntnv[u] + 1;      // plan.go:238-245
                        else q.ntn[(size_t)(tl < size ? nidL[tl] : NX) * N + nid[u]] = ntnv[u] + 1;
                    }
                }
            }
        }
        if (n_low > 0) {                             // a chosen node leaves its lower priority state (plan.go:294-297)
            for (int e = 0; e < kChainLow; e++) {
                const int le = REC(kCLow + e), lt = REC(kCLowState + e);
                if (le < 0) continue;
#pragma unroll
                for (int c = 0; c < KM; c++) {
                    if (c < k && chosen_l[c] == le) {
#pragma unroll
                        for (int u = 0; u < NPTC; u++) dt[u] -= le == lane + 64 * u ? w : 0;
                        if (lane == 0) q.cnt[lt * NX + chosen[c]] -= w;
                    }
                }
            }
        }
        PH(9);
        {
            unsigned changed_m = 0;
#pragma unroll
            for (int u = 0; u < NPTC; u++) {
                cntv[u] += dc[u];
                totv[u] += dt[u];
                if (dc[u] | dt[u]) changed_m |= 1u << u;
                if (FAST && (cntv[u] >= (1 << 15) || cntv[u] <= -(1 << 15))) range_bad = true;
            }
            if (__ballot(changed_m != 0)) {
#pragma unroll
                for (int u = 0; u < NPTC; u++) {
                    double gn = FAST ? (double)cntv[u]       // no quotients, no weights: plan.go:664-670 only
                                     : chain_score(cntv[u], 0, totv[u], (hasw_m >> u) & 1, wv[u], NP, 0.0,
                                                   q.booster_kind, lp_tab, ff_tab);
                    g[u] = ((changed_m >> u) & 1) ? gn : g[u];
                    if ((changed_m >> u) & 1) {
                        const int i = lane + 64 * u;
                        gL[i] = g[u]; cntL[i] = cntv[u]; totL[i] = totv[u];
                    }
                }
            }
        }
        PH(10);
        {
            // did this step keep its nodes?  then the next ones probably do, too
            bool same = ((cn >> 24) & 1) && !((cn >> 25) & 1) && (cn & 0xff) == n_out;
            if (same) {
#pragma unroll
                for (int c = 0; c < KM; c++) if (c < n_out && REC(kCOwn + c) != chosen_l[c]) same = false;
            }
            try_spec = same;
            gmin_dirty = true;
        }
        if (lane == 0) {
            int* o = outbuf + b * q.OW;
            o[0] = n_out;
#pragma unroll
            for (int c = 0; c < KM; c++) if (c < k) o[1 + c] = chosen[c];
        }
        PH(11);
#undef REC
        BLANCE_WAVE_SYNC();
        b++;
        if (__ballot(range_bad)) { escaped = true; stop_range = true; break; }
      }
      BLANCE_WAVE_SYNC();
      PH(18);
      stop_at = base + b;
      // flat mode keeps the steps done before a stop; a region chain's pass is redone as a whole
      const int n_done = (!escaped || q.flat) ? b : 0;
      prev_base = base; prev_done = n_done;          // (written out by the four waves when the next stage starts)
      PH(19);
    }
    if (prev_done > 0) post_service(2, 0);           // the last stage's outputs
    if (lane == 0) ctl[0] = 0;                       // the helpers leave
    lds_barrier();
    if (!escaped) while (ev_cur < ev_end) apply_event();      // nodes that leave after this region's last step
    if (__ballot(range_bad)) { escaped = true; stop_range = true; }
    PH_DUMP(cend - cbeg);
    if (lane == 0 && spec_batches) { atomicAdd(&q.flags[2], spec_steps); atomicAdd(&q.flags[3], spec_batches); }
    if (escaped) {
        if (lane == 0) { q.flags[1] = 1; q.flags[4] = stop_at; q.flags[5] = stop_range ? 1 : 0; }
        if (!q.flat) return;
        // the rest of the pass continues from global memory: hand over the LDS rows
        if (!FAST && NP > 0 && q.ntn_in_lds) {
            BLANCE_WAVE_SYNC();
            for (int i = lane; i < (size + 1) * size; i += 64) {
                const int row = i / size, col = i - row * size;
                const int cn = nidL[col];
                if (cn >= 0 && cn < N) q.ntn[(size_t)(row < size ? nidL[row] : NX) * N + cn] = ntn_l[row * ST + col];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < NPTC; u++)
        if (nid[u] >= 0) q.cnt[s * NX + nid[u]] = cntv[u];
}


// ---------------------------------------------------------------------------
// k_pass_chain_blank: the chain kernel reduced to the compact-key loop, for passes
// in which EVERY step is blank (the replica pass of a fresh plan's first sweep:
// NumPartitions == 0, no node weights, partitions that hold no node of this or a
// lower priority state).  Few live values, no spills: one dependent step costs k
// wave minima and little else.  Anything outside its envelope sets flags[1] and
// changes nothing; the host then runs k_pass_chain from the same state.
// ---------------------------------------------------------------------------
template <int NPTC, int KM>
__global__ __launch_bounds__(64) void k_pass_chain_blank(ChainParams q) {
    BLANCE_DYN_LDS(lds);
    const int lane = threadIdx.x;
    const int rg = q.region_base + blockIdx.x;
    const int lo = q.reg_lo[rg], hi = q.reg_hi[rg], size = hi - lo;
    const int cbeg = q.seg_beg ? q.seg_beg[rg] : q.reg_off[rg], cend = q.seg_end ? q.seg_end[rg] : q.reg_off[rg + 1];
    if (cbeg >= cend) return;
    const int k = q.k;
    int* nidL = (int*)lds;                           // [size]
    int* recbuf = nidL + size;                       // [kChainStage][kCW]
    int* outbuf = recbuf + kChainStage * kCW;        // [kChainStage][OW] leaf indices, turned into node ids at the flush
    int nid[NPTC], cntv[NPTC], cls[NPTC];
    unsigned alive_m = 0;
    bool bad = size > 256 || q.NP != 0 || (q.ev_off && q.ev_off[rg] != q.ev_off[rg + 1]);
    int mx = 0;
#pragma unroll
    for (int u = 0; u < NPTC; u++) {
        const int pos = lo + lane + 64 * u;
        nid[u] = -2; cntv[u] = 0; cls[u] = -1;
        if (pos < hi) {
            const int n = q.leaf_node[pos];
            if (n >= 0) {
                nid[u] = n;
                cntv[u] = q.cnt[q.s * q.NX + n];
                if (q.node_has_weight[n]) bad = true;
                if (n < q.N && q.alive[n]) alive_m |= 1u << u;
                cls[u] = q.leaf_cls[pos];
                if (cntv[u] <= -kCompactMax / 2 || cntv[u] >= kCompactMax / 2) bad = true;
            }
            const int cs = q.cls_size[pos];          // sizes are stored per class index at reg_lo + c
            if (cs > mx) mx = cs;
            nidL[lane + 64 * u] = nid[u];
        }
    }
    __syncthreads();
    {   // node ids must rise with the leaf index; k classes must never cover the region
        int prev = -1;
        bool mono = true;
        if (lane == 0)
            for (int i = 0; i < size; i++) { const int n = nidL[i]; if (n >= 0) { if (n <= prev) mono = false; prev = n; } }
        if (!__builtin_amdgcn_readlane(mono ? 1 : 0, 0)) bad = true;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { int t = __shfl_xor(mx, off, 64); mx = t > mx ? t : mx; }
        if ((long long)mx * (k + 1) >= (long long)size) bad = true;
    }
    unsigned key[NPTC];
#pragma unroll
    for (int u = 0; u < NPTC; u++) {
        const unsigned c9 = cls[u] < 0 ? 511u : (unsigned)cls[u];
        key[u] = ((alive_m >> u) & 1) ? (((unsigned)(2 * cntv[u] + (1 << 14)) << 17) | ((unsigned)(lane + 64 * u) << 9) | c9)
                                      : kKeyNone;
    }
    bool failed = __ballot(bad) != 0;
    for (int base = cbeg; base < cend && !failed; base += kChainStage) {
      const int nbs = cend - base < kChainStage ? cend - base : kChainStage;     // steps staged this round
      for (int i = lane; i < nbs * kCW; i += 64) recbuf[i] = q.crec[(size_t)base * kCW + i];
      __syncthreads();
      for (int sub = 0; sub < nbs && !failed; sub += 64) {                         // 64 steps: lane r keeps step r's words
        const int nb = nbs - sub < 64 ? nbs - sub : 64;
        const bool active = lane < nb;
        const int* rp = recbuf + (sub + (active ? lane : 0)) * kCW;
        const int w0 = recbuf[sub * kCW + 1];
        const int tcv = rp[6];
        int hv[kChainHigh];
#pragma unroll
        for (int j = 0; j < kChainHigh; j++) hv[j] = rp[kCHigh + j];
        const bool blank = (rp[5] & 0xff00ff) == 0 && rp[1] == w0 && (tcv >= 0 || q.flat);
        if (__ballot(active && !blank) || w0 <= 0 || w0 >= 64) { failed = true; break; }
        const bool any_high = __ballot(active && (rp[5] & 0xff00) != 0) != 0;
        const unsigned bump = (unsigned)(2 * w0) << 17;
        // failure conditions are accumulated and tested once per batch: a bad step only
        // produces garbage that is thrown away with the whole launch
        unsigned none_acc = kKeyNone;                // becomes 0 if some minimum found no candidate
        int anchor_acc = 0;                          // sign bit set if some anchor had no exclude class
        int my_w[KM];                                // lane r keeps the picks of step r
#pragma unroll
        for (int c = 0; c < KM; c++) my_w[c] = 0;
        for (int r = 0; r < nb; r++) {
            int acls = __builtin_amdgcn_readlane(tcv, r);
            unsigned excl_m = 0;
            if (any_high) {                          // plan.go:146-154
#pragma unroll
                for (int j = 0; j < kChainHigh; j++) {
                    const int hj = __builtin_amdgcn_readlane(hv[j], r);
#pragma unroll
                    for (int u = 0; u < NPTC; u++) excl_m |= (hj == lane + 64 * u ? 1u : 0u) << u;
                }
            }
            unsigned picked_m = 0;
            int wl[KM];
#pragma unroll
            for (int slot = 0; slot < KM; slot++) {
                wl[slot] = 0;
                if (slot < k) {
                    if (!(q.flat && slot == 0)) anchor_acc |= acls;
#pragma unroll
                    for (int u = 0; u < NPTC; u++) excl_m |= (cls[u] == acls ? 1u : 0u) << u;    // (leaves without a class are no candidates)
                    unsigned km = kKeyNone;
#pragma unroll
                    for (int u = 0; u < NPTC; u++) {
                        const unsigned kv = ((excl_m >> u) & 1) ? kKeyNone : key[u];
                        km = kv < km ? kv : km;
                    }
                    const unsigned kb = wave_min_u32_bcast(km);
                    none_acc = ~kb < none_acc ? ~kb : none_acc;
#pragma unroll
                    for (int u = 0; u < NPTC; u++) picked_m |= (key[u] == kb ? 1u : 0u) << u;
                    wl[slot] = (int)((kb >> 9) & 0xff);
                    const int wc = (int)(kb & 0x1ff);
                    acls = wc == 511 ? -1 : wc;
                }
            }
#pragma unroll
            for (int u = 0; u < NPTC; u++) {
                const bool p = (picked_m >> u) & 1;
                key[u] += p ? bump : 0u;
                cntv[u] += p ? w0 : 0;
            }
#pragma unroll
            for (int c = 0; c < KM; c++) my_w[c] = lane == r ? wl[c] : my_w[c];
        }
        bool esc = none_acc == 0 || anchor_acc < 0;
#pragma unroll
        for (int u = 0; u < NPTC; u++) if (cntv[u] >= kCompactMax / 2) esc = true;   // a batch adds at most 64 * 63
        if (__ballot(esc)) { failed = true; break; }
        if (active) {
#pragma unroll
            for (int c = 0; c < KM; c++) if (c < k) outbuf[(sub + lane) * q.OW + 1 + c] = my_w[c];
        }
      }
      if (failed) break;
      __syncthreads();
      for (int i = lane; i < nbs * q.OW; i += 64) {
          const int c = i % q.OW;
          q.out[(size_t)base * q.OW + i] = c == 0 ? k : nidL[outbuf[i]];
      }
      __syncthreads();
    }
    if (failed) {
        if (lane == 0) q.flags[1] = 1;
        return;
    }
    // every region must succeed before any of them may publish its counters: publish to
    // the scratch copy; the host commits it when no chain failed
#pragma unroll
    for (int u = 0; u < NPTC; u++)
        if (nid[u] >= 0) q.cnt_out[q.s * q.NX + nid[u]] = cntv[u];
}

// ---------------------------------------------------------------------------
// k_pass_chain_planes: the all-blank pass (see k_pass_chain_blank) as a SCALAR automaton.
// With NumPartitions == 0 and no node weights a leaf's score is its count, and every pick adds the
// same weight w0: the leaves of a region are kept as bit planes in SGPRs -- plane j holds the
// live leaves whose count is base + j * w0, one bit per leaf in leaf order (= node id order,
// the tie-break of plan.go:617-628).  A pick is: first plane with a bit outside the exclude mask
// (s_andn2_b64 / s_or_b64), first such bit (s_ff1_i32_b64), move the bit one plane up.  No
// cross-lane reduction, no VALU on the dependent chain except the lane reads of the step's and
// the pick's exclude masks (v_readlane_b32).  The exclude masks of 64 steps are prepared lane
// parallel (lane r: step r), the records of the next 64 steps are fetched while this batch is walked.
// Envelope: <= 64 W leaves, every live leaf has an exclude class, counts within kPlanes levels of
// each other at any time, one partition weight for the whole chain.  Outside it: flags[1], nothing
// published (as k_pass_chain_blank).
// ---------------------------------------------------------------------------
constexpr int kPlanes = 4;

// One pick of the plane automaton: the lowest plane with a leaf outside E, its lowest such leaf, moved
// one plane up; MORE: the leaf's exclude class joins E for the step's next pick (lm: the class mask
// of every leaf, lane l of lm[u] = leaf 64 u + l).  Returns the leaf; negative: no candidate at
// all (-1) or the leaf left the planes (-2).
// planes_pick_from<J>: the general form, plane J upwards.
template <int W, int J, bool MORE>
__device__ __forceinline__ int planes_pick_from(unsigned long long (&P)[kPlanes][W], unsigned long long (&E)[W],
                                             const unsigned (&lm)[W][2 * W]) {
    typedef unsigned long long u64;
    u64 m[W], any = 0;
#pragma unroll
    for (int u = 0; u < W; u++) { m[u] = P[J][u] & ~E[u]; any |= m[u]; }
    if (any != 0) {
#pragma unroll
        for (int u = 0; u < W; u++) {
            if (u == W - 1 || m[u] != 0) {
                const int b = __builtin_ctzll(m[u]);
                const u64 bit = 1ull << b;
                P[J][u] ^= bit;
                if (J + 1 < kPlanes) P[J + 1][u] |= bit;
                if (MORE) {
#pragma unroll
                    for (int v = 0; v < W; v++)
                        E[v] |= (u64)(unsigned)__builtin_amdgcn_readlane((int)lm[u][2 * v], b) |
                                ((u64)(unsigned)__builtin_amdgcn_readlane((int)lm[u][2 * v + 1], b) << 32);
                }
                return J + 1 < kPlanes ? 64 * u + b : -2;
            }
        }
    }
    if constexpr (J + 1 < kPlanes) return planes_pick_from<W, J + 1, MORE>(P, E, lm);
    return -1;
}

#ifndef BLANCE_SIMT_EMU
// planes_walk_w2<K>: the steps of a batch for regions of up to 128 leaves (two 64-bit words per plane)
// as long as every pick finds its leaf on plane 0 or 1 -- the hand-scheduled scalar loop of this kernel.
// One wave issues an instruction every ~4.5 cycles and pays ~10 / ~25 cycles for a branch not taken /
// taken (measured, tools/dev_lat_micro.hip), so the loop is laid out by instruction count: the
// expected case (plane 0 has the leaf) falls through every rare-case branch, the two words of a plane
// are two code paths (one branch) instead of selects, the last pick's paths each carry their own loop
// tail, picks from plane 1 (a quarter of the steps at BASELINE config 3: the end of every round, when
// the lowest level is left in excluded racks only) sit behind the loop and jump back.
// Per step: 4 v_readlane (the step's exclude mask), per pick 3 + 4 scalar instructions, 4 v_readlane
// (the picked leaf's class mask) and 2 s_or unless it is the step's last pick, 1 v_writelane.
// Leaves the loop (a) at r == nb; (b) before a pick that has no candidate on planes 0 and 1, or finds
// plane 0 empty: `slot` = its index, E = the exclude mask so far -- the caller finishes that step with
// planes_pick_from, drops an empty plane 0, and comes back.
// (the step counter lives in M0 inside the loop: it is the lane select of every v_readlane / v_writelane of the step)
#define BLANCE_PL_HEAD                                           \
    "s_mov_b32 m0, %[r]\n"                                       \
    "0:\n\t"                                                     \
    "v_readlane_b32 s44, %[ex0], m0\n\t"                         \
    "v_readlane_b32 s45, %[ex1], m0\n\t"                         \
    "v_readlane_b32 s46, %[ex2], m0\n\t"                         \
    "v_readlane_b32 s47, %[ex3], m0\n\t"
// candidates of plane (PL, PH) in s[52:53] / s[54:55]; word 0 has one: fall through (s_andn2 leaves SCC = result != 0,
// so the expected case costs two instructions and a branch not taken); else to label W1, which checks word 1
#define BLANCE_PL_FIND(PL, PH, W1)                               \
    "s_andn2_b64 s[54:55], %[" #PH "], s[46:47]\n\t"             \
    "s_andn2_b64 s[52:53], %[" #PL "], s[44:45]\n\t"             \
    "s_cbranch_scc0 " W1 "\n\t"
// (at label W1) word 1 has a candidate: fall through; else to label NONE
#define BLANCE_PL_FIND_W1(NONE)                                  \
    "s_cmp_lg_u64 s[54:55], 0\n\t"                               \
    "s_cbranch_scc0 " NONE "\n\t"
// lowest candidate of word T (s56 = its bit index) leaves plane word FROM for plane word TO
#define BLANCE_PL_TAKE(T, FROM, TO)                              \
    "s_ff1_i32_b64 s56, " T "\n\t"                               \
    "s_lshl_b64 " T ", 1, s56\n\t"                               \
    "s_xor_b64 %[" #FROM "], %[" #FROM "], " T "\n\t"            \
    "s_or_b64 %[" #TO "], %[" #TO "], " T "\n\t"
// the picked leaf's exclude class joins the step's mask: its class mask read from the lanes (any classes) ...
#define BLANCE_PL_CLASS_0                                        \
    "v_readlane_b32 s48, %[lm00], s56\n\t"                       \
    "v_readlane_b32 s49, %[lm01], s56\n\t"                       \
    "v_readlane_b32 s50, %[lm02], s56\n\t"                       \
    "v_readlane_b32 s51, %[lm03], s56\n\t"                       \
    "s_or_b64 s[44:45], s[44:45], s[48:49]\n\t"                  \
    "s_or_b64 s[46:47], s[46:47], s[50:51]\n\t"
#define BLANCE_PL_CLASS_1                                        \
    "v_readlane_b32 s48, %[lm10], s56\n\t"                       \
    "v_readlane_b32 s49, %[lm11], s56\n\t"                       \
    "v_readlane_b32 s50, %[lm12], s56\n\t"                       \
    "v_readlane_b32 s51, %[lm13], s56\n\t"                       \
    "s_or_b64 s[44:45], s[44:45], s[48:49]\n\t"                  \
    "s_or_b64 s[46:47], s[46:47], s[50:51]\n\t"
// ... or made by arithmetic when every class is an aligned run of S = 2^e <= 64 leaves (racks of equal size): the run
// of the picked bit inside its own word -- 3 scalar instructions instead of 4 lane reads and 2 ors
#define BLANCE_PL_CLASS_A0                                       \
    "s_andn2_b32 s48, s56, %[sm1]\n\t"                           \
    "s_lshl_b64 s[48:49], %[sones], s48\n\t"                     \
    "s_or_b64 s[44:45], s[44:45], s[48:49]\n\t"
#define BLANCE_PL_CLASS_A1                                       \
    "s_andn2_b32 s48, s56, %[sm1]\n\t"                           \
    "s_lshl_b64 s[48:49], %[sones], s48\n\t"                     \
    "s_or_b64 s[46:47], s[46:47], s[48:49]\n\t"
#define BLANCE_PL_TAIL                                           \
    "s_add_u32 m0, m0, 1\n\t"                                    \
    "s_cmp_lt_u32 m0, %[nb]\n\t"                                 \
    "s_cbranch_scc1 0b\n\t"                                      \
    "s_branch 7f\n"
// both words' paths of a pick from plane (PL, PH) into (QL, QH); CONT: what follows the pick
#define BLANCE_PL_WORDS_MORE(SLOT, PL, PH, QL, QH, W1, NONE, CONT, C0, C1) \
    BLANCE_PL_TAKE("s[52:53]", PL, QL) C0                        \
    "v_writelane_b32 %[w" #SLOT "], s56, m0\n\t"                 \
    "s_branch " CONT "\n"                                        \
    W1 ":\n\t"                                                   \
    BLANCE_PL_FIND_W1(NONE)                                      \
    BLANCE_PL_TAKE("s[54:55]", PH, QH) C1                        \
    "s_or_b32 s56, s56, 64\n\t"                                  \
    "v_writelane_b32 %[w" #SLOT "], s56, m0\n\t"
#define BLANCE_PL_WORDS_LAST(SLOT, PL, PH, QL, QH, W1, NONE)     \
    BLANCE_PL_TAKE("s[52:53]", PL, QL)                           \
    "v_writelane_b32 %[w" #SLOT "], s56, m0\n\t"                 \
    BLANCE_PL_TAIL                                               \
    W1 ":\n\t"                                                   \
    BLANCE_PL_FIND_W1(NONE)                                      \
    BLANCE_PL_TAKE("s[54:55]", PH, QH)                           \
    "s_or_b32 s56, s56, 64\n\t"                                  \
    "v_writelane_b32 %[w" #SLOT "], s56, m0\n\t"                 \
    BLANCE_PL_TAIL
// in the loop: a pick from plane 0 (labels 2<slot>, 3<slot>); behind the loop: the same pick from plane 1.  A pick that
// is not the step's last carries, on its word-0 path, its OWN copy of the rest of the step (REST: it ends in loop tails
// and never falls through) instead of a taken branch over the word-1 path; the word-1 path and the plane-1 picks
// continue at 3<slot>.  (Numeric labels repeat across the copies: a reference binds to the nearest definition in its
// direction, and every copy of a continuation is the same code.)
#define BLANCE_PL_PICK_MORE(SLOT, C0, C1, REST)                  \
    BLANCE_PL_FIND(p0l, p0h, "2" #SLOT "f")                      \
    BLANCE_PL_TAKE("s[52:53]", p0l, p1l) C0                      \
    "v_writelane_b32 %[w" #SLOT "], s56, m0\n\t"                 \
    REST                                                         \
    "2" #SLOT ":\n\t"                                            \
    BLANCE_PL_FIND_W1("5" #SLOT "f")                             \
    BLANCE_PL_TAKE("s[54:55]", p0h, p1h) C1                      \
    "s_or_b32 s56, s56, 64\n\t"                                  \
    "v_writelane_b32 %[w" #SLOT "], s56, m0\n"                   \
    "3" #SLOT ":\n\t"
#define BLANCE_PL_PICK_LAST(SLOT)                                \
    BLANCE_PL_FIND(p0l, p0h, "2" #SLOT "f")                      \
    BLANCE_PL_WORDS_LAST(SLOT, p0l, p0h, p1l, p1h, "2" #SLOT, "5" #SLOT "f")
#define BLANCE_PL_SLOW_HEAD(SLOT)                                \
    "5" #SLOT ":\n\t"                                            \
    "s_or_b64 s[48:49], %[p0l], %[p0h]\n\t"                      \
    "s_cbranch_scc0 9" #SLOT "f\n\t"                             \
    BLANCE_PL_FIND(p1l, p1h, "6" #SLOT "f")
#define BLANCE_PL_SLOW_MORE(SLOT, C0, C1)                        \
    BLANCE_PL_SLOW_HEAD(SLOT)                                    \
    BLANCE_PL_WORDS_MORE(SLOT, p1l, p1h, p2l, p2h, "6" #SLOT, "9" #SLOT "f", "3" #SLOT "b", C0, C1) \
    "s_branch 3" #SLOT "b\n"                                     \
    "9" #SLOT ":\n\ts_mov_b32 %[slot], " #SLOT "\n\ts_branch 7f\n"
#define BLANCE_PL_SLOW_LAST(SLOT)                                \
    BLANCE_PL_SLOW_HEAD(SLOT)                                    \
    BLANCE_PL_WORDS_LAST(SLOT, p1l, p1h, p2l, p2h, "6" #SLOT, "9" #SLOT "f") \
    "9" #SLOT ":\n\ts_mov_b32 %[slot], " #SLOT "\n\ts_branch 7f\n"
#define BLANCE_PL_EXIT "7:\n\ts_mov_b32 %[r], m0\n\ts_mov_b64 %[elo], s[44:45]\n\ts_mov_b64 %[ehi], s[46:47]\n"
#define BLANCE_PL_OPERANDS                                                                                          \
    [p0l] "+s"(P[0][0]), [p0h] "+s"(P[0][1]), [p1l] "+s"(P[1][0]), [p1h] "+s"(P[1][1]), [p2l] "+s"(P[2][0]),         \
    [p2h] "+s"(P[2][1]), [r] "+s"(r), [slot] "=&s"(slot), [elo] "=&s"(E[0]), [ehi] "=&s"(E[1])
#define BLANCE_PL_INPUTS                                                                                            \
    [nb] "s"(nb), [ex0] "v"(ex[0]), [ex1] "v"(ex[1]), [ex2] "v"(ex[2]), [ex3] "v"(ex[3]),                           \
    [lm00] "v"(lm[0][0]), [lm01] "v"(lm[0][1]), [lm02] "v"(lm[0][2]), [lm03] "v"(lm[0][3]),                         \
    [lm10] "v"(lm[1][0]), [lm11] "v"(lm[1][1]), [lm12] "v"(lm[1][2]), [lm13] "v"(lm[1][3]),                         \
    [sm1] "s"(cls_m1), [sones] "s"(cls_ones)
#define BLANCE_PL_CLOBBER "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "m0", "scc"

// ARITH: the exclude classes are aligned runs of cls_m1 + 1 = 2^e leaves (cls_ones = that many one bits)
template <int K, bool ARITH>
__device__ __forceinline__ void planes_walk_w2(unsigned long long (&P)[kPlanes][2], unsigned long long (&E)[2], int& r, int nb,
                                               int& slot, const unsigned (&ex)[4], const unsigned (&lm)[2][4], int (&my_w)[K],
                                               int cls_m1_in, unsigned long long cls_ones_in) {
    // (wave-uniform by construction; the asm wants them in SGPRs)
    const int cls_m1 = __builtin_amdgcn_readfirstlane(cls_m1_in);
    const unsigned long long cls_ones = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)cls_ones_in) |
                                        ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(cls_ones_in >> 32)) << 32);
#define BLANCE_PL_R3(C0, C1) BLANCE_PL_PICK_LAST(3)
#define BLANCE_PL_R2(C0, C1) BLANCE_PL_PICK_MORE(2, C0, C1, BLANCE_PL_R3(C0, C1)) BLANCE_PL_R3(C0, C1)
#define BLANCE_PL_R1(C0, C1) BLANCE_PL_PICK_MORE(1, C0, C1, BLANCE_PL_R2(C0, C1)) BLANCE_PL_R2(C0, C1)
#define BLANCE_PL_BODY(C0, C1)                                                                                                        \
    if constexpr (K == 1) {                                                                                                           \
        asm volatile(BLANCE_PL_HEAD BLANCE_PL_PICK_LAST(0) BLANCE_PL_SLOW_LAST(0) BLANCE_PL_EXIT                                      \
                     : BLANCE_PL_OPERANDS, [w0] "+v"(my_w[0]) : BLANCE_PL_INPUTS : BLANCE_PL_CLOBBER);                                \
    } else if constexpr (K == 2) {                                                                                                    \
        asm volatile(BLANCE_PL_HEAD BLANCE_PL_PICK_MORE(0, C0, C1, BLANCE_PL_PICK_LAST(1)) BLANCE_PL_PICK_LAST(1)                     \
                     BLANCE_PL_SLOW_MORE(0, C0, C1) BLANCE_PL_SLOW_LAST(1) BLANCE_PL_EXIT                                             \
                     : BLANCE_PL_OPERANDS, [w0] "+v"(my_w[0]), [w1] "+v"(my_w[1]) : BLANCE_PL_INPUTS : BLANCE_PL_CLOBBER);            \
    } else if constexpr (K == 3) {                                                                                                    \
        asm volatile(BLANCE_PL_HEAD                                                                                                   \
                     BLANCE_PL_PICK_MORE(0, C0, C1, BLANCE_PL_PICK_MORE(1, C0, C1, BLANCE_PL_PICK_LAST(2)) BLANCE_PL_PICK_LAST(2))    \
                     BLANCE_PL_PICK_MORE(1, C0, C1, BLANCE_PL_PICK_LAST(2)) BLANCE_PL_PICK_LAST(2)                                    \
                     BLANCE_PL_SLOW_MORE(0, C0, C1) BLANCE_PL_SLOW_MORE(1, C0, C1) BLANCE_PL_SLOW_LAST(2) BLANCE_PL_EXIT              \
                     : BLANCE_PL_OPERANDS, [w0] "+v"(my_w[0]), [w1] "+v"(my_w[1]), [w2] "+v"(my_w[2])                                 \
                     : BLANCE_PL_INPUTS : BLANCE_PL_CLOBBER);                                                                         \
    } else {                                                                                                                          \
        asm volatile(BLANCE_PL_HEAD BLANCE_PL_PICK_MORE(0, C0, C1, BLANCE_PL_R1(C0, C1)) BLANCE_PL_R1(C0, C1)                         \
                     BLANCE_PL_SLOW_MORE(0, C0, C1) BLANCE_PL_SLOW_MORE(1, C0, C1)                                                    \
                     BLANCE_PL_SLOW_MORE(2, C0, C1) BLANCE_PL_SLOW_LAST(3) BLANCE_PL_EXIT                                             \
                     : BLANCE_PL_OPERANDS, [w0] "+v"(my_w[0]), [w1] "+v"(my_w[1]), [w2] "+v"(my_w[2]), [w3] "+v"(my_w[3])             \
                     : BLANCE_PL_INPUTS : BLANCE_PL_CLOBBER);                                                                         \
    }
    if constexpr (ARITH) { BLANCE_PL_BODY(BLANCE_PL_CLASS_A0, BLANCE_PL_CLASS_A1) }
    else { BLANCE_PL_BODY(BLANCE_PL_CLASS_0, BLANCE_PL_CLASS_1) }
#undef BLANCE_PL_BODY
#undef BLANCE_PL_R1
#undef BLANCE_PL_R2
#undef BLANCE_PL_R3
}
#endif

template <int W, int K, bool ARITH>
__global__ __launch_bounds__(64) void k_pass_chain_planes(ChainParams q) {
    BLANCE_DYN_LDS(lds);
    typedef unsigned long long u64;
    const int lane = threadIdx.x;
    const int rg = q.region_base + blockIdx.x;
    const int lo = q.reg_lo[rg], hi = q.reg_hi[rg], size = hi - lo;
    const int cbeg = q.seg_beg ? q.seg_beg[rg] : q.reg_off[rg], cend = q.seg_end ? q.seg_end[rg] : q.reg_off[rg + 1];
    if (cbeg >= cend) return;
    const int k = K;
    int* nidL = (int*)lds;                                 // [64 W] node id of a leaf
    unsigned* cmL = (unsigned*)(nidL + 64 * W);            // [64 W classes][2 W words] leaves of an exclude class
    bool bad = size > 64 * W || q.k != K || q.NP != 0 || q.flat != 0 || (q.ev_off && q.ev_off[rg] != q.ev_off[rg + 1]);
    for (int i = lane; i < 64 * W * 2 * W; i += 64) cmL[i] = 0;
    for (int i = lane; i < 64 * W; i += 64) nidL[i] = -2;
    __syncthreads();
    int nid[W], cntv[W], cls[W];
    unsigned alive_m = 0;
    int mx = 0;
#pragma unroll
    for (int u = 0; u < W; u++) {
        const int pos = lo + lane + 64 * u;
        nid[u] = -2; cntv[u] = 0; cls[u] = -1;
        if (pos < hi) {
            const int n = q.leaf_node[pos];
            if (n >= 0) {
                nid[u] = n;
                cntv[u] = q.cnt[q.s * q.NX + n];
                if (q.node_has_weight[n]) bad = true;
                if (n < q.N && q.alive[n]) alive_m |= 1u << u;
                cls[u] = q.leaf_cls[pos];
                if (cls[u] >= 64 * W) bad = true;
                if (cls[u] >= 0) atomicOr((int*)&cmL[cls[u] * 2 * W + 2 * u + (lane >> 5)], (int)(1u << (lane & 31)));
                else if ((alive_m >> u) & 1) bad = true;      // a live leaf without an exclude class
            }
            const int cs = q.cls_size[pos];                // sizes are stored per class index at reg_lo + c
            if (cs > mx) mx = cs;
            nidL[lane + 64 * u] = nid[u];
        }
    }
    __syncthreads();
    {   // node ids must rise with the leaf index; k classes must never cover the region
        int prev = -1;
        bool mono = true;
        if (lane == 0)
            for (int i = 0; i < size; i++) { const int n = nidL[i]; if (n >= 0) { if (n <= prev) mono = false; prev = n; } }
        if (!__builtin_amdgcn_readlane(mono ? 1 : 0, 0)) bad = true;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { int t = __shfl_xor(mx, off, 64); mx = t > mx ? t : mx; }
        if ((long long)mx * (k + 1) >= (long long)size) bad = true;
    }
    // the one partition weight of this chain, the lowest count, every live leaf's level above it
    const int w0 = q.crec[(size_t)cbeg * kCW + 1];
    int base_cnt = INT_MAX;
#pragma unroll
    for (int u = 0; u < W; u++) if (((alive_m >> u) & 1) && cntv[u] < base_cnt) base_cnt = cntv[u];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { int t = __shfl_xor(base_cnt, off, 64); base_cnt = t < base_cnt ? t : base_cnt; }
    if (w0 <= 0 || w0 > (1 << 20) || base_cnt == INT_MAX) bad = true;
    int level[W];
#pragma unroll
    for (int u = 0; u < W; u++) {
        level[u] = -1;
        if (((alive_m >> u) & 1) && !bad) {
            const long long dlt = (long long)cntv[u] - base_cnt;
            if (dlt % w0 != 0 || dlt / w0 >= kPlanes) bad = true;
            else level[u] = (int)(dlt / w0);
        }
    }
    if (__ballot(bad)) {
        if (lane == 0) q.flags[1] = 1;
        return;
    }
    // planes (wave-uniform: SGPRs), and for every leaf the mask of its exclude class
    u64 P[kPlanes][W];
#pragma unroll
    for (int j = 0; j < kPlanes; j++)
#pragma unroll
        for (int u = 0; u < W; u++) P[j][u] = __ballot(level[u] == j);
    unsigned lm[W][2 * W];
#pragma unroll
    for (int u = 0; u < W; u++)
#pragma unroll
        for (int x = 0; x < 2 * W; x++) lm[u][x] = cls[u] >= 0 ? cmL[cls[u] * 2 * W + x] : 0u;
    // ARITH (chosen by the host from the rule's class table, q.cls_run = S): the exclude classes are aligned runs of
    // S = 2^e <= 64 leaves, so a class mask is a shift, not a lane read.  Checked against the masks themselves.
    int cls_m1 = 0;
    u64 cls_ones = 0;
    if (ARITH) {
        const int S = q.cls_run;
        bool okc = W == 2 && S >= 1 && S <= 64 && (S & (S - 1)) == 0;
        cls_m1 = okc ? S - 1 : 0;
        cls_ones = !okc ? 0ull : S == 64 ? ~0ull : ((1ull << S) - 1);
#pragma unroll
        for (int u = 0; u < W; u++) {
            if (!((alive_m >> u) & 1)) continue;
            const u64 want = cls_ones << (lane & ~cls_m1);
#pragma unroll
            for (int v = 0; v < W; v++) {
                const u64 have = (u64)lm[u][2 * v] | ((u64)lm[u][2 * v + 1] << 32);
                if (have != (v == u ? want : 0ull)) okc = false;
            }
        }
        if (__ballot(!okc)) {
            if (lane == 0) q.flags[1] = 1;
            return;
        }
    }
    int shifts = 0;                                        // planes dropped below: plane j is level shifts + j
    bool failed = false;
    PH_DECL;
#ifdef BLANCE_PHASE_PROF
    int ph_exits = 0;
#endif
    // step records of a batch, lane r: step r -- weight, counts word, top's exclude class, higher priority leaves
    int rw, rc5, rtc, rhv[kChainHigh];
    {
        const int nb0 = cend - cbeg < 64 ? cend - cbeg : 64;
        const int* rp = q.crec + (size_t)(cbeg + (lane < nb0 ? lane : 0)) * kCW;
        rw = rp[1]; rc5 = rp[5]; rtc = rp[6];
#pragma unroll
        for (int j = 0; j < kChainHigh; j++) rhv[j] = rp[kCHigh + j];
    }
    for (int base = cbeg; base < cend; base += 64) {
        const int nb = cend - base < 64 ? cend - base : 64;
        const bool active = lane < nb;
        const int w = rw, c5 = rc5, tcv = rtc;
        int hv[kChainHigh];
#pragma unroll
        for (int j = 0; j < kChainHigh; j++) hv[j] = rhv[j];
        if (base + 64 < cend) {                            // the next batch's words travel while this one is walked
            const int nbn = cend - base - 64 < 64 ? cend - base - 64 : 64;
            const int* rp = q.crec + (size_t)(base + 64 + (lane < nbn ? lane : 0)) * kCW;
            rw = rp[1]; rc5 = rp[5]; rtc = rp[6];
#pragma unroll
            for (int j = 0; j < kChainHigh; j++) rhv[j] = rp[kCHigh + j];
        }
        const bool blank = (c5 & 0xff00ff) == 0 && w == w0 && tcv >= 0 && tcv < 64 * W;
        if (__ballot(active && !blank)) { failed = true; break; }
        // the step's exclude mask: its top priority node's class, its higher priority nodes (plan.go:146-154)
        unsigned ex[2 * W];
#pragma unroll
        for (int x = 0; x < 2 * W; x++) ex[x] = active ? cmL[tcv * 2 * W + x] : 0u;
        if (__ballot(active && (c5 & 0xff00) != 0)) {
#pragma unroll
            for (int j = 0; j < kChainHigh; j++) {
                const int h = hv[j];
#pragma unroll
                for (int x = 0; x < 2 * W; x++) ex[x] |= (h >= 0 && (h >> 5) == x) ? 1u << (h & 31) : 0u;
            }
        }
        int my_w[K];                                       // lane r keeps the picks of step r
#pragma unroll
        for (int c = 0; c < K; c++) my_w[c] = 0;
        int trouble = 0;                                   // sign bit: a pick found no candidate / left the planes
        int r = 0;
        PH(0);
        while (r < nb) {
            u64 E[W];
            int slot0 = 0;
            bool resumed = false;
#ifndef BLANCE_SIMT_EMU
            if constexpr (W == 2) {                        // the scalar loop; comes back where a pick needs more than plane 0
                planes_walk_w2<K, ARITH>(P, E, r, nb, slot0, ex, lm, my_w, cls_m1, cls_ones);
                PH(1);
                if (r >= nb) break;
                resumed = true;
#ifdef BLANCE_PHASE_PROF
                ph_exits++;
#endif
            }
#endif
            if (!resumed) {
#pragma unroll
                for (int u = 0; u < W; u++)
                    E[u] = (u64)(unsigned)__builtin_amdgcn_readlane((int)ex[2 * u], r) |
                           ((u64)(unsigned)__builtin_amdgcn_readlane((int)ex[2 * u + 1], r) << 32);
            }
#pragma unroll
            for (int slot = 0; slot < K; slot++) {
                if (slot >= slot0) {
                    // the next pick also avoids this pick's class (plan.go:185-212)
                    const int f = slot + 1 < K ? planes_pick_from<W, 0, true>(P, E, lm) : planes_pick_from<W, 0, false>(P, E, lm);
                    trouble |= f;
                    my_w[slot] = write_lane(f, r, my_w[slot]);
                }
            }
            u64 p0 = 0;
#pragma unroll
            for (int u = 0; u < W; u++) p0 |= P[0][u];
            if (p0 == 0) {                                 // nobody left on the lowest level: drop it
#pragma unroll
                for (int j = 0; j + 1 < kPlanes; j++)
#pragma unroll
                    for (int u = 0; u < W; u++) P[j][u] = P[j + 1][u];
#pragma unroll
                for (int u = 0; u < W; u++) P[kPlanes - 1][u] = 0;
                shifts++;
            }
            r++;
            PH(2);
        }
        if (trouble < 0 || shifts > (1 << 24)) { failed = true; break; }
        if (active) {
            int32_t* o = q.out + (size_t)(base + lane) * q.OW;
            o[0] = K;
#pragma unroll
            for (int c = 0; c < K; c++) o[1 + c] = nidL[my_w[c]];
        }
        PH(3);
    }
#ifdef BLANCE_PHASE_PROF
    if (blockIdx.x == 0 && threadIdx.x == 0) printf("[planes] %d steps, %d returns of the scalar loop, %d planes dropped\n", cend - cbeg, ph_exits, shifts);
#endif
    PH_DUMP(cend - cbeg);
    if (failed) {
        if (lane == 0) q.flags[1] = 1;
        return;
    }
    // every region must succeed before any of them may publish its counters: publish to
    // the scratch copy; the host commits it when no chain failed
#pragma unroll
    for (int u = 0; u < W; u++) {
        if (nid[u] < 0) continue;
        int c = cntv[u];
#pragma unroll
        for (int j = 0; j < kPlanes; j++)
            if ((P[j][u] >> lane) & 1) c = (int)((long long)base_cnt + (long long)(shifts + j) * w0);
        q.cnt_out[q.s * q.NX + nid[u]] = c;
    }
}

}  // namespace blance
