// k_period: an all-blank chain pass (k_pass_chain_planes) whose step records repeat with a period T.
// Part of blance_hip.hip (one translation unit); DESIGN.md section 4.1c.  On by default (options.reserved[2] & 256 or
// BLANCE_PERIODIC=0 turn it off).
#pragma once

namespace blance {

// ============================================================================
// The all-blank pass of a region is a deterministic automaton: its state is the
// region's load counters (NumPartitions == 0, no node weights: a leaf's score is
// its counter, plan.go:634-689), its input per step the words 1..23 of the step's
// compact record (weight, top priority node's exclude class, higher priority
// leaves: plan.go:146-154, :185-212) -- word 0, the step's index in pass order,
// is not read by the chain.  Scores are compared inside the region only, so a
// state is the same state after every live leaf's counter grew by the same d.
//
// If the records repeat with period T from the chain's first step up to step
// `limit`, and the counters after 2T steps are the counters after T steps plus a
// uniform d, then by induction step t >= 2T (t < limit) picks what step t - T
// picked: the chain is walked for 2T steps, the rest of the periodic stretch is
// copied data-parallel, its counters follow in closed form, and the chain kernel
// walks whatever lies behind `limit`.  BASELINE config 3: T = 128 (the primaries
// cycle through a zone), 32,768 steps per region -> 256 walked, 32,512 copied.
// Nothing is assumed: both conditions are checked on the device for every region,
// and a region that fails either is walked in full by the same kernel.
// ============================================================================
constexpr int kPeriodCap = 4096;                 // longest period looked for
constexpr int kPeriodMinRounds = 4;              // a region joins if its periodic stretch is at least this many periods

// per region arrays of the period buffer, B ints each
enum { kPT = 0, kPLimit, kPOk, kPD, kPBeg1, kPEnd1, kPBeg2, kPEnd2, kPBeg3, kPEnd3, kPWords };

__device__ __forceinline__ bool period_same_input(const int32_t* a, const int32_t* b) {
    bool same = true;
#pragma unroll
    for (int j = 1; j < kCW; j++) same = same && a[j] == b[j];
    return same;
}

// grid: gx workgroups per region over its steps: T[rg] = smallest t > 0 whose record equals the first step's
__global__ void k_period_find(int B, int gx, const int32_t* reg_off, const int32_t* crec, int32_t* pb) {
    const int rg = blockIdx.x / gx;
    const int cbeg = reg_off[rg], len = reg_off[rg + 1] - cbeg;
    const int t = (blockIdx.x % gx) * blockDim.x + threadIdx.x;
    if (t < 1 || t >= len || t > kPeriodCap) return;
    if (period_same_input(crec + (size_t)cbeg * kCW, crec + (size_t)(cbeg + t) * kCW)) atomicMin(pb + kPT * B + rg, t);
}

// limit[rg] = first step t >= T whose record differs from step t - T's (the chain's length if none).  One thread per 16 bytes
// of the region's records (a record is kCW / 4 = 6 of them; word 0 of a record, the step's pass index, does not count): the
// loads of a wave are one contiguous 1 KB, and the piece T records back is in the L2 (12 KB behind at T = 128) -- the 100 MB
// of config 3's records cross the HBM interface once.  (One thread per RECORD, each reading its 96 bytes, ran at 0.5 TB/s.)
__global__ void k_period_verify(int B, int gx, const int32_t* reg_off, const int32_t* crec, int32_t* pb) {
    static_assert(kCW % 4 == 0, "records are whole 16-byte pieces");
    constexpr int per = kCW / 4;
    const int rg = blockIdx.x / gx;
    const int cbeg = reg_off[rg], len = reg_off[rg + 1] - cbeg;
    const int T = pb[kPT * B + rg];
    const long long qi = (long long)(blockIdx.x % gx) * blockDim.x + threadIdx.x;
    const long long t = qi / per;
    if (T < 1 || T > kPeriodCap || t < T || t >= len) return;
    const int4* r4 = (const int4*)(crec + (size_t)cbeg * kCW);
    const int4 a = r4[qi], b = r4[qi - (long long)per * T];
    const bool same = a.y == b.y && a.z == b.z && a.w == b.w && (qi % per == 0 || a.x == b.x);
    if (!same) atomicMin(pb + kPLimit * B + rg, (int)t);
}

// test knob (BLANCE_PERIODIC_CUT=n): the periodic stretch ends after n steps at the latest -- any prefix of a periodic
// stretch is one; what lies behind is walked by the chain kernel (the path a chain with a non-periodic tail takes)
__global__ void k_period_clamp(int B, int n, int32_t* pb) {
    const int rg = blockIdx.x * blockDim.x + threadIdx.x;
    if (rg < B && pb[kPLimit * B + rg] > n) pb[kPLimit * B + rg] = n;
}

__global__ void k_period_init(int B, const int32_t* reg_off, int32_t* pb) {
    const int rg = blockIdx.x * blockDim.x + threadIdx.x;
    if (rg >= B) return;
    pb[kPT * B + rg] = INT_MAX;
    pb[kPLimit * B + rg] = reg_off[rg + 1] - reg_off[rg];
    pb[kPOk * B + rg] = 0;
    pb[kPD * B + rg] = INT_MIN;
}

// the first two segments: [0, T) and [T, 2T) of a region that joins, the whole chain and nothing of one that does not
__global__ void k_period_segments(int B, const int32_t* reg_off, int32_t* pb) {
    const int rg = blockIdx.x * blockDim.x + threadIdx.x;
    if (rg >= B) return;
    const int cbeg = reg_off[rg], cend = reg_off[rg + 1];
    const int T = pb[kPT * B + rg], limit = pb[kPLimit * B + rg];
    const bool joins = T >= 1 && T <= kPeriodCap && (long long)limit >= (long long)kPeriodMinRounds * T;
    pb[kPOk * B + rg] = joins ? 1 : 0;
    pb[kPBeg1 * B + rg] = cbeg;
    pb[kPEnd1 * B + rg] = joins ? cbeg + T : cend;
    pb[kPBeg2 * B + rg] = joins ? cbeg + T : cend;
    pb[kPEnd2 * B + rg] = joins ? cbeg + 2 * T : cend;
    pb[kPBeg3 * B + rg] = cend;
    pb[kPEnd3 * B + rg] = cend;
}

// are the counters after 2T steps those after T steps plus the same d on every live leaf?  Three small launches over
// the leaves (gx workgroups per region over its leaves): the largest difference, every difference against it, and
// per region the verdict with the third segment: behind `limit` if so, behind 2T if not.
__global__ void k_period_state_max(int B, int gx, int s, int N, int NX, const int32_t* reg_lo, const int32_t* reg_hi,
                                   const int32_t* leaf_node, const uint8_t* alive, const int32_t* cnt1, const int32_t* cnt2,
                                   int32_t* pb) {
    const int rg = blockIdx.x / gx;
    if (!pb[kPOk * B + rg]) return;
    const int pos = reg_lo[rg] + (blockIdx.x % gx) * blockDim.x + threadIdx.x;
    if (pos >= reg_hi[rg]) return;
    const int n = leaf_node[pos];
    if (n < 0) return;
    const int d = cnt2[s * NX + n] - cnt1[s * NX + n];
    if (n < N && alive[n]) atomicMax(pb + kPD * B + rg, d);
    else if (d != 0) atomicMin(pb + kPOk * B + rg, 0);      // (a leaf that is no candidate never moves)
}

__global__ void k_period_state_check(int B, int gx, int s, int N, int NX, const int32_t* reg_lo, const int32_t* reg_hi,
                                     const int32_t* leaf_node, const uint8_t* alive, const int32_t* cnt1, const int32_t* cnt2,
                                     int32_t* pb) {
    const int rg = blockIdx.x / gx;
    if (!pb[kPOk * B + rg]) return;
    const int pos = reg_lo[rg] + (blockIdx.x % gx) * blockDim.x + threadIdx.x;
    if (pos >= reg_hi[rg]) return;
    const int n = leaf_node[pos];
    if (n >= 0 && n < N && alive[n] && cnt2[s * NX + n] - cnt1[s * NX + n] != pb[kPD * B + rg]) atomicMin(pb + kPOk * B + rg, 0);
}

// (flags[1]: a chain of the walks so far had to escape -- it published nothing, the host will redo the whole pass with
// k_pass_chain from the saved counters: then nothing is copied either.  Without this a walk that escaped in its second
// period left the counters where the first period had put them, "the same d = 0 on every leaf" passed for a verdict, and
// the outputs of steps nobody had written were replicated and counted -- node ids out of whatever the buffer held.)
__global__ void k_period_verdict(int B, const int32_t* reg_off, const int32_t* flags, int32_t* pb) {
    const int rg = blockIdx.x * blockDim.x + threadIdx.x;
    if (rg >= B) return;
    const int cbeg = reg_off[rg], cend = reg_off[rg + 1];
    const int T = pb[kPT * B + rg], limit = pb[kPLimit * B + rg], d = pb[kPD * B + rg];
    const bool joined = T >= 1 && T <= kPeriodCap && (long long)limit >= (long long)kPeriodMinRounds * T;
    if (!joined) return;                            // (segment 1 was the whole chain, segments 2 and 3 are empty)
    const bool ok = pb[kPOk * B + rg] != 0 && d != INT_MIN && d >= 0 && flags[1] == 0;
    pb[kPOk * B + rg] = ok ? 1 : 0;
    pb[kPBeg3 * B + rg] = ok ? cbeg + limit : cbeg + 2 * T;
    pb[kPEnd3 * B + rg] = cend;
}

// grid: gx workgroups per region over its steps: step t in [2T, limit) emits what step T + (t - T) % T emitted
__global__ void k_period_replicate(int B, int gx, int OW, const int32_t* reg_off, const int32_t* pb, int32_t* out) {
    const int rg = blockIdx.x / gx;
    if (!pb[kPOk * B + rg]) return;
    const int cbeg = reg_off[rg];
    const int T = pb[kPT * B + rg], limit = pb[kPLimit * B + rg];
    const int t = 2 * T + (blockIdx.x % gx) * blockDim.x + threadIdx.x;
    if (t >= limit) return;
    const int32_t* src = out + (size_t)(cbeg + T + (t - T) % T) * OW;
    int32_t* dst = out + (size_t)(cbeg + t) * OW;
    for (int j = 0; j < OW; j++) dst[j] = src[j];
}

// one workgroup per region: the counters behind the copied stretch -- d per full period on every live leaf,
// and the picks of the stretch's last, partial period one by one
__global__ void k_period_counts(int B, int s, int N, int NX, int OW, const int32_t* reg_off, const int32_t* reg_lo,
                                const int32_t* reg_hi, const int32_t* leaf_node, const uint8_t* alive, const int32_t* crec,
                                const int32_t* out, const int32_t* pb, int32_t* cnt) {
    const int rg = blockIdx.x;
    if (!pb[kPOk * B + rg]) return;
    const int cbeg = reg_off[rg];
    const int T = pb[kPT * B + rg], limit = pb[kPLimit * B + rg], d = pb[kPD * B + rg];
    const int copied = limit - 2 * T, full = copied / T, rest = copied % T;
    const int lo = reg_lo[rg], hi = reg_hi[rg];
    for (int pos = lo + (int)threadIdx.x; pos < hi; pos += blockDim.x) {
        const int n = leaf_node[pos];
        if (n >= 0 && n < N && alive[n]) cnt[s * NX + n] += d * full;
    }
    __syncthreads();
    const int w0 = crec[(size_t)cbeg * kCW + 1];
    for (int j = threadIdx.x; j < rest; j += blockDim.x) {
        const int32_t* o = out + (size_t)(cbeg + T + j) * OW;
        const int n_out = o[0] & 0xffff;
        for (int c = 0; c < n_out; c++) {
            const int n = o[1 + c];
            if (n >= 0 && n < NX) atomicAdd(cnt + s * NX + n, w0);
        }
    }
}

}  // namespace blance
