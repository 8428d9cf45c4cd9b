// Translation unit of k_pass_chain / k_pass_chain_blank: region chains and their launch wrappers.
#include "dev_prelude.h"
#include "k_pass_chain.h"
#include "k_stay.h"

namespace blance {

template <int NPTC, int KM, bool FAST>
static void launch_chain_v(hipStream_t stream, const ChainParams& q, size_t lds, int waves) {
    auto kern = k_pass_chain<NPTC, KM, FAST>;
    // (the walking wave and its helpers for the stay test, k_pass_chain.h; never more than the instance's launch bounds)
    if (waves > chain_waves_max<NPTC>()) waves = chain_waves_max<NPTC>();
    BLANCE_LAUNCH(kern, q.n_launch, 64 * waves, lds, stream, q);
}

template <int NPTC, int KM>
static void launch_chain_mode(hipStream_t stream, const ChainParams& q, size_t lds, bool fast, int waves) {
    if (fast) launch_chain_v<NPTC, KM, true>(stream, q, lds, waves);
    else launch_chain_v<NPTC, KM, false>(stream, q, lds, waves);
}

// (a stage = 64 steps per wave: records and outputs; a table of top priority nodes per wave)
static size_t chain_lds_base(const ChainParams& q, int max_size, int waves) {
    return sizeof(double) * (kLpTab + kFfTab + (size_t)max_size) + sizeof(int32_t) * 7 * (size_t)max_size +
           sizeof(int32_t) * 64 * waves * (size_t)(kCW + q.OW) + sizeof(int32_t) * (waves * ((size_t)max_size + 1) + kChainCtl) + 64;
}

static bool chain_rows_fit(const ChainParams& q, int max_size, int waves) {
    const size_t ntn_bytes = sizeof(int32_t) * (size_t)(max_size + 1) * (max_size + 1);
    return ntn_bytes <= 100 * 1024 && chain_lds_base(q, max_size, waves) + ntn_bytes <= 156 * 1024;
}

// the region's nodeToNodeCounts rows live in LDS when they fit beside the rest (160 KB per CU; beside the four-wave
// workgroup's smaller stage if not beside the eight-wave one's); flat mode may insist on global rows
bool chain_rows_in_lds(const ChainParams& q, int max_size) {
    if (!q.flat || q.ntn_in_lds) return chain_rows_fit(q, max_size, 4);
    return false;
}

// waves of a region's workgroup: eight (rounds of 512 steps) unless the rows then no longer fit in LDS; q.waves = 4 / 8 insists
static int chain_waves(const ChainParams& q, int max_size) {
    if (q.waves == 4 || q.waves == kChainWaves) return q.waves;
    const bool rows = q.NP > 0 && chain_rows_in_lds(q, max_size);
    return (!rows || chain_rows_fit(q, max_size, kChainWaves)) ? kChainWaves : 4;
}

// one wave64 per region; lanes own NPTC leaves each, k <= KM picks per step
bool launch_chain(hipStream_t stream, ChainParams& q, int max_size, bool fast) {
    int nptc = (max_size + 63) / 64;
    size_t ntn_bytes = sizeof(int32_t) * (size_t)(max_size + 1) * (max_size + 1);
    q.ntn_in_lds = chain_rows_in_lds(q, max_size);
    int waves = chain_waves(q, max_size);
    if (q.NP > 0 && q.ntn_in_lds && !chain_rows_fit(q, max_size, waves)) waves = 4;      // (insisted on eight: not at the rows' expense)
    if (nptc > 4) waves = 4;                         // (the <8, ..> instances, chain_waves_max: such a wave needs more than half a SIMD's registers)
    size_t lds = chain_lds_base(q, max_size, waves);
    if (q.NP > 0 && q.ntn_in_lds) lds += ntn_bytes;
    if (q.k <= 2) {
        if (nptc <= 2) launch_chain_mode<2, 2>(stream, q, lds, fast, waves);
        else if (nptc <= 4) launch_chain_mode<4, 2>(stream, q, lds, fast, waves);
        else if (nptc <= 8) launch_chain_mode<8, 2>(stream, q, lds, fast, waves);
        else return false;
    } else if (q.k <= 4) {
        if (nptc <= 2) launch_chain_mode<2, 4>(stream, q, lds, fast, waves);
        else if (nptc <= 4) launch_chain_mode<4, 4>(stream, q, lds, fast, waves);
        else if (nptc <= 8) launch_chain_mode<8, 4>(stream, q, lds, fast, waves);
        else return false;
    } else {
        return false;
    }
    return true;
}

void launch_chain_blank(hipStream_t stream, const ChainParams& q, int max_size) {
    const int nptc = (max_size + 63) / 64, B = q.n_launch;
    const size_t lds = sizeof(int32_t) * ((size_t)max_size + kChainStage * (size_t)(kCW + q.OW)) + 64;
    if (q.k <= 2) {
        if (nptc <= 2) { auto kern = k_pass_chain_blank<2, 2>; BLANCE_LAUNCH(kern, B, 64, lds, stream, q); }
        else { auto kern = k_pass_chain_blank<4, 2>; BLANCE_LAUNCH(kern, B, 64, lds, stream, q); }
    } else {
        if (nptc <= 2) { auto kern = k_pass_chain_blank<2, 4>; BLANCE_LAUNCH(kern, B, 64, lds, stream, q); }
        else { auto kern = k_pass_chain_blank<4, 4>; BLANCE_LAUNCH(kern, B, 64, lds, stream, q); }
    }
}


template <int W, bool ARITH>
static bool launch_planes_w(hipStream_t stream, const ChainParams& q, size_t lds) {
    const int B = q.n_launch;
    switch (q.k) {
    case 1: { auto kern = k_pass_chain_planes<W, 1, ARITH>; BLANCE_LAUNCH(kern, B, 64, lds, stream, q); return true; }
    case 2: { auto kern = k_pass_chain_planes<W, 2, ARITH>; BLANCE_LAUNCH(kern, B, 64, lds, stream, q); return true; }
    case 3: { auto kern = k_pass_chain_planes<W, 3, ARITH>; BLANCE_LAUNCH(kern, B, 64, lds, stream, q); return true; }
    case 4: { auto kern = k_pass_chain_planes<W, 4, ARITH>; BLANCE_LAUNCH(kern, B, 64, lds, stream, q); return true; }
    }
    return false;
}

bool launch_chain_planes(hipStream_t stream, const ChainParams& q, int max_size) {
    // two 64-bit words per plane: regions of up to 128 leaves (wider regions: k_pass_chain_blank)
    if (max_size > 128 || q.k > 4 || q.k < 1) return false;
    const size_t lds = sizeof(int32_t) * (64 * 2 + 64 * 2 * 2 * 2) + 64;
    return q.cls_run > 0 ? launch_planes_w<2, true>(stream, q, lds) : launch_planes_w<2, false>(stream, q, lds);
}


bool launch_stay_by_top(hipStream_t stream, const StayParams& q, int n_wgs, int max_size) {
    if (max_size > kStayMaxLeaves || q.k < 1 || q.k > 4 || n_wgs < 1) return false;
    // (a workgroup per kStayWaves top priority nodes, a wave each: kStaySplit workgroups per entry of the work table)
    const size_t lds = sizeof(int32_t) * ((size_t)max_size * (7 + kStayWaves) + kStayWaves + 2) + sizeof(double) * kStayWaves + 64;
    if (q.k <= 2) { auto kern = k_stay_by_top<2>; BLANCE_LAUNCH(kern, n_wgs * kStaySplit, 64 * kStayWaves, lds, stream, q); }
    else { auto kern = k_stay_by_top<4>; BLANCE_LAUNCH(kern, n_wgs * kStaySplit, 64 * kStayWaves, lds, stream, q); }
    return true;
}

}  // namespace blance
