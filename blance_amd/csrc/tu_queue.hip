// Translation unit of k_pass_queue: flat passes (k <= 2) on one wave64, the candidates as a sorted window over the lanes.
#include "dev_prelude.h"
#include "k_pass_queue.h"

namespace blance {

size_t queue_bits_words(int NX) {
    const size_t NXp = (size_t)((NX + 63) / 64) * 64, BW = ((NXp >> 5) + 3) & ~(size_t)3;
    return (size_t)(NX + 1) * BW;
}

bool launch_pass_queue(hipStream_t stream, PassParams q) {
    if (q.rule_begin < q.rule_end || q.NX > kQueueMaxNodes || q.NX < 1 || q.k < 1 || q.k > 2 || !q.stop) return false;
    if (q.NP > 0 && !q.ntn_bits) return false;
    const size_t lds = queue_lds_bytes(q.NX, q.RW);
    if (lds > 160 * 1024) return false;
    auto kern = k_pass_queue<2>;
    // one workgroup: the walking wave and its helper waves
    BLANCE_LAUNCH(kern, 1, 64 * kQueueWaves, lds, stream, q);
    return true;
}

}  // namespace blance
