// Translation unit of k_pass_pool: flat passes on one wave64 whose lanes hold the pool of the smallest nodes.
#include "dev_prelude.h"
#include "k_pass_tree.h"
#include "k_pass_par.h"
#include "k_pass_pool.h"

namespace blance {

bool launch_pass_pool(hipStream_t stream, PassParams q) {
    if (q.rule_begin < q.rule_end || q.NX > kTreeMaxNodes || q.NX < 1 || q.k < 1 || q.k > 2 || !q.stop_at) return false;
    const size_t lds = pool_lds_bytes(q.NX, q.RW);
    if (lds > 160 * 1024) return false;
    auto kern = k_pass_pool<2>;
    BLANCE_LAUNCH(kern, 1, 64, lds, stream, q);
    return true;
}

}  // namespace blance
