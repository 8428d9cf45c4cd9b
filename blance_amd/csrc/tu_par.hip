// Translation unit of k_pass_par: flat passes whose steps are resolved in parallel by the lanes of one wave64.
#include "dev_prelude.h"
#include "k_pass_tree.h"
#include "k_pass_par.h"

namespace blance {

bool launch_pass_par(hipStream_t stream, PassParams q) {
    if (q.rule_begin < q.rule_end || q.NX > kTreeMaxNodes || q.NX < 1 || q.k < 1 || q.k > 2 || !q.stop_at) return false;
    const size_t lds = par_lds_bytes(q.NX, q.RW);
    if (lds > 160 * 1024) return false;
    q.spec = getenv("BLANCE_PAR_STAY") ? 8 : 0;      // developer knob: never give the pass up
    auto kern = k_pass_par<2>;
    BLANCE_LAUNCH(kern, 1, 64, lds, stream, q);
    return true;
}

}  // namespace blance
