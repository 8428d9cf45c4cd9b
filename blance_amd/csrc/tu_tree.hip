// Translation unit of k_pass_tree: flat passes on one wave64 with bound-ordered candidates.
#include "dev_prelude.h"
#include "k_pass_tree.h"

namespace blance {

bool launch_pass_tree(hipStream_t stream, PassParams q, int knobs) {
    if (q.rule_begin < q.rule_end || q.NX > kTreeMaxNodes || q.NX < 1 || q.k < 1 || q.k > 4) return false;
    const size_t lds = tree_lds_bytes(q.NX, q.RW);
    if (lds > 160 * 1024) return false;
    q.spec = ((knobs & 1) ? 2 : 0) | ((knobs & 2) ? 4 : 0);   // test knobs: dense general steps, no short general steps
    if (q.k <= 2) { auto kern = k_pass_tree<2>; BLANCE_LAUNCH(kern, 1, 64, lds, stream, q); }
    else { auto kern = k_pass_tree<4>; BLANCE_LAUNCH(kern, 1, 64, lds, stream, q); }
    return true;
}

}  // namespace blance
