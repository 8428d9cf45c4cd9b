// Developer / test aid, NOT a planner: a shared library with the entry points of include/blance_hip.h whose blance_plan()
// fills the result with a synthetic, well-formed map (state m of partition p gets constraints[m] distinct nodes by a fixed
// formula; 3 sweeps, converged).  It exists so that the HOST side of the C++ mirror -- interning, un-interning, the stores into
// the caller's maps (blance_amd/csrc/host/blance_api.cpp) -- can be timed and stress-tested at a million partitions on a
// machine without a GPU (the emulated kernels would take hours at that size).  Nothing in the product loads it.
//   g++ -O2 -shared -fPIC -o /tmp/libblance_stub.so tests/tools/stub_plan_lib.cpp
//   blance_amd/lib/blance_host_cli /tmp/libblance_stub.so bench 3
#include <stdint.h>
#include <string.h>

#include "../../include/blance_hip.h"

extern "C" {

int blance_abi_version(void) { return BLANCE_ABI_VERSION; }
int blance_is_emulated(void) { return 1; }
const char* blance_last_error(void) { return "stub"; }
int blance_validate(const blance_problem*) { return BLANCE_OK; }

int64_t blance_result_capacity(const blance_problem* pb) {
    int64_t k = 0;
    for (int m = 0; m < pb->n_states; m++) k += pb->state_constraints[m] > 0 ? pb->state_constraints[m] : 0;
    return (int64_t)pb->n_parts * k + 1;
}

int blance_ctx_create(const blance_options*, blance_ctx** out) {
    *out = (blance_ctx*)new int(1);
    return BLANCE_OK;
}
void blance_ctx_destroy(blance_ctx* c) { delete (int*)c; }

int blance_plan(blance_ctx*, const blance_problem* pb, blance_result* res) {
    const int P = pb->n_parts, M = pb->n_states, N = pb->n_nodes;
    int64_t at = 0;
    res->out_off[0] = 0;
    for (int p = 0; p < P; p++) {
        int taken = 0;
        for (int m = 0; m < M; m++) {
            const int k = pb->state_constraints[m] > 0 ? pb->state_constraints[m] : 0;
            for (int j = 0; j < k && taken < N; j++, taken++) res->out_nodes[at++] = (int32_t)(((int64_t)p * 7 + taken * 131) % N);
            res->out_kind[(size_t)p * M + m] = k > 0 ? BLANCE_LIST_SET : BLANCE_LIST_ABSENT;
            res->out_off[(size_t)p * M + m + 1] = (int32_t)at;
        }
    }
    res->n_warnings = 0;
    res->iterations = 3;
    res->converged = 1;
    res->device_ms = 0.0;
    res->total_ms = 0.0;
    return BLANCE_OK;
}

}  // extern "C"
