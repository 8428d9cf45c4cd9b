// Translation unit of k_pass_seq: the exact workgroup pass and its launch wrapper.
#include "dev_prelude.h"
#include "k_pass_seq.h"

namespace blance {

template <int T, int NPT, bool HIER, int KM>
static void launch_pass_v(hipStream_t stream, const PassParams& q, size_t lds) {
    auto kern = k_pass_seq<T, NPT, HIER, KM>;
    BLANCE_LAUNCH(kern, 1, T, lds, stream, q);
}

template <int T, int NPT>
static void launch_pass(hipStream_t stream, PassParams q, bool allow_spec) {
    size_t lds = sizeof(RedSlot) * 2 * (T / 64) + sizeof(double) * kLpTab + 64;
    // flat passes: LDS mirrors for the verified-stay speculation (k_pass_seq.h)
    const size_t mirrors = sizeof(int32_t) * (3 * (size_t)q.NX + 4) + 32;
    const bool rules = q.rule_begin < q.rule_end;
    q.spec = (!rules && allow_spec && lds + mirrors <= 150 * 1024) ? 1 : 0;
    if (q.spec) lds += mirrors;
    if (rules) {
        if (q.k <= 2) launch_pass_v<T, NPT, true, 2>(stream, q, lds);
        else launch_pass_v<T, NPT, true, kMaxK>(stream, q, lds);
    } else {
        if (q.k <= 2) launch_pass_v<T, NPT, false, 2>(stream, q, lds);
        else launch_pass_v<T, NPT, false, kMaxK>(stream, q, lds);
    }
}

int launch_pass_seq(hipStream_t stream, PassParams q, int force_threads, bool allow_spec) {
    // T threads own NPT nodes each (register resident); one workgroup runs the pass.  A step is a
    // chain of dependent instructions in every wave (about 11 cycles each, measured).
    const int NX = q.NX > 0 ? q.NX : 1;
    // measured per general step (us): 1,024 nodes: 2.25 with 256 x 4, 1.93 with 512 x 2; 4,096 nodes: 3.43 with
    // 512 x 8, 2.92 with 1024 x 4 -- two waves per SIMD hide each other's latency, more nodes per thread cost more
    int T = force_threads;
    if (T != 64 && T != 256 && T != 512 && T != 1024) T = NX <= 256 ? 64 : (NX <= 1024 ? 512 : 1024);
    if (T == 64 && NX > 256) T = 256;
    if ((T == 256 || T == 512) && NX > 1024) T = 1024;
    const int npt = (NX + T - 1) / T;
    if (T == 64) {
        if (npt <= 1) launch_pass<64, 1>(stream, q, allow_spec);
        else launch_pass<64, 4>(stream, q, allow_spec);
    } else if (T == 256) {
        if (npt <= 1) launch_pass<256, 1>(stream, q, allow_spec);
        else launch_pass<256, 4>(stream, q, allow_spec);
    } else if (T == 512) {
        launch_pass<512, 2>(stream, q, allow_spec);
    } else {
        if (npt <= 2) launch_pass<1024, 2>(stream, q, allow_spec);
        else if (npt <= 4) launch_pass<1024, 4>(stream, q, allow_spec);
        else if (npt <= 8) launch_pass<1024, 8>(stream, q, allow_spec);
        else return -1;
    }
    return 0;
}

}  // namespace blance
