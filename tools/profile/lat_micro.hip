// Developer aid: dependent-chain latencies of one wave64 on gfx950 (cycles per instruction by s_memtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
__global__ void k(unsigned long long* out, int iters) {
    unsigned long long t0, t1;
    int lane = threadIdx.x;
    // 1. s_add_u32 dependent
    unsigned s = (unsigned)iters;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) asm volatile(REP64("s_add_u32 %0, %0, 1\n") : "+s"(s) : : "scc", "vcc");
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[0] = t1 - t0;
    // 2. s_ff1 / s_lshl_b64 / s_xor_b64 dependent triple
    unsigned long long m = 0xfffffffffffffff0ull + s, bit; int f;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++)
        asm volatile(REP8("s_ff1_i32_b64 %1, %0\n s_lshl_b64 %2, 1, %1\n s_xor_b64 %0, %0, %2\n s_or_b64 %0, %0, 1\n s_andn2_b64 %2, %0, %2\n s_or_b64 %0, %0, %2\n s_cmp_lg_u64 %0, 0\n s_cselect_b32 %1, %1, 0\n") : "+s"(m), "=s"(f), "=s"(bit) : : "scc", "vcc");
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[1] = t1 - t0;
    // 3. v_readlane -> s_add -> v_writelane round trip
    int v = lane, sv = 0;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++)
        asm volatile(REP8("v_readlane_b32 %1, %0, 3\n s_add_u32 %1, %1, 1\n v_writelane_b32 %0, %1, 3\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n") : "+v"(v), "+s"(sv) : : "scc", "vcc");
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[2] = t1 - t0;
    // 4. v_readlane x4 independent then s_or of them
    int a = lane, b = lane * 3, c = lane * 5, d = lane * 7; int s0, s1, s2, s3;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++)
        asm volatile(REP8("v_readlane_b32 %4, %0, 5\n v_readlane_b32 %5, %1, 5\n v_readlane_b32 %6, %2, 5\n v_readlane_b32 %7, %3, 5\n s_or_b32 %4, %4, %5\n s_or_b32 %6, %6, %7\n s_or_b32 %4, %4, %6\n s_add_u32 %4, %4, 1\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : : "scc", "vcc");
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[3] = t1 - t0 + (s0 & 0);
    // 5. taken branch
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++)
        asm volatile(REP8("s_cmp_eq_u32 %0, %0\n s_cbranch_scc1 1f\n s_nop 0\n 1: s_cmp_eq_u32 %0, %0\n s_cbranch_scc1 2f\n s_nop 0\n 2: s_cmp_eq_u32 %0, %0\n s_cbranch_scc1 3f\n s_nop 0\n 3: s_cmp_eq_u32 %0, %0\n s_cbranch_scc1 4f\n s_nop 0\n 4:\n") : "+s"(s) : : "scc", "vcc");
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[4] = t1 - t0;
    // 6. not-taken branch
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++)
        asm volatile(REP8("s_cmp_lg_u32 %0, %0\n s_cbranch_scc1 1f\n 1: s_cmp_lg_u32 %0, %0\n s_cbranch_scc1 2f\n 2: s_cmp_lg_u32 %0, %0\n s_cbranch_scc1 3f\n 3: s_cmp_lg_u32 %0, %0\n s_cbranch_scc1 4f\n 4:\n") : "+s"(s) : : "scc", "vcc");
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[5] = t1 - t0;
    // 7. dependent v_min_u32 dpp chain (row_shr:1) with required nops
    unsigned x = lane * 2654435761u;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++)
        asm volatile(REP8("s_nop 1\n v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n s_nop 1\n v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n s_nop 1\n v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n") : "+v"(x) : : "scc", "vcc");
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[6] = t1 - t0 + (x & 0);
    // 8. dependent plain VALU v_add_u32
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) asm volatile(REP64("v_add_u32 %0, %0, 1\n") : "+v"(x) : : "scc", "vcc");
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[7] = t1 - t0 + (x & 0);
    // 9. s_andn2_b64 / s_or_b64 / s_cmp / s_cselect independent-ish pairs
    unsigned long long p = s, q2 = s * 3ull, r2;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++)
        asm volatile(REP8("s_andn2_b64 %2, %0, %1\n s_or_b64 %0, %0, %2\n s_andn2_b64 %2, %1, %0\n s_or_b64 %1, %1, %2\n s_andn2_b64 %2, %0, %1\n s_or_b64 %0, %0, %2\n s_andn2_b64 %2, %1, %0\n s_or_b64 %1, %1, %2\n") : "+s"(p), "+s"(q2), "=s"(r2) : : "scc", "vcc");
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[8] = t1 - t0 + (p & 0);
    if (lane == 0) out[9] = s + m + v + x + a + b + c + d + (unsigned)q2;
}
int main() {
    unsigned long long* d; hipMalloc(&d, 128);
    const int iters = 2000;
    unsigned long long h[16];
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k, 1, 64, 0, 0, d, iters); hipDeviceSynchronize(); }
    hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
    const char* name[] = {"s_add dependent", "salu 8-op pick-like group (per op)", "readlane->s_add->writelane(+5 nop) (per group of 8)", "4 readlane + 4 salu (per op)",
                          "taken branch pair (per cmp+branch)", "not-taken branch pair (per cmp+branch)", "dpp min stage incl s_nop (per stage)", "v_add dependent", "salu b64 dependent"};
    const double per[] = {64, 64, 8, 64, 32, 32, 32, 64, 64};
    // s_memtime: shader cycles
    for (int i = 0; i < 9; i++) printf("%-55s raw %llu  per-unit %.2f ticks\n", name[i], h[i], (double)h[i] / (iters * per[i]));
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("clock rate kHz %d (ticks = shader cycles: a dependent scalar add every 4.5 ticks)\n", clk);
    return 0;
}
