// TEST INFRASTRUCTURE ONLY -- a minimal SIMT emulator so that the kernels of
// blance_amd/csrc/blance_hip.hip can be executed on a machine without a GPU.
//
// This is not a product path and not a portability layer: the product is built
// by hipcc for gfx950 only and fails loudly without a device.  The emulator
// exists because the development container has no GPU; it lets `-m "not gpu"`
// tests run the *same kernel source* (thread-for-thread, barrier-for-barrier)
// against the oracle before GPU minutes are spent.  Each GPU thread of a
// barrier-using kernel is a fiber; __syncthreads() parks it until the whole
// workgroup arrived; wave64 cross-lane builtins exchange through a per-wave
// scratch area.  Single OS thread, deterministic, only meant for tiny problems.
#pragma once
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#define BLANCE_SIMT_EMU 1

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__
#define HIP_KERNEL_NAME(...) __VA_ARGS__

struct int4 { int x, y, z, w; };                    // (used for 16-byte copies only)

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};

namespace emu {
// GPU threads of one workgroup are fibers of the calling OS thread,
// resumed round-robin; a barrier parks the fiber until its generation advances.
constexpr size_t kStackBytes = 256 * 1024;
struct Barrier { int expected = 0, arrived = 0; unsigned gen = 0; };
// minimal x86-64 context switch (callee-saved registers + stack pointer; defined in
// emu_lib.cpp) -- ucontext's swapcontext costs two sigprocmask system calls per switch
extern "C" void emu_switch(void** save_sp, void* new_sp);

struct Fiber {
    void* sp = nullptr;
    unsigned char* stack = nullptr;       // from a pool reused across launches (never zero-filled)
    bool done = false;
    Barrier* wait_bar = nullptr;
    unsigned wait_gen = 0;
    dim3 tid;
};
struct Block {
    int n_threads = 0;
    Barrier bar;
    std::vector<Barrier> wave_bar;
    std::vector<uint64_t> xch;      // [n_waves][64] exchange slots
    unsigned char* lds = nullptr;
    std::vector<Fiber> fibers;
    void* sched_sp = nullptr;
    int cur = -1;
    std::function<void()> body;
};
// one launch at a time: several contexts may be driven from several host threads (tests/test_dist.py), the
// emulator's state is global -- launches take g_launch_mu (plain globals keep threadIdx a load, not a TLS call)
extern Block* t_block;
extern dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
extern std::mutex g_launch_mu;

inline void barrier_wait(Barrier& b) {
    Block* blk = t_block;
    if (b.expected <= 1) return;
    if (++b.arrived == b.expected) { b.arrived = 0; b.gen++; return; }
    Fiber& f = blk->fibers[blk->cur];
    f.wait_bar = &b;
    f.wait_gen = b.gen;
    emu_switch(&f.sp, blk->sched_sp);
}
inline void block_barrier() { if (t_block) barrier_wait(t_block->bar); }
inline void wave_barrier() { if (t_block) barrier_wait(t_block->wave_bar[t_threadIdx.x >> 6]); }
// all lanes publish v, then read lane `src`
inline uint64_t wave_exchange(uint64_t v, int src) {
    if (!t_block) return v;                       // serial (no-sync) kernel: single lane semantics
    int w = t_threadIdx.x >> 6, l = t_threadIdx.x & 63;
    uint64_t* x = &t_block->xch[(size_t)w * 64];
    x[l] = v;
    wave_barrier();
    uint64_t r = x[src & 63];
    wave_barrier();
    return r;
}
inline void fiber_entry() {
    Block* blk = t_block;
    blk->body();
    Fiber& f = blk->fibers[blk->cur];
    f.done = true;
    emu_switch(&f.sp, blk->sched_sp);
    abort();
}

template <class F>
void launch_threads(dim3 grid, dim3 block, size_t lds_bytes, F body) {
    std::lock_guard<std::mutex> launch_lock(g_launch_mu);
    t_blockDim = block; t_gridDim = grid;
    for (unsigned b = 0; b < grid.x; b++) {
        Block blk;
        blk.n_threads = (int)block.x;
        int n_waves = (int)((block.x + 63) / 64);
        blk.bar.expected = (int)block.x;
        blk.wave_bar.resize(n_waves);
        for (int w = 0; w < n_waves; w++) {
            int lanes = (int)block.x - w * 64;
            blk.wave_bar[w].expected = lanes > 64 ? 64 : lanes;
        }
        blk.xch.assign((size_t)n_waves * 64, 0);
        std::vector<unsigned char> lds(lds_bytes + 64, 0);
        blk.lds = lds.data();
        blk.body = body;
        blk.fibers.resize(block.x);
        t_block = &blk;
        t_blockIdx = dim3(b);
        for (unsigned t = 0; t < block.x; t++) {
            Fiber& f = blk.fibers[t];
            static std::vector<unsigned char*> pool;
            if (pool.size() <= t) pool.resize(t + 1, nullptr);
            if (!pool[t]) pool[t] = (unsigned char*)malloc(kStackBytes);
            f.stack = pool[t];
            f.tid = dim3(t);
            // initial frame: six callee-saved registers, then the entry point as return address
            uintptr_t top = ((uintptr_t)f.stack + kStackBytes) & ~(uintptr_t)15;
            void** frame = (void**)(top - 64);
            for (int r = 0; r < 6; r++) frame[r] = nullptr;
            frame[6] = (void*)fiber_entry;
            frame[7] = nullptr;
            f.sp = frame;
        }
        int remaining = (int)block.x;
        while (remaining > 0) {
            bool progressed = false;
            for (unsigned t = 0; t < block.x; t++) {
                Fiber& f = blk.fibers[t];
                if (f.done) continue;
                if (f.wait_bar) {
                    if (f.wait_bar->gen == f.wait_gen) continue;
                    f.wait_bar = nullptr;
                }
                blk.cur = (int)t;
                t_threadIdx = f.tid;
                emu_switch(&blk.sched_sp, f.sp);
                progressed = true;
                if (f.done) remaining--;
            }
            if (!progressed) { fprintf(stderr, "emu: deadlock (divergent barrier)\n"); abort(); }
        }
        t_block = nullptr;
    }
}

template <class F>
void launch_serial(dim3 grid, dim3 block, F body) {
    std::lock_guard<std::mutex> launch_lock(g_launch_mu);
    t_block = nullptr;
    t_blockDim = block; t_gridDim = grid;
    for (unsigned b = 0; b < grid.x; b++)
        for (unsigned t = 0; t < block.x; t++) {
            t_threadIdx = dim3(t); t_blockIdx = dim3(b);
            body();
        }
}
}  // namespace emu

#define threadIdx emu::t_threadIdx
#define blockIdx emu::t_blockIdx
#define blockDim emu::t_blockDim
#define gridDim emu::t_gridDim

#define BLANCE_DYN_LDS(ptr) unsigned char* ptr = emu::t_block->lds
#define BLANCE_LAUNCH(kern, grid, block, lds, stream, ...) \
    emu::launch_threads(dim3(grid), dim3(block), (size_t)(lds), [&]() { kern(__VA_ARGS__); })
#define BLANCE_LAUNCH_NOSYNC(kern, grid, block, lds, stream, ...) \
    emu::launch_serial(dim3(grid), dim3(block), [&]() { kern(__VA_ARGS__); })

inline void __syncthreads() { emu::block_barrier(); }
#define BLANCE_WAIT_VMEM() ((void)0)
#define BLANCE_WAVE_SYNC() emu::wave_barrier()   /* fibers are not in lockstep: model it */

inline int __shfl_xor(int v, int mask, int = 64) {
    return (int)(uint32_t)emu::wave_exchange((uint32_t)v, (threadIdx.x & 63) ^ mask);
}
inline double __shfl_xor(double v, int mask, int = 64) {
    uint64_t u;
    memcpy(&u, &v, 8);
    u = emu::wave_exchange(u, (threadIdx.x & 63) ^ mask);
    memcpy(&v, &u, 8);
    return v;
}
inline int __shfl_up(int v, int delta, int = 64) {
    int l = threadIdx.x & 63;
    int r = (int)(uint32_t)emu::wave_exchange((uint32_t)v, l >= delta ? l - delta : l);
    return r;
}
inline int __shfl(int v, int src, int = 64) { return (int)(uint32_t)emu::wave_exchange((uint32_t)v, src); }
inline int __builtin_amdgcn_readlane(int v, int lane) { return (int)(uint32_t)emu::wave_exchange((uint32_t)v, lane); }
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }   // only used on wave-uniform values
// DPP lane selects used by wave_argmin: quad_perm (ctrl < 0x100), row_half_mirror, row_mirror
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int, bool) {
    int l = threadIdx.x & 63, from;
    bool write = (row_mask >> (l >> 4)) & 1;          // rows not in row_mask keep `old`
    if (ctrl < 0x100) from = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);
    else if (ctrl == 0x141) from = (l & ~7) | (7 - (l & 7));
    else if (ctrl == 0x140) from = (l & ~15) | (15 - (l & 15));
    else if (ctrl == 0x142) { from = (l & ~15) - 1; if (from < 0) { from = l; write = false; } }   // row_bcast:15
    else if (ctrl == 0x143) { from = 31; if (l < 32) write = false; }                             // row_bcast:31
    else if (ctrl == 0x138) { from = l - 1; if (from < 0) { from = l; write = false; } }            // wave_shr:1: lane i <- lane i - 1
    else if (ctrl == 0x130) { from = l + 1; if (from > 63) { from = l; write = false; } }           // wave_shl:1: lane i <- lane i + 1
    else { fprintf(stderr, "emu: dpp ctrl %x not modelled\n", ctrl); abort(); }
    int got = (int)(uint32_t)emu::wave_exchange((uint32_t)src, from);
    return write ? got : old;
}
inline unsigned long long __ballot(int pred) {
    if (!emu::t_block) return pred ? 1ull : 0ull;
    int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    uint64_t* x = &emu::t_block->xch[(size_t)w * 64];
    x[l] = pred ? 1 : 0;
    emu::wave_barrier();
    unsigned long long m = 0;
    int lanes = emu::t_block->wave_bar[w].expected;
    for (int i = 0; i < lanes; i++) if (x[i]) m |= 1ull << i;
    emu::wave_barrier();
    return m;
}
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline long long __double_as_longlong(double d) { long long x; memcpy(&x, &d, 8); return x; }

inline double __longlong_as_double(long long x) { double d; memcpy(&d, &x, 8); return d; }
inline double __hiloint2double(int hi, int lo) {
    uint64_t u = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
    double d; memcpy(&d, &u, 8); return d;
}
inline int __double2loint(double d) { uint64_t u; memcpy(&u, &d, 8); return (int)(uint32_t)u; }
inline int __double2hiint(double d) { uint64_t u; memcpy(&u, &d, 8); return (int)(uint32_t)(u >> 32); }

inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline int atomicOr(int* p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline unsigned long long atomicOr(unsigned long long* p, unsigned long long v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline int atomicMin(int* p, int v) {
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
inline int atomicMax(int* p, int v) {
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v > old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}

// ---- host runtime: device memory is host memory -----------------------------
typedef int hipError_t;
typedef void* hipStream_t;
struct EmuEvent { std::chrono::steady_clock::time_point t; };
typedef EmuEvent* hipEvent_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); return *p ? hipSuccess : 1; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n) { *p = malloc(n); return *p ? hipSuccess : 1; }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }      // (launches run when enqueued: in order)
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new EmuEvent(); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
