// call_arena.hpp: the region allocator and the global operator new / delete in front of it.
#include "call_arena.hpp"

#include <stdint.h>
#include <stdlib.h>
#include <sys/mman.h>

#include <atomic>
#include <mutex>
#include <new>

namespace blance {
namespace arena {
namespace {

constexpr size_t kChunk = 4u << 20;             // chunks are kChunk-aligned: a pointer's chunk header is a mask away
constexpr size_t kMaxSmall = kChunk / 4;        // larger requests go to malloc
constexpr size_t kHdr = 64;
constexpr int64_t kBias = 1ll << 40;            // a chunk that is being filled cannot reach a live count of 0
constexpr uint32_t kMaxChunks = 16384;          // 64 GB of address space, touched only as far as it is used

struct Hdr {
    std::atomic<int64_t> live;                  // objects handed out and not yet deleted (+ kBias while the chunk is being filled)
};

// The reserved range [g_lo, g_hi); empty (g_hi == 0) until reserve() ran.  operator delete on any thread reads the pair
// while the first Scope of the process may still be reserving: g_lo is stored first, g_hi is published with release and
// read with acquire, so a reader sees either an empty range or both bounds.
std::atomic<uintptr_t> g_lo{1}, g_hi{0};
uint32_t g_chunks = 0;                          // chunks the range holds
std::once_flag g_once;
std::mutex g_mu;                                // the pool
uint32_t g_pool[kMaxChunks];
uint32_t g_npool = 0, g_fresh = 0;              // pooled chunk indices (LIFO: the warmest first); indices never handed out yet
std::atomic<uint32_t> g_in_use{0};

void reserve() {
    for (size_t want = (size_t)kMaxChunks * kChunk; want >= (64u << 20); want >>= 1) {
        void* p = mmap(nullptr, want + kChunk, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) continue;
        const uintptr_t lo = ((uintptr_t)p + kChunk - 1) & ~(uintptr_t)(kChunk - 1);
        g_chunks = (uint32_t)(want / kChunk);
        g_lo.store(lo, std::memory_order_relaxed);
        g_hi.store(lo + want, std::memory_order_release);
        return;
    }
}

struct ThreadState {
    char* cur = nullptr;                        // chunk being filled by this thread
    size_t off = 0;
    int64_t count = 0;                          // objects handed out of cur
    int depth = 0;                              // open Scopes
};
thread_local ThreadState t;

void recycle(Hdr* h) {
    const uint32_t idx = (uint32_t)(((uintptr_t)h - g_lo.load(std::memory_order_relaxed)) / kChunk);
    std::lock_guard<std::mutex> g(g_mu);
    g_pool[g_npool++] = idx;
    g_in_use.fetch_sub(1, std::memory_order_relaxed);
}

void retire_current() {                         // cur is full (or its thread ends): from now on the count can reach 0
    if (!t.cur) return;
    Hdr* h = (Hdr*)t.cur;
    if (h->live.fetch_add(t.count - kBias, std::memory_order_acq_rel) + (t.count - kBias) == 0) recycle(h);
    t.cur = nullptr;
    t.off = 0;
    t.count = 0;
}

bool next_chunk() {
    retire_current();
    uint32_t idx;
    {
        std::lock_guard<std::mutex> g(g_mu);
        if (g_npool) idx = g_pool[--g_npool];
        else if (g_fresh < g_chunks) idx = g_fresh++;
        else return false;
        g_in_use.fetch_add(1, std::memory_order_relaxed);
    }
    t.cur = (char*)(g_lo.load(std::memory_order_relaxed) + (uintptr_t)idx * kChunk);
    ((Hdr*)t.cur)->live.store(kBias, std::memory_order_release);
    t.off = kHdr;
    t.count = 0;
    return true;
}

struct ThreadEnd {                              // a thread that ends hands its partly filled chunk over
    bool armed = false;
    ~ThreadEnd() { retire_current(); t.depth = 0; }
};
thread_local ThreadEnd t_end;

inline void* take(size_t n) {                   // nullptr: not from the region
    if (t.depth == 0 || n > kMaxSmall) return nullptr;
    n = (n + 15) & ~(size_t)15;
    if (n == 0) n = 16;
    if (!t.cur || t.off + n > kChunk) {
        t_end.armed = true;                     // (first use on this thread registers the destructor)
        if (!next_chunk()) return nullptr;
    }
    void* p = t.cur + t.off;
    t.off += n;
    t.count++;
    return p;
}

inline bool give(void* p) {                     // true: it was the region's
    const uintptr_t a = (uintptr_t)p;
    const uintptr_t hi = g_hi.load(std::memory_order_acquire);       // (0 while nothing is reserved: nothing is the region's)
    if (a >= hi || a < g_lo.load(std::memory_order_relaxed)) return false;
    Hdr* h = (Hdr*)(a & ~(uintptr_t)(kChunk - 1));
    if (h->live.fetch_sub(1, std::memory_order_acq_rel) == 1) recycle(h);
    return true;
}

}  // namespace

bool available() {
    std::call_once(g_once, reserve);
    return g_hi.load(std::memory_order_acquire) > g_lo.load(std::memory_order_relaxed);
}

Scope::Scope() {
    outer_ = available();
    if (outer_) t.depth++;
}
Scope::~Scope() {
    if (outer_) t.depth--;
}

Stats stats() {
    std::lock_guard<std::mutex> g(g_mu);
    return Stats{g_in_use.load(), g_npool, g_fresh};
}

void trim() {
    std::lock_guard<std::mutex> g(g_mu);
    for (uint32_t i = 0; i < g_npool; i++) madvise((void*)(g_lo.load(std::memory_order_relaxed) + (uintptr_t)g_pool[i] * kChunk), kChunk, MADV_DONTNEED);
}

void* allocate(size_t n) { return take(n); }
bool release(void* p) { return give(p); }

}  // namespace arena
}  // namespace blance

// ---- the replaceable allocation functions ([new.delete]): the region first, malloc otherwise
namespace {
inline void* plain(size_t n) {
    for (;;) {
        void* p = malloc(n ? n : 1);
        if (p) return p;
        std::new_handler h = std::get_new_handler();
        if (!h) throw std::bad_alloc();
        h();
    }
}
inline void* aligned(size_t n, size_t al) {
    void* p = nullptr;
    if (posix_memalign(&p, al < sizeof(void*) ? sizeof(void*) : al, n ? n : 1) != 0) throw std::bad_alloc();
    return p;
}
}  // namespace

namespace blance { namespace arena { void* allocate(size_t n); bool release(void* p); } }

void* operator new(size_t n) {
    void* p = blance::arena::allocate(n);
    return p ? p : plain(n);
}
void* operator new[](size_t n) {
    void* p = blance::arena::allocate(n);
    return p ? p : plain(n);
}
void* operator new(size_t n, const std::nothrow_t&) noexcept {
    void* p = blance::arena::allocate(n);
    return p ? p : malloc(n ? n : 1);
}
void* operator new[](size_t n, const std::nothrow_t&) noexcept {
    void* p = blance::arena::allocate(n);
    return p ? p : malloc(n ? n : 1);
}
void* operator new(size_t n, std::align_val_t al) {
    if ((size_t)al <= 16) return operator new(n);
    return aligned(n, (size_t)al);
}
void* operator new[](size_t n, std::align_val_t al) {
    if ((size_t)al <= 16) return operator new(n);
    return aligned(n, (size_t)al);
}
void* operator new(size_t n, std::align_val_t al, const std::nothrow_t&) noexcept {
    try { return operator new(n, al); } catch (...) { return nullptr; }
}
void* operator new[](size_t n, std::align_val_t al, const std::nothrow_t&) noexcept {
    try { return operator new(n, al); } catch (...) { return nullptr; }
}

void operator delete(void* p) noexcept {
    if (p && !blance::arena::release(p)) free(p);
}
void operator delete[](void* p) noexcept { operator delete(p); }
void operator delete(void* p, size_t) noexcept { operator delete(p); }
void operator delete[](void* p, size_t) noexcept { operator delete(p); }
void operator delete(void* p, const std::nothrow_t&) noexcept { operator delete(p); }
void operator delete[](void* p, const std::nothrow_t&) noexcept { operator delete(p); }
void operator delete(void* p, std::align_val_t) noexcept { operator delete(p); }
void operator delete[](void* p, std::align_val_t) noexcept { operator delete(p); }
void operator delete(void* p, size_t, std::align_val_t) noexcept { operator delete(p); }
void operator delete[](void* p, size_t, std::align_val_t) noexcept { operator delete(p); }
void operator delete(void* p, std::align_val_t, const std::nothrow_t&) noexcept { operator delete(p); }
void operator delete[](void* p, std::align_val_t, const std::nothrow_t&) noexcept { operator delete(p); }
