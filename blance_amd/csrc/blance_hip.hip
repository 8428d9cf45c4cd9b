// MI355X (gfx950) implementation of blance's planNextMapEx (plan.go:23-58) behind
// the C ABI of include/blance_hip.h.  One blance_plan() call runs the whole
// convergence loop on the device; the host only sequences kernels and reads
// one convergence word per sweep.  See DESIGN.md for the kernel inventory.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC
// (fp64 scores must keep the reference's operation order, plan.go:634-689).
#include "dev_prelude.h"

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <new>
#include <thread>
#include <type_traits>
#include <string>
#include <vector>

#include "k_flat.h"
#include "k_sweep.h"
#include "k_period.h"
#ifdef BLANCE_SIMT_EMU          /* the emulator build is one translation unit */
#include "tu_seq.hip"
#include "tu_tree.hip"
#include "tu_queue.hip"
#include "tu_chain.hip"
#endif


// ============================================================================
// Host side: context, upload, the sweep driver, download
// ============================================================================
using namespace blance;

static thread_local std::string g_last_error;

static int fail(int code, const char* fmt, const char* a = "", long b = 0) {
    char buf[512];
    snprintf(buf, sizeof buf, fmt, a, b);
    g_last_error = buf;
    return code;
}

#define HIPTRY(expr)                                                                  \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess)                                                         \
            return fail(BLANCE_ERR_DEVICE, "%s failed: line %ld", hipGetErrorString(e_), __LINE__); \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap && p) return 0;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes < 256 ? 256 : bytes;
        if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return -1; }
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return (T*)p; }
};

// ---- page-locked host memory (include/blance_hip.h: blance_host_alloc).  hipHostMalloc costs milliseconds for the
// sizes of a plan, so freed blocks are kept (up to kPinCacheMax bytes) and handed out again.
struct CopySeg { void* dst; const void* src; size_t bytes; };
namespace {
constexpr size_t kPinCacheMax = (size_t)2 << 30;
struct PinBlock { void* p; size_t cap; };
std::mutex g_pin_mu;
std::vector<PinBlock> g_pin_free;
std::map<uintptr_t, size_t> g_pin_live;          // blocks handed out: base -> capacity
size_t g_pin_cached = 0;

void* pin_alloc(size_t bytes) {
    const size_t want = ((bytes ? bytes : 1) + 65535) & ~(size_t)65535;
    {
        std::lock_guard<std::mutex> g(g_pin_mu);
        size_t best = g_pin_free.size();
        for (size_t i = 0; i < g_pin_free.size(); i++)
            if (g_pin_free[i].cap >= want && g_pin_free[i].cap <= 2 * want + (1u << 20) &&
                (best == g_pin_free.size() || g_pin_free[i].cap < g_pin_free[best].cap)) best = i;
        if (best < g_pin_free.size()) {
            PinBlock b = g_pin_free[best];
            g_pin_free.erase(g_pin_free.begin() + (long)best);
            g_pin_cached -= b.cap;
            g_pin_live[(uintptr_t)b.p] = b.cap;
            return b.p;
        }
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, want) != hipSuccess || !p) { (void)hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> g(g_pin_mu);
    g_pin_live[(uintptr_t)p] = want;
    return p;
}
void pin_free(void* p) {
    if (!p) return;
    size_t cap = 0;
    {
        std::lock_guard<std::mutex> g(g_pin_mu);
        auto it = g_pin_live.find((uintptr_t)p);
        if (it == g_pin_live.end()) return;          // not one of ours
        cap = it->second;
        g_pin_live.erase(it);
        if (g_pin_cached + cap <= kPinCacheMax) { g_pin_free.push_back(PinBlock{p, cap}); g_pin_cached += cap; return; }
    }
    (void)hipHostFree(p);
}
void pin_trim() {
    std::vector<PinBlock> drop;
    {
        std::lock_guard<std::mutex> g(g_pin_mu);
        drop.swap(g_pin_free);
        g_pin_cached = 0;
    }
    for (const PinBlock& b : drop) (void)hipHostFree(b.p);
}
// does [p, p + bytes) lie in page-locked memory the device can copy from / to directly?
bool host_ptr_pinned(const void* p, size_t bytes) {
    const uintptr_t a = (uintptr_t)p;
    {
        std::lock_guard<std::mutex> g(g_pin_mu);
        auto it = g_pin_live.upper_bound(a);
        if (it != g_pin_live.begin()) {
            --it;
            if (a >= it->first && a + bytes <= it->first + it->second) return true;
        }
    }
#ifndef BLANCE_SIMT_EMU
    if (bytes >= ((size_t)1 << 20)) {                // (memory the caller registered itself: worth a query for big arrays only)
        // both ends: an array that only starts inside a registered range is not DMA'd as if all of it were page-locked
        hipPointerAttribute_t at, at_end;
        if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (at.type != hipMemoryTypeHost) return false;
        if (hipPointerGetAttributes(&at_end, (const char*)p + bytes - 1) != hipSuccess) { (void)hipGetLastError(); return false; }
        return at_end.type == hipMemoryTypeHost && at_end.hostPointer != nullptr && at.hostPointer != nullptr &&
               (const char*)at_end.hostPointer - (const char*)at.hostPointer == (ptrdiff_t)(bytes - 1);
    }
#endif
    return false;
}

// n bytes copied by a few threads (a pageable array on its way into / out of the staging buffer: one core moves ~10 GB/s,
// the link five times that)
void copy_threaded(const std::vector<CopySeg>& segs) {
    size_t total = 0;
    for (const CopySeg& g : segs) total += g.bytes;
    unsigned T = 1;
    if (total >= ((size_t)4 << 20)) {
        const unsigned hw = std::thread::hardware_concurrency();
        T = hw >= 16 ? 6 : hw >= 8 ? 4 : hw >= 4 ? 2 : 1;
        const unsigned by_size = (unsigned)(total >> 21);
        if (by_size < T) T = by_size ? by_size : 1;
    }
    auto work = [&](unsigned t) {
        const size_t lo = total * t / T, hi = total * (t + 1) / T;
        size_t pos = 0;
        for (const CopySeg& g : segs) {
            const size_t b = pos, e = pos + g.bytes;
            pos = e;
            if (e <= lo || b >= hi) continue;
            const size_t from = lo > b ? lo - b : 0, to = (hi < e ? hi : e) - b;
            memcpy((char*)g.dst + from, (const char*)g.src + from, to - from);
        }
    };
    std::vector<std::thread> th;
    struct Join { std::vector<std::thread>& v; ~Join() { for (auto& t : v) if (t.joinable()) t.join(); } } join{th};
    unsigned started = 1;
    try {
        for (unsigned t = 1; t < T; t++) { th.emplace_back(work, t); started++; }
    } catch (...) {}                                  // (no more threads: this one does the rest)
    work(0);
    for (unsigned t = started; t < T; t++) work(t);
}
}  // namespace

extern "C" void* blance_host_alloc(size_t bytes) { return pin_alloc(bytes); }
extern "C" void blance_host_free(void* p) { pin_free(p); }
extern "C" void blance_host_trim(void) { pin_trim(); }
static std::atomic<int> g_live_contexts{0};

struct HostStage {                                   // a page-locked staging buffer of the context
    void* p = nullptr;
    size_t cap = 0, used = 0;
    void release() { if (p) pin_free(p); p = nullptr; cap = used = 0; }
};

struct blance_ctx {
    int device = 0;
    int engine = BLANCE_ENGINE_AUTO;
    int force_threads = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::mutex mu;
    bool uploaded = false;
    bool planned = false;
    // one plan on several ranks (include/blance_hip.h "one plan on several GPUs")
    blance_comm comm{0, 1, nullptr, nullptr, nullptr};
    void* rccl_comm = nullptr;      // ncclComm_t of blance_comm_init_rccl
    DevBuf scan_sums;               // tile totals of launch_scan_excl
    DevBuf scan_part;               // k_flat_scan's per-wave results
    DevBuf topkey, top_counts, top_off, top_order;   // k_stay_by_top: steps grouped by the leaf of their top priority node
    // That grouping made AHEAD of the sweep that wants it: on a second stream, beside the chain kernel of the sweep before (a
    // few dozen workgroups on 256 CUs).  It holds as long as the grouping by region it was made from does (group_epoch).
    hipStream_t side = nullptr;
    hipEvent_t side_go = nullptr, side_done = nullptr;
    DevBuf side_sums;               // (launch_scan_excl's tile totals on that stream)
    bool side_pending = false;      // work on `side` the planner's stream has not waited for yet
    int top_group_state = -1;       // top_off / top_order are those of this state's chain order ...
    int64_t top_group_epoch = -1, group_epoch = 0;   // ... as grouped at this count of regroupings
    std::vector<int64_t> last_stays; // [state] steps the last chain pass of that state committed as verified stays
    bool no_stay_top = false;       // test knob (& 64): never k_stay_by_top
    bool force_stay_top = false;    // test knob (& 128): try k_stay_by_top in every chain pass with NumPartitions > 0
    bool periodic = true;           // an all-blank chain pass with periodic records walks two periods (k_period.h); off: & 256, or BLANCE_PERIODIC=0
    int64_t periodic_passes = 0;
    int periodic_cut = 0;           // test knob BLANCE_PERIODIC_CUT (k_period_clamp)
    DevBuf cnt_base, xbuf, gath;    // sharded pass: loads at pass start, [flags | load change], gathered output slices
    std::vector<int32_t> h_reg_off; // host copy of the chain offsets (slice sizes of the all-gather)
    int chain_group_state = -1;     // the state whose chain pass last grouped the steps by region (chain_order, chain_oi, reg_off) ...
    bool chain_group_static = false; // ... and whether it did so from the static order (sweeps >= 2)
    bool tops_moved = true;         // this sweep's top-state pass was not (known to be) one run of stays
    bool flags_clean = false;       // the chain passes' flag words (scalars[4 .. 11]) are zero: nothing has set one since the last fill
    bool rowcount_clean = false;    // f_row_count is zero (k_flat_row_count adds to it)
    bool top_prio_strict = false;   // every other state with constraints > 0 has a priority strictly behind the top state's
    // every partition gets the same partitionSorter category in sweep 1 (decided at upload): the pass order IS the static
    // order, no k_category / stable partition (plan.go:542-561) -- in the sweep's first state pass / in all of them
    bool uniform_first = false, uniform_all = false;
    bool trace = false;             // BLANCE_TRACE, read once at context creation
    int dump_sweep = -1;            // BLANCE_DUMP_SWEEP (developer aid), likewise
    DevBuf dl_off, dl_nodes;        // blance_download: the result as CSR, compacted on the device
    HostStage stage;                // pageable arrays of the caller pass through this page-locked buffer
    // small readbacks (flag words, counts) land in a page-locked block first: a D2H copy into pageable memory is staged by
    // the runtime and costs several microseconds more, thirteen times per call
    void* rb_buf = nullptr;
    size_t rb_used = 0;
    struct RbItem { void* dst; size_t off, bytes; };
    std::vector<RbItem> rb_items;
    DevBuf vres, vseen;             // blance_upload: the device's part of the validation (k_validate_parts)
    DevBuf mv[11];                  // blance_calc_moves: inputs, per-partition slices, offsets, compacted outputs (kept between calls)
    int64_t comm_calls = 0, comm_bytes = 0;
    std::vector<hipEvent_t> comm_events;     // begin / end pairs around the collectives of the current plan (RCCL path)
    size_t comm_events_used = 0;
    double comm_ms = 0.0;                    // device time between those pairs, all plans so far
    int64_t n_syncs = 0, plan_syncs = 0;     // stream_sync() calls so far / inside the last plan
    // known at upload (no readback needed for them in the first sweep's first pass): the partitions to assign hold no node at
    // all; no load counter starts above zero (no extra loads, nothing counted from prevMap)
    bool assign_empty = false, counts_start_zero = false;
    int chain_waves = 0;                     // k_pass_chain's workgroup: 0 = 8 waves when the LDS is there, else 4 (BLANCE_CHAIN_WAVES=4|8)
    int speculate = 1;                       // host decisions taken before their words are read back (BLANCE_SPECULATE=0|1|fail)
    int64_t spec_refuted = 0;                // ... and how often one had to be taken back

    // host copy of the small parts of the problem
    blance_problem h{};
    std::vector<int32_t> state_priority, state_constraints, rule_off;
    int L = 1, np_later = 0, n_alive = 0, any_removed = 0;
    int chain_min_parts = 2048;
    int any_node_weight = 0;
    bool no_fast_keys = false;      // a chain left the packed keys' range during this pass
    bool no_seq_spec = false;       // test knob (options.reserved[2] & 1): k_pass_seq without stay speculation
    bool no_tree = false;           // test knob (& 2): flat passes never on k_pass_tree
    bool tree_dense = false;        // test knob (& 4): k_pass_tree scores every node in every general step
    bool tree_always = false;       // test knob (& 8): k_pass_tree even when a k_pass_seq workgroup size is forced
    bool tree_long = false;         // test knob (& 16): k_pass_tree decodes the record in every general step
    bool no_planes = false;         // test knob (& 32): the all-blank chain pass on k_pass_chain_blank, not k_pass_chain_planes
    bool no_queue = false;          // test knob (& 512): flat passes with k <= 2 on k_pass_tree, never on k_pass_queue
    bool queue_general = false;     // test knob (& 1024): k_pass_queue without its lean walk
    bool queue_no_asm = false;      // test knob (& 4096): k_pass_queue's lean walk as compiled C++ only
    bool queue_force_dense = false; // test knob (& 2048): every general step of k_pass_queue scores every node
    bool queue_exact_rebuild = false; // test knob (& 8192): k_pass_queue's window always rebuilt by the exact selection
    bool queue_bits_self = false;   // test knob (& 32768): k_pass_queue's walking wave copies the row bit maps itself (no helper)
    bool shard_one_rank = false;    // test knob (& 16384): a communicator of ONE rank takes the sharded branch of a chain pass, so that
                                    // both collectives really execute (ncclAllReduce / ncclAllGather on a one-GPU box)
    DevBuf ntn_bits;                // k_pass_queue: one bit per nodeToNodeCounts entry, zeroed with the matrix
    bool bits_stale = false;        // another kernel bumped the matrix in this pass: k_ntn_bits before k_pass_queue goes on
    // nodeToNodeCounts (67 MB at config 3) is zeroed lazily: only a pass that reads or bumps the matrix in HBM pays for it
    // (region chains keep their rows in LDS, a pass that is one run of stays needs none of it)
    bool pass_ntn_ready = false;    // this pass has been given its zeroed matrix already (plan.go:266)
    int64_t queue_launches = 0, queue_stops = 0, queue_moved = 0, queue_exact = 0, queue_rebuilds = 0, queue_dense = 0;
    struct RuleRegions {           // regions the rule cuts the leaves into (chains), if it does
        bool ok = false;
        int n_regions = 0, max_size = 0;
        DevBuf node_region, reg_lo, reg_hi, leaf_cls, cls_size;
        DevBuf wg_region, wg_chunk;     // k_stay_by_top: entry b of its work table = the tops at leaves reg_lo + 64 chunk .. + 63 of its region
        int n_stay_wgs = 0, n_leaves = 0;
        int cls_run = 0;                // S if every exclude class is an aligned run of S = 2^e <= 64 node-carrying leaves, else 0
    };
    std::vector<RuleRegions> rule_regions;
    DevBuf leaf_node, regid, chain_order, bucket_counts, reg_off, cnt_save, crec;
    DevBuf period, cnt_p1;          // k_period.h: per-region period tables, the counters after the first period
    DevBuf fl_iota, fl_zero, fl_one, fl_reglo, fl_reghi;   // the whole cluster as one region (flat single chain)
    DevBuf n_ev, chain_oi, ev_key, ev_oi, ev_leaf, ev_w, ev_perm, ev_off, ev_counts;   // chain events
    bool flat_chain_ok = false;
    DevBuf f_tot, f_g, f_top_g, f_top_n, f_row_count, f_m, f_moff, f_keys_a, f_keys_b, f_vals_a, f_vals_b, f_hist, f_comp;
    int64_t steps_batched = 0;
    int64_t out_capacity = 0;

    // device: problem
    DevBuf node_removed, node_added, node_weight, node_has_weight, alive, zeros_nx, node_leaf_pos;
    DevBuf alive_ids, alive_rank;   // the nodes of nodesNext in id order; a node's place in that list (-1: not in it)
    DevBuf part_order, part_weight, part_has_weight, part_in_prev, part_never_equal;
    DevBuf a_off, a_nodes, a_kind, p_off, p_nodes, p_kind;
    DevBuf load_state, load_node, load_weight, load_first;
    DevBuf rule_inc, rule_exc, vparent, vlo, vhi, anchors;
    DevBuf state_stick, state_has_stick;
    // device: working state
    DevBuf live, live_len, live_kind, prv, prv_len, prv_kind, in_prev, never_equal;
    DevBuf cnt, ntn, cat, order, chunk_counts, rec, out, warn_part, warn_state, scalars;
    // scalars: [0] warn_count, [1] not_match, [2] err
    int32_t iterations = 0, converged = 0;
    int64_t n_warnings = 0, steps_total = 0, kernel_launches = 0, pass_launches = 0;
    double device_ms = 0.0, pass_ms = 0.0;
    std::vector<hipEvent_t> pass_events;     // begin/end pairs around every pass kernel
    std::vector<int> pass_kind;              // 0 = one pass kernel, 1 = flat bulk driver
    double flat_ms = 0.0, blank_ms = 0.0, stay_ms = 0.0;
    int64_t flat_passes = 0, blank_launches = 0, stay_launches = 0;

    void free_all() {
        DevBuf* all[] = {&node_removed, &node_added, &node_weight, &node_has_weight, &alive, &zeros_nx,
                         &node_leaf_pos, &part_order, &part_weight, &part_has_weight, &part_in_prev,
                         &part_never_equal, &a_off, &a_nodes, &a_kind, &p_off, &p_nodes, &p_kind,
                         &load_state, &load_node, &load_weight, &load_first, &rule_inc, &rule_exc,
                         &vparent, &vlo, &vhi, &anchors, &state_stick, &state_has_stick, &live,
                         &live_len, &live_kind, &prv, &prv_len, &prv_kind, &in_prev, &never_equal,
                         &cnt, &ntn, &cat, &order, &chunk_counts, &rec, &out, &warn_part,
                         &warn_state, &scalars};
        for (DevBuf* b : all) b->release();
        for (auto& rr : rule_regions) { rr.node_region.release(); rr.reg_lo.release(); rr.reg_hi.release(); rr.leaf_cls.release(); rr.cls_size.release(); rr.wg_region.release(); rr.wg_chunk.release(); }
        rule_regions.clear();
        topkey.release(); top_counts.release(); top_off.release(); top_order.release();
        alive_ids.release(); alive_rank.release(); side_sums.release();
        cnt_base.release(); xbuf.release(); gath.release(); scan_sums.release(); scan_part.release(); ntn_bits.release();
        dl_off.release(); dl_nodes.release(); vres.release(); vseen.release();
        stage.release();
        if (rb_buf) pin_free(rb_buf);
        rb_buf = nullptr;
        for (DevBuf& b : mv) b.release();
        DevBuf* more[] = {&leaf_node, &regid, &chain_order, &bucket_counts, &reg_off, &cnt_save, &crec, &period, &cnt_p1, &n_ev, &chain_oi,
                          &ev_key, &ev_oi, &ev_leaf, &ev_w, &ev_perm, &ev_off, &ev_counts, &fl_iota, &fl_zero,
                          &fl_one, &fl_reglo, &fl_reghi, &f_tot, &f_g,
                          &f_top_g, &f_top_n, &f_row_count, &f_m, &f_moff, &f_keys_a, &f_keys_b, &f_vals_a,
                          &f_vals_b, &f_hist, &f_comp};
        for (DevBuf* b : more) b->release();
    }
};

static void comm_release(blance_ctx* c);
// no exception crosses the C boundary (std::bad_alloc from a staging vector, say)
template <class F>
static int guarded(F f) {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return fail(BLANCE_ERR_DEVICE, "out of host memory");
    } catch (...) {
        return fail(BLANCE_ERR_DEVICE, "unexpected exception");
    }
}
extern "C" int blance_abi_version(void) { return BLANCE_ABI_VERSION; }
extern "C" const char* blance_last_error(void) { return g_last_error.c_str(); }

extern "C" int64_t blance_result_capacity(const blance_problem* pb) {
    if (!pb || !pb->assign_off || !pb->state_constraints) return 0;
    int64_t cap = 0;
    const int64_t PM = (int64_t)pb->n_parts * pb->n_states;
    for (int64_t idx = 0; idx < PM; idx++) {
        int len = pb->assign_off[idx + 1] - pb->assign_off[idx];
        int k = pb->state_constraints[idx % pb->n_states];
        cap += len > k ? len : k;
    }
    return cap;
}

// blance_validate in four parts, in the order of its checks: head (sizes, pointers), parts (the O(P) loops: CSR shape, ids,
// the order's permutation -- blance_upload runs these on the device, k_validate_parts), tail_a (loads, rules, hierarchy),
// tail_b (what needs the longest list and the weight sums).
static int validate_head(const blance_problem* pb) {
    if (!pb) return fail(BLANCE_ERR_BAD_ARG, "null problem");
    const int N = pb->n_nodes, NX = pb->n_nodes_ext, M = pb->n_states, P = pb->n_parts;
    if (N < 0 || NX < N || M < 0 || P < 0 || pb->n_prev < 0 || pb->n_loads < 0 || pb->n_rules < 0 ||
        pb->max_iterations < 0)
        return fail(BLANCE_ERR_BAD_ARG, "negative or inconsistent sizes");
    if ((int64_t)P * (M > 0 ? M : 1) > (int64_t)INT32_MAX / 4) return fail(BLANCE_ERR_UNSUPPORTED, "P*M too large");
    if (M > kMaxStates) return fail(BLANCE_ERR_UNSUPPORTED, "more than 16 model states");
    if (M > 0 && (pb->top_state < 0 || pb->top_state >= M)) return fail(BLANCE_ERR_BAD_ARG, "top_state out of range");
    const void* need[] = {pb->state_priority, pb->state_constraints, pb->state_stickiness, pb->state_has_stickiness,
                          pb->node_removed, pb->node_added, pb->node_weight, pb->node_has_weight, pb->part_order,
                          pb->part_weight, pb->part_has_weight, pb->part_in_prev, pb->part_prev_never_equal,
                          pb->assign_off, pb->assign_nodes, pb->assign_kind, pb->prev_off, pb->prev_nodes,
                          pb->prev_kind, pb->load_state, pb->load_node, pb->load_weight, pb->load_first_sweep_only,
                          pb->rule_off, pb->rule_inc, pb->rule_exc, pb->node_leaf_pos};
    for (const void* q : need) if (!q) return fail(BLANCE_ERR_BAD_ARG, "null array pointer");
    if (pb->assign_off[0] != 0 || pb->prev_off[0] != 0) return fail(BLANCE_ERR_BAD_ARG, "CSR offsets must start at 0");
    return BLANCE_OK;
}

struct PartsSummary { int L; long long fresh, cap, sumw, aprev; };

static int validate_parts_host(const blance_problem* pb, PartsSummary* ps) {
    const int NX = pb->n_nodes_ext, M = pb->n_states, P = pb->n_parts;
    const int64_t PM = (int64_t)P * M;
    for (int64_t i = 0; i < PM; i++) {
        if (pb->assign_off[i + 1] < pb->assign_off[i] || pb->prev_off[i + 1] < pb->prev_off[i])
            return fail(BLANCE_ERR_BAD_ARG, "CSR offsets not monotone");
        if (pb->assign_kind[i] > BLANCE_LIST_SET || pb->prev_kind[i] > BLANCE_LIST_SET)
            return fail(BLANCE_ERR_BAD_ARG, "bad list kind");
        if (pb->assign_off[i + 1] - pb->assign_off[i] > 0xffff || pb->prev_off[i + 1] - pb->prev_off[i] > 0xffff)
            return fail(BLANCE_ERR_UNSUPPORTED, "state list longer than 65535");
    }
    for (int64_t i = 0; i < pb->assign_off[PM]; i++)
        if (pb->assign_nodes[i] < 0 || pb->assign_nodes[i] >= NX) return fail(BLANCE_ERR_BAD_ARG, "assign node id out of range");
    for (int64_t i = 0; i < pb->prev_off[PM]; i++)
        if (pb->prev_nodes[i] < 0 || pb->prev_nodes[i] >= NX) return fail(BLANCE_ERR_BAD_ARG, "prev node id out of range");
    {
        std::vector<uint8_t> seen((size_t)P, 0);
        for (int i = 0; i < P; i++) {
            int p = pb->part_order[i];
            if (p < 0 || p >= P || seen[p]) return fail(BLANCE_ERR_BAD_ARG, "part_order is not a permutation");
            seen[p] = 1;
        }
    }
    ps->L = 0; ps->fresh = ps->cap = ps->sumw = ps->aprev = 0;
    for (int64_t i = 0; i < PM; i++) {
        int a = pb->assign_off[i + 1] - pb->assign_off[i], b = pb->prev_off[i + 1] - pb->prev_off[i];
        if (a > ps->L) ps->L = a;
        if (b > ps->L) ps->L = b;
        const int k = pb->state_constraints[i % M];
        ps->cap += a > k ? a : k;
    }
    auto la = [](long long v) { return v < 0 ? -v : v; };
    for (int p = 0; p < P; p++) {
        const long long w = (!pb->partition_weights_nil && pb->part_has_weight[p]) ? la(pb->part_weight[p]) : 1;
        ps->sumw += w;
        if (pb->part_in_prev[p]) ps->aprev += w * (pb->prev_off[(int64_t)(p + 1) * M] - pb->prev_off[(int64_t)p * M]);
        else ps->fresh++;
    }
    return BLANCE_OK;
}

static int validate_tail_a(const blance_problem* pb) {
    const int NX = pb->n_nodes_ext, M = pb->n_states;
    for (int i = 0; i < pb->n_loads; i++)
        if (pb->load_state[i] < 0 || pb->load_state[i] > M || pb->load_node[i] < 0 || pb->load_node[i] >= NX)
            return fail(BLANCE_ERR_BAD_ARG, "load entry out of range");
    if (pb->rule_off[0] != 0) return fail(BLANCE_ERR_BAD_ARG, "rule_off must start at 0");
    for (int m = 0; m < M; m++) {
        int k = pb->state_constraints[m];
        if (k > kMaxK) return fail(BLANCE_ERR_UNSUPPORTED, "constraints > 8 for a state");
        if (pb->rule_off[m + 1] < pb->rule_off[m]) return fail(BLANCE_ERR_BAD_ARG, "rule_off not monotone");
        if (!pb->hierarchy_rules_nil && k > 0 && (pb->rule_off[m + 1] - pb->rule_off[m]) * k > kMaxAnchors - 1)
            return fail(BLANCE_ERR_UNSUPPORTED, "more than 8 hierarchy picks per partition and state");
    }
    if (M > 0 && pb->rule_off[M] > pb->n_rules) return fail(BLANCE_ERR_BAD_ARG, "rule_off exceeds n_rules");
    if (!pb->hierarchy_rules_nil) {
        const int VX = pb->n_vertices;
        if (VX <= NX || !pb->vertex_parent || !pb->vertex_leaf_lo || !pb->vertex_leaf_hi)
            return fail(BLANCE_ERR_BAD_ARG, "hierarchy arrays missing");
        if (pb->vertex_empty < 0 || pb->vertex_empty >= VX) return fail(BLANCE_ERR_BAD_ARG, "vertex_empty out of range");
        for (int v = 0; v < VX; v++) {
            if (pb->vertex_parent[v] < 0 || pb->vertex_parent[v] >= VX) return fail(BLANCE_ERR_BAD_ARG, "vertex_parent out of range");
            if (pb->vertex_leaf_lo[v] < 0 || pb->vertex_leaf_hi[v] <= pb->vertex_leaf_lo[v] || pb->vertex_leaf_hi[v] > VX)
                return fail(BLANCE_ERR_BAD_ARG, "vertex leaf interval empty or beyond the number of vertices");
        }
        for (int n = 0; n < NX; n++)
            if (pb->node_leaf_pos[n] < -1 || pb->node_leaf_pos[n] >= VX) return fail(BLANCE_ERR_BAD_ARG, "node_leaf_pos out of range");
        for (int r = 0; r < pb->n_rules; r++)
            if (pb->rule_inc[r] < 0 || pb->rule_exc[r] < 0 || pb->rule_inc[r] > 64 || pb->rule_exc[r] > 64)
                return fail(BLANCE_ERR_UNSUPPORTED, "hierarchy rule level outside 0..64");
    }
    if (pb->booster_kind != BLANCE_BOOSTER_NONE && pb->booster_kind != BLANCE_BOOSTER_CBGT)
        return fail(BLANCE_ERR_UNSUPPORTED, "unknown booster kind");
    if (NX > 1024 * 8) return fail(BLANCE_ERR_UNSUPPORTED, "more than 8192 node names (register-resident tables)");
    return BLANCE_OK;
}

static int validate_tail_b(const blance_problem* pb, const PartsSummary& ps) {
    const int N = pb->n_nodes, NX = pb->n_nodes_ext, M = pb->n_states;
    int L = 1;
    for (int m = 0; m < M; m++) if (pb->state_constraints[m] > L) L = pb->state_constraints[m];
    if (ps.L > L) L = ps.L;
    if (kRecHead + M * (1 + L) > 64) return fail(BLANCE_ERR_UNSUPPORTED, "step record wider than 64 words (states x list length)");
    {   // the load tables are int32: bound every sum a plan can form (a shim doing its own interning gets the check too)
        long long abs_load = ps.aprev, ksum = 0;
        auto la = [](long long v) { return v < 0 ? -v : v; };
        for (int i = 0; i < pb->n_loads; i++) abs_load += la(pb->load_weight[i]);
        for (int m = 0; m < M; m++) ksum += pb->state_constraints[m] > 0 ? pb->state_constraints[m] : 0;
        abs_load += ps.sumw * (ksum > 1 ? ksum : 1) * 2;
        if (abs_load > 2147483647LL) return fail(BLANCE_ERR_UNSUPPORTED, "partition weights overflow the int32 load tables");
    }
    if ((int64_t)(NX + 1) * (N > 0 ? N : 1) * 4 > (int64_t)64 << 30) return fail(BLANCE_ERR_UNSUPPORTED, "nodeToNodeCounts matrix > 64 GiB");
    return BLANCE_OK;
}

extern "C" int blance_validate(const blance_problem* pb) {
    return guarded([&]() -> int {
    int st = validate_head(pb);
    if (st) return st;
    PartsSummary ps;
    if ((st = validate_parts_host(pb, &ps))) return st;
    if ((st = validate_tail_a(pb))) return st;
    return validate_tail_b(pb, ps);
    });
}

extern "C" int blance_ctx_create(const blance_options* opt, blance_ctx** out) {
    if (!out) return fail(BLANCE_ERR_BAD_ARG, "null out");
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(BLANCE_ERR_NO_DEVICE, "no HIP device visible");
    int dev = opt ? opt->device_id : 0;
    if (dev < 0 || dev >= n_dev) return fail(BLANCE_ERR_BAD_ARG, "device_id out of range");
    HIPTRY(hipSetDevice(dev));
    blance_ctx* c = new blance_ctx();
    c->device = dev;
    c->engine = opt ? opt->engine : BLANCE_ENGINE_AUTO;
    c->force_threads = opt ? opt->reserved[0] : 0;
    if (opt && opt->reserved[1] > 0) c->chain_min_parts = opt->reserved[1];
    c->no_seq_spec = opt && (opt->reserved[2] & 1);
    c->no_tree = opt && (opt->reserved[2] & 2);
    c->tree_dense = opt && (opt->reserved[2] & 4);
    c->tree_always = opt && (opt->reserved[2] & 8);
    c->tree_long = opt && (opt->reserved[2] & 16);
    c->no_planes = opt && (opt->reserved[2] & 32);
    c->no_queue = opt && (opt->reserved[2] & 512);
    c->queue_general = opt && (opt->reserved[2] & 1024);
    c->queue_force_dense = opt && (opt->reserved[2] & 2048);
    c->queue_no_asm = opt && (opt->reserved[2] & 4096);
    c->queue_exact_rebuild = opt && (opt->reserved[2] & 8192);
    c->shard_one_rank = opt && (opt->reserved[2] & 16384);
    c->queue_bits_self = opt && (opt->reserved[2] & 32768);
    c->no_stay_top = opt && (opt->reserved[2] & 64);
    c->force_stay_top = opt && (opt->reserved[2] & 128);
    c->periodic = !(opt && (opt->reserved[2] & 256));
    if (const char* pe = getenv("BLANCE_PERIODIC")) c->periodic = atoi(pe) != 0;      // BLANCE_PERIODIC=0: the way out
    if (const char* pc = getenv("BLANCE_PERIODIC_CUT")) c->periodic_cut = atoi(pc);
    c->trace = getenv("BLANCE_TRACE") != nullptr;
    if (const char* cw = getenv("BLANCE_CHAIN_WAVES")) c->chain_waves = atoi(cw);
    if (const char* sp = getenv("BLANCE_SPECULATE")) c->speculate = !strcmp(sp, "fail") ? 2 : atoi(sp) != 0;   // 0: every decision read back first
    if (const char* ds = getenv("BLANCE_DUMP_SWEEP")) c->dump_sweep = atoi(ds);
    if (hipStreamCreate(&c->stream) != hipSuccess || hipEventCreate(&c->ev0) != hipSuccess ||
        hipEventCreate(&c->ev1) != hipSuccess ||
        hipEventCreate(&c->side_go) != hipSuccess || hipEventCreate(&c->side_done) != hipSuccess) {
        delete c;
        return fail(BLANCE_ERR_DEVICE, "stream/event creation failed");
    }
    *out = c;
    g_live_contexts++;
    return BLANCE_OK;
}

extern "C" void blance_ctx_destroy(blance_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->side) (void)hipStreamSynchronize(c->side);
    comm_release(c);
    c->free_all();
    for (hipEvent_t e : c->pass_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->comm_events) (void)hipEventDestroy(e);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->side_go) (void)hipEventDestroy(c->side_go);
    if (c->side_done) (void)hipEventDestroy(c->side_done);
    if (c->side) (void)hipStreamDestroy(c->side);
    delete c;
    if (--g_live_contexts == 0) pin_trim();          // the process's last context: the cache of page-locked blocks goes too
}

constexpr size_t kRbBytes = 64 * 1024;
// device -> host of a few words, complete after the next stream_sync()
static hipError_t read_back(blance_ctx* c, void* dst, const void* dev, size_t bytes) {
    if (!bytes) return hipSuccess;
    if (!c->rb_buf) c->rb_buf = pin_alloc(kRbBytes);
    const size_t need = (bytes + 15) & ~(size_t)15;
    if (!c->rb_buf || c->rb_used + need > kRbBytes) return hipMemcpyAsync(dst, dev, bytes, hipMemcpyDeviceToHost, c->stream);
    char* at = (char*)c->rb_buf + c->rb_used;
    hipError_t e = hipMemcpyAsync(at, dev, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e != hipSuccess) return e;
    c->rb_items.push_back(blance_ctx::RbItem{dst, c->rb_used, bytes});
    c->rb_used += need;
    return hipSuccess;
}
static hipError_t stream_sync(blance_ctx* c) {
    c->n_syncs++;
    if (c->trace) fprintf(stderr, "[blance] host synchronisation %lld (%zu words read back)\n", (long long)c->n_syncs, c->rb_used / 4);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess)
        for (const blance_ctx::RbItem& it : c->rb_items) memcpy(it.dst, (const char*)c->rb_buf + it.off, it.bytes);
    c->rb_items.clear();
    c->rb_used = 0;
    return e;
}

// An entry point's error exit: nothing is in flight any more (no DMA into the caller's arrays after the call returns) and
// no read-back is left queued -- its destination is a local of a frame that is gone by now, so it is dropped, not copied.
static void rb_discard(blance_ctx* c) {
    c->rb_items.clear();
    c->rb_used = 0;
}
static int settle(blance_ctx* c, int st) {
    if (st) {
        if (c->stream && hipStreamSynchronize(c->stream) != hipSuccess) (void)hipGetLastError();
        rb_discard(c);
    }
    return st;
}

// ---- host <-> device copies.  An array in page-locked memory (blance_host_alloc, or registered by the caller) is copied by
// DMA where it lies; a pageable one passes through the context's page-locked staging buffer -- small ones at once, big ones
// (>= 1 MB) by a few threads at flush().  Nothing of the caller's is read after the stream synchronisation that ends the call.
struct Mover {
    blance_ctx* c;
    bool to_device;
    std::vector<CopySeg> host_copy;                 // pending host side copies (caller <-> staging)
    std::vector<CopySeg> dma;                       // the DMAs that go with them (to_device: after the host copy; else before)
    Mover(blance_ctx* ctx, bool up) : c(ctx), to_device(up) {}
    int reserve(size_t bytes);                       // staging space for `bytes` more
    int copy(void* dst, const void* src, size_t bytes);
    int flush();                                     // to_device: host copies done and DMAs enqueued on return
    int finish();                                    // device -> host: DMAs done, then the host copies
};
static int stage_grow(blance_ctx* c, size_t need) {  // (the stream is idle, nothing pending)
    HostStage& st = c->stage;
    size_t cap = st.cap * 2 > need ? st.cap * 2 : need;
    cap = (cap + ((size_t)1 << 20)) & ~(((size_t)1 << 20) - 1);
    st.release();
    st.p = pin_alloc(cap);
    if (!st.p) return fail(BLANCE_ERR_DEVICE, "page-locked staging buffer: hipHostMalloc failed");
    st.cap = cap;
    st.used = 0;
    return 0;
}
int Mover::reserve(size_t bytes) {
    HostStage& st = c->stage;
    if (st.used + bytes <= st.cap) return 0;
    int e = to_device ? flush() : finish();
    if (e) return e;
    HIPTRY(stream_sync(c));
    if (bytes <= st.cap) { st.used = 0; return 0; }
    return stage_grow(c, bytes);
}
int Mover::copy(void* dst, const void* src, size_t bytes) {
    if (!bytes) return 0;
    const void* host = to_device ? src : dst;
    if (host_ptr_pinned(host, bytes)) {
        HIPTRY(hipMemcpyAsync(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, c->stream));
        return 0;
    }
    const size_t need = (bytes + 255) & ~(size_t)255;
    int e = reserve(need);
    if (e) return e;
    HostStage& st = c->stage;
    char* at = (char*)st.p + st.used;
    st.used += need;
    if (to_device) {
        if (bytes < ((size_t)1 << 20)) {
            memcpy(at, src, bytes);
            HIPTRY(hipMemcpyAsync(dst, at, bytes, hipMemcpyHostToDevice, c->stream));
        } else {
            host_copy.push_back(CopySeg{at, src, bytes});
            dma.push_back(CopySeg{dst, at, bytes});
        }
    } else {
        HIPTRY(hipMemcpyAsync(at, src, bytes, hipMemcpyDeviceToHost, c->stream));
        host_copy.push_back(CopySeg{dst, at, bytes});
    }
    return 0;
}
int Mover::flush() {
    if (!to_device) return 0;
    if (!host_copy.empty()) copy_threaded(host_copy);
    host_copy.clear();
    for (const CopySeg& g : dma) HIPTRY(hipMemcpyAsync(g.dst, g.src, g.bytes, hipMemcpyHostToDevice, c->stream));
    dma.clear();
    return 0;
}
int Mover::finish() {
    if (to_device) return flush();
    if (host_copy.empty()) return 0;
    HIPTRY(stream_sync(c));
    copy_threaded(host_copy);
    host_copy.clear();
    return 0;
}

template <class T>
static int put(Mover& mv, DevBuf& b, const T* src, size_t n) {
    if (b.reserve(n * sizeof(T))) return fail(BLANCE_ERR_DEVICE, "hipMalloc failed");
    return mv.copy(b.p, src, n * sizeof(T));
}
#define PUT(buf, src, n) do { int e__ = put(up, c->buf, src, (size_t)(n)); if (e__) return e__; } while (0)
#define RESERVE(buf, bytes) do { if (c->buf.reserve((size_t)(bytes))) return fail(BLANCE_ERR_DEVICE, "hipMalloc failed"); } while (0)

static inline int cdiv(int64_t a, int b) { return (int)((a + b - 1) / b); }

// up to four zero fills and one copy of int32 words in one launch (k_sweep.h: k_fill_copy)
struct FillCopyJob {
    FillCopy a{};
    int nz = 0;
    int64_t most = 0;
    void zero(void* p, int64_t words) {
        if (words <= 0) return;
        a.z[nz] = (int32_t*)p; a.zn[nz] = (int32_t)words; nz++;
        if (words > most) most = words;
    }
    void copy(void* dst, const void* src, int64_t words) {
        a.cd = (int32_t*)dst; a.cs = (const int32_t*)src; a.cn = (int32_t)words;
        if (words > most) most = words;
    }
};
static int run_fill_copy(blance_ctx* c, FillCopyJob& j);

static int run_fill_copy(blance_ctx* c, FillCopyJob& j) {
    if (j.most > 0) BLANCE_LAUNCH_NOSYNC(k_fill_copy, cdiv(j.most, 256), 256, 0, c->stream, j.a);
    return 0;
}

static int upload_inner(blance_ctx* c, const blance_problem* pb);
// copies may still be reading the caller's arrays (and the staging buffer) when an
// error cuts the upload short: never return with copies in flight
static int upload_locked(blance_ctx* c, const blance_problem* pb) {
    rb_discard(c);                                   // (left behind by a call that ended in an exception)
    return settle(c, upload_inner(c, pb));
}

static int upload_inner(blance_ctx* c, const blance_problem* pb) {
    int st = validate_head(pb);
    if (st) return st;
    HIPTRY(hipSetDevice(c->device));
    c->uploaded = false;
    c->planned = false;
    c->h = *pb;
    const int N = pb->n_nodes, NX = pb->n_nodes_ext, M = pb->n_states, P = pb->n_parts;
    const int64_t PM = (int64_t)P * M;
    c->state_priority.assign(pb->state_priority, pb->state_priority + M);
    c->state_constraints.assign(pb->state_constraints, pb->state_constraints + M);
    c->rule_off.assign(pb->rule_off, pb->rule_off + M + 1);
    c->top_prio_strict = M > 0;
    for (int m = 0; m < M; m++)
        if (m != pb->top_state && pb->state_priority[m] <= pb->state_priority[pb->top_state]) c->top_prio_strict = false;
    std::vector<uint8_t> alive((size_t)NX + 1, 0);
    std::vector<int32_t> alive_ids, alive_rank((size_t)NX + 1, -1);      // the nodes of nodesNext by id, and a node's place among them
    c->n_alive = 0;
    c->any_removed = 0;
    for (int n = 0; n < NX; n++) {
        if (pb->node_removed[n]) c->any_removed = 1;
        if (n < N && !pb->node_removed[n]) { alive[n] = 1; alive_rank[n] = c->n_alive++; alive_ids.push_back(n); }
    }
    alive_ids.push_back(-1);                                              // (never empty)

    Mover up(c, true);
    c->stage.used = 0;                               // (the stream is idle between calls)
    {   // staging space for everything that may be pageable, asked for once
        const size_t per_part = 4 + 4 + 1 + 1 + 1, per_pm = 4 + 1 + 4 + 1;
        const size_t want = (size_t)P * per_part + (size_t)(PM + 1) * per_pm + (size_t)NX * 32 + (size_t)pb->n_loads * 13 + ((size_t)4 << 20);
        if (c->stage.cap < want && (st = stage_grow(c, want))) return st;
    }
    PUT(node_removed, pb->node_removed, NX);
    PUT(node_added, pb->node_added, NX);
    PUT(node_weight, pb->node_weight, NX);
    PUT(node_has_weight, pb->node_has_weight, NX);
    PUT(alive, alive.data(), NX);
    PUT(alive_ids, alive_ids.data(), alive_ids.size());
    PUT(alive_rank, alive_rank.data(), NX);
    PUT(node_leaf_pos, pb->node_leaf_pos, NX);
    RESERVE(zeros_nx, NX + 1);
    HIPTRY(hipMemsetAsync(c->zeros_nx.p, 0, (size_t)NX + 1, c->stream));
    PUT(part_order, pb->part_order, P);
    PUT(part_weight, pb->part_weight, P);
    PUT(part_has_weight, pb->part_has_weight, P);
    PUT(part_in_prev, pb->part_in_prev, P);
    PUT(part_never_equal, pb->part_prev_never_equal, P);
    PUT(a_off, pb->assign_off, PM + 1);
    PUT(a_kind, pb->assign_kind, PM);
    PUT(p_off, pb->prev_off, PM + 1);
    PUT(p_kind, pb->prev_kind, PM);
    PUT(load_state, pb->load_state, pb->n_loads);
    PUT(load_node, pb->load_node, pb->n_loads);
    PUT(load_weight, pb->load_weight, pb->n_loads);
    PUT(load_first, pb->load_first_sweep_only, pb->n_loads);
    PUT(state_stick, pb->state_stickiness, M);
    PUT(state_has_stick, pb->state_has_stickiness, M);
    PUT(rule_inc, pb->rule_inc, pb->n_rules);
    PUT(rule_exc, pb->rule_exc, pb->n_rules);
    if ((st = up.flush())) return st;

    // ---- the O(P) checks of blance_validate and the sizes they yield, on the device (k_validate_parts)
    RESERVE(vres, 64);
    RESERVE(vseen, sizeof(uint32_t) * ((size_t)P / 32 + 2));
    HIPTRY(hipMemsetAsync(c->vres.p, 0, 64, c->stream));
    HIPTRY(hipMemsetAsync(c->vseen.p, 0, sizeof(uint32_t) * ((size_t)P / 32 + 1), c->stream));
    if (P > 0) {
        ValidateParams vp;
        memset(&vp, 0, sizeof vp);
        vp.P = P; vp.M = M; vp.weights_nil = pb->partition_weights_nil;
        for (int m = 0; m < M; m++) vp.k[m] = pb->state_constraints[m];
        vp.a_off = c->a_off.as<int32_t>(); vp.a_kind = c->a_kind.as<uint8_t>();
        vp.p_off = c->p_off.as<int32_t>(); vp.p_kind = c->p_kind.as<uint8_t>();
        vp.part_order = c->part_order.as<int32_t>(); vp.part_weight = c->part_weight.as<int32_t>();
        vp.part_has_weight = c->part_has_weight.as<uint8_t>(); vp.part_in_prev = c->part_in_prev.as<uint8_t>();
        vp.seen = c->vseen.as<uint32_t>(); vp.res = c->vres.as<int32_t>();
        const int vblocks = cdiv(PM > P ? PM : P, 256);
        BLANCE_LAUNCH(k_validate_parts, vblocks < 1024 ? vblocks : 1024, 256, 4 * 8 * sizeof(unsigned long long), c->stream, vp);
    }
    int32_t vr[16] = {0};
    HIPTRY(read_back(c, vr, c->vres.p, sizeof vr));
    const int tail_a = validate_tail_a(pb);          // (the host's share, while the device works)
    const std::string tail_a_text = g_last_error;
    HIPTRY(stream_sync(c));
    if (vr[3]) {                                      // the first list the host's loop would have refused, and why
        const int check = (INT_MAX - vr[3]) & 3;
        if (check == 0) return fail(BLANCE_ERR_BAD_ARG, "CSR offsets not monotone");
        if (check == 1) return fail(BLANCE_ERR_BAD_ARG, "bad list kind");
        return fail(BLANCE_ERR_UNSUPPORTED, "state list longer than 65535");
    }
    // the lists' payloads: their lengths are the last offsets, which are sound now
    const int64_t na = pb->assign_off[PM], np = pb->prev_off[PM];
    {
        // partitionSorter's category (plan.go:542-561) is "0" only for partitions with nodes in nodesToRemove.  Without any:
        // "2" for every partition when nodesToAdd == nil; "1" for every partition when nodesToAdd names no node.  A fresh
        // plan (the partitions to assign hold nothing): "1" for every partition in the sweep's first state pass (nothing held,
        // nothing in nodesToAdd), and -- when EVERY node of nodesNext is in nodesToAdd -- "2" for every partition in the later
        // ones: the first pass gave each partition at least one node (slot 0 always finds a candidate among >= 1 nodes, by
        // the rule or by the fallback of plan.go:216-218; nothing is excluded as "higher priority" yet) and that node is in
        // nodesToAdd.  The category is computed from the LIVE lists, so it changes from pass to pass of a sweep.
        bool any_added = false, all_alive_added = true;
        for (int n = 0; n < NX; n++) {
            if (pb->node_added[n]) any_added = true;
            else if (n < N && !pb->node_removed[n]) all_alive_added = false;
        }
        const bool never = !c->any_removed && (pb->nodes_to_add_nil || !any_added);
        c->uniform_first = never || (!c->any_removed && na == 0);
        c->uniform_all = never || (!c->any_removed && na == 0 && all_alive_added && c->n_alive >= 1);
        if (getenv("BLANCE_NO_UNIFORM_CATEGORY")) c->uniform_first = c->uniform_all = false;   // (tests: the partition kernels on such inputs too)
    }
    PUT(a_nodes, pb->assign_nodes, na);
    PUT(p_nodes, pb->prev_nodes, np);
    if ((st = up.flush())) return st;
    if (na > 0) BLANCE_LAUNCH(k_validate_ids, cdiv(na, 256), 256, 0, c->stream, (long long)na, NX, c->a_nodes.as<int32_t>(), kVErrAssignId, c->vres.as<int32_t>());
    if (np > 0) BLANCE_LAUNCH(k_validate_ids, cdiv(np, 256), 256, 0, c->stream, (long long)np, NX, c->p_nodes.as<int32_t>(), kVErrPrevId, c->vres.as<int32_t>());
    if (na > 0 || np > 0) {
        HIPTRY(read_back(c, vr, c->vres.p, sizeof(int32_t)));
        HIPTRY(stream_sync(c));
    }
    if (vr[0] & kVErrAssignId) return fail(BLANCE_ERR_BAD_ARG, "assign node id out of range");
    if (vr[0] & kVErrPrevId) return fail(BLANCE_ERR_BAD_ARG, "prev node id out of range");
    if (vr[0] & kVErrOrder) return fail(BLANCE_ERR_BAD_ARG, "part_order is not a permutation");
    if (tail_a) { g_last_error = tail_a_text; return tail_a; }
    PartsSummary ps;
    {
        long long r64[3];
        memcpy(r64, vr + 4, sizeof r64);
        ps.L = vr[1]; ps.fresh = vr[2]; ps.cap = r64[0]; ps.sumw = r64[1]; ps.aprev = r64[2];
    }
    if ((st = validate_tail_b(pb, ps))) return st;
    int L = 1;
    for (int m = 0; m < M; m++) if (pb->state_constraints[m] > L) L = pb->state_constraints[m];
    if (ps.L > L) L = ps.L;
    c->L = L;
    c->np_later = pb->n_prev + (int)ps.fresh;              // plan.go:50
    c->assign_empty = na == 0;
    c->counts_start_zero = pb->n_loads == 0 && ps.aprev == 0;
    c->out_capacity = ps.cap;

    for (auto& rr : c->rule_regions) { rr.node_region.release(); rr.reg_lo.release(); rr.reg_hi.release(); rr.leaf_cls.release(); rr.cls_size.release(); rr.wg_region.release(); rr.wg_chunk.release(); }
    c->rule_regions.clear();
    c->any_node_weight = 0;
    c->last_stays.assign((size_t)M, 0);
    for (int n = 0; n < NX; n++) if (pb->node_has_weight[n]) c->any_node_weight = 1;
    if (!pb->hierarchy_rules_nil) {
        // leaf-interval table of every (rule, anchor): plan.go:723-734, :755-774
        const int R = pb->n_rules;
        std::vector<AnchorSet> tab((size_t)(R > 0 ? R : 1) * (NX + 1));
        for (int r = 0; r < R; r++)
            for (int a = 0; a <= NX; a++) {
                int v = a == NX ? pb->vertex_empty : a;
                int vi = v, ve = v;
                for (int l = pb->rule_inc[r]; l > 0; l--) vi = pb->vertex_parent[vi];   // findAncestor
                for (int l = pb->rule_exc[r]; l > 0; l--) ve = pb->vertex_parent[ve];
                AnchorSet st;
                st.alo = pb->vertex_leaf_lo[vi]; st.ahi = pb->vertex_leaf_hi[vi];
                st.blo = pb->vertex_leaf_lo[ve]; st.bhi = pb->vertex_leaf_hi[ve];
                tab[(size_t)r * (NX + 1) + a] = st;
            }
        PUT(anchors, tab.data(), tab.size());
        int n_leaves = 1;
        for (int v = 0; v < pb->n_vertices; v++) if (pb->vertex_leaf_hi[v] > n_leaves) n_leaves = pb->vertex_leaf_hi[v];
        std::vector<int32_t> leaf_node((size_t)n_leaves, -1);
        for (int a = 0; a < NX; a++)
            if (pb->node_leaf_pos[a] >= 0 && pb->node_leaf_pos[a] < n_leaves) leaf_node[pb->node_leaf_pos[a]] = a;
        PUT(leaf_node, leaf_node.data(), leaf_node.size());
        // Does the rule cut the leaves into regions?  Every node whose leaf lies in
        // a region must have exactly that region as its include set.
        c->rule_regions.resize(R);
        for (int r = 0; r < R; r++) {
            blance_ctx::RuleRegions& rr = c->rule_regions[r];
            const AnchorSet* t = &tab[(size_t)r * (NX + 1)];
            std::vector<std::pair<int, int>> iv;
            for (int a = 0; a < NX; a++) {
                int lp = pb->node_leaf_pos[a];
                if (lp >= 0 && t[a].alo <= lp && lp < t[a].ahi) iv.emplace_back(t[a].alo, t[a].ahi);
            }
            std::sort(iv.begin(), iv.end());
            iv.erase(std::unique(iv.begin(), iv.end()), iv.end());
            bool ok = iv.size() >= 2;
            for (size_t i = 1; i < iv.size() && ok; i++) if (iv[i].first < iv[i - 1].second) ok = false;
            std::vector<int32_t> node_region((size_t)NX, -1), rlo, rhi;
            int max_size = 0;
            if (ok) {
                for (auto& x : iv) {
                    rlo.push_back(x.first); rhi.push_back(x.second);
                    if (x.second - x.first > max_size) max_size = x.second - x.first;
                }
                for (int a = 0; a < NX && ok; a++) {
                    int lp = pb->node_leaf_pos[a];
                    if (lp < 0) continue;
                    size_t j = std::upper_bound(rlo.begin(), rlo.end(), lp) - rlo.begin();
                    if (j == 0 || lp >= rhi[j - 1]) continue;
                    if (t[a].alo != rlo[j - 1] || t[a].ahi != rhi[j - 1]) ok = false;
                    node_region[a] = (int)j - 1;
                }
            }
            if (max_size > kChainMaxLeaves) ok = false;
            // Exclude classes: inside a region the anchors' exclude intervals must be
            // pairwise disjoint (racks inside a zone), so "leaf is excluded by anchor a"
            // is "leaf has a's class".  Intervals that cover the region get class -1.
            std::vector<int32_t> leaf_cls((size_t)n_leaves, -1), cls_size((size_t)n_leaves, 0);
            int cls_run = -1;
            for (size_t g = 0; g < rlo.size() && ok; g++) {
                std::vector<std::pair<int, int>> cl;
                for (int lp = rlo[g]; lp < rhi[g]; lp++) {
                    int a = leaf_node[lp];
                    if (a < 0) continue;
                    int bl = t[a].blo, bh = t[a].bhi;
                    if (rlo[g] <= bl && bh <= rhi[g] && bh - bl < rhi[g] - rlo[g]) cl.emplace_back(bl, bh);
                }
                std::sort(cl.begin(), cl.end());
                cl.erase(std::unique(cl.begin(), cl.end()), cl.end());
                for (size_t i = 1; i < cl.size() && ok; i++) if (cl[i].first < cl[i - 1].second) ok = false;
                for (size_t i = 0; i < cl.size() && ok; i++) {       // equal, aligned, power-of-two runs? (k_pass_chain_planes)
                    const int sz = cl[i].second - cl[i].first;
                    if (cls_run == -1) cls_run = sz;
                    if (sz != cls_run || sz < 1 || sz > 64 || (sz & (sz - 1)) || (cl[i].first - rlo[g]) % sz) cls_run = 0;
                    for (int lp = cl[i].first; lp < cl[i].second && cls_run > 0; lp++) if (leaf_node[lp] < 0) cls_run = 0;
                }
                for (size_t i = 0; i < cl.size() && ok; i++) cls_size[rlo[g] + i] = cl[i].second - cl[i].first;
                for (int lp = rlo[g]; lp < rhi[g] && ok; lp++) {
                    int a = leaf_node[lp];
                    if (a < 0) continue;
                    // the class whose interval holds this leaf must be the node's own exclude interval
                    size_t j = std::upper_bound(cl.begin(), cl.end(), std::make_pair(lp, INT_MAX)) - cl.begin();
                    if (j > 0 && lp < cl[j - 1].second) {
                        if (t[a].blo != cl[j - 1].first || t[a].bhi != cl[j - 1].second) ok = false;
                        leaf_cls[lp] = (int)j - 1;
                    }
                }
            }
            rr.ok = ok;
            rr.n_regions = ok ? (int)rlo.size() : 0;
            rr.max_size = max_size;
            rr.n_leaves = n_leaves;
            rr.cls_run = ok && cls_run > 0 ? cls_run : 0;
            std::vector<int32_t> wg_region, wg_chunk;
            if (ok)
                for (size_t g = 0; g < rlo.size(); g++)
                    for (int ch = 0; ch * 64 < rhi[g] - rlo[g]; ch++) { wg_region.push_back((int32_t)g); wg_chunk.push_back(ch); }
            rr.n_stay_wgs = (int)wg_region.size();
            if (ok) {
                if (put(up, rr.wg_region, wg_region.data(), wg_region.size())) return BLANCE_ERR_DEVICE;
                if (put(up, rr.wg_chunk, wg_chunk.data(), wg_chunk.size())) return BLANCE_ERR_DEVICE;
                if (put(up, rr.node_region, node_region.data(), node_region.size())) return BLANCE_ERR_DEVICE;
                if (put(up, rr.reg_lo, rlo.data(), rlo.size())) return BLANCE_ERR_DEVICE;
                if (put(up, rr.reg_hi, rhi.data(), rhi.size())) return BLANCE_ERR_DEVICE;
                if (put(up, rr.leaf_cls, leaf_cls.data(), leaf_cls.size())) return BLANCE_ERR_DEVICE;
                if (put(up, rr.cls_size, cls_size.data(), cls_size.size())) return BLANCE_ERR_DEVICE;
            }
            if ((st = up.flush())) return st;             // this rule's staging vectors go out of scope (their bytes are in the staging buffer)
        }
        if ((st = up.flush())) return st;                 // ... and the anchor / leaf tables
    }
    const int RW = kRecHead + M * (1 + L);       // header + per-state lists
    int kmax = 1;
    for (int m = 0; m < M; m++) if (pb->state_constraints[m] > kmax) kmax = pb->state_constraints[m];
    RESERVE(live, sizeof(int32_t) * (size_t)(PM * L + 1));
    RESERVE(live_len, sizeof(int32_t) * (size_t)(PM + 1));
    RESERVE(live_kind, (size_t)PM + 1);
    RESERVE(prv, sizeof(int32_t) * (size_t)(PM * L + 1));
    RESERVE(prv_len, sizeof(int32_t) * (size_t)(PM + 1));
    RESERVE(prv_kind, (size_t)PM + 1);
    RESERVE(in_prev, (size_t)P + 1);
    RESERVE(never_equal, (size_t)P + 1);
    RESERVE(cnt, sizeof(int32_t) * (size_t)(M + 1) * (NX + 1));
    RESERVE(ntn, sizeof(int32_t) * (size_t)(NX + 1) * (N > 0 ? N : 1));
    RESERVE(cat, (size_t)P + 1);
    RESERVE(order, sizeof(int32_t) * ((size_t)P + 1));
    RESERVE(chunk_counts, sizeof(int32_t) * 3 * (size_t)(cdiv(P, kPartChunk) + 1));
    RESERVE(rec, sizeof(int32_t) * ((size_t)P * RW + 64));
    RESERVE(out, sizeof(int32_t) * ((size_t)P * (1 + kmax) + 1));
    RESERVE(warn_part, sizeof(int32_t) * (size_t)(PM + 1));
    RESERVE(warn_state, sizeof(int32_t) * (size_t)(PM + 1));
    RESERVE(scalars, 256);           // [16..17] k_pass_queue's stop words, [18..25] its statistics (long long x 4)
    RESERVE(ntn_bits, sizeof(uint32_t) * (queue_bits_words(NX) + 4));
    {
        int maxB = 1;
        for (auto& rr : c->rule_regions) if (rr.ok && rr.n_regions > maxB) maxB = rr.n_regions;
        RESERVE(regid, sizeof(int32_t) * ((size_t)P + 1));
        RESERVE(chain_order, sizeof(int32_t) * ((size_t)P + 1));
        RESERVE(bucket_counts, sizeof(int32_t) * ((size_t)maxB * (cdiv(P, kPartChunk) + 1) + 1));
        RESERVE(reg_off, sizeof(int32_t) * ((size_t)maxB + 2));
        RESERVE(cnt_save, sizeof(int32_t) * (size_t)(M + 1) * (NX + 1));
        if (maxB > 1) {
            const size_t emax = (size_t)P * L + 1;
            RESERVE(n_ev, sizeof(int32_t) * ((size_t)P + 2));
            RESERVE(chain_oi, sizeof(int32_t) * ((size_t)P + 1));
            RESERVE(ev_key, sizeof(int32_t) * emax);
            RESERVE(ev_oi, sizeof(int32_t) * emax);
            RESERVE(ev_leaf, sizeof(int32_t) * emax);
            RESERVE(ev_w, sizeof(int32_t) * emax);
            RESERVE(ev_perm, sizeof(int32_t) * emax);
            RESERVE(ev_off, sizeof(int32_t) * ((size_t)maxB + 2));
            RESERVE(ev_counts, sizeof(int32_t) * ((size_t)maxB * (cdiv((int64_t)emax, kPartChunk) + 1) + 1));
        }
        c->flat_chain_ok = NX >= 1 && NX <= 256 && L <= kChainOwn;   // wider: k_pass_seq is faster (measured)
        if (maxB > 1 || c->flat_chain_ok) RESERVE(crec, sizeof(int32_t) * ((size_t)P * kCW + 64));
        if (c->flat_chain_ok) {
            std::vector<int32_t> iota((size_t)NX + 1), zero((size_t)NX + 1, 0), one((size_t)NX + 1, 1);
            for (int i = 0; i <= NX; i++) iota[i] = i;
            int32_t lo0 = 0, hi0 = NX;
            PUT(fl_iota, iota.data(), iota.size());
            PUT(fl_zero, zero.data(), zero.size());
            PUT(fl_one, one.data(), one.size());
            PUT(fl_reglo, &lo0, 1);
            PUT(fl_reghi, &hi0, 1);
            if ((st = up.flush())) return st;             // iota / zero / one / lo0 / hi0 are locals
        }
        RESERVE(f_tot, sizeof(int32_t) * ((size_t)NX + 1));
        RESERVE(f_g, sizeof(double) * ((size_t)NX + 1));
        RESERVE(f_top_g, sizeof(double) * kTopList);
        RESERVE(f_top_n, sizeof(int32_t) * kTopList);
        RESERVE(f_row_count, sizeof(int32_t) * ((size_t)NX + 2));
        RESERVE(f_m, sizeof(int32_t) * ((size_t)N + 2));
        RESERVE(f_moff, sizeof(int32_t) * ((size_t)N + 2));
        RESERVE(f_keys_a, sizeof(unsigned long long) * (2 * (size_t)P + 4));      // fresh runs: up to 2 picks per step, + 1
        RESERVE(f_keys_b, sizeof(unsigned long long) * (2 * (size_t)P + 4));
        RESERVE(f_vals_a, sizeof(int32_t) * (2 * (size_t)P + 4));
        RESERVE(f_vals_b, sizeof(int32_t) * (2 * (size_t)P + 4));
        RESERVE(f_hist, sizeof(int32_t) * 256 * ((size_t)cdiv(2 * (int64_t)P + 4, kSortTile) + 1));
    }
    if ((st = up.flush())) return st;
    HIPTRY(stream_sync(c));
    // the caller's arrays are not retained: drop the host pointers
    blance_problem& h = c->h;
    h.state_priority = h.state_constraints = h.state_stickiness = nullptr;
    h.state_has_stickiness = h.node_removed = h.node_added = nullptr;
    c->uploaded = true;
    return BLANCE_OK;
}

// plan.go:266: a state pass starts from an empty nodeToNodeCounts.  Called by whatever is about to read or bump the matrix
// in HBM (NumPartitions > 0); the first such call of a pass zeroes it (a pass that never calls this -- region chains with their
// rows in LDS, a pass that is one run of stays -- does not pay for the 67 MB).
static int ntn_prepare(blance_ctx* c) {
    if (!c->pass_ntn_ready) {
        const blance_problem& h = c->h;
        HIPTRY(hipMemsetAsync(c->ntn.p, 0, sizeof(int32_t) * (size_t)(h.n_nodes_ext + 1) * (h.n_nodes > 0 ? h.n_nodes : 1), c->stream));
        HIPTRY(hipMemsetAsync(c->ntn_bits.p, 0, sizeof(uint32_t) * queue_bits_words(h.n_nodes_ext), c->stream));
        c->pass_ntn_ready = true;
        c->bits_stale = false;
    }
    return 0;
}
#define NTNTRY() do { int e__ = ntn_prepare(c); if (e__) return e__; } while (0)

// A state pass (or a sub-range of one) in order.  Flat passes (no hierarchy rule for the state) of up
// to kTreeMaxNodes node names: one wave64 with bound-ordered candidates (k_pass_tree.h) -- the cost
// of a step does not grow with the cluster; everything else: the workgroup pass k_pass_seq.
static int dispatch_pass_tree_or_seq(blance_ctx* c, const PassParams& q) {
    if (q.NP > 0) NTNTRY();
    c->bits_stale = true;
    const bool tree = !c->no_tree && c->engine != BLANCE_ENGINE_SEQUENTIAL && (c->force_threads == 0 || c->tree_always);
    if (tree && launch_pass_tree(c->stream, q, (c->tree_dense ? 1 : 0) | (c->tree_long ? 2 : 0))) {
        if (c->trace) fprintf(stderr, "[blance] k_pass_tree state %d steps [%d, %d) k %d\n", q.s, q.beg, q.end, q.k);
        return 0;
    }
    if (launch_pass_seq(c->stream, q, c->force_threads, !c->no_seq_spec && c->engine != BLANCE_ENGINE_SEQUENTIAL))
        return fail(BLANCE_ERR_UNSUPPORTED, "too many nodes for the register-resident pass");
    return 0;
}

// Flat passes with k <= 2 first go to k_pass_queue (k_pass_queue.h: the candidates as a sorted window over the lanes of
// one wave64).  It stops at a step it does not take (q.stop); k_pass_tree / k_pass_seq then do a few steps -- more after
// every stop that comes soon after the last one -- and the queue kernel takes over again.
static int dispatch_pass(blance_ctx* c, const PassParams& q0) {
    const bool queue = !c->no_queue && !c->no_tree && c->engine != BLANCE_ENGINE_SEQUENTIAL && (c->force_threads == 0 || c->tree_always) &&
                       !c->tree_dense && !c->tree_long && q0.k <= 2 && q0.rule_begin >= q0.rule_end && q0.NX >= 1 && q0.NX <= 4096;
    if (!queue) return dispatch_pass_tree_or_seq(c, q0);
    PassParams q = q0;
    if (q.NP > 0) NTNTRY();
    int32_t* scal = c->scalars.as<int32_t>();
    q.ntn_bits = c->ntn_bits.as<uint32_t>();
    q.stop = scal + 16;
    q.qstats = (long long*)(scal + 18);
    q.spec = (c->queue_general ? 8 : 0) | (c->queue_force_dense ? 16 : 0) | (c->queue_no_asm ? 32 : 0) | (c->queue_exact_rebuild ? 64 : 0) |
             (c->queue_bits_self ? 128 : 0);
    int pos = q0.beg, chunk = 64;
    while (pos < q0.end) {
        q.beg = pos; q.end = q0.end;
        if (q.NP > 0 && c->bits_stale) {
            const long long words = (long long)queue_bits_words(q.NX);
            BLANCE_LAUNCH(k_ntn_bits, cdiv(q.NX + 1, 4), 256, 0, c->stream, q.N, q.NX + 1, (int)(words / (q.NX + 1)), q.ntn, q.ntn_bits);
            c->bits_stale = false;
        }
        if (!launch_pass_queue(c->stream, q)) { q.beg = pos; return dispatch_pass_tree_or_seq(c, q); }
        int32_t st[2] = {0, 0};
        HIPTRY(read_back(c, st, scal + 16, sizeof st));
        HIPTRY(stream_sync(c));
        c->queue_launches++;
        if (c->trace) fprintf(stderr, "[blance] k_pass_queue state %d steps [%d, %d) k %d: stopped at %d (%d)\n", q.s, pos, q0.end, q.k, st[0], st[1]);
        if (st[0] < pos || st[0] > q0.end) return fail(BLANCE_ERR_DEVICE, "k_pass_queue returned a position outside its range");
        if (st[0] >= q0.end) break;
        c->queue_stops++;
        chunk = st[0] - pos < 4096 ? (chunk < 65536 ? chunk * 2 : chunk) : 64;
        PassParams t = q0;
        t.beg = st[0];
        t.end = q0.end - st[0] < chunk ? q0.end : st[0] + chunk;
        const int e = dispatch_pass_tree_or_seq(c, t);
        if (e) return e;
        pos = t.end;
    }
    return 0;
}

// exclusive scan of n ints on the planner's stream (k_sweep.h: one workgroup for short arrays, tile
// totals + per-tile scans over the whole chip for long ones)
static int launch_scan_excl_on(blance_ctx* c, hipStream_t stream, DevBuf& sums, int n, int32_t* data) {
    if (n <= 4 * kScanTile) {
        BLANCE_LAUNCH(k_scan_excl, 1, 1024, 256, stream, n, data);
        return 0;
    }
    const int tiles = cdiv(n, kScanTile);
    if (sums.reserve(sizeof(int32_t) * ((size_t)tiles + 1))) return fail(BLANCE_ERR_DEVICE, "hipMalloc failed");
    BLANCE_LAUNCH(k_scan_tile_sums, tiles, 1024, 256, stream, n, data, sums.as<int32_t>());
    BLANCE_LAUNCH(k_scan_excl, 1, 1024, 256, stream, tiles, sums.as<int32_t>());
    BLANCE_LAUNCH(k_scan_apply, tiles, 1024, 256, stream, n, data, sums.as<int32_t>());
    return 0;
}
static int launch_scan_excl(blance_ctx* c, int n, int32_t* data) { return launch_scan_excl_on(c, c->stream, c->scan_sums, n, data); }
#define SCANTRY(n, data) do { int e__ = launch_scan_excl(c, (n), (data)); if (e__) return e__; } while (0)



// stable LSD radix sort of the n (key, value) pairs in the f_*_a buffers; *sorted_vals = the buffer the sorted
// values ended in (a or b: no copy back), *other_vals = the other one (free for the caller)
// (known_varying: the key bits that can differ at all, when the caller can tell -- no varbits launch, no round trip)
static int radix_sort_pairs(blance_ctx* c, int n, int64_t* launches, int32_t** sorted_vals, int32_t** other_vals,
                            const unsigned long long* known_varying) {
    const int n_tiles = cdiv(n, kSortTile);
    unsigned long long* ka = c->f_keys_a.as<unsigned long long>();
    unsigned long long* kb = c->f_keys_b.as<unsigned long long>();
    int32_t* va = c->f_vals_a.as<int32_t>();
    int32_t* vb = c->f_vals_b.as<int32_t>();
    // byte positions equal in every key need no pass (scores of one pass share most of their bits)
    unsigned long long* vbits = (unsigned long long*)(c->scalars.as<int32_t>() + 14);
    unsigned long long varying = 0;
    if (known_varying) {
        varying = *known_varying;
    } else {
        HIPTRY(hipMemsetAsync(vbits, 0, sizeof varying, c->stream));
        BLANCE_LAUNCH(k_sort_varbits, cdiv(n, 256 * kVarbitsPer), 256, 0, c->stream, n, ka, vbits);
        HIPTRY(read_back(c, &varying, vbits, sizeof varying));
        HIPTRY(stream_sync(c));
        *launches += 1;
    }
    int done = 0;
    for (int shift = 0; shift < 64; shift += 8) {
        if (((varying >> shift) & 0xff) == 0) continue;
        BLANCE_LAUNCH(k_sort_hist, n_tiles, 64, 1024 + 64, c->stream, n, shift, ka, n_tiles, c->f_hist.as<int32_t>());
        SCANTRY(256 * n_tiles, c->f_hist.as<int32_t>());
        BLANCE_LAUNCH(k_sort_scatter, n_tiles, 64, 1024 + 64, c->stream, n, shift, ka, va, kb, vb, n_tiles,
                      c->f_hist.as<int32_t>());
        std::swap(ka, kb);
        std::swap(va, vb);
        *launches += 3;
        done++;
    }
    (void)done;
    *sorted_vals = va;                               // (the loop swapped the roles after every pass)
    *other_vals = vb;
    return 0;
}

static bool dispatch_chain(blance_ctx* c, ChainParams& q, int max_size);

// Steps [beg, end) of a flat pass on ONE wave64 (clusters of <= 256 node names): the
// chain kernel with the whole cluster as its region and every node its own exclude
// class.  Where the chain cannot go on exactly it stops; k_pass_seq does a few steps
// and the chain resumes.  crec must hold the pass's compact records in pass order.
static int run_flat_chain(blance_ctx* c, PassParams q, int beg, int end, bool lds_rows, int32_t* scal,
                          int64_t* launches) {
    hipStream_t sm = c->stream;
    if (q.NP > 0) NTNTRY();
    c->bits_stale = true;
    ChainParams cq;
    memset(&cq, 0, sizeof cq);
    cq.N = q.N; cq.NX = q.NX; cq.M = q.M; cq.L = q.L; cq.s = q.s; cq.k = q.k; cq.NP = q.NP; cq.OW = q.OW;
    cq.booster_kind = q.booster_kind;
    cq.n_regions = 1; cq.n_launch = 1; cq.flat = 1; cq.waves = c->chain_waves;
    cq.reg_lo = c->fl_reglo.as<int32_t>(); cq.reg_hi = c->fl_reghi.as<int32_t>();
    cq.reg_off = c->reg_off.as<int32_t>();
    cq.leaf_node = c->fl_iota.as<int32_t>(); cq.leaf_cls = c->fl_iota.as<int32_t>(); cq.cls_size = c->fl_one.as<int32_t>();
    cq.alive = q.alive; cq.node_weight = q.node_weight; cq.node_has_weight = q.node_has_weight;
    cq.cnt = q.cnt; cq.ntn = q.ntn; cq.crec = c->crec.as<int32_t>(); cq.out = q.out; cq.flags = scal + 4;
    int pos = beg;
    c->chain_group_state = -1;                       // (reg_off is this chain's range from here on)
    c->group_epoch++;
    while (pos < end) {
        int32_t range[2] = {pos, end};
        HIPTRY(hipMemcpyAsync(c->reg_off.p, range, sizeof range, hipMemcpyHostToDevice, sm));
        HIPTRY(hipMemsetAsync(scal + 4, 0, 32, sm));
        c->flags_clean = false;
        cq.ntn_in_lds = lds_rows ? 1 : 0;
        if (!dispatch_chain(c, cq, q.NX)) return fail(BLANCE_ERR_UNSUPPORTED, "flat chain shape");
        int32_t fl[8] = {0};
        HIPTRY(read_back(c, fl, scal + 4, sizeof fl));
        HIPTRY(stream_sync(c));
        *launches += 1;
        lds_rows = false;                          // a stopped chain handed its rows to global memory
        if (fl[0]) return 1;                       // a step the compact record cannot hold: caller falls back
        if (!fl[1]) break;                         // ran to the end
        if (fl[5]) c->no_fast_keys = true;
        int stop = fl[4];
        int nseq = end - stop < 16 ? end - stop : 16;
        q.beg = stop; q.end = stop + nseq;
        int e = dispatch_pass(c, q);
        if (e) return e;
        *launches += 1;
        pos = stop + nseq;
    }
    return 0;
}

// The compact records a flat single chain walks (clusters of <= 256 names), made when a pass first needs them: a pass the
// bulk runs settle entirely (config 2: every pass) never pays for the gather and its round trip.
struct FlatChainPrep {
    bool possible = false, done = false, ok = false;
    DevProblem d;
    int m = 0, higher_mask = 0;
    const int32_t* order = nullptr;
};
static int flat_chain_prepare(blance_ctx* c, FlatChainPrep& fc, int64_t* launches) {
    if (fc.done || !fc.possible) return 0;
    fc.done = true;
    const blance_problem& h = c->h;
    hipStream_t sm = c->stream;
    int32_t* scal = c->scalars.as<int32_t>();
    HIPTRY(hipMemsetAsync(scal + 4, 0, 32, sm));
    c->flags_clean = false;
    BLANCE_LAUNCH(k_gather_chain, cdiv(h.n_parts, 256), 256, sizeof(int32_t) * 256 * (kCW + 1) + 64, sm, fc.d, fc.m, h.top_state, fc.higher_mask,
                         fc.order, (const int32_t*)nullptr, c->state_stick.as<int32_t>(),
                         c->state_has_stick.as<uint8_t>(), c->fl_iota.as<int32_t>(),
                         c->fl_zero.as<int32_t>(), c->fl_reglo.as<int32_t>(), c->fl_iota.as<int32_t>(),
                         c->fl_one.as<int32_t>(), 1,
                         c->crec.as<int32_t>(), scal + 4, (int32_t*)nullptr);
    int32_t bad = 0;
    HIPTRY(read_back(c, &bad, scal + 4, sizeof bad));
    HIPTRY(stream_sync(c));
    *launches += 1;
    fc.ok = !bad;                                  // (bad: some step does not fit the compact record)
    return 0;
}

// A flat pass (no hierarchy rule for the state): runs of certain stays and of
// fresh identical partitions are resolved in bulk, the rest by k_pass_seq in
// order on sub-ranges.  See the "Flat bulk engine" comment above the kernels.
// opening: a pass of a plan's first sweep whose steps the host knows to be fresh and alike without looking -- the first pass
// over partitions that hold nothing, or the second when the first gave every partition ONE node in a state of higher priority
// and NumPartitions == 0 (k_flat_scan's test for "fresh": such a node is just not a candidate).  *whole_known: the pass was
// such a run from its first step to its last.
static int run_flat_pass(blance_ctx* c, PassParams q, int32_t* scal, int64_t* launches, int64_t* batched,
                         FlatChainPrep& fc, bool opening, bool* whole_known, bool settled, bool* nothing_to_apply, bool rows_counted) {
    *whole_known = false;
    *nothing_to_apply = false;
    hipStream_t sm = c->stream;
    c->bits_stale = true;                           // (the bulk kernels below bump nodeToNodeCounts, not k_pass_queue's bit maps)
    const int P = q.P;
    FlatParams fq;
    memset(&fq, 0, sizeof fq);
    fq.N = q.N; fq.NX = q.NX; fq.M = q.M; fq.L = q.L; fq.P = P; fq.s = q.s; fq.k = q.k; fq.top_state = q.top_state;
    fq.NP = q.NP; fq.RW = q.RW; fq.OW = q.OW; fq.higher_mask = q.higher_mask; fq.booster_kind = q.booster_kind;
    fq.alive = q.alive; fq.node_weight = q.node_weight; fq.node_has_weight = q.node_has_weight;
    fq.cnt = q.cnt; fq.tot = c->f_tot.as<int32_t>(); fq.g = c->f_g.as<double>();
    fq.top_g = c->f_top_g.as<double>(); fq.top_n = c->f_top_n.as<int32_t>();
    fq.row_count = c->f_row_count.as<int32_t>();
    fq.ntn = q.ntn; fq.rec = q.rec; fq.out = q.out; fq.scan = scal + 8;
    fq.int_keys = (q.NP == 0 && !c->any_node_weight) ? 1 : 0;
    if (q.NP > 0 && !rows_counted) {                // only read by the stay test when NP > 0; (else: k_gather has counted)
        if (!c->rowcount_clean) HIPTRY(hipMemsetAsync(c->f_row_count.p, 0, sizeof(int32_t) * ((size_t)q.NX + 1), sm));
        c->rowcount_clean = false;
        BLANCE_LAUNCH(k_flat_row_count, cdiv(P, 256), 256, 0, sm, fq, c->f_row_count.as<int32_t>());
        *launches += 1;
    }
    // bulk paths have fixed costs (a host round trip, a sort): short runs stay sequential
    const int kMinStayRun = c->chain_min_parts < 64 ? c->chain_min_parts : 64;
    const int kMinFreshRun = c->chain_min_parts < 512 ? c->chain_min_parts : 512;
    int pos = 0, seq_batch = 256;
    bool dirty = true;
    while (pos < P) {
        c->bits_stale = true;                       // (conservative: a bulk commit may have run since the last k_pass_queue)
        int32_t got[2] = {0, 0};
        // A plan from nothing: the partitions to assign hold no node and have no weights of their own -- no step of the
        // opening pass is a stay (k_flat_scan: a stay keeps the ONE node the partition holds) and every step is fresh and
        // identical to the first (weight 1, nothing held anywhere): the scan's answer without the scan.
        const bool known_run = opening && pos == 0 && c->assign_empty && c->h.partition_weights_nil && c->speculate > 0;
        if (known_run) {
            got[0] = 0; got[1] = P;
        } else {
            if (dirty) {                            // the load totals and the smallest partition-independent scores: what the scan tests against
                BLANCE_LAUNCH(k_flat_prepare, 1, 1024, sizeof(RedSlot) * 32 + 64, sm, fq, c->f_tot.as<int32_t>(),
                              c->f_g.as<double>(), c->f_top_g.as<double>(), c->f_top_n.as<int32_t>());
                dirty = false;
                *launches += 1;
            }
            const int scan_blocks = cdiv(P - pos, 256);
            fq.scan_waves = scan_blocks * 4;
            if (c->scan_part.reserve(sizeof(int32_t) * 2 * ((size_t)fq.scan_waves + 1))) return fail(BLANCE_ERR_DEVICE, "hipMalloc failed");
            fq.scan_part = c->scan_part.as<int32_t>();
            BLANCE_LAUNCH(k_flat_scan, scan_blocks, 256, 0, sm, fq, pos, P);
            BLANCE_LAUNCH(k_flat_scan_min, 1, 1024, 256, sm, fq.scan_waves, (const int32_t*)fq.scan_part, scal + 8);
            HIPTRY(read_back(c, got, scal + 8, sizeof got));
            HIPTRY(stream_sync(c));
            *launches += 1;
        }
        int first_nonstay = got[0] > P ? P : got[0], first_nonfresh = got[1] > P ? P : got[1];
        if (first_nonstay - pos >= kMinStayRun || (first_nonstay == P && first_nonstay > pos)) {
            const bool whole = pos == 0 && first_nonstay == P;      // the pass is one run of stays: no one reads the matrix
            if (whole && q.s == q.top_state && q.k == 1) c->tops_moved = false;      // (k > 1: a stay may still reorder the list, plan.go:126-138 reads its first node)
            if (q.NP > 0 && !whole) NTNTRY();
            // One run of certain stays with ONE node each, as the first pass of a sweep >= 2 (`settled`: every present list is a
            // non-nil slice, plan.go:418): applying it (plan.go:290-299) changes nothing -- the partition keeps the node it holds,
            // which is in no other list of the partition (k_flat_scan's test), so no list is filtered and no kind changes.
            // Neither the outputs nor k_scatter are needed then.
            if (whole && q.k == 1 && settled) {
                *nothing_to_apply = true;
                *batched += P;
                break;
            }
            BLANCE_LAUNCH_NOSYNC(k_flat_commit_stay, cdiv(first_nonstay - pos, 256), 256, 0, sm, fq, pos, first_nonstay, whole ? 0 : 1);
            *launches += 1;
            *batched += first_nonstay - pos;
            pos = first_nonstay;
            continue;
        }
        if (first_nonfresh - pos >= kMinFreshRun && c->n_alive > 0) {
            int R = first_nonfresh - pos;
            // NumPartitions == 0: steps may each exclude one node (k_fresh_excl); one more element of the
            // exclusion-free sequence is needed then
            // NumPartitions == 0: steps may exclude one node each and take two (k_fresh_excl); the exclusion-free
            // sequence then needs k elements per step and one more
            const bool excl = q.NP == 0 && (q.higher_mask != 0 || q.k == 2);
            const int RS = excl ? q.k * R + q.k : R;
            if (q.NP > 0) NTNTRY();                 // (the fresh run reads and bumps row "" of the matrix)
            int32_t *sorted_vals = nullptr, *other_vals = nullptr;
            // Integer keys (NumPartitions == 0, no node weights) from counters that all start at zero, step weight 1
            // (known_run): node n's elements are (0, n), (1, n), (2, n) ..; the RS smallest in (key, node) order are the A
            // nodes of nodesNext by id, again and again -- the sorted sequence and every node's share of it without
            // threshold search, emission and sort (the empty cluster's greedy plan is a round robin).
            if (known_run && fq.int_keys && c->counts_start_zero) {
                sorted_vals = c->f_vals_a.as<int32_t>();
                other_vals = c->f_vals_b.as<int32_t>();
                BLANCE_LAUNCH_NOSYNC(k_fresh_cycle, cdiv(RS > q.N ? RS : q.N, 256), 256, 0, sm, RS, c->n_alive, q.N,
                                     c->alive_ids.as<int32_t>(), c->alive_rank.as<int32_t>(), sorted_vals, c->f_m.as<int32_t>());
                *launches += 1;
            } else {
                if (dirty) {
                    BLANCE_LAUNCH(k_flat_prepare, 1, 1024, sizeof(RedSlot) * 32 + 64, sm, fq, c->f_tot.as<int32_t>(),
                                  c->f_g.as<double>(), c->f_top_g.as<double>(), c->f_top_n.as<int32_t>());
                    dirty = false;
                    *launches += 1;
                }
                BLANCE_LAUNCH(k_fresh_threshold, 1, 1024, 16384 + 64, sm, fq, pos, RS, c->f_m.as<int32_t>(),
                              c->f_moff.as<int32_t>());
                BLANCE_LAUNCH_NOSYNC(k_fresh_emit, cdiv(RS, 256), 256, 0, sm, fq, pos, RS, c->f_moff.as<int32_t>(),
                                     c->f_keys_a.as<unsigned long long>(), c->f_vals_a.as<int32_t>());
                const int e = radix_sort_pairs(c, RS, launches, &sorted_vals, &other_vals, nullptr);
                if (e) return e;
            }
            const int32_t* picks = sorted_vals;
            if (excl) {
                int32_t bad = INT_MAX;
                HIPTRY(hipMemcpyAsync(scal + 10, &bad, sizeof bad, hipMemcpyHostToDevice, sm));
                // (threads of a few steps each: coalesced record reads; up to 64 workgroups)
                int G = cdiv(R, 4 * 1024);
                G = G < 1 ? 1 : G > 64 ? 64 : G;
                RESERVE(f_comp, (size_t)G * 1024 + 64 + 64);
                unsigned char* comp = c->f_comp.as<unsigned char>();
                BLANCE_LAUNCH(k_fresh_excl_scan, G, 1024, 2048 + 64, sm, fq, pos, R, sorted_vals, comp, comp + (size_t)G * 1024);
                BLANCE_LAUNCH_NOSYNC(k_fresh_excl_apply, G, 1024, 0, sm, fq, pos, R, sorted_vals, comp, comp + (size_t)G * 1024, other_vals, scal + 10);
                HIPTRY(read_back(c, &bad, scal + 10, sizeof bad));
                HIPTRY(stream_sync(c));
                *launches += 1;
                if (bad < R) R = bad;               // a pending node came up again: the run ends before that step
                picks = other_vals;
                HIPTRY(hipMemsetAsync(c->f_m.p, 0, sizeof(int32_t) * ((size_t)q.N + 1), sm));
                if (R > 0) BLANCE_LAUNCH_NOSYNC(k_fresh_hist, cdiv((int64_t)q.k * R, 256), 256, 0, sm, q.k * R, picks, c->f_m.as<int32_t>());
            }
            if (R > 0) {
                BLANCE_LAUNCH_NOSYNC(k_fresh_commit_steps, cdiv(R, 256), 256, 0, sm, fq, pos, R, picks);
                BLANCE_LAUNCH_NOSYNC(k_fresh_commit_nodes, cdiv(q.N, 256), 256, 0, sm, fq, pos, c->f_m.as<int32_t>(), q.cnt);
                *launches += 4;
                *batched += R;
                if (known_run && R == P) *whole_known = true;
                pos += R;
                dirty = true;
                seq_batch = 256;
                continue;
            }
        }
        int B = P - pos < seq_batch ? P - pos : seq_batch;
        int e = flat_chain_prepare(c, fc, launches);
        if (e) return e;
        if (fc.ok) {                                  // small cluster: one wave64 walks the batch
            e = run_flat_chain(c, q, pos, pos + B, false, scal, launches);
            if (e < 0) return e;
        } else {
            q.beg = pos; q.end = pos + B;
            e = dispatch_pass(c, q);
            if (e) return e;
            *launches += 1;
        }
        pos += B;
        dirty = true;
        if (seq_batch < (1 << 20)) seq_batch *= 2;
    }
    return 0;
}

static bool dispatch_chain(blance_ctx* c, ChainParams& q, int max_size) {
    const bool fast = q.NP == 0 && !c->any_node_weight && !c->no_fast_keys;
    return launch_chain(c->stream, q, max_size, fast);
}

// developer aid: BLANCE_DUMP_SWEEP=<i> prints every step's choice of sweep i (pass order)
static int dump_pass(blance_ctx* c, int sweep, int state, int P, int OW, const int32_t* idx_dev /* or null */) {
    if (c->dump_sweep != sweep) return 0;
    std::vector<int32_t> out((size_t)P * OW), idx((size_t)P);
    HIPTRY(hipMemcpyAsync(out.data(), c->out.p, sizeof(int32_t) * out.size(), hipMemcpyDeviceToHost, c->stream));
    if (idx_dev) HIPTRY(hipMemcpyAsync(idx.data(), idx_dev, sizeof(int32_t) * P, hipMemcpyDeviceToHost, c->stream));
    HIPTRY(stream_sync(c));
    std::vector<int> at((size_t)P);
    for (int i = 0; i < P; i++) at[idx_dev ? idx[i] : i] = i;
    for (int oi = 0; oi < P; oi++) {
        const int32_t* o = &out[(size_t)at[oi] * OW];
        fprintf(stderr, "[dump] sweep %d state %d step %d:", sweep, state, oi);
        for (int j = 0; j < OW; j++) fprintf(stderr, " %d", o[j]);
        fprintf(stderr, "\n");
    }
    return 0;
}

// ---- collectives of a sharded plan ---------------------------------------------------------
#ifndef BLANCE_SIMT_EMU
#include <dlfcn.h>
struct Id128 { char b[128]; };                  // ncclUniqueId, passed by value
namespace {
// RCCL is bound at run time (dlopen): a single-GPU caller never loads it.  The two enum values
// this file passes are part of NCCL's stable ABI (nccl.h: ncclDataType_t, ncclRedOp_t).
constexpr int kNcclInt32 = 2;                   // ncclInt32 == ncclInt
constexpr int kNcclSum = 0;                     // ncclSum
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommAbort)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
}
static Rccl g_rccl;
static std::mutex g_rccl_mu;
static int rccl_load() {
    std::lock_guard<std::mutex> g(g_rccl_mu);
    if (g_rccl.lib) return 0;
    void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return fail(BLANCE_ERR_COMM, "librccl.so not found: %s", dlerror());
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(lib, "ncclCommInitRank");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(lib, "ncclAllReduce");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(lib, "ncclAllGather");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(lib, "ncclCommDestroy");
    g_rccl.CommAbort = (decltype(g_rccl.CommAbort))dlsym(lib, "ncclCommAbort");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.AllGather || !g_rccl.CommDestroy)
        return fail(BLANCE_ERR_COMM, "librccl.so lacks an entry point");
    g_rccl.lib = lib;
    return 0;
}
#endif

extern "C" int blance_is_emulated(void) {
#ifdef BLANCE_SIMT_EMU
    return 1;
#else
    return 0;
#endif
}

extern "C" int blance_comm_unique_id(void* id_out_128) {
#ifndef BLANCE_SIMT_EMU
    if (!id_out_128) return fail(BLANCE_ERR_BAD_ARG, "null id buffer");
    int st = rccl_load();
    if (st) return st;
    int e = g_rccl.GetUniqueId(id_out_128);
    if (e) return fail(BLANCE_ERR_COMM, "ncclGetUniqueId: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?");
    return BLANCE_OK;
#else
    (void)id_out_128;
    return fail(BLANCE_ERR_COMM, "no RCCL in the emulator build");
#endif
}

extern "C" int blance_comm_init_rccl(blance_ctx* c, int32_t n_ranks, int32_t rank, const void* id_128) {
    return guarded([&]() -> int {
#ifndef BLANCE_SIMT_EMU
    if (!c || !id_128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(BLANCE_ERR_BAD_ARG, "bad communicator arguments");
    std::lock_guard<std::mutex> g(c->mu);
    int st = rccl_load();
    if (st) return st;
    HIPTRY(hipSetDevice(c->device));
    if (c->rccl_comm) { g_rccl.CommDestroy(c->rccl_comm); c->rccl_comm = nullptr; }
    Id128 id;
    memcpy(id.b, id_128, sizeof id.b);
    void* comm = nullptr;
    int e = g_rccl.CommInitRank(&comm, n_ranks, id, rank);
    if (e) return fail(BLANCE_ERR_COMM, "ncclCommInitRank: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?");
    c->rccl_comm = comm;
    c->comm = blance_comm{rank, n_ranks, nullptr, nullptr, nullptr};
    return BLANCE_OK;
#else
    (void)c; (void)n_ranks; (void)rank; (void)id_128;
    return fail(BLANCE_ERR_COMM, "no RCCL in the emulator build");
#endif
    });
}

extern "C" int blance_comm_set(blance_ctx* c, const blance_comm* comm) {
    return guarded([&]() -> int {
    if (!c) return fail(BLANCE_ERR_BAD_ARG, "null ctx");
    std::lock_guard<std::mutex> g(c->mu);
    if (!comm) { c->comm = blance_comm{0, 1, nullptr, nullptr, nullptr}; return BLANCE_OK; }
    if (comm->n_ranks < 1 || comm->rank < 0 || comm->rank >= comm->n_ranks || (comm->n_ranks > 1 && !comm->allreduce_sum_i32))
        return fail(BLANCE_ERR_BAD_ARG, "bad communicator");
    c->comm = *comm;
    return BLANCE_OK;
    });
}

extern "C" int blance_comm_stats(blance_ctx* c, int64_t* calls, int64_t* words) {
    if (!c) return fail(BLANCE_ERR_BAD_ARG, "null ctx");
    std::lock_guard<std::mutex> g(c->mu);
    if (calls) *calls = c->comm_calls;
    if (words) *words = c->comm_bytes / 4;
    return BLANCE_OK;
}

extern "C" int blance_comm_time_ms(blance_ctx* c, double* ms) {
    if (!c || !ms) return fail(BLANCE_ERR_BAD_ARG, "null argument");
    std::lock_guard<std::mutex> g(c->mu);
    *ms = c->comm_ms;
    return BLANCE_OK;
}

static void comm_release(blance_ctx* c) {
#ifndef BLANCE_SIMT_EMU
    if (c->rccl_comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->rccl_comm);
#endif
    c->rccl_comm = nullptr;
}

// a sharded call failed on this rank after the others may have entered a collective: RCCL
// communicators are aborted so that no rank waits forever (the communicator is invalid afterwards)
static void comm_abort(blance_ctx* c) {
#ifndef BLANCE_SIMT_EMU
    if (c->rccl_comm && g_rccl.CommAbort) { g_rccl.CommAbort(c->rccl_comm); c->rccl_comm = nullptr; }
#endif
    (void)c;
}

// the device time a collective takes on the planner's stream: an event on either side (summed up when the plan ends)
static int comm_mark(blance_ctx* c) {
    if (c->comm_events_used == c->comm_events.size()) {
        hipEvent_t ev;
        HIPTRY(hipEventCreate(&ev));
        c->comm_events.push_back(ev);
    }
    HIPTRY(hipEventRecord(c->comm_events[c->comm_events_used++], c->stream));
    return 0;
}
static void comm_sum_up(blance_ctx* c) {            // (after the stream has been synchronised)
    for (size_t i = 0; i + 1 < c->comm_events_used; i += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->comm_events[i], c->comm_events[i + 1]) == hipSuccess) c->comm_ms += ms;
        else (void)hipGetLastError();
    }
    c->comm_events_used = 0;
}

// in-place int32 sum over the ranks, ordered with the kernels of the planner's stream
static int comm_allreduce(blance_ctx* c, int32_t* buf, int64_t n) {
    if ((c->comm.n_ranks <= 1 && !c->shard_one_rank) || n <= 0) return 0;
    c->comm_calls++;
    c->comm_bytes += n * 4;
    if (c->comm.allreduce_sum_i32) {
        HIPTRY(stream_sync(c));
        if (c->comm.allreduce_sum_i32(c->comm.user, buf, n)) return fail(BLANCE_ERR_COMM, "the caller's all-reduce failed");
        return 0;
    }
#ifndef BLANCE_SIMT_EMU
    if (!c->rccl_comm) return fail(BLANCE_ERR_COMM, "no communicator");
    if (comm_mark(c)) return BLANCE_ERR_DEVICE;
    int e = g_rccl.AllReduce(buf, buf, (size_t)n, kNcclInt32, kNcclSum, c->rccl_comm, c->stream);
    if (e) return fail(BLANCE_ERR_COMM, "ncclAllReduce: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?");
    if (comm_mark(c)) return BLANCE_ERR_DEVICE;
    return 0;
#else
    return fail(BLANCE_ERR_COMM, "no communicator");
#endif
}

// in-place all-gather of n_ranks blocks of `per_rank` int32 values (this rank's block filled in)
static int comm_allgather(blance_ctx* c, int32_t* buf, int64_t per_rank) {
    if ((c->comm.n_ranks <= 1 && !c->shard_one_rank) || per_rank <= 0) return 0;
    c->comm_calls++;
    c->comm_bytes += per_rank * 4 * c->comm.n_ranks;
    if (c->comm.allreduce_sum_i32) {
        HIPTRY(stream_sync(c));
        if (!c->comm.allgather_i32) return fail(BLANCE_ERR_COMM, "no all-gather hook");
        if (c->comm.allgather_i32(c->comm.user, buf, per_rank)) return fail(BLANCE_ERR_COMM, "the caller's all-gather failed");
        return 0;
    }
#ifndef BLANCE_SIMT_EMU
    if (!c->rccl_comm) return fail(BLANCE_ERR_COMM, "no communicator");
    if (comm_mark(c)) return BLANCE_ERR_DEVICE;
    int e = g_rccl.AllGather(buf + (size_t)c->comm.rank * per_rank, buf, (size_t)per_rank, kNcclInt32, c->rccl_comm, c->stream);
    if (e) return fail(BLANCE_ERR_COMM, "ncclAllGather: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?");
    if (comm_mark(c)) return BLANCE_ERR_DEVICE;
    return 0;
#else
    return fail(BLANCE_ERR_COMM, "no communicator");
#endif
}
#define COMMTRY(expr) do { int e__ = (expr); if (e__) return e__; } while (0)

static DevProblem dev_problem(blance_ctx* c) {
    const blance_problem& h = c->h;
    DevProblem d;
    d.N = h.n_nodes; d.NX = h.n_nodes_ext; d.M = h.n_states; d.L = c->L; d.P = h.n_parts;
    d.weights_nil = h.partition_weights_nil;
    d.node_removed = c->zeros_nx.as<uint8_t>();
    d.node_added = c->zeros_nx.as<uint8_t>();
    d.part_weight = c->part_weight.as<int32_t>();
    d.part_has_weight = c->part_has_weight.as<uint8_t>();
    d.live = c->live.as<int32_t>(); d.live_len = c->live_len.as<int32_t>(); d.live_kind = c->live_kind.as<uint8_t>();
    d.prv = c->prv.as<int32_t>(); d.prv_len = c->prv_len.as<int32_t>(); d.prv_kind = c->prv_kind.as<uint8_t>();
    d.in_prev = c->in_prev.as<uint8_t>(); d.never_equal = c->never_equal.as<uint8_t>();
    return d;
}

// ---- one state pass as region chains ----------------------------------------
// Header of the buffer the ranks of a sharded plan sum up after such a pass (collective A):
// [0..7] the chain kernels' flags, [8] poison (a rank failed), then the load-vector change.
constexpr int kXHead = 16;
struct ChainPassArgs {
    DevProblem d;
    int m, k, NP, OW, RW, higher_mask, r0, it;
    const int32_t* order;          // the pass order: the stable partition of sweep 1, the static order itself afterwards
    bool same_tops;                // no partition has changed its top priority node since this state's last chain pass
};

// this rank failed before collective A of a sharded chain pass: take part in it with the poison word set
static void comm_poison(blance_ctx* c) {
    const blance_problem& h = c->h;
    const size_t n = kXHead + (size_t)(h.n_states + 1) * h.n_nodes_ext;
    if (c->xbuf.reserve(sizeof(int32_t) * (n + 1))) return;
    if (hipMemsetAsync(c->xbuf.p, 0, sizeof(int32_t) * n, c->stream) != hipSuccess) return;
    const int32_t one = 1;
    if (hipMemcpyAsync(c->xbuf.as<int32_t>() + 8, &one, sizeof one, hipMemcpyHostToDevice, c->stream) != hipSuccess) return;
    if (stream_sync(c) != hipSuccess) return;
    (void)comm_allreduce(c, c->xbuf.as<int32_t>(), (int64_t)n);
    (void)stream_sync(c);
}

// The host round trips a chain pass may save, and what it left open.
//   allow_spec: the classification of a pass that reuses its grouping is ASSUMED clean (no events, no orphans) instead of
//     read back; the words are checked with the pass's own flags, and `redo` says the assumption was wrong: nothing of the
//     pass stands, the caller runs it again without.
//   defer: the pass's own verdict (k_stay_by_top's, the all-blank kernel's, the chain kernel's flags) is not read here
//     either: the pass is taken to stand, k_scatter is enqueued behind a Gate on those words, and `pending` tells the
//     caller which words to look at with its next readback (the sweep's convergence word: one round trip for both).  A
//     verdict that comes back bad has changed nothing but the counters (cnt_save holds them): the caller runs the pass
//     again with no_stay / no_lean / skip set accordingly.
struct ChainRun {
    bool allow_spec = true, defer = false, no_stay = false, no_lean = false, skip = false;
    bool redo = false;
    int pending = 0;               // 0: settled; 1: k_stay_by_top's verdict, 2: the all-blank kernel's, 3: the chain kernel's is still on the device
    bool spec = false;             // the classification was assumed
    Gate gate = kNoGate;
};
constexpr int kFlagStayMoved = 22, kFlagForced = 23;     // words of scal + 4: k_stay_by_top's "not all stays"; BLANCE_SPECULATE=fail

// 0 = ok (*done tells whether the pass was made; if not, the counters are as before and the caller
// runs the pass in order), < 0 = error.
static int run_chain_pass_once(blance_ctx* c, const ChainPassArgs& a, int64_t* launches_io, int64_t* batched_io, int* n_pass_io,
                               bool* done, bool* a_done, ChainRun& run) {
    const blance_problem& h = c->h;
    const DevProblem& d = a.d;
    const int N = h.n_nodes, NX = h.n_nodes_ext, M = h.n_states, P = h.n_parts, L = c->L;
    const int m = a.m, k = a.k, NP = a.NP, OW = a.OW;
    hipStream_t sm = c->stream;
    int32_t* scal = c->scalars.as<int32_t>();
    int64_t launches = 0;
    int& n_pass = *n_pass_io;
    blance_ctx::RuleRegions& rr = c->rule_regions[a.r0];
    const int B = rr.n_regions, nbc = cdiv(P, kPartChunk);
    const int G = c->comm.n_ranks, rank = c->comm.rank;
    const bool sharded = (G > 1 || c->shard_one_rank) && B >= G;
    if (!c->flags_clean) HIPTRY(hipMemsetAsync(scal + 4, 0, 32, sm));
    c->flags_clean = false;
    BLANCE_LAUNCH_NOSYNC(k_chain_classify, cdiv(P + 1, 256), 256, 0, sm, d, m, h.top_state,
                         a.order, rr.node_region.as<int32_t>(), c->regid.as<int32_t>(),
                         c->n_ev.as<int32_t>(), scal + 4);
    int nbits = 1;
    while ((1 << nbits) < B) nbits++;
    // The steps grouped by the region of their top priority node (a stable counting sort of the pass order).  A sweep whose
    // top-state pass was one run of stays has moved no top priority node, and from sweep 2 on the pass order is the static
    // order: the grouping of this state's last chain pass -- chain_order, chain_oi, reg_off -- still stands (config 3's
    // third sweep: six launches less).
    const bool regroup = !(a.same_tops && c->chain_group_state == m && c->chain_group_static && a.order == c->part_order.as<int32_t>() &&
                           c->h_reg_off.size() == (size_t)B + 1);
    if (regroup) {
        BLANCE_LAUNCH(k_part_count, nbc, 64, sizeof(int32_t) * B + 64, sm, P, c->regid.as<int32_t>(),
                      (const uint8_t*)nullptr, (const int32_t*)nullptr, nbc, B, c->bucket_counts.as<int32_t>());
        SCANTRY(B * nbc, c->bucket_counts.as<int32_t>());
        BLANCE_LAUNCH_NOSYNC(k_region_offsets, cdiv(B + 1, 64), 64, 0, sm, B, nbc, P,
                             c->bucket_counts.as<int32_t>(), c->reg_off.as<int32_t>());
        BLANCE_LAUNCH(k_part_scatter, nbc, 64, sizeof(int32_t) * B + 64, sm, P, c->regid.as<int32_t>(),
                      (const uint8_t*)nullptr, (const int32_t*)nullptr, a.order, nbc, B, nbits,
                      c->bucket_counts.as<int32_t>(), c->chain_order.as<int32_t>(), c->chain_oi.as<int32_t>());
        c->chain_group_state = m;
        c->group_epoch++;
        c->chain_group_static = a.order == c->part_order.as<int32_t>();
    } else if (c->trace) fprintf(stderr, "[blance] chain pass state %d: the grouping by region of the last sweep stands\n", m);
    // events: how many?  (also: is every step region-local at all, are there orphan nodes); a sharded
    // plan reads the chain offsets in the same round trip (the slice sizes of collective B)
    // A pass that reuses its grouping has nothing else to learn from this round trip: with the top priority nodes where they
    // were, events and orphans come from this state's own nodes having left their partition's region since -- rare enough to
    // assume there are none and to look at flags[6], flags[7] only when the pass's own flags come back.
    const bool spec = run.allow_spec && !regroup && !sharded && c->speculate > 0;
    const bool defer = run.defer && !sharded && c->speculate > 0;
    run.spec = spec;
    run.pending = 0;
    run.redo = false;
    auto gate_on = [&](uint32_t words) {               // (the words of scal + 4 a deferred verdict depends on)
        Gate g;
        g.flags = scal + 4;
        g.mask = words | (1u << kFlagForced) | (spec ? (1u << 6) | (1u << 7) : 0u);
        return g;
    };
    int32_t n_events = 0, cfl[8] = {0};
    if (!spec) {
        HIPTRY(read_back(c, cfl, scal + 4, sizeof cfl));
        if (regroup) {                                     // (always: the next sweep may reuse the grouping, the host's copy with it)
            c->h_reg_off.resize((size_t)B + 1);
            HIPTRY(read_back(c, c->h_reg_off.data(), c->reg_off.p, sizeof(int32_t) * ((size_t)B + 1)));
        }
        HIPTRY(stream_sync(c));
    }
    // (speculate == 2, tests: every assumption is treated as refuted)
    auto refuted = [&](const int32_t* f) { return spec && (f[6] || f[7] || c->speculate == 2); };
    if (!cfl[0] && cfl[7]) {                            // rare: nodes outside their partition's region
        SCANTRY(P + 1, c->n_ev.as<int32_t>());   // -> event slots
        HIPTRY(read_back(c, &n_events, c->n_ev.as<int32_t>() + P, sizeof n_events));
        HIPTRY(stream_sync(c));
    }
    if (c->trace)
        fprintf(stderr, spec ? "[blance] chain pass state %d: classification assumed clean (checked with the pass's flags)\n" :
                               "[blance] chain pass state %d: %d events, not-local %d, orphans %d\n", m, n_events, cfl[0], cfl[6]);
    const size_t cnt_words = (size_t)(M + 1) * NX;
    {   // one launch: no events yet, the counters this pass starts from (what a redo restores), k_stay_by_top's flag
        FillCopyJob fj;
        fj.zero(c->ev_off.p, (int64_t)B + 1);
        fj.zero(scal + 4 + kFlagStayMoved, 1);
        fj.copy(c->cnt_save.p, c->cnt.p, (int64_t)cnt_words);
        if (run_fill_copy(c, fj)) return BLANCE_ERR_DEVICE;
    }
    if (!cfl[0] && n_events > 0) {
        const int nec = cdiv(n_events, kPartChunk);
        BLANCE_LAUNCH_NOSYNC(k_chain_ev_fill, cdiv(P, 256), 256, 0, sm, d, m, h.top_state, a.order,
                             rr.node_region.as<int32_t>(), rr.reg_lo.as<int32_t>(), c->node_leaf_pos.as<int32_t>(),
                             c->regid.as<int32_t>(), c->n_ev.as<int32_t>(), c->ev_key.as<int32_t>(),
                             c->ev_oi.as<int32_t>(), c->ev_leaf.as<int32_t>(), c->ev_w.as<int32_t>());
        BLANCE_LAUNCH(k_part_count, nec, 64, sizeof(int32_t) * B + 64, sm, n_events, c->ev_key.as<int32_t>(),
                      (const uint8_t*)nullptr, (const int32_t*)nullptr, nec, B, c->ev_counts.as<int32_t>());
        SCANTRY(B * nec, c->ev_counts.as<int32_t>());
        BLANCE_LAUNCH_NOSYNC(k_region_offsets, cdiv(B + 1, 64), 64, 0, sm, B, nec, n_events,
                             c->ev_counts.as<int32_t>(), c->ev_off.as<int32_t>());
        BLANCE_LAUNCH(k_part_scatter, nec, 64, sizeof(int32_t) * B + 64, sm, n_events, c->ev_key.as<int32_t>(),
                      (const uint8_t*)nullptr, (const int32_t*)nullptr, (const int32_t*)nullptr, nec, B, nbits,
                      c->ev_counts.as<int32_t>(), c->ev_perm.as<int32_t>(), (int32_t*)nullptr);
        launches += 5;
    }
    // A pass of stays only (the last sweep of every plan that converges)?  Worth a try when the state's pass of the
    // sweep before was one but for a few steps: k_stay_by_top checks every step in parallel.
    const bool stay_fits = !sharded && !c->no_stay_top && NP > 0 && !cfl[0] && !cfl[6] && !cfl[7] && rr.max_size <= kStayMaxLeaves && rr.n_stay_wgs > 0;
    const bool try_stay = stay_fits && !run.no_stay && (c->force_stay_top || c->last_stays[m] * 100 >= (int64_t)P * 99);
    // k_stay_by_top's work list (the steps grouped by the leaf of their top priority node: four launches over all steps)
    // depends on the grouping by region and on the top priority nodes only.  One made for this state at the same count of
    // regroupings still holds; and a pass that does NOT try k_stay_by_top makes it for the next sweep's on the second
    // stream, beside its own chain kernel.
    const bool group_stands = c->top_group_state == m && c->top_group_epoch == c->group_epoch;
    // (The second stream is made when it is first wanted, and only in a process of one or two planners: a stream is a
    // hardware queue, and with many contexts planning at once on one GPU -- bench.py's replicas: 16 contexts, 32 queues -- the
    // planners' own streams end up sharing queues and their long kernels run one after the other.)
    bool group_ahead = stay_fits && !try_stay && !group_stands && c->speculate > 0 && a.it + 1 < h.max_iterations &&
                       (c->side || g_live_contexts.load() <= 2);
    if (group_ahead && !c->side && hipStreamCreate(&c->side) != hipSuccess) { c->side = nullptr; group_ahead = false; (void)hipGetLastError(); }
    const int BL = rr.n_leaves;
    if (try_stay || group_ahead) {
        RESERVE(topkey, sizeof(int32_t) * ((size_t)P + 1));
        RESERVE(top_order, sizeof(int32_t) * ((size_t)P + 1));
        RESERVE(top_off, sizeof(int32_t) * ((size_t)rr.n_leaves + 2));
        RESERVE(top_counts, sizeof(int32_t) * ((size_t)rr.n_leaves * nbc + 1));
        if (c->side_pending) {                         // (what the second stream reads and writes is about to be used here)
            HIPTRY(hipStreamWaitEvent(sm, c->side_done, 0));
            c->side_pending = false;
        }
    }
    const bool group_now = (try_stay && !group_stands) || group_ahead;
    BLANCE_LAUNCH(k_gather_chain, cdiv(P, 256), 256, sizeof(int32_t) * 256 * (kCW + 1) + 64, sm, d, m, h.top_state, a.higher_mask,
                         c->chain_order.as<int32_t>(), c->chain_oi.as<int32_t>(), c->state_stick.as<int32_t>(),
                         c->state_has_stick.as<uint8_t>(), c->node_leaf_pos.as<int32_t>(),
                         rr.node_region.as<int32_t>(), rr.reg_lo.as<int32_t>(), rr.leaf_cls.as<int32_t>(),
                         rr.cls_size.as<int32_t>(), 0,
                         c->crec.as<int32_t>(), scal + 4, group_now ? c->topkey.as<int32_t>() : (int32_t*)nullptr);
    // steps grouped by the leaf of their top priority node, pass order inside a group (stable counting sort)
    auto group_by_top = [&](hipStream_t st, DevBuf& sums) -> int {
        int lbits = 1;
        while ((1 << lbits) < BL) lbits++;
        BLANCE_LAUNCH(k_part_count, nbc, 64, sizeof(int32_t) * BL + 64, st, P, c->topkey.as<int32_t>(),
                      (const uint8_t*)nullptr, (const int32_t*)nullptr, nbc, BL, c->top_counts.as<int32_t>());
        const int se = launch_scan_excl_on(c, st, sums, BL * nbc, c->top_counts.as<int32_t>());
        if (se) return se;
        BLANCE_LAUNCH_NOSYNC(k_region_offsets, cdiv(BL + 1, 64), 64, 0, st, BL, nbc, P, c->top_counts.as<int32_t>(), c->top_off.as<int32_t>());
        BLANCE_LAUNCH(k_part_scatter, nbc, 64, sizeof(int32_t) * BL + 64, st, P, c->topkey.as<int32_t>(),
                      (const uint8_t*)nullptr, (const int32_t*)nullptr, (const int32_t*)nullptr, nbc, BL, lbits,
                      c->top_counts.as<int32_t>(), c->top_order.as<int32_t>(), (int32_t*)nullptr);
        c->top_group_state = m;
        c->top_group_epoch = c->group_epoch;
        launches += 5;
        return 0;
    };
    if (group_ahead) {
        HIPTRY(hipEventRecord(c->side_go, sm));        // (the keys are written)
        HIPTRY(hipStreamWaitEvent(c->side, c->side_go, 0));
        const int ge = group_by_top(c->side, c->side_sums);
        if (ge) return ge;
        HIPTRY(hipEventRecord(c->side_done, c->side));
        c->side_pending = true;
        if (c->trace) fprintf(stderr, "[blance] chain pass state %d: the steps grouped by top priority node for the next sweep, on the second stream\n", m);
    }
    ChainParams cq;
    memset(&cq, 0, sizeof cq);
    cq.N = N; cq.NX = NX; cq.M = M; cq.L = L; cq.s = m; cq.k = k;
    cq.NP = NP; cq.OW = OW; cq.booster_kind = h.booster_kind;
    cq.n_regions = B;
    // (a plan's first sweep is where the moves are: a stay round of 512 steps that commits a short prefix costs more than one of
    // 256 -- the last wave looks its node up in seven tables; the rebalance of config 3 after a tenth of the nodes left: 59.5
    // against 56.7 ms for that pass.  Eight waves from the second sweep on, when the LDS is there.)
    cq.waves = c->chain_waves ? c->chain_waves : (a.it == 0 ? 4 : 0);
    cq.cls_run = rr.cls_run;
    // a sharded plan: this rank walks the chains of its slice of the regions
    auto slice_lo = [&](int r) { return (int)((int64_t)B * r / G); };
    cq.region_base = sharded ? slice_lo(rank) : 0;
    cq.n_launch = sharded ? slice_lo(rank + 1) - cq.region_base : B;
    cq.reg_lo = rr.reg_lo.as<int32_t>(); cq.reg_hi = rr.reg_hi.as<int32_t>();
    cq.reg_off = c->reg_off.as<int32_t>();
    cq.leaf_node = c->leaf_node.as<int32_t>();
    cq.leaf_cls = rr.leaf_cls.as<int32_t>();
    cq.cls_size = rr.cls_size.as<int32_t>();
    cq.alive = c->alive.as<uint8_t>();
    cq.node_weight = c->node_weight.as<int32_t>();
    cq.node_has_weight = c->node_has_weight.as<uint8_t>();
    cq.cnt = c->cnt.as<int32_t>(); cq.ntn = c->ntn.as<int32_t>();
    cq.crec = c->crec.as<int32_t>(); cq.out = c->out.as<int32_t>();
    cq.flags = scal + 4;
    cq.ev_off = c->ev_off.as<int32_t>(); cq.ev_perm = c->ev_perm.as<int32_t>();
    cq.ev_oi = c->ev_oi.as<int32_t>(); cq.ev_leaf = c->ev_leaf.as<int32_t>(); cq.ev_w = c->ev_w.as<int32_t>();
    if (cfl[6])                                        // nodes of this state that lie in no region
        BLANCE_LAUNCH_NOSYNC(k_chain_orphans, cdiv(P, 256), 256, 0, sm, d, m, h.top_state, a.order,
                             rr.node_region.as<int32_t>(), c->cnt.as<int32_t>());
    cq.cnt_out = cq.cnt;
    const bool gather_out = sharded && (c->comm.allgather_i32 || !c->comm.allreduce_sum_i32);
    if (sharded) {                                     // the loads every rank starts this pass from (orphans included)
        RESERVE(cnt_base, sizeof(int32_t) * (cnt_words + 1));
        RESERVE(xbuf, sizeof(int32_t) * (kXHead + cnt_words + 1));
        HIPTRY(hipMemcpyAsync(c->cnt_base.p, c->cnt.p, sizeof(int32_t) * cnt_words, hipMemcpyDeviceToDevice, sm));
        if (!gather_out) HIPTRY(hipMemsetAsync(c->out.p, 0, sizeof(int32_t) * (size_t)P * OW, sm));
    }
    HIPTRY(hipEventRecord(c->pass_events[2 * n_pass], sm));
    bool stayed = false;
    if (try_stay) {
        if (!group_stands) {
            const int ge = group_by_top(sm, c->scan_sums);
            if (ge) return ge;
        } else if (c->trace) fprintf(stderr, "[blance] chain pass state %d: the steps' grouping by top priority node stands\n", m);
        StayParams sq;
        memset(&sq, 0, sizeof sq);
        sq.N = N; sq.NX = NX; sq.M = M; sq.s = m; sq.k = k; sq.NP = NP; sq.OW = OW; sq.booster_kind = h.booster_kind;
        sq.wg_region = rr.wg_region.as<int32_t>(); sq.wg_chunk = rr.wg_chunk.as<int32_t>();
        sq.reg_lo = rr.reg_lo.as<int32_t>(); sq.reg_hi = rr.reg_hi.as<int32_t>();
        sq.leaf_node = c->leaf_node.as<int32_t>(); sq.leaf_cls = rr.leaf_cls.as<int32_t>(); sq.cls_size = rr.cls_size.as<int32_t>();
        sq.alive = c->alive.as<uint8_t>(); sq.node_weight = c->node_weight.as<int32_t>(); sq.node_has_weight = c->node_has_weight.as<uint8_t>();
        sq.cnt = c->cnt.as<int32_t>(); sq.crec = c->crec.as<int32_t>();
        sq.top_off = c->top_off.as<int32_t>(); sq.top_order = c->top_order.as<int32_t>();
        sq.out = c->out.as<int32_t>(); sq.flag = scal + 4 + kFlagStayMoved;
        if (launch_stay_by_top(sm, sq, rr.n_stay_wgs, rr.max_size)) {
            launches += 1;
            if (defer) {
                stayed = true;                          // (until the caller's readback says otherwise)
                run.pending = 1;
                run.gate = gate_on(1u | (1u << kFlagStayMoved));
                if (c->trace) fprintf(stderr, "[blance] chain pass state %d: stays per top priority node, the verdict is read with the sweep's\n", m);
            } else {
                int32_t sf[8] = {0}, moved = 0;         // [0] a step is not region-local (k_gather_chain); moved: not all stays
                HIPTRY(read_back(c, sf, scal + 4, sizeof sf));
                HIPTRY(read_back(c, &moved, scal + 4 + kFlagStayMoved, sizeof moved));
                HIPTRY(stream_sync(c));
                if (refuted(sf)) {                      // (k_stay_by_top changes no counter)
                    run.redo = true;
                    *launches_io += launches;
                    return 0;
                }
                stayed = !sf[0] && !moved;
                if (c->trace) fprintf(stderr, "[blance] chain pass state %d: stays verified per top priority node: %s\n", m, stayed ? "all of them" : "no");
            }
        }
    }
    if (stayed) {
        HIPTRY(hipEventRecord(c->pass_events[2 * n_pass + 1], sm));
        c->pass_kind.resize(n_pass + 1);
        c->pass_kind[n_pass] = 3;                      // 3: k_stay_by_top verified the pass
        n_pass++;
        c->last_stays[m] = P;
        if (dump_pass(c, a.it, m, P, OW, c->chain_oi.as<int32_t>())) return BLANCE_ERR_DEVICE;
        BLANCE_LAUNCH_NOSYNC(k_scatter, cdiv(P, 256), 256, 0, sm, d, m, OW, c->chain_order.as<int32_t>(), c->out.as<int32_t>(),
                             run.pending ? run.gate : kNoGate);
        launches++;
        *batched_io += P;
        *done = true;
        *launches_io += launches;
        return 0;
    }
    // a fresh plan's first sweep: every step blank -> the lean kernel; it either does this rank's
    // whole slice or changes nothing that is not restored below (a rank-local decision: the full
    // kernel makes the same choices)
    bool lean = false;
    if (NP == 0 && !run.no_lean && !c->any_node_weight && rr.max_size <= 256 && k <= 4 && !cfl[0]) {
        bool walked = false;
        if (c->periodic && !sharded && n_events == 0 && !cfl[6] && !cfl[7]) {
            // k_period.h: regions whose records repeat are walked for two periods; the rest of the periodic stretch is
            // copied, what lies behind it is walked -- all decided on the device, region by region
            int max_len = 0;
            for (int r = 0; r < B; r++) max_len = std::max(max_len, (int)(c->h_reg_off[r + 1] - c->h_reg_off[r]));
            if (max_len >= kPeriodMinRounds * 2) {
                RESERVE(period, sizeof(int32_t) * ((size_t)kPWords * B + 1));
                RESERVE(cnt_p1, sizeof(int32_t) * (cnt_words + 1));
                int32_t* pb = c->period.as<int32_t>();
                const int gx = cdiv(max_len, 256), gl = cdiv(rr.max_size, 64);
                BLANCE_LAUNCH_NOSYNC(k_period_init, cdiv(B, 64), 64, 0, sm, B, cq.reg_off, pb);
                const int gf = cdiv(std::min(max_len, kPeriodCap + 1), 256);
                BLANCE_LAUNCH_NOSYNC(k_period_find, gf * B, 256, 0, sm, B, gf, cq.reg_off, cq.crec, pb);
                const int gv = cdiv((long long)max_len * (kCW / 4), 256);
                BLANCE_LAUNCH_NOSYNC(k_period_verify, gv * B, 256, 0, sm, B, gv, cq.reg_off, cq.crec, pb);
                if (c->periodic_cut > 0) BLANCE_LAUNCH_NOSYNC(k_period_clamp, cdiv(B, 64), 64, 0, sm, B, c->periodic_cut, pb);
                BLANCE_LAUNCH_NOSYNC(k_period_segments, cdiv(B, 64), 64, 0, sm, B, cq.reg_off, pb);
                ChainParams sq = cq;
                sq.seg_beg = pb + (size_t)kPBeg1 * B; sq.seg_end = pb + (size_t)kPEnd1 * B;
                // (the plane automaton for regions of up to 128 leaves, the lane-minimum kernel for wider ones)
                auto walk = [&](const ChainParams& p) {
                    if (c->no_planes || !launch_chain_planes(sm, p, rr.max_size)) launch_chain_blank(sm, p, rr.max_size);
                };
                {
                    walk(sq);
                    walked = true;
                    HIPTRY(hipMemcpyAsync(c->cnt_p1.p, c->cnt.p, sizeof(int32_t) * cnt_words, hipMemcpyDeviceToDevice, sm));
                    sq.seg_beg = pb + (size_t)kPBeg2 * B; sq.seg_end = pb + (size_t)kPEnd2 * B;
                    walk(sq);
                    BLANCE_LAUNCH_NOSYNC(k_period_state_max, gl * B, 64, 0, sm, B, gl, m, N, NX, cq.reg_lo, cq.reg_hi, cq.leaf_node,
                                         cq.alive, c->cnt_p1.as<int32_t>(), cq.cnt, pb);
                    BLANCE_LAUNCH_NOSYNC(k_period_state_check, gl * B, 64, 0, sm, B, gl, m, N, NX, cq.reg_lo, cq.reg_hi, cq.leaf_node,
                                         cq.alive, c->cnt_p1.as<int32_t>(), cq.cnt, pb);
                    BLANCE_LAUNCH_NOSYNC(k_period_verdict, cdiv(B, 64), 64, 0, sm, B, cq.reg_off, cq.flags, pb);
                    BLANCE_LAUNCH_NOSYNC(k_period_replicate, gx * B, 256, 0, sm, B, gx, OW, cq.reg_off, pb, cq.out);
                    BLANCE_LAUNCH(k_period_counts, B, 256, 0, sm, B, m, N, NX, OW, cq.reg_off, cq.reg_lo, cq.reg_hi, cq.leaf_node,
                                  cq.alive, cq.crec, cq.out, pb, cq.cnt);
                    sq.seg_beg = pb + (size_t)kPBeg3 * B; sq.seg_end = pb + (size_t)kPEnd3 * B;
                    walk(sq);
                    launches += 11;
                    c->periodic_passes++;
                    if (c->trace) {
                        std::vector<int32_t> hp((size_t)kPWords * B);
                        HIPTRY(hipMemcpyAsync(hp.data(), pb, sizeof(int32_t) * hp.size(), hipMemcpyDeviceToHost, sm));
                        HIPTRY(stream_sync(c));
                        int64_t copied = 0; int n_ok = 0;
                        for (int r = 0; r < B; r++)
                            if (hp[(size_t)kPOk * B + r]) { n_ok++; copied += hp[(size_t)kPLimit * B + r] - 2 * hp[(size_t)kPT * B + r]; }
                        fprintf(stderr, "[blance] chain pass state %d: periodic records in %d of %d regions (period %d in the first), %lld of %d steps copied\n",
                                m, n_ok, B, hp[(size_t)kPT * B], (long long)copied, P);
                    }
                }
            }
        }
        if (!walked && (c->no_planes || !launch_chain_planes(sm, cq, rr.max_size))) launch_chain_blank(sm, cq, rr.max_size);
        int32_t fl[8] = {0};
        launches++;
        if (defer) {                                    // (taken to have done the pass until the caller's readback says otherwise)
            run.pending = 2;
            run.gate = gate_on(1u | 2u);
        } else {
            HIPTRY(read_back(c, fl, scal + 4, sizeof fl));
            HIPTRY(stream_sync(c));
            if (refuted(fl)) {
                HIPTRY(hipMemcpyAsync(c->cnt.p, c->cnt_save.p, sizeof(int32_t) * cnt_words, hipMemcpyDeviceToDevice, sm));
                run.redo = true;
                *launches_io += launches;
                return 0;
            }
        }
        if (c->trace) fprintf(stderr, "[blance] chain pass state %d: all-blank kernel (%s) %s\n", m, c->no_planes ? "lanes" : "planes",
                              run.pending ? "launched, its flags are read with the sweep's" : !fl[0] && !fl[1] ? "did the pass" : "escaped");
        if (!fl[0] && !fl[1]) {
            lean = true;
        } else if (!fl[0]) {                            // not all blank: the full kernel, from the same state
            HIPTRY(hipMemcpyAsync(c->cnt.p, sharded ? c->cnt_base.p : c->cnt_save.p, sizeof(int32_t) * cnt_words,
                                  hipMemcpyDeviceToDevice, sm));
            if (cfl[6] && !sharded)
                BLANCE_LAUNCH_NOSYNC(k_chain_orphans, cdiv(P, 256), 256, 0, sm, d, m, h.top_state,
                                     a.order, rr.node_region.as<int32_t>(), c->cnt.as<int32_t>());
            HIPTRY(hipMemsetAsync(scal + 4, 0, 16, sm));
        }
    }
    if (!lean) {
        if (NP > 0 && !chain_rows_in_lds(cq, rr.max_size)) NTNTRY();      // (rows in LDS: the matrix in HBM is not touched)
        if (!dispatch_chain(c, cq, rr.max_size)) return fail(BLANCE_ERR_UNSUPPORTED, "region chain shape");
    }
    HIPTRY(hipEventRecord(c->pass_events[2 * n_pass + 1], sm));
    launches += 8;
    c->pass_kind.resize(n_pass + 1);
    c->pass_kind[n_pass] = lean ? 2 : 0;           // 2: the all-blank kernel did the pass
    n_pass++;
    int32_t fl[kXHead] = {0};
    if (sharded) {
        // collective A: [flags | this rank's change of the load vector]
        int32_t* xb = c->xbuf.as<int32_t>();
        HIPTRY(hipMemsetAsync(xb, 0, sizeof(int32_t) * kXHead, sm));
        HIPTRY(hipMemcpyAsync(xb, scal + 4, 32, hipMemcpyDeviceToDevice, sm));
        BLANCE_LAUNCH_NOSYNC(k_vec_sub, cdiv((int64_t)cnt_words, 256), 256, 0, sm, (int)cnt_words, c->cnt.as<int32_t>(),
                             c->cnt_base.as<int32_t>(), xb + kXHead);
        *a_done = true;
        COMMTRY(comm_allreduce(c, xb, (int64_t)(kXHead + cnt_words)));
        HIPTRY(read_back(c, fl, xb, sizeof fl));
        launches++;
    } else if (!lean) {                              // (the all-blank kernel's flags were read above: all clear)
        if (defer) {
            run.pending = 3;
            run.gate = gate_on(1u | 2u);
        } else {
            HIPTRY(read_back(c, fl, scal + 4, 32));
        }
    }
    if (!run.pending && (sharded || !lean)) HIPTRY(stream_sync(c));
    if (fl[8]) return fail(BLANCE_ERR_COMM, "another rank of the sharded plan failed");
    if (!run.pending && !lean && refuted(fl)) {      // (the all-blank kernel's words were checked above)
        c->pass_ntn_ready = false;
        HIPTRY(hipMemcpyAsync(c->cnt.p, c->cnt_save.p, sizeof(int32_t) * cnt_words, hipMemcpyDeviceToDevice, sm));
        n_pass--;
        run.redo = true;
        *launches_io += launches;
        return 0;
    }
    if (c->trace && !run.pending)
        fprintf(stderr, "[blance] chain pass state %d: %d of %d steps committed as verified stays in %d batches\n",
                m, fl[2], P, fl[3]);
    c->last_stays[m] = (!fl[0] && !fl[1]) ? fl[2] : 0;          // (a pending verdict: the caller fills this in)
    if (!fl[0] && !fl[1]) {
        if (sharded) {
            // every rank's chains wrote their own regions' loads and their own steps' outputs
            int32_t* xb = c->xbuf.as<int32_t>();
            BLANCE_LAUNCH_NOSYNC(k_vec_add, cdiv((int64_t)cnt_words, 256), 256, 0, sm, (int)cnt_words, c->cnt_base.as<int32_t>(),
                                 xb + kXHead, c->cnt.as<int32_t>());
            launches++;
            if (gather_out) {
                // collective B: a rank's steps are contiguous in chain order
                const int32_t* ro = c->h_reg_off.data();
                int64_t per = 0;
                for (int r = 0; r < G; r++) {
                    const int64_t len = (int64_t)(ro[slice_lo(r + 1)] - ro[slice_lo(r)]) * OW;
                    if (len > per) per = len;
                }
                if (per > 0) {
                    RESERVE(gath, sizeof(int32_t) * ((size_t)per * G + 1));
                    int32_t* gb = c->gath.as<int32_t>();
                    const int64_t mine = (int64_t)(ro[slice_lo(rank + 1)] - ro[slice_lo(rank)]) * OW;
                    if (mine > 0)
                        HIPTRY(hipMemcpyAsync(gb + (size_t)rank * per, c->out.as<int32_t>() + (size_t)ro[slice_lo(rank)] * OW,
                                              sizeof(int32_t) * (size_t)mine, hipMemcpyDeviceToDevice, sm));
                    COMMTRY(comm_allgather(c, gb, per));
                    for (int r = 0; r < G; r++) {
                        const int64_t len = (int64_t)(ro[slice_lo(r + 1)] - ro[slice_lo(r)]) * OW;
                        if (r != rank && len > 0)
                            HIPTRY(hipMemcpyAsync(c->out.as<int32_t>() + (size_t)ro[slice_lo(r)] * OW, gb + (size_t)r * per,
                                                  sizeof(int32_t) * (size_t)len, hipMemcpyDeviceToDevice, sm));
                    }
                }
            } else {
                COMMTRY(comm_allreduce(c, c->out.as<int32_t>(), (int64_t)P * OW));
            }
        }
        if (dump_pass(c, a.it, m, P, OW, c->chain_oi.as<int32_t>())) return BLANCE_ERR_DEVICE;
        BLANCE_LAUNCH_NOSYNC(k_scatter, cdiv(P, 256), 256, 0, sm, d, m, OW, c->chain_order.as<int32_t>(), c->out.as<int32_t>(),
                             run.pending ? run.gate : kNoGate);
        launches++;
        *batched_io += P;
        *done = true;
    } else {                                        // not region-local after all: redo in order
        c->pass_ntn_ready = false;                  // (chains of big regions keep their rows in global memory: zeroed again on demand)
        HIPTRY(hipMemcpyAsync(c->cnt.p, c->cnt_save.p, sizeof(int32_t) * cnt_words, hipMemcpyDeviceToDevice, sm));
    }
    *launches_io += launches;
    return 0;
}

static int run_chain_pass(blance_ctx* c, const ChainPassArgs& a, int64_t* launches_io, int64_t* batched_io, int* n_pass_io,
                          bool* done, bool* a_done, ChainRun& run) {
    int e = run_chain_pass_once(c, a, launches_io, batched_io, n_pass_io, done, a_done, run);
    if (e || !run.redo) return e;
    c->spec_refuted++;
    if (c->trace) fprintf(stderr, "[blance] chain pass state %d: the assumed classification did not hold, the pass runs again\n", a.m);
    run.allow_spec = false;
    run.defer = false;
    return run_chain_pass_once(c, a, launches_io, batched_io, n_pass_io, done, a_done, run);
}

static int plan_locked(blance_ctx* c, blance_result* res) {
    if (!c->uploaded) return fail(BLANCE_ERR_BAD_ARG, "no problem uploaded");
    HIPTRY(hipSetDevice(c->device));
    const blance_problem& h = c->h;
    const int N = h.n_nodes, NX = h.n_nodes_ext, M = h.n_states, P = h.n_parts, L = c->L;
    const int64_t PM = (int64_t)P * M;
    const int RW = kRecHead + M * (1 + L);       // header + per-state lists
    hipStream_t sm = c->stream;
    int32_t* scal = c->scalars.as<int32_t>();
    int64_t launches = 0, steps = 0, batched = 0;
    int n_pass = 0;

    DevProblem d = dev_problem(c);
    c->last_stays.assign((size_t)(M > 0 ? M : 1), 0);
    c->queue_launches = c->queue_stops = 0;
    c->comm_events_used = 0;
    c->chain_group_state = -1;
    c->group_epoch++;
    c->top_group_state = -1;
    if (c->side_pending) { HIPTRY(hipStreamSynchronize(c->side)); c->side_pending = false; }
    const int64_t syncs0 = c->n_syncs;

    HIPTRY(hipEventRecord(c->ev0, sm));
    HIPTRY(hipMemsetAsync(scal, 0, 256, sm));
    if (c->speculate == 2) {                                         // (tests: every deferred verdict comes back bad, every gate is closed)
        static const int32_t one = 1;
        HIPTRY(hipMemcpyAsync(scal + 4 + kFlagForced, &one, sizeof one, hipMemcpyHostToDevice, sm));
    }
    if (P > 0) {
        BLANCE_LAUNCH_NOSYNC(k_flags_init, cdiv(P, 256), 256, 0, sm, P, c->part_in_prev.as<uint8_t>(), c->part_never_equal.as<uint8_t>(),
                             d.in_prev, d.never_equal);
    }
    int iterations = 0, converged = 0;
    int32_t hs[32] = {0};                                            // the scalar words read back after every sweep
    int m_last = -1;                                                 // the last state a sweep makes a pass for
    for (int m = 0; m < M; m++)
        if (c->state_constraints[m] > 0 && P > 0) m_last = m;
    for (int it = 0; it < h.max_iterations; it++) {                 // plan.go:32
        const bool first = it == 0;
        d.node_removed = first ? c->node_removed.as<uint8_t>() : c->zeros_nx.as<uint8_t>();   // plan.go:53-55
        d.node_added = first ? c->node_added.as<uint8_t>() : c->zeros_nx.as<uint8_t>();
        const int add_nil = first ? h.nodes_to_add_nil : 0;
        const int any_removed = first ? c->any_removed : 0;
        const int NP = first ? h.n_prev : c->np_later;
        c->tops_moved = true;                                       // until this sweep's top-state pass turns out to be one run of stays
        {   // one launch: warn_count / not_match, the chain flags, stateNodeCounts (plan.go:94), the flat passes' row counts
            FillCopyJob fj;
            fj.zero(scal, 2);
            fj.zero(scal + 4, 8);
            fj.zero(c->cnt.p, (int64_t)(M + 1) * (NX + 1));
            if (NP > 0 && c->f_row_count.p) fj.zero(c->f_row_count.p, (int64_t)NX + 1);
            if (run_fill_copy(c, fj)) return BLANCE_ERR_DEVICE;
            c->flags_clean = true;
            c->rowcount_clean = NP > 0 && c->f_row_count.p;
        }
        if (PM > 0) {
            if (first)
                BLANCE_LAUNCH_NOSYNC(k_live_init, cdiv(PM, 256), 256, 0, sm, d, c->a_off.as<int32_t>(),
                                     c->a_nodes.as<int32_t>(), c->a_kind.as<uint8_t>(), c->p_off.as<int32_t>(),
                                     c->p_nodes.as<int32_t>(), c->p_kind.as<uint8_t>());
            else
                BLANCE_LAUNCH_NOSYNC(k_live_refresh, cdiv(PM, 256), 256, 0, sm, d);
            launches++;
        }
        // stateNodeCounts = countStateNodes(prevMap), plan.go:94 (zeroed above)
        if (h.n_loads > 0) {
            BLANCE_LAUNCH_NOSYNC(k_count_loads, cdiv(h.n_loads, 256), 256, 0, sm, h.n_loads, NX, first ? 0 : 1,
                                 c->load_state.as<int32_t>(), c->load_node.as<int32_t>(),
                                 c->load_weight.as<int32_t>(), c->load_first.as<uint8_t>(), c->cnt.as<int32_t>());
            launches++;
        }
        if (PM > 0) {
            BLANCE_LAUNCH_NOSYNC(k_count_prev, cdiv(PM, 256), 256, 0, sm, d, c->cnt.as<int32_t>());
            launches++;
        }
        int passes_this_sweep = 0;
        int known_passes = 0, known_state = -1, known_k = 0;        // run_flat_pass's `opening`
        bool known_broken = false;
        // The sweep's last pass, when it is a chain pass, leaves its verdict on the device (ChainRun::defer) and the words
        // come back with the convergence word: one round trip for both.  A bad verdict brings the loop back for that
        // state alone (retry says how), with everything the pass enqueued behind its gate undone by never having run.
        int m_from = 0;
        ChainRun retry;
        bool retrying = false, counted = false;
        for (;;) {
        ChainRun pend;
        int64_t batched0 = batched, steps0 = steps;
        int n_pass0 = n_pass, passes0 = passes_this_sweep;
        for (int m = m_from; m < M; m++) {                          // plan.go:307-324
            const int k = c->state_constraints[m];
            if (k <= 0 || P == 0) continue;
            if (m == m_last) { batched0 = batched; steps0 = steps; n_pass0 = n_pass; passes0 = passes_this_sweep; }
            const int n_chunks = cdiv(P, kPartChunk);
            const bool sort_cat = first && !(passes_this_sweep == 0 ? c->uniform_first : c->uniform_all);
            passes_this_sweep++;
            if (sort_cat) {
                BLANCE_LAUNCH_NOSYNC(k_category, cdiv(P, 256), 256, 0, sm, d, m, any_removed, add_nil, c->cat.as<uint8_t>());
                BLANCE_LAUNCH(k_part_count, n_chunks, 64, 64, sm, P, (const int32_t*)nullptr, c->cat.as<uint8_t>(),
                              c->part_order.as<int32_t>(), n_chunks, 3, c->chunk_counts.as<int32_t>());
                SCANTRY(3 * n_chunks, c->chunk_counts.as<int32_t>());
                BLANCE_LAUNCH(k_part_scatter, n_chunks, 64, 64, sm, P, (const int32_t*)nullptr, c->cat.as<uint8_t>(),
                              c->part_order.as<int32_t>(), c->part_order.as<int32_t>(), n_chunks, 3, 2,
                              c->chunk_counts.as<int32_t>(), c->order.as<int32_t>(), (int32_t*)nullptr);
            } else {
                // sweeps >= 2 run with nodesToRemove = nodesToAdd = [] (non-nil, plan.go:53-55): no partition's
                // nodes are in either, so every category is "1" (plan.go:542-561) and the pass order is the
                // static order itself
            }
            // (the same holds in sweep 1 when every partition has one category, known at upload: uniform_category)
            const int32_t* order = sort_cat ? c->order.as<int32_t>() : c->part_order.as<int32_t>();
            c->pass_ntn_ready = false;                              // nodeToNodeCounts := fresh (plan.go:266), zeroed when first needed
            const int OW = 1 + k;
            int higher_mask = 0;
            for (int t = 0; t < M; t++)
                if (c->state_priority[t] < c->state_priority[m]) higher_mask |= 1 << t;
            const int r0 = c->rule_off[m], r1 = c->rule_off[m + 1];
            while (c->pass_events.size() < 2 * (size_t)(n_pass + 2)) {
                hipEvent_t ev;
                HIPTRY(hipEventCreate(&ev));
                c->pass_events.push_back(ev);
            }

            // ---- region chains, when the state's single hierarchy rule allows them
            bool done = false;
            if (c->engine != BLANCE_ENGINE_SEQUENTIAL && !h.hierarchy_rules_nil && r1 - r0 == 1 &&
                c->rule_regions[r0].ok && P >= c->chain_min_parts && k <= 4 && !(retrying && retry.skip)) {
                // (same_tops: the top state's pass was one run of stays AND no other state can take a partition's top priority
                // node away in between -- plan.go:146-154 keeps only nodes of STRICTLY higher priority states out of a pass)
                ChainPassArgs ca{d, m, k, NP, OW, RW, higher_mask, r0, it, order,
                                 !first && !c->tops_moved && m != h.top_state && c->top_prio_strict};
                bool a_done = false;
                ChainRun run;
                if (retrying) run = retry;
                run.defer = !retrying && m == m_last;
                known_broken = true;
                const int e = run_chain_pass(c, ca, &launches, &batched, &n_pass, &done, &a_done, run);
                if (!e && run.pending) pend = run;
                if (e) {
                    const bool sharded = (c->comm.n_ranks > 1 || c->shard_one_rank) && c->rule_regions[r0].n_regions >= c->comm.n_ranks;
                    if (sharded && !a_done) comm_poison(c);       // the other ranks are (or will be) in collective A
                    if (sharded) comm_abort(c);
                    return e;
                }
            }
            if (!done) {
            // (what kind of pass this will be, before its records are made: the flat bulk driver's row bound rides on the gather)
            const bool flat_state = h.hierarchy_rules_nil || r1 == r0;
            const bool bulk = c->engine != BLANCE_ENGINE_SEQUENTIAL && flat_state && (k == 1 || (k == 2 && NP == 0)) &&
                              P >= c->chain_min_parts;
            const bool count_rows = bulk && NP > 0 && c->f_row_count.p;
            if (count_rows) {
                if (!c->rowcount_clean) HIPTRY(hipMemsetAsync(c->f_row_count.p, 0, sizeof(int32_t) * ((size_t)NX + 1), sm));
                c->rowcount_clean = false;
            }
            BLANCE_LAUNCH(k_gather, cdiv(P, 256), 256, sizeof(int32_t) * 256 * (RW | 1) + 64, sm, d, m, h.top_state, RW, order,
                                 c->state_stick.as<int32_t>(), c->state_has_stick.as<uint8_t>(), c->rec.as<int32_t>(),
                                 count_rows ? c->f_row_count.as<int32_t>() : (int32_t*)nullptr, NX);
            PassParams q;
            memset(&q, 0, sizeof q);
            q.N = N; q.NX = NX; q.M = M; q.L = L; q.P = P; q.s = m; q.k = k; q.top_state = h.top_state;
            q.NP = NP; q.RW = RW; q.OW = OW;
            q.higher_mask = higher_mask;
            q.hier = !h.hierarchy_rules_nil;
            q.rule_begin = r0; q.rule_end = r1;
            q.booster_kind = h.booster_kind;
            q.n_alive = c->n_alive;
            q.vertex_empty_anchor = NX;
            q.alive = c->alive.as<uint8_t>();
            q.node_weight = c->node_weight.as<int32_t>();
            q.node_has_weight = c->node_has_weight.as<uint8_t>();
            q.node_leaf_pos = c->node_leaf_pos.as<int32_t>();
            q.anchors = c->anchors.as<AnchorSet>();
            q.cnt = c->cnt.as<int32_t>();
            q.ntn = c->ntn.as<int32_t>();
            q.rec = c->rec.as<int32_t>();
            q.out = c->out.as<int32_t>();
            q.warn_part = c->warn_part.as<int32_t>();
            q.warn_state = c->warn_state.as<int32_t>();
            q.warn_count = scal + 0;
            q.err = scal + 2;
            q.spec_count = (long long*)(scal + 12);
            q.beg = 0; q.end = P;
            // a flat pass (no rule for the state) of a small cluster can run on one wave64
            bool flat_chain = c->engine != BLANCE_ENGINE_SEQUENTIAL && flat_state && c->flat_chain_ok && k <= 4 &&
                              P >= c->chain_min_parts;
            c->no_fast_keys = false;
            FlatChainPrep fc;
            fc.possible = flat_chain; fc.d = d; fc.m = m; fc.higher_mask = higher_mask; fc.order = order;
            if (flat_chain && !bulk) {               // (the bulk driver asks for the records when a sub-range needs them)
                const int pe = flat_chain_prepare(c, fc, &launches);
                if (pe) return pe;
                flat_chain = fc.ok;
            }
            HIPTRY(hipEventRecord(c->pass_events[2 * n_pass], sm));
            int e;
            bool nothing_to_apply = false;               // (run_flat_pass: a pass of stays that leaves every list as it is)
            c->pass_kind.resize(n_pass + 1);
            // the flat bulk driver: k = 1, and the first sweep of a fresh plan (NumPartitions == 0) with k = 2
            if (bulk) {
                c->pass_kind[n_pass] = 1;
                // (what the sweep's passes so far have been: known_passes of them fresh runs known in advance, the last one
                // of state known_state with known_k picks a step)
                const bool opening = first && !retrying && !known_broken &&
                                     (known_passes == 0 || (known_passes == 1 && known_k == 1 && NP == 0 && ((higher_mask >> known_state) & 1)));
                bool whole_known = false;
                const bool settled = !first && passes_this_sweep == 1 && !retrying && c->dump_sweep < 0 && c->speculate > 0;
                e = run_flat_pass(c, q, scal, &launches, &batched, fc, opening, &whole_known, settled, &nothing_to_apply, count_rows);
                if (whole_known) { known_passes++; known_state = m; known_k = k; }
                else known_broken = true;
            } else if (flat_chain) {
                known_broken = true;
                c->pass_kind[n_pass] = 0;
                const size_t rows = sizeof(int32_t) * (size_t)(NX + 1) * (NX + 1);
                e = run_flat_chain(c, q, 0, P, NP > 0 && rows <= 100 * 1024, scal, &launches);
                if (e > 0) e = fail(BLANCE_ERR_DEVICE, "flat chain refused a checked pass");
                if (!e) batched += P;
            } else {
                known_broken = true;
                c->pass_kind[n_pass] = 0;
                e = dispatch_pass(c, q);
            }
            if (e) return e;
            HIPTRY(hipEventRecord(c->pass_events[2 * n_pass + 1], sm));
            n_pass++;
            if (dump_pass(c, it, m, P, q.OW, nullptr)) return BLANCE_ERR_DEVICE;
            if (!nothing_to_apply)
                BLANCE_LAUNCH_NOSYNC(k_scatter, cdiv(P, 256), 256, 0, sm, d, m, q.OW, order, c->out.as<int32_t>(), kNoGate);
            }
            launches += 7;
            steps += P;
        }
        if (!counted) iterations++;
        counted = true;
        // convergence (plan.go:36-45) + write-back (plan.go:49-52)
        if (P > 0) {
            BLANCE_LAUNCH(k_converge, cdiv(P, 256), 256, 0, sm, d, scal + 1, pend.pending ? pend.gate : kNoGate);   // (uses a wave ballot)
            launches++;
        }
        // one readback per sweep: the convergence word with the warnings count, the chain flags (a deferred verdict) and --
        // complete with the last sweep -- the statistics words behind them (steps k_pass_seq committed as verified stays,
        // the queue kernel's counters)
        HIPTRY(read_back(c, hs, scal, sizeof hs));
        HIPTRY(stream_sync(c));
        HIPTRY(hipGetLastError());
        if (!pend.pending) break;
        {
            const int32_t* f = hs + 4;
            const bool refuted = (pend.spec && (f[6] || f[7])) || f[kFlagForced];
            const bool bad = pend.pending == 1 ? (f[0] || f[kFlagStayMoved]) : (f[0] || f[1]);
            if (c->trace)
                fprintf(stderr, "[blance] chain pass state %d: deferred verdict (%s): %s; %d verified stays in %d batches\n", m_last,
                        pend.pending == 1 ? "k_stay_by_top" : pend.pending == 2 ? "all-blank kernel" : "chain kernel",
                        refuted ? "assumption refuted" : bad ? "the pass did not stand" : "stands", f[2], f[3]);
            if (!refuted && !bad) {
                if (pend.pending == 3) c->last_stays[m_last] = f[2];
                break;
            }
            // nothing behind the gate ran: the live lists and prevMap are as the pass found them; the counters are not
            c->spec_refuted++;
            retry = ChainRun();
            if (refuted) retry.allow_spec = false;
            else if (pend.pending == 1) retry.no_stay = true;
            else if (pend.pending == 2) retry.no_lean = true;
            else retry.skip = true;                                 // (straight to the pass in order)
            if (pend.pending != 1 || refuted) {
                c->pass_ntn_ready = false;
                HIPTRY(hipMemcpyAsync(c->cnt.p, c->cnt_save.p, sizeof(int32_t) * (size_t)(M + 1) * NX, hipMemcpyDeviceToDevice, sm));
            }
            batched = batched0; steps = steps0; n_pass = n_pass0; passes_this_sweep = passes0;
            retrying = true;
            m_from = m_last;
        }
        }
        if (hs[2]) return fail(BLANCE_ERR_UNSUPPORTED, "hierarchy fold overflowed the device's interval budget");
        c->n_warnings = hs[0];
        if (!hs[1]) { converged = 1; break; }
    }
    HIPTRY(hipEventRecord(c->ev1, sm));
    {
        long long spec = 0, qs[4] = {0, 0, 0, 0};
        memcpy(&spec, hs + 12, sizeof spec);
        memcpy(qs, hs + 18, sizeof qs);
        HIPTRY(hipEventSynchronize(c->ev1));
        comm_sum_up(c);
        c->queue_moved = qs[0]; c->queue_exact = qs[1]; c->queue_rebuilds = qs[2]; c->queue_dense = qs[3];
        if (c->trace || getenv("BLANCE_QUEUE_STATS"))
            fprintf(stderr, "[blance] k_pass_queue: %lld launches, %lld stops, %lld moving steps (%lld with matrix reads, %lld scoring every node), %lld window rebuilds\n",
                    (long long)c->queue_launches, (long long)c->queue_stops, qs[0], qs[1], qs[3], qs[2]);
        batched += spec;
        if (batched > steps) batched = steps;     // the flat chain counts its whole pass already
        if (c->trace) fprintf(stderr, "[blance] sequential passes: %lld verified stays\n", spec);
    }
    float ms = 0.f;
    HIPTRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    double pass_ms = 0.0, flat_ms = 0.0, blank_ms = 0.0, stay_ms = 0.0;
    int n_kernel_pass = 0, n_flat = 0, n_blank = 0, n_stay = 0;
    for (int i = 0; i < n_pass; i++) {
        float pm = 0.f;
        HIPTRY(hipEventElapsedTime(&pm, c->pass_events[2 * i], c->pass_events[2 * i + 1]));
        if (c->pass_kind[i] != 1) { pass_ms += pm; n_kernel_pass++; } else { flat_ms += pm; n_flat++; }
        if (c->pass_kind[i] == 2) { blank_ms += pm; n_blank++; }
        if (c->pass_kind[i] == 3) { stay_ms += pm; n_stay++; }
        if (c->trace)
            fprintf(stderr, "[blance] pass %d (%s): %.3f ms\n", i,
                    c->pass_kind[i] == 1 ? "flat bulk driver" : c->pass_kind[i] == 2 ? "all-blank chain kernel" :
                    c->pass_kind[i] == 3 ? "k_stay_by_top" : "pass kernel", pm);
    }
    c->pass_ms = pass_ms;
    c->pass_launches = n_kernel_pass;
    c->flat_ms = flat_ms;
    c->flat_passes = n_flat;
    c->blank_ms = blank_ms;
    c->blank_launches = n_blank;
    c->stay_ms = stay_ms;
    c->stay_launches = n_stay;
    c->iterations = iterations;
    c->converged = converged;
    c->device_ms = ms;
    c->plan_syncs = c->n_syncs - syncs0;
    c->steps_total = steps;
    c->steps_batched = batched;
    c->kernel_launches = launches;
    c->planned = true;
    if (iterations == 0) c->n_warnings = 0;
    if (res) {
        res->iterations = iterations;
        res->converged = converged;
        res->device_ms = ms;
        res->total_ms = ms;
        res->steps_total = steps;
        res->steps_sequential = steps - batched;
        res->steps_batched = batched;
        res->kernel_launches = launches;
        res->n_warnings = c->n_warnings;
        res->pass_kernel_ms = pass_ms;
        res->pass_kernel_launches = n_kernel_pass;
        res->flat_pass_ms = flat_ms;
        res->flat_passes = n_flat;
        res->blank_pass_ms = blank_ms;
        res->blank_pass_launches = n_blank;
        res->stay_pass_ms = stay_ms;
        res->stay_pass_launches = n_stay;
        res->host_syncs = c->plan_syncs;
    }
    return BLANCE_OK;
}

static int download_locked(blance_ctx* c, blance_result* res) {
    if (!c->planned) return fail(BLANCE_ERR_BAD_ARG, "nothing planned yet");
    if (!res || !res->out_off || !res->out_nodes || !res->out_kind || !res->warn_part || !res->warn_state)
        return fail(BLANCE_ERR_BAD_ARG, "null result buffers");
    HIPTRY(hipSetDevice(c->device));
    const blance_problem& h = c->h;
    const int M = h.n_states, P = h.n_parts, L = c->L;
    const size_t PM = (size_t)P * M;
    if (c->n_warnings > res->warn_capacity) return fail(BLANCE_ERR_CAPACITY, "warn_capacity too small");
    if (c->iterations == 0) {
        // MaxIterationsPerPlan <= 0: planNextMapEx returns (nil, nil) -- nothing to report (plan.go:32-58)
        for (size_t i = 0; i <= PM; i++) res->out_off[i] = 0;
        for (size_t i = 0; i < PM; i++) res->out_kind[i] = BLANCE_LIST_ABSENT;
        res->n_warnings = 0;
        res->iterations = 0;
        res->converged = 0;
        return BLANCE_OK;
    }
    res->out_off[0] = 0;
    Mover down(c, false);
    c->stage.used = 0;
    int e = 0;
    if (PM) {
        // the CSR is made on the device (lengths -> exclusive scan -> gather) and lands in the caller's arrays: directly when
        // they are page-locked, through the staging buffer otherwise
        DevProblem d = dev_problem(c);
        RESERVE(dl_off, sizeof(int32_t) * (PM + 2));
        BLANCE_LAUNCH_NOSYNC(k_result_len, cdiv((int64_t)PM + 1, 256), 256, 0, c->stream, d, c->dl_off.as<int32_t>());
        SCANTRY((int)PM + 1, c->dl_off.as<int32_t>());
        int32_t total = 0;
        HIPTRY(read_back(c, &total, c->dl_off.as<int32_t>() + PM, sizeof total));
        HIPTRY(stream_sync(c));
        if (total > res->out_capacity) return fail(BLANCE_ERR_CAPACITY, "out_capacity too small");
        RESERVE(dl_nodes, sizeof(int32_t) * ((size_t)total + 1));
        BLANCE_LAUNCH_NOSYNC(k_result_gather, cdiv((int64_t)PM, 256), 256, 0, c->stream, d, c->dl_off.as<int32_t>(),
                             c->dl_nodes.as<int32_t>());
        if ((e = down.copy(res->out_off, c->dl_off.p, sizeof(int32_t) * (PM + 1)))) return e;
        if (total && (e = down.copy(res->out_nodes, c->dl_nodes.p, sizeof(int32_t) * (size_t)total))) return e;
        if ((e = down.copy(res->out_kind, c->live_kind.p, PM))) return e;
    }
    if (c->n_warnings) {
        if ((e = down.copy(res->warn_part, c->warn_part.p, sizeof(int32_t) * (size_t)c->n_warnings))) return e;
        if ((e = down.copy(res->warn_state, c->warn_state.p, sizeof(int32_t) * (size_t)c->n_warnings))) return e;
    }
    if ((e = down.finish())) return e;
    HIPTRY(stream_sync(c));
    res->n_warnings = c->n_warnings;
    res->iterations = c->iterations;
    res->converged = c->converged;
    res->device_ms = c->device_ms;
    res->steps_total = c->steps_total;
    res->steps_sequential = c->steps_total - c->steps_batched;
    res->steps_batched = c->steps_batched;
    res->kernel_launches = c->kernel_launches;
    res->pass_kernel_ms = c->pass_ms;
    res->pass_kernel_launches = c->pass_launches;
    res->flat_pass_ms = c->flat_ms;
    res->flat_passes = c->flat_passes;
    res->blank_pass_ms = c->blank_ms;
    res->blank_pass_launches = c->blank_launches;
    res->stay_pass_ms = c->stay_ms;
    res->stay_pass_launches = c->stay_launches;
    res->host_syncs = c->plan_syncs;
    return BLANCE_OK;
}

extern "C" int blance_calc_moves(blance_ctx* c, const blance_moves_problem* pb, blance_moves_result* res) {
    return guarded([&]() -> int {
    if (!c || !pb || !res) return fail(BLANCE_ERR_BAD_ARG, "null argument");
    std::lock_guard<std::mutex> g(c->mu);
    const int P = pb->n_parts, M = pb->n_states;
    if (P < 0 || M < 0 || !pb->beg_off || !pb->end_off || !pb->beg_nodes || !pb->end_nodes || !res->op_off ||
        !res->op_node || !res->op_state || !res->op_kind)
        return fail(BLANCE_ERR_BAD_ARG, "null array pointer or negative size");
    const size_t PS = (size_t)P * (M + 1);
    if (pb->beg_off[0] != 0 || pb->end_off[0] != 0) return fail(BLANCE_ERR_BAD_ARG, "CSR offsets must start at 0");
    for (size_t i = 0; i < PS; i++)
        if (pb->beg_off[i + 1] < pb->beg_off[i] || pb->end_off[i + 1] < pb->end_off[i])
            return fail(BLANCE_ERR_BAD_ARG, "CSR offsets not monotone");
    const int64_t nb = pb->beg_off[PS], ne = pb->end_off[PS], cap = nb + ne;
    if (cap > res->capacity) return fail(BLANCE_ERR_CAPACITY, "moves capacity too small");
    if (cap > (int64_t)INT32_MAX) return fail(BLANCE_ERR_UNSUPPORTED, "more than 2^31 list entries");
    HIPTRY(hipSetDevice(c->device));
    DevBuf &boff = c->mv[0], &bnod = c->mv[1], &eoff = c->mv[2], &enod = c->mv[3], &onode = c->mv[4], &ostate = c->mv[5],
           &okind = c->mv[6], &nmov = c->mv[7], &cnode = c->mv[8], &cstate = c->mv[9], &ckind = c->mv[10];
    auto up = [&](DevBuf& b, const int32_t* src, size_t n) -> int {
        if (b.reserve(sizeof(int32_t) * (n + 1))) return fail(BLANCE_ERR_DEVICE, "hipMalloc failed");
        if (n) HIPTRY(hipMemcpyAsync(b.p, src, sizeof(int32_t) * n, hipMemcpyHostToDevice, c->stream));
        return 0;
    };
    int e;
    if ((e = up(boff, pb->beg_off, PS + 1)) || (e = up(bnod, pb->beg_nodes, (size_t)nb)) ||
        (e = up(eoff, pb->end_off, PS + 1)) || (e = up(enod, pb->end_nodes, (size_t)ne))) {
        (void)stream_sync(c);
        return e;
    }
    if (onode.reserve(sizeof(int32_t) * ((size_t)cap + 1)) || ostate.reserve(sizeof(int32_t) * ((size_t)cap + 1)) ||
        okind.reserve(sizeof(int32_t) * ((size_t)cap + 1)) || nmov.reserve(sizeof(int32_t) * ((size_t)P + 2)) ||
        cnode.reserve(sizeof(int32_t) * ((size_t)cap + 1)) || cstate.reserve(sizeof(int32_t) * ((size_t)cap + 1)) ||
        ckind.reserve(sizeof(int32_t) * ((size_t)cap + 1))) {
        (void)stream_sync(c);
        return fail(BLANCE_ERR_DEVICE, "hipMalloc failed");
    }
    MovesParams q;
    q.P = P; q.M = M; q.favor_min_nodes = pb->favor_min_nodes;
    q.beg_off = boff.as<int32_t>(); q.beg_nodes = bnod.as<int32_t>();
    q.end_off = eoff.as<int32_t>(); q.end_nodes = enod.as<int32_t>();
    q.op_node = onode.as<int32_t>(); q.op_state = ostate.as<int32_t>(); q.op_kind = okind.as<int32_t>();
    q.n_moves = nmov.as<int32_t>();
    HIPTRY(hipEventRecord(c->ev0, c->stream));
    res->op_off[0] = 0;
    if (P > 0) {
        // per-partition slices -> offsets (exclusive scan of the move counts) -> packed on the device
        HIPTRY(hipMemsetAsync(nmov.as<int32_t>() + P, 0, sizeof(int32_t), c->stream));
        BLANCE_LAUNCH_NOSYNC(k_calc_moves, cdiv(P, 256), 256, 0, c->stream, q);
        SCANTRY(P + 1, nmov.as<int32_t>());
        BLANCE_LAUNCH_NOSYNC(k_moves_compact, cdiv(P, 256), 256, 0, c->stream, q, nmov.as<int32_t>(), cnode.as<int32_t>(),
                             cstate.as<int32_t>(), ckind.as<int32_t>());
    }
    HIPTRY(hipEventRecord(c->ev1, c->stream));
    if (P > 0) {
        HIPTRY(hipMemcpyAsync(res->op_off, nmov.p, sizeof(int32_t) * ((size_t)P + 1), hipMemcpyDeviceToHost, c->stream));
        HIPTRY(stream_sync(c));
        const int64_t total = res->op_off[P];
        if (total > 0) {
            HIPTRY(hipMemcpyAsync(res->op_node, cnode.p, sizeof(int32_t) * total, hipMemcpyDeviceToHost, c->stream));
            HIPTRY(hipMemcpyAsync(res->op_state, cstate.p, sizeof(int32_t) * total, hipMemcpyDeviceToHost, c->stream));
            HIPTRY(hipMemcpyAsync(res->op_kind, ckind.p, sizeof(int32_t) * total, hipMemcpyDeviceToHost, c->stream));
        }
    }
    HIPTRY(stream_sync(c));
    float ms = 0.f;
    HIPTRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    res->device_ms = ms;
    return BLANCE_OK;
    });
}

extern "C" int blance_upload(blance_ctx* c, const blance_problem* pb) {
    return guarded([&]() -> int {
    if (!c) return fail(BLANCE_ERR_BAD_ARG, "null ctx");
    std::lock_guard<std::mutex> g(c->mu);
    return upload_locked(c, pb);
    });
}

extern "C" int blance_plan_resident(blance_ctx* c, blance_result* res) {
    return guarded([&]() -> int {
    if (!c) return fail(BLANCE_ERR_BAD_ARG, "null ctx");
    std::lock_guard<std::mutex> g(c->mu);
    rb_discard(c);
    return settle(c, plan_locked(c, res));
    });
}

extern "C" int blance_plan_stats_get(blance_ctx* c, blance_plan_stats* st) {
    return guarded([&]() -> int {
    if (!c || !st) return fail(BLANCE_ERR_BAD_ARG, "null argument");
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->planned) return fail(BLANCE_ERR_BAD_ARG, "nothing planned yet");
    const blance_problem& h = c->h;
    const int N = h.n_nodes, NX = h.n_nodes_ext, M = h.n_states, P = h.n_parts;
    if (st->n_states < M || !st->load_min || !st->load_max || !st->load_sum || !st->load_sumsq || !st->nodes_used ||
        !st->unmet_slots)
        return fail(BLANCE_ERR_BAD_ARG, "stats arrays missing or shorter than n_states");
    HIPTRY(hipSetDevice(c->device));
    hipStream_t sm = c->stream;
    DevBuf load, out, cons, unmet, roff;
    struct Free { DevBuf* b[5]; ~Free() { for (DevBuf* x : b) x->release(); } } fr{{&load, &out, &cons, &unmet, &roff}};
    if (load.reserve(sizeof(int32_t) * ((size_t)M * (NX > 0 ? NX : 1) + 1)) || out.reserve(sizeof(long long) * ((size_t)M * 5 + 1)) ||
        cons.reserve(sizeof(int32_t) * ((size_t)M + 1)) || unmet.reserve(sizeof(long long) * (2 * (size_t)M + 1)) ||
        roff.reserve(sizeof(int32_t) * ((size_t)M + 2)))
        return fail(BLANCE_ERR_DEVICE, "hipMalloc failed");
    std::vector<long long> host((size_t)M * 5 + 1), hun(2 * (size_t)M + 1, 0);
    int n_next = 0;
    if (c->iterations > 0 && M > 0) {
        HIPTRY(hipMemsetAsync(load.p, 0, sizeof(int32_t) * (size_t)M * (NX > 0 ? NX : 1), sm));
        HIPTRY(hipMemsetAsync(unmet.p, 0, sizeof(long long) * 2 * (size_t)M, sm));
        HIPTRY(hipMemcpyAsync(cons.p, c->state_constraints.data(), sizeof(int32_t) * (size_t)M, hipMemcpyHostToDevice, sm));
        DevProblem d = dev_problem(c);
        if ((int64_t)P * M > 0)
            BLANCE_LAUNCH_NOSYNC(k_stats_load, cdiv((int64_t)P * M, 256), 256, 0, sm, d, load.as<int32_t>(), cons.as<int32_t>(),
                                 unmet.as<unsigned long long>());
        if (!h.hierarchy_rules_nil && h.n_rules > 0 && (int64_t)P * M > 0) {      // rule violations (words M .. 2M - 1 of `unmet`)
            HIPTRY(hipMemcpyAsync(roff.p, c->rule_off.data(), sizeof(int32_t) * ((size_t)M + 1), hipMemcpyHostToDevice, sm));
            BLANCE_LAUNCH_NOSYNC(k_stats_rules, cdiv((int64_t)P * M, 256), 256, 0, sm, d, h.top_state, roff.as<int32_t>(),
                                 c->anchors.as<AnchorSet>(), c->node_leaf_pos.as<int32_t>(), unmet.as<unsigned long long>() + M);
        }
        BLANCE_LAUNCH(k_stats_reduce, M, 256, sizeof(long long) * 5 * 256 + 64, sm, N, NX, c->alive.as<uint8_t>(), load.as<int32_t>(), out.as<long long>());
        HIPTRY(hipMemcpyAsync(host.data(), out.p, sizeof(long long) * (size_t)M * 5, hipMemcpyDeviceToHost, sm));
        HIPTRY(hipMemcpyAsync(hun.data(), unmet.p, sizeof(long long) * 2 * (size_t)M, hipMemcpyDeviceToHost, sm));
        HIPTRY(stream_sync(c));
    }
    if (c->iterations > 0) n_next = c->n_alive;
    st->n_nodes_next = n_next;
    for (int m = 0; m < M; m++) {
        const bool any = c->iterations > 0 && n_next > 0;
        st->load_min[m] = any ? host[(size_t)m * 5 + 0] : 0;
        st->load_max[m] = any ? host[(size_t)m * 5 + 1] : 0;
        st->load_sum[m] = any ? host[(size_t)m * 5 + 2] : 0;
        st->load_sumsq[m] = any ? host[(size_t)m * 5 + 3] : 0;
        st->nodes_used[m] = any ? (int32_t)host[(size_t)m * 5 + 4] : 0;
        st->unmet_slots[m] = c->iterations > 0 ? hun[(size_t)m] : 0;
        if (st->rule_violations) st->rule_violations[m] = c->iterations > 0 ? hun[(size_t)M + m] : 0;
    }
    return BLANCE_OK;
    });
}

extern "C" int blance_download(blance_ctx* c, blance_result* res) {
    return guarded([&]() -> int {
    if (!c) return fail(BLANCE_ERR_BAD_ARG, "null ctx");
    std::lock_guard<std::mutex> g(c->mu);
    rb_discard(c);
    return settle(c, download_locked(c, res));
    });
}

extern "C" int blance_plan(blance_ctx* c, const blance_problem* pb, blance_result* res) {
    return guarded([&]() -> int {
    if (!c) return fail(BLANCE_ERR_BAD_ARG, "null ctx");
    if (!res) return fail(BLANCE_ERR_BAD_ARG, "null result");
    std::lock_guard<std::mutex> g(c->mu);
    hipEvent_t t0, t1;
    HIPTRY(hipSetDevice(c->device));
    HIPTRY(hipEventCreate(&t0));
    HIPTRY(hipEventCreate(&t1));
    HIPTRY(hipEventRecord(t0, c->stream));
    int st = upload_locked(c, pb);
    if (!st) st = settle(c, plan_locked(c, res));
    if (!st) st = settle(c, download_locked(c, res));
    if (!st) {
        (void)hipEventRecord(t1, c->stream);
        (void)hipEventSynchronize(t1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, t0, t1);
        res->total_ms = ms;
    }
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    return st;
    });
}

