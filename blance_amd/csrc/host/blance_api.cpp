// Host-side mirror of the reference's planner API over the C ABI: interning of
// strings to the int32 SoA problem (the work the reference does implicitly with Go
// maps keyed by strings), the blance_plan() call, un-interning, and the
// caller-visible mutations of plan.go:49-52.  Counterpart of blance_amd/problem.py;
// tests/test_host_cpp.py drives both on the reference's golden inputs.
#include "blance_api.hpp"

#include <chrono>

#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <exception>
#include <system_error>
#include <atomic>
#include <functional>
#include <mutex>
#include <thread>
#include <set>
#include <unordered_map>

#include "../../../include/blance_hip.h"
#ifdef BLANCE_CALL_ARENA
#include "call_arena.hpp"
#endif

namespace blance {

int MaxIterationsPerPlan = 10;                 // plan.go:21
Booster NodeScoreBooster = Booster::None;      // plan.go:693
bool CustomNodeSorterIsDefault = true;         // plan.go:580

namespace {

#ifdef BLANCE_CALL_ARENA
using ArenaScope = arena::Scope;        // allocations of the enclosing block, this thread: from the call's region
#else
struct ArenaScope { ArenaScope() {} };  // (call_arena.cpp not linked in: plain malloc)
#endif

struct Abi {
    int (*validate)(const blance_problem*) = nullptr;
    int64_t (*result_capacity)(const blance_problem*) = nullptr;
    int (*ctx_create)(const blance_options*, blance_ctx**) = nullptr;
    void (*ctx_destroy)(blance_ctx*) = nullptr;
    int (*plan)(blance_ctx*, const blance_problem*, blance_result*) = nullptr;
    const char* (*last_error)(void) = nullptr;
    int (*calc_moves)(blance_ctx*, const blance_moves_problem*, blance_moves_result*) = nullptr;
    std::shared_ptr<struct Scratch> scratch;                // per Library: the call's flat arrays, kept between calls
};
std::unordered_map<void*, Abi> g_abi;

struct Unsupported {
    std::string why;
};

// name -> dense id: open addressing over the names themselves (FNV-1a), no node allocation per entry --
// a plan of a million partitions interns a few million node references
struct Intern {
    std::vector<std::string> names;
    std::vector<int32_t> slots;                  // id or -1; capacity a power of two, load <= 1/2
    static uint64_t hash(const std::string& s) {
        uint64_t h = 1469598103934665603ull;
        for (unsigned char ch : s) { h ^= ch; h *= 1099511628211ull; }
        return h ^ (h >> 29);
    }
    void grow() {
        const size_t cap = slots.empty() ? 64 : slots.size() * 2;
        slots.assign(cap, -1);
        for (size_t i = 0; i < names.size(); i++) {
            size_t at = hash(names[i]) & (cap - 1);
            while (slots[at] >= 0) at = (at + 1) & (cap - 1);
            slots[at] = (int32_t)i;
        }
    }
    int find(const std::string& s) const {
        if (slots.empty()) return -1;
        const size_t mask = slots.size() - 1;
        for (size_t at = hash(s) & mask;; at = (at + 1) & mask) {
            const int32_t id = slots[at];
            if (id < 0) return -1;
            if (names[(size_t)id] == s) return id;
        }
    }
    bool has(const std::string& s) const { return find(s) >= 0; }
    int add(const std::string& s) {
        int id = find(s);
        if (id >= 0) return id;
        if ((names.size() + 1) * 2 > slots.size()) grow();
        id = (int)names.size();
        names.push_back(s);
        const size_t mask = slots.size() - 1;
        size_t at = hash(s) & mask;
        while (slots[at] >= 0) at = (at + 1) & mask;
        slots[at] = id;
        return id;
    }
    int at(const std::string& s) const { return find(s); }
};

bool atoi_go(const std::string& s, long long* out) {       // strconv.Atoi, plan.go:525
    size_t i = 0;
    if (s.empty()) return false;
    if (s[0] == '+' || s[0] == '-') i = 1;
    if (i >= s.size()) return false;
    for (size_t j = i; j < s.size(); j++) if (s[j] < '0' || s[j] > '9') return false;
    if (s.size() - i <= 18) {                               // cannot overflow: no library call
        long long v = 0;
        for (size_t j = i; j < s.size(); j++) v = v * 10 + (s[j] - '0');
        *out = s[0] == '-' ? -v : v;
        return true;
    }
    errno = 0;
    char* end = nullptr;
    long long v = strtoll(s.c_str(), &end, 10);
    if (errno != 0) return false;
    *out = v;
    return true;
}

std::string pad10(long long v) {
    char buf[40];
    snprintf(buf, sizeof buf, "%10lld", v);
    return buf;
}

// stateNameSorter.Less, plan.go:459-470
bool state_less(const PartitionModel& model, const std::string& a, const std::string& b) {
    auto ia = model.find(a), ib = model.find(b);
    if (ia != model.end() && ib != model.end() && ia->second && ib->second &&
        ia->second->Priority < ib->second->Priority)
        return true;
    return a < b;
}

struct Flat {
    blance_problem pb{};
    std::vector<int32_t> state_priority, state_constraints, state_stickiness;
    std::vector<uint8_t> state_has_stickiness;
    std::vector<uint8_t> node_removed, node_added, node_has_weight;
    std::vector<int32_t> node_weight;
    std::vector<int32_t> part_order, part_weight;
    std::vector<uint8_t> part_has_weight, part_in_prev, never_equal;
    std::vector<int32_t> a_off, a_nodes, p_off, p_nodes;
    std::vector<uint8_t> a_kind, p_kind;
    std::vector<int32_t> load_state, load_node, load_weight;
    std::vector<uint8_t> load_first;
    std::vector<int32_t> rule_off, rule_inc, rule_exc, v_parent, v_lo, v_hi, leaf_pos;
    std::vector<std::string> node_names, state_names, part_names;
    // where partition i's pointer sits in partitionsToAssign / in prevMap (nullptr: not there): the stores of
    // plan.go:49-52 go straight to these instead of walking the trees again (map insertions move no element)
    std::vector<PartitionPtr*> assign_slot, prev_slot;
    // build()'s own temporaries
    std::vector<const Partition*> pparts;
    std::vector<long long> name_num;
    std::vector<uint64_t> sort_key, sort_key2;
    std::vector<int32_t> sort_idx, sort_idx2;
    void reset() {                                          // empty, capacity kept
        pb = blance_problem{};
        for (auto* v : {&state_priority, &state_constraints, &state_stickiness, &node_weight, &part_order, &part_weight, &a_off,
                        &a_nodes, &p_off, &p_nodes, &load_state, &load_node, &load_weight, &rule_off, &rule_inc, &rule_exc,
                        &v_parent, &v_lo, &v_hi, &leaf_pos, &sort_idx, &sort_idx2})
            v->clear();
        for (auto* v : {&state_has_stickiness, &node_removed, &node_added, &node_has_weight, &part_has_weight, &part_in_prev,
                        &never_equal, &a_kind, &p_kind, &load_first})
            v->clear();
        node_names.clear(); state_names.clear(); part_names.clear(); assign_slot.clear(); prev_slot.clear();
        pparts.clear(); name_num.clear(); sort_key.clear(); sort_key2.clear();
    }
};

// What a call needs besides its result: ~200 MB of flat arrays at a million partitions.  Fresh allocations of that size
// are fresh mappings whose pages fault in one by one; a Library keeps them between calls (a second concurrent call on the
// same Library finds them taken and uses its own).
struct Scratch {
    std::mutex mu;
    Flat flat;
    std::vector<int32_t> out_off, out_nodes, warn_part, warn_state;
    std::vector<uint8_t> out_kind;
    std::vector<PartitionPtr> parts, stored;
};

// Worker threads that cannot take the process down: a worker's exception (std::bad_alloc, say) is carried to the calling
// thread and rethrown there after every thread has been joined; if the system gives no more threads the caller does the
// rest of the work itself.  fn(i) for i in [0, n): the caller runs i = 0.
struct Joiner {
    std::vector<std::thread>& th;
    ~Joiner() { for (auto& x : th) if (x.joinable()) x.join(); }
};
template <class Fn>
void run_workers(int n, Fn fn) {
    std::vector<std::thread> th;
    std::exception_ptr first;
    std::mutex mu;
    auto guarded = [&](int i) {
        try { fn(i); } catch (...) { std::lock_guard<std::mutex> g(mu); if (!first) first = std::current_exception(); }
    };
    int started = 1;
    {
        Joiner join{th};
        try {
            for (int i = 1; i < n; i++) { th.emplace_back(guarded, i); started++; }
        } catch (const std::system_error&) {}        // no more threads
        guarded(0);
        for (int i = started; i < n; i++) guarded(i);
    }
    if (first) std::rethrow_exception(first);
}

template <class T>
const T* ptr(const std::vector<T>& v) {
    static T dummy[1] = {};
    return v.empty() ? dummy : v.data();
}

// The entries of a std::map in key order, as pointers.  A million-entry tree is a million dependent cache misses when it
// is walked with an iterator; its subtrees are independent, so a few threads walk one each.  That needs the tree's root, which
// the interface does not give: libstdc++'s iterator exposes its node (`_M_node`), the header's parent is the root (stl_tree.h).
// With any other library this is the plain walk.
template <class Map>
void collect_in_order(const Map& m, std::vector<const typename Map::value_type*>& out, int n_threads, size_t min_size = 4096) {
    using V = typename Map::value_type;
    out.clear();
    out.reserve(m.size());
#if defined(__GLIBCXX__) && !defined(_GLIBCXX_DEBUG)
    using Base = std::_Rb_tree_node_base;
    using Node = std::_Rb_tree_node<V>;
    if (n_threads > 1 && m.size() >= min_size) {
        const Base* root = m.end()._M_node->_M_parent;
        struct Task { const Base* n; bool subtree; std::vector<const V*> got; };
        std::vector<Task> tasks;                            // in key order: subtrees of depth-5 nodes and the nodes above them
        struct Frame { const Base* n; int d; bool left_done; };
        std::vector<Frame> st{{root, 0, false}};
        while (!st.empty()) {
            Frame fr = st.back();
            st.pop_back();
            if (!fr.n) continue;
            if (fr.d == 5) { tasks.push_back(Task{fr.n, true, {}}); continue; }
            if (!fr.left_done) {
                st.push_back(Frame{fr.n, fr.d, true});
                st.push_back(Frame{fr.n->_M_left, fr.d + 1, false});
            } else {
                tasks.push_back(Task{fr.n, false, {}});
                st.push_back(Frame{fr.n->_M_right, fr.d + 1, false});
            }
        }
        std::atomic<size_t> next{0};
        auto work = [&]() {
            std::vector<const Base*> stack;
            for (;;) {
                const size_t ti = next.fetch_add(1);
                if (ti >= tasks.size()) break;
                Task& t = tasks[ti];
                if (!t.subtree) continue;
                const Base* n = t.n;
                stack.clear();
                while (n || !stack.empty()) {               // in-order, children fetched ahead
                    while (n) {
                        if (n->_M_left) __builtin_prefetch(n->_M_left);
                        if (n->_M_right) __builtin_prefetch(n->_M_right);
                        stack.push_back(n);
                        n = n->_M_left;
                    }
                    n = stack.back();
                    stack.pop_back();
                    t.got.push_back(static_cast<const Node*>(n)->_M_valptr());
                    n = n->_M_right;
                }
            }
        };
        run_workers(n_threads, [&](int) { work(); });
        for (auto& t : tasks) {
            if (t.subtree) out.insert(out.end(), t.got.begin(), t.got.end());
            else out.push_back(static_cast<const Node*>(t.n)->_M_valptr());
        }
        if (out.size() == m.size()) return;
        out.clear();                                        // (not the layout this was written for: the plain walk)
    }
#endif
    (void)n_threads;
    for (auto& kv : m) out.push_back(&kv);
}

void build(Flat& f, const PartitionMap* prevMapIn, const PartitionMap& assign, const std::vector<std::string>& nodesAll,
           const StringList& nodesToRemove, const StringList& nodesToAdd, const PartitionModel& model,
           const PlanNextMapOptions& o) {
    static const PartitionMap empty_map;
    const bool tr_ = getenv("BLANCE_HOST_TRACE") != nullptr;
    auto t0_ = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!tr_) return;
        auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[host] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0_).count());
        t0_ = t1;
    };
    if (!prevMapIn && !assign.empty())
        throw Unsupported{"nil prevMap with partitions to assign (reference panics, plan.go:50)"};
    const PartitionMap& prevMap = prevMapIn ? *prevMapIn : empty_map;

    // ---- states: sortStateNames, plan.go:437-474
    std::vector<std::string> states;
    for (auto& kv : model) {
        if (!kv.second) throw Unsupported{"nil *PartitionModelState"};
        states.push_back(kv.first);
    }
    for (auto& a : states)
        for (auto& b : states)
            if (a != b && state_less(model, a, b) && state_less(model, b, a))
                throw Unsupported{"state priority order contradicts state name order"};
    for (size_t i = 1; i < states.size(); i++)
        for (size_t j = i; j > 0 && state_less(model, states[j], states[j - 1]); j--) std::swap(states[j], states[j - 1]);
    const int M = (int)states.size();
    std::unordered_map<std::string, int> sid;
    for (int i = 0; i < M; i++) sid[states[i]] = i;
    bool any_pass = false;
    for (auto& s : states) {
        const auto& ms = model.at(s);
        int k = ms->Constraints;
        if (o.ModelStateConstraints) {                      // plan.go:314-319
            auto it = o.ModelStateConstraints->find(s);
            if (it != o.ModelStateConstraints->end()) k = it->second;
        }
        f.state_priority.push_back(ms->Priority);
        f.state_constraints.push_back(k);
        if (k > 0) any_pass = true;
    }
    int top_state = 0;
    if (M) {
        int mn = *std::min_element(f.state_priority.begin(), f.state_priority.end());
        int n_top = 0;
        for (int i = M - 1; i >= 0; i--) if (f.state_priority[i] == mn) { top_state = i; n_top++; }
        if (n_top > 1 && any_pass) throw Unsupported{"several states share the top priority"};
    }

    // ---- nodes
    Intern nodes;
    for (auto& n : nodesAll) {
        if (nodes.has(n)) throw Unsupported{"duplicate node name in nodesAll"};
        nodes.add(n);
    }
    const int N = (int)nodes.names.size();

    // ---- partitions.  The maps are ordered by name (std::map, like the sorted walk the shim makes of Go's maps):
    // prevMap and PartitionWeights are joined with partitionsToAssign by merging the three name-ordered sequences, not by
    // a lookup per name.  At a million partitions this is the call's largest host cost, and all of it is waiting for
    // memory (tree nodes and Partition objects lie wherever the caller allocated them), so it is done by a few threads:
    // (1) the maps' entries are collected in name order, subtree by subtree (collect_in_order); (2) the partitions are cut
    // into contiguous ranges; every range is flattened by one thread into its own buffers, which (3) are appended in order.
    // One range (small inputs) is the same code without threads.
    const bool weights_nil = !o.PartitionWeights.has_value();
    int n_threads = 1;
    const bool threads_forced = getenv("BLANCE_HOST_THREADS") != nullptr;          // (tests: the threaded paths on small maps)
    if (const char* e = getenv("BLANCE_HOST_THREADS")) n_threads = std::max(1, atoi(e));
    else if (assign.size() >= 65536) n_threads = (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
    using PmItem = const PartitionMap::value_type*;
    using WItem = const std::map<std::string, int>::value_type*;
    std::vector<PmItem> items, pitems;
    std::vector<WItem> witems;
    const size_t min_tree = threads_forced ? 8 : 4096;
    collect_in_order(assign, items, n_threads, min_tree);
    collect_in_order(prevMap, pitems, n_threads, min_tree);
    if (!weights_nil) collect_in_order(*o.PartitionWeights, witems, n_threads, min_tree);
    const int P = (int)items.size();
    mark("maps in name order");
    std::vector<const Partition*>& pparts = f.pparts;
    std::vector<long long>& name_num = f.name_num;          // the name as a non-negative number (plan.go:525), or -1
    std::vector<std::string>& pnames = f.part_names;
    pnames.resize((size_t)P);
    pparts.resize((size_t)P);
    name_num.resize((size_t)P);
    f.assign_slot.resize((size_t)P);
    f.prev_slot.assign((size_t)P, nullptr);
    f.part_weight.assign(P, 1);
    f.part_has_weight.assign(P, 0);
    f.part_in_prev.assign(P, 0);
    f.never_equal.assign(P, 0);
    f.a_kind.resize((size_t)P * M);
    f.p_kind.resize((size_t)P * M);
    f.a_off.resize((size_t)P * M + 1);
    f.p_off.resize((size_t)P * M + 1);
    f.a_off[0] = 0;
    f.p_off[0] = 0;
    const bool any_removed = nodesToRemove && !nodesToRemove->empty();
    auto labs64 = [](long long v) { return v < 0 ? -v : v; };
    static const std::map<std::string, StringList> no_states;
    // the model's states inside a NodesByState map: both are ordered by name -> one pass, no lookups
    std::vector<int> by_name(M);                            // state ids in name order
    for (int m = 0; m < M; m++) by_name[m] = m;
    std::sort(by_name.begin(), by_name.end(), [&](int a, int b) { return states[a] < states[b]; });

    struct Range {
        int lo = 0, hi = 0;
        std::vector<int32_t> a_nodes, p_nodes;              // the range's part of the CSR arrays (offsets relative to it)
        std::vector<int32_t> load_state, load_node, load_weight;
        std::vector<uint8_t> load_first;
        long long abs_load = 0;
        bool unknown_node = false;                          // a node name outside nodesAll: ids are given out in ONE order
        bool failed = false;
        Unsupported err;
    };
    const int n_ranges = std::max(1, std::min(n_threads, P > 0 ? P : 1));
    std::vector<Range> ranges((size_t)n_ranges);
    for (int r = 0; r < n_ranges; r++) {
        ranges[(size_t)r].lo = (int)((long long)P * r / n_ranges);
        ranges[(size_t)r].hi = (int)((long long)P * (r + 1) / n_ranges);
    }
    // single: the one thread may give new ids to names it meets (the reference's order of first appearance)
    auto flatten = [&](Range& R, bool single, bool first_range, bool last_range) {
        auto node_id = [&](const std::string& x) -> int32_t {
            if (single) return nodes.add(x);
            const int id = nodes.find(x);
            if (id < 0) { R.unknown_node = true; return 0; }
            return id;
        };
        std::vector<const StringList*> lists(M);
        std::vector<char> present(M);
        auto split = [&](const std::map<std::string, StringList>& nbs, bool* foreign) {
            for (int m = 0; m < M; m++) { lists[m] = nullptr; present[m] = 0; }
            int bi = 0;
            for (auto& kv : nbs) {
                while (bi < M && states[by_name[bi]] < kv.first) bi++;
                if (bi < M && states[by_name[bi]] == kv.first) { lists[by_name[bi]] = &kv.second; present[by_name[bi]] = 1; }
                else *foreign = true;
            }
        };
        auto prev_only = [&](const std::string& name, const PartitionPtr& pp) {     // partitions only in prevMap
            if (!pp) throw Unsupported{"nil *Partition in prevMap"};
            long long w = 1;
            if (!weights_nil) {
                auto it = o.PartitionWeights->find(name);
                if (it != o.PartitionWeights->end()) w = it->second;
            }
            if (pp->NodesByState)
                for (auto& sl : *pp->NodesByState) {
                    if (!sl.second) continue;
                    auto is = sid.find(sl.first);
                    for (auto& x : *sl.second) {
                        R.load_state.push_back(is == sid.end() ? M : is->second); R.load_node.push_back(node_id(x));
                        R.load_weight.push_back((int32_t)w); R.load_first.push_back(0);
                        R.abs_load += labs64(w);
                    }
                }
        };
        if (R.lo >= R.hi && !(first_range && last_range)) return;
        // where the range starts in the other two sequences; the first range also takes what lies in front of it
        size_t ip = 0, pe = pitems.size(), iw = 0, we = witems.size();
        if (!first_range && R.lo < P) {
            const std::string& k = items[(size_t)R.lo]->first;
            ip = (size_t)(std::lower_bound(pitems.begin(), pitems.end(), k, [](PmItem a, const std::string& b) { return a->first < b; }) - pitems.begin());
            iw = (size_t)(std::lower_bound(witems.begin(), witems.end(), k, [](WItem a, const std::string& b) { return a->first < b; }) - witems.begin());
        }
        if (!last_range && R.hi < P) {
            const std::string& k = items[(size_t)R.hi]->first;
            pe = (size_t)(std::lower_bound(pitems.begin(), pitems.end(), k, [](PmItem a, const std::string& b) { return a->first < b; }) - pitems.begin());
        }
        R.a_nodes.reserve((size_t)(R.hi - R.lo) * 4);
        R.p_nodes.reserve((size_t)(R.hi - R.lo) * 4);
        for (int i = R.lo; i < R.hi; i++) {
            if (i + 16 < R.hi) {
                __builtin_prefetch(items[(size_t)i + 16]);
                if (items[(size_t)i + 8]->second) __builtin_prefetch(items[(size_t)i + 8]->second.get());   // the objects lie wherever the caller allocated them
            }
            const auto& kv = *items[(size_t)i];
            if (!kv.second) throw Unsupported{"nil *Partition in partitionsToAssign"};
            const std::string& name = pnames[(size_t)i] = kv.first;
            long long v = 0;
            name_num[(size_t)i] = atoi_go(name, &v) && v >= 0 ? v : -1;
            const Partition& pa = *kv.second;
            pparts[(size_t)i] = &pa;
            f.assign_slot[(size_t)i] = const_cast<PartitionPtr*>(&kv.second);
            while (iw < we && witems[iw]->first < name) ++iw;
            if (iw < we && witems[iw]->first == name) { f.part_weight[i] = witems[iw]->second; f.part_has_weight[i] = 1; }
            if (pa.Name != name) throw Unsupported{"partition key != Partition.Name"};
            const auto& nbs = pa.NodesByState ? *pa.NodesByState : no_states;
            bool foreign = false;
            split(nbs, &foreign);
            if (foreign) throw Unsupported{"partition carries a state that is not in the model"};
            for (int m = 0; m < M; m++) {
                uint8_t kind = BLANCE_LIST_ABSENT;
                if (present[m]) {
                    const StringList& l = *lists[m];
                    if (l) {
                        const size_t first = R.a_nodes.size();
                        for (auto& x : *l) R.a_nodes.push_back(node_id(x));
                        for (size_t a = first; a < R.a_nodes.size(); a++)
                            for (size_t b = first; b < a; b++)
                                if (R.a_nodes[a] == R.a_nodes[b] && !R.unknown_node) throw Unsupported{"duplicate node inside a state list"};
                    }
                    kind = l ? BLANCE_LIST_SET : BLANCE_LIST_NIL;
                }
                f.a_kind[(size_t)i * M + m] = kind;
                f.a_off[(size_t)i * M + m + 1] = (int32_t)R.a_nodes.size();
            }
            while (ip < pe && pitems[ip]->first < name) { prev_only(pitems[ip]->first, pitems[ip]->second); ++ip; }
            const long long w = f.part_weight[i];
            if (ip >= pe || pitems[ip]->first != name) {
                if (any_removed && any_pass)
                    throw Unsupported{"nodesToRemove non-empty but a partition is missing from prevMap (plan.go:545)"};
                for (int m = 0; m < M; m++) { f.p_kind[(size_t)i * M + m] = BLANCE_LIST_ABSENT; f.p_off[(size_t)i * M + m + 1] = (int32_t)R.p_nodes.size(); }
                continue;
            }
            const PartitionPtr& pptr = pitems[ip]->second;
            if (!pptr) throw Unsupported{"nil *Partition in prevMap"};
            f.part_in_prev[i] = 1;
            f.prev_slot[(size_t)i] = const_cast<PartitionPtr*>(&pptr);
            const Partition& pp = *pptr;
            ++ip;
            if (!pp.NodesByState || pp.Name != name) f.never_equal[i] = 1;
            const auto& pn = pp.NodesByState ? *pp.NodesByState : no_states;
            foreign = false;
            split(pn, &foreign);
            for (int m = 0; m < M; m++) {
                uint8_t kind = BLANCE_LIST_ABSENT;
                if (present[m]) {
                    const StringList& l = *lists[m];
                    kind = l ? BLANCE_LIST_SET : BLANCE_LIST_NIL;
                    if (l)
                        for (auto& x : *l) { R.p_nodes.push_back(node_id(x)); R.abs_load += labs64(w); }
                }
                f.p_kind[(size_t)i * M + m] = kind;
                f.p_off[(size_t)i * M + m + 1] = (int32_t)R.p_nodes.size();
            }
            if (foreign)
                for (auto& kv2 : pn) {
                    if (sid.count(kv2.first)) continue;
                    f.never_equal[i] = 1;
                    if (kv2.second)
                        for (auto& x : *kv2.second) {
                            R.load_state.push_back(M); R.load_node.push_back(node_id(x));
                            R.load_weight.push_back((int32_t)w); R.load_first.push_back(1);
                            R.abs_load += labs64(w);
                        }
                }
        }
        for (; ip < pe; ++ip) prev_only(pitems[ip]->first, pitems[ip]->second);
    };
    auto run_range = [&](int r, bool single) {
        Range& R = ranges[(size_t)r];
        try {
            flatten(R, single, r == 0, r == n_ranges - 1);
        } catch (const Unsupported& u) {
            R.failed = true;
            R.err = u;
        }
    };
    bool again = false;
    if (n_ranges > 1) {
        run_workers(n_ranges, [&](int r) { run_range(r, false); });
        for (auto& R : ranges) if (R.unknown_node) again = true;
    }
    if (n_ranges == 1 || again) {
        // names outside nodesAll get their ids in the order one walk meets them: all of it again as one range
        ranges.assign(1, Range());
        ranges[0].lo = 0;
        ranges[0].hi = P;
        const int keep = n_ranges;
        (void)keep;
        Range& R = ranges[0];
        try {
            flatten(R, true, true, true);
        } catch (const Unsupported& u) {
            R.failed = true;
            R.err = u;
        }
    }
    for (auto& R : ranges) if (R.failed) throw R.err;        // (the first range in name order that met one)
    long long abs_load = 0;
    {
        size_t na = 0, np = 0, nl = 0;
        for (auto& R : ranges) { na += R.a_nodes.size(); np += R.p_nodes.size(); nl += R.load_state.size(); }
        f.a_nodes.resize(na); f.p_nodes.resize(np);
        f.load_state.reserve(nl); f.load_node.reserve(nl); f.load_weight.reserve(nl); f.load_first.reserve(nl);
        size_t ba = 0, bp = 0;
        for (auto& R : ranges) {
            if (!R.a_nodes.empty()) memcpy(f.a_nodes.data() + ba, R.a_nodes.data(), R.a_nodes.size() * sizeof(int32_t));
            if (!R.p_nodes.empty()) memcpy(f.p_nodes.data() + bp, R.p_nodes.data(), R.p_nodes.size() * sizeof(int32_t));
            if (ba || bp)
                for (size_t j = (size_t)R.lo * M + 1; j <= (size_t)R.hi * M; j++) { f.a_off[j] += (int32_t)ba; f.p_off[j] += (int32_t)bp; }
            ba += R.a_nodes.size(); bp += R.p_nodes.size();
            f.load_state.insert(f.load_state.end(), R.load_state.begin(), R.load_state.end());
            f.load_node.insert(f.load_node.end(), R.load_node.begin(), R.load_node.end());
            f.load_weight.insert(f.load_weight.end(), R.load_weight.begin(), R.load_weight.end());
            f.load_first.insert(f.load_first.end(), R.load_first.begin(), R.load_first.end());
            abs_load += R.abs_load;
        }
    }
    mark("partitions -> ids");
    {
        long long sumw = 0, maxw = 0, ksum = 0;
        for (int i = 0; i < P; i++) { sumw += labs64(f.part_weight[i]); maxw = std::max<long long>(maxw, labs64(f.part_weight[i])); }
        for (int k : f.state_constraints) ksum += std::max(k, 0);
        abs_load += sumw * std::max<long long>(1, ksum) * 2;
        if (abs_load > 2147483647LL || maxw > 2147483647LL)
            throw Unsupported{"partition weights overflow the device's int32 load tables"};
    }

    // ---- node attributes (may still add names that are not in nodesAll)
    if (nodesToRemove) for (auto& x : *nodesToRemove) nodes.add(x);
    if (nodesToAdd) for (auto& x : *nodesToAdd) nodes.add(x);
    if (o.NodeWeights) for (auto& kv : *o.NodeWeights) nodes.add(kv.first);
    const int NX = (int)nodes.names.size();
    f.node_removed.assign(NX, 0); f.node_added.assign(NX, 0); f.node_weight.assign(NX, 0); f.node_has_weight.assign(NX, 0);
    if (nodesToRemove) for (auto& x : *nodesToRemove) f.node_removed[nodes.at(x)] = 1;
    if (nodesToAdd) for (auto& x : *nodesToAdd) f.node_added[nodes.at(x)] = 1;
    if (o.NodeWeights)
        for (auto& kv : *o.NodeWeights) { f.node_weight[nodes.at(kv.first)] = kv.second; f.node_has_weight[nodes.at(kv.first)] = 1; }

    f.state_stickiness.assign(M, 0); f.state_has_stickiness.assign(M, 0);
    if (o.StateStickiness)
        for (auto& kv : *o.StateStickiness) {
            auto it = sid.find(kv.first);
            if (it != sid.end()) { f.state_stickiness[it->second] = kv.second; f.state_has_stickiness[it->second] = 1; }
        }

    // ---- hierarchy rules and the tree as DFS leaf intervals (plan.go:703-774)
    const bool rules_nil = !o.HierarchyRules_.has_value();
    int VX = 0, v_empty = 0;
    f.rule_off.push_back(0);
    if (!rules_nil) {
        for (int m = 0; m < M; m++) {
            auto it = o.HierarchyRules_->find(states[m]);
            if (it != o.HierarchyRules_->end())
                for (auto& r : it->second) {
                    if (!r) throw Unsupported{"nil *HierarchyRule"};
                    f.rule_inc.push_back(std::max(r->IncludeLevel, 0));   // findAncestor: `for level > 0`
                    f.rule_exc.push_back(std::max(r->ExcludeLevel, 0));
                }
            f.rule_off.push_back((int32_t)f.rule_inc.size());
            int k = f.state_constraints[m];
            if (k > 0 && (f.rule_off[m + 1] - f.rule_off[m]) * k > 64) throw Unsupported{"more than 64 hierarchy picks"};
        }
        if (nodes.has("")) throw Unsupported{"\"\" used as a node name"};
        Intern v = nodes;
        static const std::map<std::string, std::string> no_hier;
        const auto& hier = o.NodeHierarchy ? *o.NodeHierarchy : no_hier;
        for (auto& kv : hier) { v.add(kv.first); v.add(kv.second); }
        v_empty = v.add("");
        VX = (int)v.names.size();
        f.v_parent.assign(VX, v_empty);                     // findAncestor: missing -> ""
        std::vector<std::vector<int>> children(VX);
        std::vector<char> has_parent(VX, 0);
        for (auto& kv : hier) {                             // std::map: children in name order (plan.go:705-715)
            int c = v.at(kv.first), p = v.at(kv.second);
            f.v_parent[c] = p;
            children[p].push_back(c);
            has_parent[c] = 1;
        }
        f.v_lo.assign(VX, -1); f.v_hi.assign(VX, -1);
        int pos = 0;
        for (int root = 0; root < VX; root++) {
            if (has_parent[root]) continue;
            std::vector<std::pair<int, size_t>> stack{{root, 0}};
            while (!stack.empty()) {
                auto [u, ci] = stack.back();
                stack.pop_back();
                if (ci == 0) {
                    f.v_lo[u] = pos;
                    if (children[u].empty()) { pos++; f.v_hi[u] = pos; continue; }   // a childless vertex is its own leaf
                }
                if (ci < children[u].size()) {
                    stack.push_back({u, ci + 1});
                    stack.push_back({children[u][ci], 0});
                } else {
                    f.v_hi[u] = pos;
                }
            }
        }
        for (int u = 0; u < VX; u++) if (f.v_lo[u] < 0 || f.v_hi[u] < 0) throw Unsupported{"cycle in NodeHierarchy"};
        f.leaf_pos.assign(NX, -1);
        for (int n = 0; n < NX; n++) if (children[n].empty()) f.leaf_pos[n] = f.v_lo[n];
    } else {
        f.rule_off.assign(M + 1, 0);
        f.leaf_pos.assign(NX, -1);
    }

    mark("nodes, hierarchy");
    // ---- static part of partitionSorter's key (plan.go:519-540): ("%10d" of 999999999 - weight, "%10d" of the name if
    // it is a non-negative number else the name, Name), compared as strings.  "%10d" renderings of values in
    // [0, 9999999999] compare like the values (digits right-aligned behind spaces), so such keys are sorted as
    // integers; a problem with any other key takes the string comparison of the reference literally.
    {
        bool simple = true;
        long long n_max = 0, w_min = 0, w_max = 0;
        for (int i = 0; i < P && simple; i++) {
            const long long v = name_num[(size_t)i];
            const long long wk = 999999999LL - (long long)f.part_weight[i];
            if (v < 0 || v > 9999999999LL || wk < 0 || wk > 9999999999LL) simple = false;
            n_max = std::max(n_max, v);
            if (i == 0) w_min = w_max = wk;
            w_min = std::min(w_min, wk); w_max = std::max(w_max, wk);
        }
        if (simple) {
            // stable LSD radix sorts of (key, index) pairs, name key first, weight key second: partitions with equal keys
            // ("7" and "007") keep the order they were taken from the map in -- ascending Name, the reference's last tie-break
            std::vector<uint64_t>&key = f.sort_key, &key2 = f.sort_key2;
            std::vector<int32_t>&idx = f.sort_idx, &idx2 = f.sort_idx2;
            key.resize((size_t)P); key2.resize((size_t)P); idx.resize((size_t)P); idx2.resize((size_t)P);
            for (int i = 0; i < P; i++) idx[(size_t)i] = i;
            // sorts idx by key[] (key[j] belongs to idx[j]); each thread counts and scatters its own slice of the input,
            // slices in order, so the sort stays stable
            const int T = (P >= 65536 || threads_forced) ? std::max(1, std::min(n_threads, P)) : 1;
            std::vector<std::vector<size_t>> cnt((size_t)T, std::vector<size_t>(2048));
            auto parallel = [&](const std::function<void(int)>& fn) { run_workers(T, fn); };
            auto radix = [&](long long mx) {
                for (int shift = 0; shift < 63 && (mx >> shift) != 0; shift += 11) {
                    parallel([&](int t) {
                        auto& c = cnt[(size_t)t];
                        std::fill(c.begin(), c.end(), 0);
                        const size_t lo = (size_t)P * t / T, hi = (size_t)P * (t + 1) / T;
                        for (size_t i = lo; i < hi; i++) c[(key[i] >> shift) & 2047]++;
                    });
                    size_t at = 0;
                    for (int d = 0; d < 2048; d++)
                        for (int t = 0; t < T; t++) { const size_t n = cnt[(size_t)t][(size_t)d]; cnt[(size_t)t][(size_t)d] = at; at += n; }
                    parallel([&](int t) {
                        auto& c = cnt[(size_t)t];
                        const size_t lo = (size_t)P * t / T, hi = (size_t)P * (t + 1) / T;
                        for (size_t i = lo; i < hi; i++) {
                            const size_t to = c[(key[i] >> shift) & 2047]++;
                            key2[to] = key[i]; idx2[to] = idx[i];
                        }
                    });
                    key.swap(key2); idx.swap(idx2);
                }
            };
            if (n_max > 0) {
                for (int i = 0; i < P; i++) key[(size_t)i] = (uint64_t)name_num[(size_t)i];
                radix(n_max);
            }
            if (w_max != w_min) {                           // one weight for all (no weights): nothing to order
                for (int i = 0; i < P; i++) key[(size_t)i] = (uint64_t)(999999999LL - (long long)f.part_weight[idx[(size_t)i]] - w_min);
                radix(w_max - w_min);
            }
            f.part_order.assign(idx.begin(), idx.end());
        } else {
            struct Key { std::string w, n; const std::string* name; int i; };
            std::vector<Key> keys;
            for (int i = 0; i < P; i++) {
                long long v = 0;
                std::string nkey = (atoi_go(pnames[i], &v) && v >= 0) ? pad10(v) : pnames[i];
                keys.push_back({pad10(999999999LL - (long long)f.part_weight[i]), nkey, &pnames[i], i});
            }
            std::sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) {
                if (a.w != b.w) return a.w < b.w;
                if (a.n != b.n) return a.n < b.n;
                if (*a.name != *b.name) return *a.name < *b.name;
                return a.i < b.i;
            });
            for (auto& k : keys) f.part_order.push_back(k.i);
        }
    }

    mark("static order");
    blance_problem& pb = f.pb;
    pb.n_nodes = N; pb.n_nodes_ext = NX; pb.n_states = M; pb.n_parts = P; pb.n_prev = (int32_t)prevMap.size();
    pb.n_loads = (int32_t)f.load_state.size(); pb.n_rules = (int32_t)f.rule_inc.size(); pb.n_vertices = VX;
    pb.max_iterations = MaxIterationsPerPlan;
    pb.partition_weights_nil = weights_nil; pb.nodes_to_add_nil = !nodesToAdd.has_value();
    pb.hierarchy_rules_nil = rules_nil; pb.booster_kind = (int32_t)NodeScoreBooster; pb.top_state = top_state;
    pb.vertex_empty = v_empty;
    pb.state_priority = ptr(f.state_priority); pb.state_constraints = ptr(f.state_constraints);
    pb.state_stickiness = ptr(f.state_stickiness); pb.state_has_stickiness = ptr(f.state_has_stickiness);
    pb.node_removed = ptr(f.node_removed); pb.node_added = ptr(f.node_added);
    pb.node_weight = ptr(f.node_weight); pb.node_has_weight = ptr(f.node_has_weight);
    pb.part_order = ptr(f.part_order); pb.part_weight = ptr(f.part_weight); pb.part_has_weight = ptr(f.part_has_weight);
    pb.part_in_prev = ptr(f.part_in_prev); pb.part_prev_never_equal = ptr(f.never_equal);
    if (P * M == 0) { f.a_off.assign(1, 0); f.p_off.assign(1, 0); }
    pb.assign_off = ptr(f.a_off); pb.assign_nodes = ptr(f.a_nodes); pb.assign_kind = ptr(f.a_kind);
    pb.prev_off = ptr(f.p_off); pb.prev_nodes = ptr(f.p_nodes); pb.prev_kind = ptr(f.p_kind);
    pb.load_state = ptr(f.load_state); pb.load_node = ptr(f.load_node); pb.load_weight = ptr(f.load_weight);
    pb.load_first_sweep_only = ptr(f.load_first);
    pb.rule_off = ptr(f.rule_off); pb.rule_inc = ptr(f.rule_inc); pb.rule_exc = ptr(f.rule_exc);
    pb.vertex_parent = ptr(f.v_parent); pb.vertex_leaf_lo = ptr(f.v_lo); pb.vertex_leaf_hi = ptr(f.v_hi);
    pb.node_leaf_pos = ptr(f.leaf_pos);
    f.node_names = nodes.names; f.state_names = states;
}

}  // namespace

bool Library::open(const std::string& path, std::string* err) {
    handle = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!handle) { if (err) *err = dlerror(); return false; }
    // the structs of include/blance_hip.h are written by the library: one built against another ABI would write past (or
    // short of) this program's blance_result
    int (*abi_version)(void) = (int (*)(void))dlsym(handle, "blance_abi_version");
    if (!abi_version || abi_version() != BLANCE_ABI_VERSION) {
        if (err) *err = "library ABI " + (abi_version ? std::to_string(abi_version()) : std::string("unknown")) +
                        ", this program was built for ABI " + std::to_string(BLANCE_ABI_VERSION);
        dlclose(handle);
        handle = nullptr;
        return false;
    }
    Abi a;
    a.validate = (int (*)(const blance_problem*))dlsym(handle, "blance_validate");
    a.result_capacity = (int64_t (*)(const blance_problem*))dlsym(handle, "blance_result_capacity");
    a.ctx_create = (int (*)(const blance_options*, blance_ctx**))dlsym(handle, "blance_ctx_create");
    a.ctx_destroy = (void (*)(blance_ctx*))dlsym(handle, "blance_ctx_destroy");
    a.plan = (int (*)(blance_ctx*, const blance_problem*, blance_result*))dlsym(handle, "blance_plan");
    a.last_error = (const char* (*)(void))dlsym(handle, "blance_last_error");
    a.calc_moves = (int (*)(blance_ctx*, const blance_moves_problem*, blance_moves_result*))dlsym(handle, "blance_calc_moves");
    if (!a.validate || !a.result_capacity || !a.ctx_create || !a.ctx_destroy || !a.plan || !a.last_error) {
        if (err) *err = "library does not export the blance C ABI";
        return false;
    }
    blance_options opt{};
    const char* eager = getenv("BLANCE_HOST_EAGER_BULK");     // tests: bulk engines for passes of any size
    if (eager && *eager) opt.reserved[1] = 1;
    blance_ctx* c = nullptr;
    if (a.ctx_create(&opt, &c) != BLANCE_OK) {               // no device: fail loudly, there is no CPU path
        if (err) *err = std::string("blance_ctx_create: ") + a.last_error();
        return false;
    }
    ctx = c;
    a.scratch = std::make_shared<Scratch>();
    g_abi[handle] = a;
    return true;
}

void Library::trim() {
    auto it = g_abi.find(handle);
    if (it == g_abi.end() || !it->second.scratch) return;
    std::unique_lock<std::mutex> held(it->second.scratch->mu, std::try_to_lock);
    if (!held.owns_lock()) return;                           // a call is using them: nothing to let go of now
    Scratch& sc = *it->second.scratch;
    sc.flat = Flat();
    std::vector<int32_t>().swap(sc.out_off); std::vector<int32_t>().swap(sc.out_nodes);
    std::vector<int32_t>().swap(sc.warn_part); std::vector<int32_t>().swap(sc.warn_state);
    std::vector<uint8_t>().swap(sc.out_kind);
    std::vector<PartitionPtr>().swap(sc.parts);
    std::vector<PartitionPtr>().swap(sc.stored);
#ifdef BLANCE_CALL_ARENA
    arena::trim();
#endif
}

void Library::close() {
    if (handle) {
        auto it = g_abi.find(handle);
        if (it != g_abi.end()) {
            if (ctx) it->second.ctx_destroy((blance_ctx*)ctx);
            g_abi.erase(it);
        }
        dlclose(handle);
    }
    handle = nullptr;
    ctx = nullptr;
}

PlanOutcome PlanNextMapEx(Library& lib, PartitionMap* prevMap, PartitionMap& partitionsToAssign,
                          const std::vector<std::string>& nodesAll, const StringList& nodesToRemove,
                          const StringList& nodesToAdd, const PartitionModel& model,
                          const PlanNextMapOptions& options) {
    PlanOutcome out;
    // callbacks of the caller's language cannot run on the device: the shim leaves such calls to plan.go
    if (!CustomNodeSorterIsDefault) { out.why = "CustomNodeSorter is not the default sorter (plan.go:580)"; return out; }
    if (NodeScoreBooster == Booster::Other) { out.why = "NodeScoreBooster is an arbitrary callback (plan.go:693)"; return out; }
    auto abi_it = g_abi.find(lib.handle);
    if (abi_it == g_abi.end()) { out.why = "library not open"; return out; }
    const Abi& abi = abi_it->second;
    Scratch own;
    std::unique_lock<std::mutex> scratch_lock(abi.scratch->mu, std::try_to_lock);
    Scratch& sc = scratch_lock.owns_lock() ? *abi.scratch : own;
    Flat& f = sc.flat;
    f.reset();
    const auto t_begin = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
    };
    try {
        build(f, prevMap, partitionsToAssign, nodesAll, nodesToRemove, nodesToAdd, model, options);
    } catch (const Unsupported& u) {
        out.why = u.why;
        return out;
    }
    // (no blance_validate / blance_result_capacity here: blance_plan makes the same checks itself -- the O(P) ones on the
    // device, ABI 5 -- and refuses with the same text; two more walks over two million offsets were 4 ms of the call)
    const int M = f.pb.n_states, P = f.pb.n_parts;
    const size_t PM = (size_t)P * M;
    int64_t cap = (int64_t)f.a_nodes.size();                 // sum of max(k, len) <= sum of len + P * sum of k
    for (int m = 0; m < M; m++) cap += (int64_t)P * (f.state_constraints[m] > 0 ? f.state_constraints[m] : 0);
    std::vector<int32_t>&out_off = sc.out_off, &out_nodes = sc.out_nodes, &warn_part = sc.warn_part, &warn_state = sc.warn_state;
    std::vector<uint8_t>& out_kind = sc.out_kind;
    out_off.resize(PM + 1); out_nodes.resize((size_t)cap + 1); warn_part.resize(PM + 1); warn_state.resize(PM + 1);
    out_kind.resize(PM + 1);
    blance_result res{};
    res.out_off = out_off.data(); res.out_nodes = out_nodes.data(); res.out_kind = out_kind.data();
    res.out_capacity = cap;
    res.warn_part = warn_part.data(); res.warn_state = warn_state.data(); res.warn_capacity = (int64_t)PM;
    out.intern_ms = ms_since(t_begin);
    const auto t_plan = std::chrono::steady_clock::now();
    if (abi.plan((blance_ctx*)lib.ctx, &f.pb, &res) != BLANCE_OK) { out.why = abi.last_error(); return out; }
    out.plan_ms = ms_since(t_plan);
    out.device_ms = res.device_ms;
    const auto t_un = std::chrono::steady_clock::now();
    out.handled = true;
    out.iterations = res.iterations;
    out.converged = res.converged != 0;
    if (res.iterations == 0) { out.nil_result = true; return out; }
    // ids -> strings.  What this costs is allocations -- per partition its Partition, two map nodes, two list buffers,
    // a node of the result map (node names fit the small-string buffer): ~7 M at config 3, twice that when the call
    // converged (below).  With glibc's malloc that was 230-450 ms and did not scale over threads (the allocator and the
    // page faults behind it serialise).  Now (call_arena.hpp, when it is linked in) they are pointer bumps in per-thread
    // chunks that are reused, warm, from call to call: the objects are built in blocks of kBlock partitions by a few
    // threads, then the result map, the stores into prevMap and the stores into partitionsToAssign run side by side.
    const bool need_store = (res.iterations > 1 || !res.converged) && prevMap;
    // What the reference leaves in the input maps are the objects of the LAST SWEEP THAT DID NOT CONVERGE: when the call
    // converged in sweep n > 1 those are sweep n - 1's -- equal in content to the returned ones (that is what converged
    // means, plan.go:36-45) but not the same objects (plan.go:334-343 makes fresh ones every sweep), so a caller that
    // edits nextMap[p] afterwards does not edit prevMap[p].  At the iteration cap the returned objects ARE the stored
    // ones (plan.go:49-52 ran on them).  One second object per partition, shared by both input maps as in the reference.
    const bool need_clones = need_store && res.converged;
    std::vector<PartitionPtr>& parts = sc.parts;
    std::vector<PartitionPtr>& stored = sc.stored;
    parts.clear();
    parts.resize((size_t)P);
    stored.clear();
    if (need_clones) stored.resize((size_t)P);
    int kBlock = 8192, n_threads = 1;
    if (const char* e = getenv("BLANCE_HOST_BLOCK")) kBlock = std::max(1, atoi(e));        // (tests: many blocks on small maps)
    if (const char* e = getenv("BLANCE_HOST_THREADS")) n_threads = std::max(1, atoi(e));
    else if (P >= 65536) n_threads = (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
    const int n_blocks = (P + kBlock - 1) / kBlock;
    n_threads = std::max(1, std::min(n_threads, n_blocks));
    std::atomic<int> next_block{0};
    std::atomic<bool> failed{false};
    auto fill = [&](Partition& part, int p) {
        part.Name = f.part_names[(size_t)p];
        part.NodesByState.emplace();
        auto& nbs = *part.NodesByState;
        for (int m = 0; m < M; m++) {
            size_t i = (size_t)p * M + m;
            if (out_kind[i] == BLANCE_LIST_ABSENT) continue;
            if (out_kind[i] == BLANCE_LIST_NIL) { nbs.emplace_hint(nbs.end(), f.state_names[m], std::nullopt); continue; }
            std::vector<std::string> lst;
            lst.reserve((size_t)(out_off[i + 1] - out_off[i]));
            for (int32_t j = out_off[i]; j < out_off[i + 1]; j++) lst.push_back(f.node_names[out_nodes[j]]);
            nbs[f.state_names[m]] = std::move(lst);
        }
    };
    auto build_blocks = [&]() {
        try {
            ArenaScope scope;
            for (;;) {
                const int b = next_block.fetch_add(1);
                if (b >= n_blocks || failed.load()) break;
                const int lo = b * kBlock, hi = std::min(P, lo + kBlock);
                // the Partition objects of a block live in ONE array (a make_shared each would be a control block each):
                // every PartitionPtr aliases its block and keeps it alive
                for (int copy = 0; copy < (need_clones ? 2 : 1); copy++) {
                    auto block = std::make_shared<std::vector<Partition>>((size_t)(hi - lo));
                    std::vector<PartitionPtr>& dst = copy ? stored : parts;
                    for (int p = lo; p < hi; p++) {
                        fill((*block)[(size_t)(p - lo)], p);
                        dst[(size_t)p] = PartitionPtr(block, &(*block)[(size_t)(p - lo)]);
                    }
                }
            }
        } catch (...) {
            failed.store(true);
        }
    };
    run_workers(n_threads, [&](int) { build_blocks(); });
    if (failed.load()) { out.handled = false; out.why = "out of memory while building the result"; parts.clear(); stored.clear(); return out; }
    out.unintern_parts_ms = ms_since(t_un);
    out.threads = n_threads;
    for (int64_t i = 0; i < res.n_warnings; i++) {           // plan.go:231-234
        const std::string& name = f.part_names[warn_part[i]];
        char buf[64];
        snprintf(buf, sizeof buf, "%d", f.state_constraints[warn_state[i]]);
        out.warnings[name].push_back(std::string("could not meet constraints: ") + buf + ", stateName: " +
                                     f.state_names[warn_state[i]] + ", partitionName: " + name);
    }
    // plan.go:49-52: every non-converged sweep stores its partitions into both input maps;
    // the last such store has the final map's content
    const std::vector<PartitionPtr>& to_store = need_clones ? stored : parts;
    // the slots recorded while the maps were read: independent stores (the tree is not walked a second time);
    // names the map does not hold yet are inserted in name order, each right after its predecessor
    // (the slot stores are not spread over threads: what they let go of is the caller's malloc'ed objects, and glibc's free
    // does not scale -- measured: 60 ms with 1, 2, 3 or 4 threads)
    auto store = [&](PartitionMap& dst, const std::vector<PartitionPtr*>& slot) {
        ArenaScope scope;                                  // (nodes this adds to the caller's map: deleted like any other)
        size_t missing = 0;
        for (int p = 0; p < P; p++) {                      // the map nodes and the objects they let go of lie wherever the
            if (p + 32 < P && slot[(size_t)p + 32]) __builtin_prefetch(slot[(size_t)p + 32]);             // caller put them
            if (p + 8 < P && slot[(size_t)p + 8]) __builtin_prefetch(slot[(size_t)p + 8]->get());
            if (slot[(size_t)p]) *slot[(size_t)p] = to_store[(size_t)p];
            else missing++;
        }
        if (!missing) return;
        if (missing < (size_t)P / 16) {                    // few: a lookup each costs less than the walk
            for (int p = 0; p < P; p++)
                if (!slot[(size_t)p]) dst.emplace(f.part_names[(size_t)p], to_store[(size_t)p]);
            return;
        }
        auto id = dst.begin();
        for (int p = 0; p < P; p++) {
            if (slot[(size_t)p]) continue;
            const std::string& name = f.part_names[(size_t)p];
            while (id != dst.end() && id->first < name) ++id;
            dst.emplace_hint(id, name, to_store[(size_t)p]);       // lands right before id, which stays the successor
        }
    };
    {
        const auto t_st = std::chrono::steady_clock::now();
        double prev_ms = 0.0, assign_ms = 0.0;
        auto store_prev = [&]() { store(*prevMap, f.prev_slot); prev_ms = ms_since(t_st); };
        auto store_assign = [&]() { store(partitionsToAssign, f.assign_slot); assign_ms = ms_since(t_st); };
        const bool two = need_store && &partitionsToAssign != prevMap;
        auto result_map = [&]() {   // filled in name order -- the order the partitions were taken from partitionsToAssign --
            ArenaScope scope;                              // so every insertion lands at the end
            for (int p = 0; p < P; p++) out.nextMap.emplace_hint(out.nextMap.end(), f.part_names[(size_t)p], parts[(size_t)p]);
            out.unintern_map_ms = ms_since(t_st);
        };
        if (need_store && n_threads > 1) {
            // the result map on this thread, the stores of plan.go:49-52 beside it (run_workers: an exception of a worker
            // is rethrown here after the join, thread shortage means this thread does the stores as well)
            run_workers(two ? 3 : 2, [&](int i) { if (i == 0) result_map(); else if (i == 1) store_prev(); else store_assign(); });
        } else {
            result_map();
            if (need_store) {
                store_prev();
                if (two) store_assign();
            }
        }
        if (getenv("BLANCE_HOST_TRACE")) fprintf(stderr, "[host] store prev %.1f ms, assign %.1f ms, result map %.1f ms (side by side)\n", prev_ms, assign_ms, out.unintern_map_ms);
        out.store_ms = std::max(prev_ms, assign_ms);        // (side by side with the result map when threads are used)
    }
    stored.clear();
    parts.clear();                                          // (the scratch keeps the capacity, not the partitions)
    out.unintern_ms = ms_since(t_un);
    return out;
}

MovesOutcome CalcPartitionMovesBatch(Library& lib, const std::vector<std::string>& states,
                                     const std::map<std::string, NodesByState>& begMap,
                                     const std::map<std::string, NodesByState>& endMap, bool favorMinNodes) {
    MovesOutcome out;
    auto it = g_abi.find(lib.handle);
    if (it == g_abi.end() || !it->second.calc_moves) { out.why = "library not open or without blance_calc_moves"; return out; }
    // partitions: the end map's, then those only in the begin map (what the orchestrator walks)
    std::vector<std::string> names;
    for (auto& kv : endMap) names.push_back(kv.first);
    for (auto& kv : begMap) if (!endMap.count(kv.first)) names.push_back(kv.first);
    std::unordered_map<std::string, int32_t> ids;
    std::vector<std::string> node_names;
    auto nid = [&](const std::string& x) {
        auto f = ids.find(x);
        if (f != ids.end()) return f->second;
        int32_t i = (int32_t)node_names.size();
        ids.emplace(x, i);
        node_names.push_back(x);
        return i;
    };
    const int M = (int)states.size();
    auto csr = [&](const std::map<std::string, NodesByState>& m, std::vector<int32_t>& off, std::vector<int32_t>& nodes) {
        off.assign(1, 0);
        static const NodesByState kNone;
        for (auto& name : names) {
            auto f = m.find(name);
            const NodesByState& nbs = f == m.end() ? kNone : f->second;
            for (auto& s : states) {
                auto l = nbs.find(s);
                if (l != nbs.end() && l->second) for (auto& x : *l->second) nodes.push_back(nid(x));
                off.push_back((int32_t)nodes.size());
            }
            for (auto& kv : nbs) {               // keys outside `states` only feed flattenNodesByState (moves.go:47-48)
                bool known = false;
                for (auto& s : states) if (s == kv.first) known = true;
                if (!known && kv.second) for (auto& x : *kv.second) nodes.push_back(nid(x));
            }
            off.push_back((int32_t)nodes.size());
        }
    };
    std::vector<int32_t> boff, bnod, eoff, enod;
    csr(begMap, boff, bnod);
    csr(endMap, eoff, enod);
    bnod.push_back(0); enod.push_back(0);        // never hand out null pointers
    const int64_t cap = (int64_t)boff.back() + eoff.back() + 1;
    std::vector<int32_t> op_off(names.size() + 1), op_node((size_t)cap), op_state((size_t)cap), op_kind((size_t)cap);
    blance_moves_problem pb{};
    pb.n_parts = (int32_t)names.size(); pb.n_states = M; pb.favor_min_nodes = favorMinNodes ? 1 : 0;
    pb.beg_off = boff.data(); pb.beg_nodes = bnod.data(); pb.end_off = eoff.data(); pb.end_nodes = enod.data();
    blance_moves_result res{};
    res.op_off = op_off.data(); res.op_node = op_node.data(); res.op_state = op_state.data(); res.op_kind = op_kind.data();
    res.capacity = cap;
    if (it->second.calc_moves((blance_ctx*)lib.ctx, &pb, &res) != BLANCE_OK) {
        out.why = std::string("blance_calc_moves: ") + it->second.last_error();
        return out;
    }
    static const char* kOps[] = {"add", "del", "promote", "demote"};     // BLANCE_OP_*
    for (size_t i = 0; i < names.size(); i++) {
        auto& v = out.moves[names[i]];
        for (int32_t j = op_off[i]; j < op_off[i + 1]; j++)
            v.push_back({node_names[(size_t)op_node[(size_t)j]], op_state[(size_t)j] < 0 ? "" : states[(size_t)op_state[(size_t)j]],
                         kOps[op_kind[(size_t)j]]});
    }
    out.ok = true;
    return out;
}

std::map<std::string, bool> StringsToMap(const std::vector<std::string>& strs) {          // misc.go:13-22
    std::map<std::string, bool> m;
    for (auto& s : strs) m[s] = true;
    return m;
}

std::vector<std::string> StringsRemoveStrings(const std::vector<std::string>& a, const std::vector<std::string>& remove) {
    auto rm = StringsToMap(remove);                                                       // misc.go:27-36
    std::vector<std::string> rv;
    for (auto& s : a) if (!rm.count(s)) rv.push_back(s);
    return rv;
}

std::vector<std::string> StringsIntersectStrings(const std::vector<std::string>& a, const std::vector<std::string>& b) {
    auto bm = StringsToMap(b);                                                            // misc.go:40-51
    std::set<std::string> seen;
    std::vector<std::string> rv;
    for (auto& s : a) if (bm.count(s) && seen.insert(s).second) rv.push_back(s);
    return rv;
}

}  // namespace blance
