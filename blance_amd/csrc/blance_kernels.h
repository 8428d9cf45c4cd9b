// Device-side data layout and kernel parameter blocks of the MI355X planner.
// Everything here is plain int32/uint8 SoA in HBM; see DESIGN.md "Data layout".
#pragma once
#include <stdint.h>

namespace blance {

constexpr int kListAbsent = 0, kListNil = 1, kListSet = 2;   // include/blance_hip.h BLANCE_LIST_*

constexpr int kMaxK = 8;         // largest supported Constraints per state
constexpr int kMaxAnchors = 9;   // hierarchy anchors per fold: 1 + (#rules of a state) * k <= 9
constexpr int kMaxStates = 16;
constexpr int kRecHead = 4;      // step-record header words: partition, weight, stickiness (fp64)
constexpr int kLpTab = 512;      // chain kernel LDS tables: c / NP for c < kLpTab ...
constexpr int kFfTab = 2048;     // ... and (0.001 * t) / NP for t < kFfTab

// One rule's include/exclude leaf intervals for one anchor (plan.go:723-734):
// leaves(findAncestor(a, IncludeLevel)) = [alo, ahi), leaves(findAncestor(a, ExcludeLevel)) = [blo, bhi).
struct AnchorSet { int32_t alo, ahi, blo, bhi; };

// Parameters of one state pass (assignStateToPartitions, plan.go:253-303).
struct PassParams {
    int32_t N, NX, M, L, P;
    int32_t beg, end;       // steps [beg, end) of the pass order run by this launch
    int32_t s;              // state id of this pass
    int32_t k;              // constraints
    int32_t top_state;
    int32_t NP;             // len(prevMap) of this sweep (plan.go:161)
    int32_t RW, OW;         // words per step record / per step output
    int32_t higher_mask;    // bit t: state t has Priority < Priority of s (plan.go:148)
    int32_t hier;           // HierarchyRules != nil
    int32_t rule_begin, rule_end;
    int32_t booster_kind;
    int32_t n_alive;
    int32_t vertex_empty_anchor;   // anchor index of "" (= NX)
    const uint8_t* alive;          // [NX] node is in nodesNext (plan.go:77)
    const int32_t* node_weight;    // [NX]
    const uint8_t* node_has_weight;
    const int32_t* node_leaf_pos;  // [NX]
    const AnchorSet* anchors;      // [n_rules][NX + 1]
    int32_t* cnt;                  // [(M + 1) * NX] stateNodeCounts
    int32_t* ntn;                  // [(NX + 1) * N] nodeToNodeCounts, zeroed per pass
    const int32_t* rec;            // [P * RW] step records in pass order
    int32_t* out;                  // [P * OW] chosen nodes per step
    int32_t* warn_part;
    int32_t* warn_state;
    int32_t* warn_count;
    int32_t* err;                  // device error word (interval overflow ...)
    int32_t spec;                  // try verified-stay speculation (flat passes; LDS mirrors sized by the host)
    long long* spec_count;         // steps committed as verified stays (statistics, may be null)
    // k_pass_queue (k_pass_queue.h) only:
    uint32_t* ntn_bits;            // [(NX + 1) * BW] one bit per nodeToNodeCounts entry: "is not zero" (BW = words per row, 16-byte rows)
    int32_t* stop;                 // out: [0] first step of [beg, end) NOT done (end: all done), [1] why (kQStop*)
    long long* qstats;             // out, may be null: [0] moving steps, [1] with matrix reads, [2] window rebuilds, [3] dense steps
};

// Parameters of a state pass run as independent per-region chains (DESIGN.md
// "Region chains"): one wave64 per hierarchy region walks that region's steps.
//
// Compact chain step record (k_gather_chain), kCW words, everything already
// translated to leaf indices local to the region:
//   1 weight   2,3 stickiness (fp64)   4 top priority node's leaf
//   0 index of the step in pass order
//   5 counts: own (inside the region) | higher << 8 | lower << 16 | (state list present) << 24
//     | (holds nodes of this state outside the region) << 25
//   6 exclude class of the top priority node
//   7..10  leaves of the nodes the partition holds in this state (-1 padded)
//   11..14 leaves of its higher priority nodes inside the region (never candidates)
//   15..18 leaves of its lower priority nodes inside the region, 19..22 their states
//   23 leaves covered by the top priority node's exclude class
constexpr int kChainOwn = 4, kChainHigh = 4, kChainLow = 4;
constexpr int kChainStage = 256;                 // steps whose records / outputs one staging round of k_pass_chain holds in LDS
constexpr int kChainMaxLeaves = 512;             // leaves of one region (one wave64, 8 leaves per lane)
constexpr int kCW = 24;
constexpr int kCOwn = 7, kCHigh = 11, kCLow = 15, kCLowState = 19;

// A launch the host enqueues BEFORE it has read the words that decide whether it should run (one round trip less per
// decision): the kernel looks at the words itself.  closed = some word flags[b], b a set bit of mask, is non-zero.
struct Gate {
    const int32_t* flags;
    uint32_t mask;
};
constexpr Gate kNoGate = {nullptr, 0u};

struct ChainParams {
    int32_t N, NX, M, L;
    int32_t s, k, NP, OW;
    int32_t booster_kind;
    int32_t n_regions, ntn_in_lds;
    int32_t region_base, n_launch; // regions [region_base, region_base + n_launch) run in this launch (a rank of a
                                   // sharded plan runs a slice of the regions)
    int32_t cls_run;               // k_pass_chain_planes: S if every exclude class of the rule is an aligned run of S = 2^e <= 64
                                   // leaves that all carry nodes (racks of equal size), else 0
    int32_t flat;                  // the whole cluster is ONE region and every node its own exclude class:
                                   // a state pass without hierarchy rules (leaf index = node id); a chain that
                                   // cannot go on stops, keeps what it did and reports the step in flags[4]
    const int32_t* reg_lo;         // [n_regions] leaf interval of the region
    const int32_t* reg_hi;
    const int32_t* reg_off;        // [n_regions + 1] step range of the region in chain order
    const int32_t* seg_beg;        // k_pass_chain_planes / k_pass_chain_blank, optional (k_period.h): [n_regions] the launch walks steps
    const int32_t* seg_end;        // [seg_beg[r], seg_end[r]) of region r's chain instead of all of it
    const int32_t* leaf_node;      // [n_leaves] node id at a leaf position, -1 if none
    const int32_t* leaf_cls;       // [n_leaves] exclude class of the leaf's node inside its region, -1 if none
    const int32_t* cls_size;       // [n_leaves] leaves covered by class c of the region at reg_lo + c
    const uint8_t* alive;
    const int32_t* node_weight;
    const uint8_t* node_has_weight;
    int32_t* cnt;
    int32_t* cnt_out;              // k_pass_chain_blank: where the new counters go (committed by the host)
    int32_t* ntn;
    // A node the partition holds in this state OUTSIDE its region leaves it (plan.go:290-293)
    // whatever the step decides: a static event for the chain that owns the node, applied
    // before that chain's first step that comes later in pass order.
    const int32_t* ev_off;         // [n_regions + 1] events of the region: ev_perm[ev_off[r] .. ev_off[r + 1])
    const int32_t* ev_perm;        // event ids grouped by region, pass order inside a region
    const int32_t* ev_oi;          // [E] pass index of the step that causes the event
    const int32_t* ev_leaf;        // [E] leaf (local to the owning region) whose counters drop
    const int32_t* ev_w;           // [E] by this much
    const int32_t* crec;           // [P * kCW] compact step records in chain order
    int32_t* out;                  // [P * OW]
    int32_t waves;                 // waves of a region's workgroup: 0 = the launcher decides (8 when the LDS is there), 4, 8
    int32_t* flags;                // [0] a step is not region-local, [1] a chain had to escape,
                                   // [2] steps committed as verified stays, [3] stay batches,
                                   // [4] flat mode: first step not done, [5] stopped by the key range
};

// k_stay_by_top (k_stay.h): a chain pass of stays verified by one thread per top priority node
constexpr int kStayMaxLeaves = 512;
struct StayParams {
    int32_t N, NX, M, s, k, NP, OW, booster_kind;
    const int32_t* wg_region;      // [grid] region of workgroup b ...
    const int32_t* wg_chunk;       // ... and which 64 leaves of it: leaf = reg_lo + 64 * chunk + lane
    const int32_t* reg_lo;
    const int32_t* reg_hi;
    const int32_t* leaf_node;
    const int32_t* leaf_cls;
    const int32_t* cls_size;
    const uint8_t* alive;
    const int32_t* node_weight;
    const uint8_t* node_has_weight;
    const int32_t* cnt;
    const int32_t* top_off;        // [n_leaves + 1]
    const int32_t* crec;           // compact step records, chain order
    const int32_t* top_order;      // chain indices grouped by top leaf
    int32_t* out;
    int32_t* flag;                 // set to 1 by any step that is not a certain stay
};


// Flat (no hierarchy rule) passes resolved in bulk: DESIGN.md "Flat bulk engine".
constexpr int kTopList = 8;      // smallest partition-independent scores kept for the stay test

struct FlatParams {
    int32_t N, NX, M, L, P;
    int32_t s, k, top_state, NP, RW, OW;
    int32_t higher_mask, booster_kind;
    const uint8_t* alive;
    const int32_t* node_weight;
    const uint8_t* node_has_weight;
    const int32_t* cnt;            // [(M + 1) * NX]
    const int32_t* tot;            // [NX] sum over states, refreshed by k_flat_prepare
    const double* g;               // [NX] partition-independent score of every node
    const double* top_g;           // [kTopList] smallest (g, node) among nodesNext
    const int32_t* top_n;
    const int32_t* row_count;      // [NX + 1] steps of this pass per top priority node
    int32_t* ntn;
    const int32_t* rec;
    int32_t* out;
    int32_t* scan;                 // [0] first step that is not a certain stay, [1] first not fresh-identical
    int32_t* scan_part;            // [2][scan_waves] per wave of k_flat_scan: its first such step, reduced by k_flat_scan_min
    int32_t scan_waves;
    int32_t int_keys;              // a fresh run's sort keys as the integers they are (NumPartitions == 0, no node has a weight: a
                                   // score is count + picks * weight exactly) instead of their fp64 images: fewer varying bytes,
                                   // one radix pass less; any order-preserving injection gives the same sorted values
};

}  // namespace blance
