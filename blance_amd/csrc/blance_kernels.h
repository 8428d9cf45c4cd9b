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

// One rule's include/exclude leaf intervals for one anchor (plan.go:723-734):
// leaves(findAncestor(a, IncludeLevel)) = [alo, ahi), leaves(findAncestor(a, ExcludeLevel)) = [blo, bhi).
struct AnchorSet { int32_t alo, ahi, blo, bhi; };

// Parameters of one state pass (assignStateToPartitions, plan.go:253-303).
struct PassParams {
    int32_t N, NX, M, L, P;
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
    const int32_t* rule_inc_unused;
    const AnchorSet* anchors;      // [n_rules][NX + 1]
    int32_t* cnt;                  // [(M + 1) * NX] stateNodeCounts
    int32_t* ntn;                  // [(NX + 1) * N] nodeToNodeCounts, zeroed per pass
    const int32_t* rec;            // [P * RW] step records in pass order
    int32_t* out;                  // [P * OW] chosen nodes per step
    int32_t* warn_part;
    int32_t* warn_state;
    int32_t* warn_count;
    int32_t* err;                  // device error word (interval overflow ...)
};

}  // namespace blance
