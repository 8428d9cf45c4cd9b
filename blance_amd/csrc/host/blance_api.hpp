// C++ host-side mirror of couchbase/blance's planner API (api.go:24-190) over the
// C ABI of include/blance_hip.h.  Same type names, argument meaning, warnings text
// and caller-visible mutations as the reference; this is what the cgo shim of
// INTEGRATION.md does in Go (no Go toolchain exists in the build image).
//
// Go's nil vs empty distinctions are kept with std::optional: a disengaged
// optional is a nil slice / nil map.
#pragma once
#include <map>
#include <memory>
#include <optional>
#include <string>
#include <vector>

namespace blance {

using StringList = std::optional<std::vector<std::string>>;          // []string, nil-able

struct Partition {                                                    // api.go:28-36
    std::string Name;
    std::optional<std::map<std::string, StringList>> NodesByState;    // map[string][]string
};
using PartitionPtr = std::shared_ptr<Partition>;
using PartitionMap = std::map<std::string, PartitionPtr>;            // api.go:24

struct PartitionModelState {                                          // api.go:46-62
    int Priority = 0;
    int Constraints = 0;
};
using PartitionModel = std::map<std::string, std::shared_ptr<PartitionModelState>>;   // api.go:41

struct HierarchyRule {                                                // api.go:96-105
    int IncludeLevel = 0;
    int ExcludeLevel = 0;
};
using HierarchyRules = std::map<std::string, std::vector<std::shared_ptr<HierarchyRule>>>;   // api.go:74

struct PlanNextMapOptions {                                           // api.go:183-190
    std::optional<std::map<std::string, int>> ModelStateConstraints;
    std::optional<std::map<std::string, int>> PartitionWeights;
    std::optional<std::map<std::string, int>> StateStickiness;
    std::optional<std::map<std::string, int>> NodeWeights;
    std::optional<std::map<std::string, std::string>> NodeHierarchy;
    std::optional<HierarchyRules> HierarchyRules_;
};

// package-level knobs of the reference (plan.go:21, :693)
extern int MaxIterationsPerPlan;
enum class Booster { None = 0, Cbgt = 1, Other = 2 };   // control_test.go:19-26 is the only booster known in the wild;
extern Booster NodeScoreBooster;                        // Other: some callback of the caller's -- not handled (plan.go:693)
extern bool CustomNodeSorterIsDefault;                  // plan.go:580; false: a caller's sorter -- not handled

using Warnings = std::map<std::string, std::vector<std::string>>;

struct PlanOutcome {
    bool handled = false;        // false: input outside the device envelope (the Go shim would run plan.go)
    std::string why;             // reason when !handled
    bool nil_result = false;     // MaxIterationsPerPlan <= 0: planNextMapEx returns (nil, nil)
    PartitionMap nextMap;
    Warnings warnings;
    int iterations = 0;
    bool converged = false;
    // where the call's time went (ms): strings -> ids, blance_plan (H2D + device + D2H), ids -> strings
    double intern_ms = 0.0, plan_ms = 0.0, unintern_ms = 0.0, device_ms = 0.0;
    // inside unintern_ms: the Partition objects (threads), the result map, the stores of plan.go:49-52 into the input maps
    double unintern_parts_ms = 0.0, unintern_map_ms = 0.0, store_ms = 0.0;
    int threads = 1;             // threads that built the result's objects (the result map and the two stores then run side by side)
};

// The C ABI entry points, resolved at run time so that one binary can drive
// libblance_hip.so (gfx950) or, in tests, the emulated build of the same kernels.
struct Library {
    void* handle = nullptr;
    void* ctx = nullptr;
    bool open(const std::string& path, std::string* err);
    void close();
    // The flat arrays of the last call (~200 MB at a million partitions) stay with the Library so that the next call does
    // not fault them in again; trim() lets them go (e.g. after one huge plan in a long-lived process).  Safe while no call
    // is running on this Library; a call that finds the arrays taken or trimmed allocates its own.
    void trim();
    ~Library() { close(); }
};

// PlanNextMapEx, api.go:147-157.  prevMap / partitionsToAssign are mutated as by
// planNextMapEx (plan.go:49-52).
PlanOutcome PlanNextMapEx(Library& lib, PartitionMap* prevMap, PartitionMap& partitionsToAssign,
                          const std::vector<std::string>& nodesAll, const StringList& nodesToRemove,
                          const StringList& nodesToAdd, const PartitionModel& model,
                          const PlanNextMapOptions& options);

// CalcPartitionMoves, moves.go:41-119 (NodeStateOp: moves.go:21-27), for every partition of the two
// maps in one device call -- the loop of OrchestrateMoves, orchestrate.go:273-287.  Partitions missing
// from one map count as holding nothing there; a nil NodesByState likewise.
struct NodeStateOp {
    std::string Node, State, Op;     // Op: "add", "del", "promote", "demote"
};
using NodesByState = std::map<std::string, StringList>;
struct MovesOutcome {
    bool ok = false;
    std::string why;
    std::map<std::string, std::vector<NodeStateOp>> moves;      // by partition name
};
MovesOutcome CalcPartitionMovesBatch(Library& lib, const std::vector<std::string>& states,
                                     const std::map<std::string, NodesByState>& begMap,
                                     const std::map<std::string, NodesByState>& endMap, bool favorMinNodes);

// misc.go:13-51
std::map<std::string, bool> StringsToMap(const std::vector<std::string>& strs);
std::vector<std::string> StringsRemoveStrings(const std::vector<std::string>& a, const std::vector<std::string>& remove);
std::vector<std::string> StringsIntersectStrings(const std::vector<std::string>& a, const std::vector<std::string>& b);

}  // namespace blance
