// the orchestrator's move selection at a million partitions -- host side, no device code.
//
// The reference rebuilds, EVERY supply round, a map node -> partitions whose next move goes to that node by
// walking all of o.mapPartitionToNextMoves (findAvailableMovesUnlocked, orchestrate.go:749-763, called at
// :521), and then asks FindMoveFunc for the best of a node's bucket once per slot, each time materialising
// the whole bucket (findNextMoves, orchestrate.go:698-714; filterNextPlausibleMovesForNode, :482-504).  With a
// planner that takes milliseconds that O(P)-per-round walk is what a million-partition rebalance waits for.
//
// MoveIndex keeps the same information incrementally:
//   * a partition sits in the bucket (node, op) of its NEXT move and is re-filed only when its Next advances
//     (orchestrate.go:689) -- O(1), swap-remove with a position table;
//   * the nodes that have any pending move are a dense list -- a supply round touches those nodes only;
//   * lowest_weight(node, count) is what filterNextPlausibleMovesForNode yields under the reference's default
//     FindMoveFunc LowestWeightPartitionMoveForNode (orchestrate.go:174-184, MoveOpWeight :187-192): `count`
//     moves in ascending op weight -- (unknown ops, weight 0), promote, demote, add, del.  Which partition comes first among equal
//     weights is not specified by the reference (it ranges over a Go map, orchestrate.go:755); any member of the
//     lowest non-empty op class is a valid answer, and that is what the equivalence test checks;
//   * bucket(node) lists everything pending for a node, for an application FindMoveFunc (O(bucket), still no
//     walk over all partitions).
// go/blance/orchestrate_index.go is the same structure in Go for the reference's package (not compilable in
// this image); this header is its compiled twin, exercised by move_index_sim.cpp / tests/test_move_index.py.
#pragma once
#include <cstdint>
#include <vector>

namespace blance {

// op classes in ascending MoveOpWeight (orchestrate.go:187-192); an op the reference's table does not know weighs 0 there
// (a missing map key) and ranks BEFORE promote: class 0
enum MoveOp : int8_t { kOpUnknown = 0, kOpPromote = 1, kOpDemote = 2, kOpAdd = 3, kOpDel = 4 };
constexpr int kMoveOpClasses = 5;

struct NodeStateOpId { int32_t node; int32_t state; int8_t op; };

struct NextMovesId {                       // NextMoves, orchestrate.go:194-212, with interned names
    std::vector<NodeStateOpId> moves;      // immutable
    int32_t next = 0;
};

class MoveIndex {
public:
    MoveIndex(int n_nodes, const std::vector<NextMovesId>* all)
        : all_(all), buckets_((size_t)n_nodes * kMoveOpClasses), pending_(n_nodes, 0), active_pos_(n_nodes, -1),
          where_(all->size(), -1) {
        for (size_t p = 0; p < all->size(); p++) insert((int32_t)p);
    }

    // nodes with at least one pending move (the key set of findAvailableMovesUnlocked's map).  The Go form hands out
    // SNAPSHOTS taken under the orchestrator's lock (go/blance/orchestrate_index.go "LOCKING"): advanced() reorders this
    // list and the buckets, so nothing here may be iterated while moves complete concurrently.
    const std::vector<int32_t>& active_nodes() const { return active_; }
    std::vector<int32_t> active_snapshot() const { return active_; }
    int64_t pending_total() const { return total_; }
    int32_t pending(int node) const { return pending_[node]; }

    // up to `count` partitions whose next move goes to `node`, ascending op weight
    void lowest_weight(int node, int count, std::vector<int32_t>* out) const {
        out->clear();
        for (int op = 0; op < kMoveOpClasses && count > 0; op++) {
            const std::vector<int32_t>& b = buckets_[(size_t)node * kMoveOpClasses + op];
            for (size_t i = 0; i < b.size() && count > 0; i++, count--) out->push_back(b[i]);
        }
    }

    // every partition whose next move goes to `node` (for an application's own FindMoveFunc)
    void bucket(int node, std::vector<int32_t>* out) const {
        out->clear();
        for (int op = 0; op < kMoveOpClasses; op++) {
            const std::vector<int32_t>& b = buckets_[(size_t)node * kMoveOpClasses + op];
            out->insert(out->end(), b.begin(), b.end());
        }
    }

    // the caller has just incremented (*all)[p].next (orchestrate.go:689): re-file the partition
    void advanced(int32_t p) {
        const NextMovesId& nm = (*all_)[p];
        const NodeStateOpId& old = nm.moves[nm.next - 1];
        remove(p, old.node, old.op);
        insert(p);
    }

private:
    // an op outside the enum (a caller with other numbering, a replayed file) is filed as "unknown", as the Go twin does
    static int op_class(int op) { return (unsigned)op < (unsigned)kMoveOpClasses ? op : (int)kOpUnknown; }
    void insert(int32_t p) {
        const NextMovesId& nm = (*all_)[p];
        if (nm.next >= (int32_t)nm.moves.size()) { where_[p] = -1; return; }     // nothing left for this partition
        const NodeStateOpId& m = nm.moves[nm.next];
        std::vector<int32_t>& b = buckets_[(size_t)m.node * kMoveOpClasses + op_class(m.op)];
        where_[p] = (int32_t)b.size();
        b.push_back(p);
        if (pending_[m.node]++ == 0) { active_pos_[m.node] = (int32_t)active_.size(); active_.push_back(m.node); }
        total_++;
    }
    void remove(int32_t p, int node, int op) {
        std::vector<int32_t>& b = buckets_[(size_t)node * kMoveOpClasses + op_class(op)];
        const int32_t at = where_[p];
        if (at < 0 || at >= (int32_t)b.size() || b[at] != p) return;     // not filed: nothing to take out
        const int32_t last = b.back();
        b[at] = last;
        where_[last] = at;
        b.pop_back();
        where_[p] = -1;
        if (--pending_[node] == 0) {
            const int32_t ap = active_pos_[node], ln = active_.back();
            active_[ap] = ln;
            active_pos_[ln] = ap;
            active_.pop_back();
            active_pos_[node] = -1;
        }
        total_--;
    }

    const std::vector<NextMovesId>* all_;
    std::vector<std::vector<int32_t>> buckets_;    // [node * 4 + op] partitions whose next move is (node, op)
    std::vector<int32_t> pending_;                 // [node] size of its four buckets together
    std::vector<int32_t> active_, active_pos_;     // nodes with pending > 0, and where each sits in that list
    std::vector<int32_t> where_;                   // [partition] position inside its bucket, -1 if done
    int64_t total_ = 0;
};

}  // namespace blance
