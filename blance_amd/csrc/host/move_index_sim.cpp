// Randomised equivalence test + timing of MoveIndex (move_index.hpp) against the reference's per-round rescan.
//   move_index_sim check <seed> <partitions> <nodes> <count>   -> "OK rounds=..." or "MISMATCH ..."
//   move_index_sim time <partitions> <nodes> <count>           -> JSON line: per-round cost of both
// The rescan is a restatement of findAvailableMovesUnlocked (orchestrate.go:749-763) and of
// filterNextPlausibleMovesForNode with LowestWeightPartitionMoveForNode (orchestrate.go:482-504, :174-184) on
// interned ids.  Rounds follow runSupplyMoves (orchestrate.go:506-590): every node with available moves is
// offered `count` of them; a random subset of the offered batches completes (Next++ for each of its
// partitions, orchestrate.go:684-691) before the next round.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <set>

#include "move_index.hpp"
using namespace blance;

static const int kOpWeight[5] = {0, 1, 2, 3, 4};       // MoveOpWeight, orchestrate.go:187-192 (enum order; 0: an op the table does not know)

// moves of one partition: a plausible CalcPartitionMoves output (1..4 steps over distinct nodes)
static void make_moves(std::mt19937_64& rng, int n_nodes, std::vector<NextMovesId>& all) {
    for (auto& nm : all) {
        const int len = (int)(rng() % 5);              // some partitions have nothing to do
        nm.moves.clear();
        nm.next = 0;
        for (int i = 0; i < len; i++) {
            NodeStateOpId m;
            m.node = (int32_t)(rng() % n_nodes);
            m.state = (int32_t)(rng() % 2);
            m.op = (int8_t)(rng() % 16 == 0 ? 0 : 1 + rng() % 4);
            nm.moves.push_back(m);
        }
    }
}

// findAvailableMovesUnlocked: node -> partitions with a next move for it
static void rescan(const std::vector<NextMovesId>& all, std::vector<std::vector<int32_t>>& avail) {
    for (auto& v : avail) v.clear();
    for (size_t p = 0; p < all.size(); p++) {
        const NextMovesId& nm = all[p];
        if (nm.next < (int32_t)nm.moves.size()) avail[nm.moves[nm.next].node].push_back((int32_t)p);
    }
}

// filterNextPlausibleMovesForNode with the default FindMoveFunc: the op weights of the picks, in pick order
static void reference_pick_weights(const std::vector<NextMovesId>& all, std::vector<int32_t> arr, int count,
                                   std::vector<int>& weights) {
    weights.clear();
    if (count <= 0) count = 1;
    if (count > (int)arr.size()) count = (int)arr.size();
    while (count-- > 0) {
        size_t r = 0;
        for (size_t i = 0; i < arr.size(); i++) {
            const int wr = kOpWeight[all[arr[r]].moves[all[arr[r]].next].op], wi = kOpWeight[all[arr[i]].moves[all[arr[i]].next].op];
            if (wr > wi) r = i;
        }
        weights.push_back(kOpWeight[all[arr[r]].moves[all[arr[r]].next].op]);
        arr[r] = arr.back();
        arr.pop_back();
    }
}

static int check(uint64_t seed, int P, int N, int count) {
    std::mt19937_64 rng(seed);
    std::vector<NextMovesId> all((size_t)P);
    make_moves(rng, N, all);
    MoveIndex ix(N, &all);
    std::vector<std::vector<int32_t>> avail((size_t)N);
    std::vector<int32_t> picks, lst;
    std::vector<int> want_w;
    long rounds = 0, moves_done = 0;
    for (;;) {
        rescan(all, avail);
        // (a) the same nodes have moves, (b) with the same partitions
        std::set<int32_t> act(ix.active_nodes().begin(), ix.active_nodes().end());
        if (act.size() != ix.active_nodes().size()) { printf("MISMATCH duplicate active node, round %ld\n", rounds); return 1; }
        int64_t total = 0;
        for (int n = 0; n < N; n++) {
            if ((avail[n].size() > 0) != (act.count(n) > 0)) { printf("MISMATCH active set, node %d round %ld\n", n, rounds); return 1; }
            total += (int64_t)avail[n].size();
            ix.bucket(n, &lst);
            std::vector<int32_t> a = avail[n], b = lst;
            std::sort(a.begin(), a.end());
            std::sort(b.begin(), b.end());
            if (a != b || (int)a.size() != ix.pending(n)) { printf("MISMATCH bucket of node %d round %ld\n", n, rounds); return 1; }
        }
        if (total != ix.pending_total()) { printf("MISMATCH total, round %ld\n", rounds); return 1; }
        if (total == 0) break;
        // (c) the picks: members of the node's bucket, distinct, op weights as the reference's picks
        std::vector<std::vector<int32_t>> offered;
        for (int32_t n : ix.active_nodes()) {
            ix.lowest_weight(n, count <= 0 ? 1 : count, &picks);
            reference_pick_weights(all, avail[n], count, want_w);
            if (picks.size() != want_w.size()) { printf("MISMATCH pick count, node %d round %ld\n", n, rounds); return 1; }
            std::set<int32_t> seen;
            for (size_t i = 0; i < picks.size(); i++) {
                const NextMovesId& nm = all[picks[i]];
                if (nm.next >= (int32_t)nm.moves.size() || nm.moves[nm.next].node != n || !seen.insert(picks[i]).second ||
                    kOpWeight[nm.moves[nm.next].op] != want_w[i]) {
                    printf("MISMATCH pick %zu of node %d round %ld\n", i, n, rounds);
                    return 1;
                }
            }
            offered.push_back(picks);
        }
        // some of the offered batches complete; at least one, as a fed mover always finishes (orchestrate.go:570-576)
        bool any = false;
        for (size_t b = 0; b < offered.size(); b++) {
            if (rng() % 3 == 0 && !(b + 1 == offered.size() && !any)) continue;
            any = true;
            for (int32_t p : offered[b]) { all[p].next++; ix.advanced(p); moves_done++; }
        }
        rounds++;
    }
    printf("OK rounds=%ld moves=%ld\n", rounds, moves_done);
    return 0;
}

static int timing(int P, int N, int count) {
    std::mt19937_64 rng(12345);
    std::vector<NextMovesId> all((size_t)P);
    make_moves(rng, N, all);
    std::vector<NextMovesId> all2 = all;
    using clk = std::chrono::steady_clock;
    // --- the reference's way: rescan + materialise + pick, every round, for `rounds` rounds
    const int rounds = 20;
    std::vector<std::vector<int32_t>> avail((size_t)N);
    std::vector<int> w;
    auto t0 = clk::now();
    long picked = 0;
    for (int r = 0; r < rounds; r++) {
        rescan(all, avail);
        for (int n = 0; n < N; n++) {
            if (avail[n].empty()) continue;
            reference_pick_weights(all, avail[n], count, w);
            picked += (long)w.size();
            // complete what was offered, the way the round after would see it
            std::vector<int32_t> arr = avail[n];
            for (size_t i = 0; i < w.size() && i < arr.size(); i++) all[arr[i]].next++;
        }
    }
    const double ref_ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count() / rounds;
    // --- the index
    auto t1 = clk::now();
    MoveIndex ix(N, &all2);
    const double build_ms = std::chrono::duration<double, std::milli>(clk::now() - t1).count();
    std::vector<int32_t> picks, nodes;
    auto t2 = clk::now();
    long picked2 = 0;
    for (int r = 0; r < rounds; r++) {
        nodes = ix.active_nodes();
        for (int32_t n : nodes) {
            ix.lowest_weight(n, count, &picks);
            picked2 += (long)picks.size();
            for (int32_t p : picks) { all2[p].next++; ix.advanced(p); }
        }
    }
    const double ix_ms = std::chrono::duration<double, std::milli>(clk::now() - t2).count() / rounds;
    printf("{\"partitions\": %d, \"nodes\": %d, \"moves_per_node_per_round\": %d, \"rounds\": %d, "
           "\"rescan_ms_per_round\": %.3f, \"index_ms_per_round\": %.4f, \"index_build_ms\": %.2f, "
           "\"moves_offered_rescan\": %ld, \"moves_offered_index\": %ld}\n",
           P, N, count, rounds, ref_ms, ix_ms, build_ms, picked, picked2);
    return 0;
}

// `replay <count>`: the partitions' move lists on stdin ("P N" then per partition "n (node state op) x n"), the
// orchestrator's supply rounds run through the index (orchestrate.go:506-590: every round each node with pending moves is
// offered its `count` lowest-weight next moves; all of them complete), the rescan of orchestrate.go:749-763 beside it.
// Prints, per round, the moves carried out as "partition:position" -- tests/test_move_index.py feeds it the move lists of
// the reference's own orchestrator fixtures (orchestrate_test.go:1049-1811).
static int replay(int count) {
    int P = 0, N = 0;
    if (scanf("%d %d", &P, &N) != 2) return 2;
    std::vector<NextMovesId> all((size_t)P);
    for (int p = 0; p < P; p++) {
        int n = 0;
        if (scanf("%d", &n) != 1) return 2;
        for (int i = 0; i < n; i++) {
            int node, state, op;
            if (scanf("%d %d %d", &node, &state, &op) != 3) return 2;
            all[(size_t)p].moves.push_back(NodeStateOpId{node, state, (int8_t)op});
        }
    }
    MoveIndex ix(N, &all);
    std::vector<std::vector<int32_t>> avail((size_t)N);
    std::vector<int32_t> picks, lst;
    printf("{\"rounds\":[");
    for (long round = 0;; round++) {
        rescan(all, avail);
        int64_t total = 0;
        for (int n = 0; n < N; n++) {
            ix.bucket(n, &lst);
            std::vector<int32_t> a = avail[n], b = lst;
            std::sort(a.begin(), a.end());
            std::sort(b.begin(), b.end());
            if (a != b) { printf("],\"error\":\"bucket of node %d differs from the rescan in round %ld\"}\n", n, round); return 1; }
            total += (int64_t)a.size();
        }
        if (total == 0) break;
        printf("%s[", round ? "," : "");
        bool first = true;
        const std::vector<int32_t> nodes = ix.active_snapshot();
        std::vector<std::pair<int32_t, int32_t>> done;
        for (int32_t n : nodes) {
            ix.lowest_weight(n, count <= 0 ? 1 : count, &picks);
            if ((int)picks.size() > (count <= 0 ? 1 : count)) { printf("],\"error\":\"too many picks\"}\n"); return 1; }
            for (int32_t p : picks) done.emplace_back(p, all[p].next);
        }
        // (a partition has one next move, so it is offered by one node only: no pick twice)
        for (auto& d : done) {
            printf("%s\"%d:%d\"", first ? "" : ",", d.first, d.second);
            first = false;
            all[d.first].next++;
            ix.advanced(d.first);
        }
        printf("]");
    }
    printf("],\"pending\":%lld}\n", (long long)ix.pending_total());
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 3 && !strcmp(argv[1], "replay")) return replay(atoi(argv[2]));
    if (argc >= 6 && !strcmp(argv[1], "check")) return check(strtoull(argv[2], nullptr, 10), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]));
    if (argc >= 5 && !strcmp(argv[1], "time")) return timing(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]));
    fprintf(stderr, "usage: move_index_sim check <seed> <P> <N> <count> | time <P> <N> <count>\n");
    return 2;
}
