// TEST / BASELINE INFRASTRUCTURE ONLY -- never linked into or called by the product path.
//
// The timing proxy for "the reference Go path" that SURVEY.md 8(d) / BASELINE.md section 4 ask for
// (no Go toolchain exists in the image): a DELIBERATELY NAIVE restatement of planNextMapEx in the
// reference's own style -- string-keyed hash maps everywhere, a fresh nodePartitionCounts map per
// findBestNodes call (plan.go:118-124), candidate filtering through freshly built sets
// (misc.go:27-51), a comparison sort whose Less evaluates Score twice (plan.go:617-628), Score doing
// its seven map lookups (plan.go:634-689), hierarchy sets by recursive findLeaves with slices
// (plan.go:723-774).  It plans the synthetic inputs of BASELINE.json configs 2 and 3 (generated
// here, as strings) and prints one line per partition ("name|primary|replica" as node positions), so
// tests/test_naive_proxy.py can check it against the id-based oracle before its time is quoted.
//
//   naive_proxy <config 2|3> <partitions> <nodes> [quiet]
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

typedef std::vector<std::string> Strs;
typedef std::unordered_map<std::string, int> Counts;
typedef std::unordered_map<std::string, Strs> NodesByState;

struct Partition { std::string Name; NodesByState nbs; };
struct ModelState { int Priority, Constraints; };
struct Rule { int IncludeLevel, ExcludeLevel; };

static std::unordered_map<std::string, bool> StringsToMap(const Strs& a) {           // misc.go:13-22
    std::unordered_map<std::string, bool> m;
    for (auto& s : a) m[s] = true;
    return m;
}
static Strs StringsRemoveStrings(const Strs& a, const Strs& rm) {                     // misc.go:27-38
    auto m = StringsToMap(rm);
    Strs r;
    for (auto& s : a) if (!m.count(s)) r.push_back(s);
    return r;
}
static Strs StringsIntersectStrings(const Strs& a, const Strs& b) {                   // misc.go:40-51
    auto m = StringsToMap(b);
    Strs r;
    for (auto& s : a) if (m.count(s)) r.push_back(s);
    return r;
}
static Strs stringsDeduplicate(const Strs& a) {                                       // misc.go:55-66
    std::unordered_map<std::string, bool> seen;
    Strs r;
    for (auto& s : a) if (!seen.count(s)) { seen[s] = true; r.push_back(s); }
    return r;
}

static std::string findAncestor(std::string node, const std::unordered_map<std::string, std::string>& parents, int level) {
    for (; level > 0; level--) {                                                      // plan.go:755-762
        auto it = parents.find(node);
        node = it == parents.end() ? std::string() : it->second;
    }
    return node;
}
static Strs findLeaves(const std::string& node, const std::unordered_map<std::string, Strs>& children) {   // plan.go:764-774
    auto it = children.find(node);
    if (it == children.end() || it->second.empty()) return Strs{node};
    Strs r;
    for (auto& c : it->second) { Strs l = findLeaves(c, children); r.insert(r.end(), l.begin(), l.end()); }
    return r;
}
static Strs includeExcludeNodes(const std::string& node, int inc, int exc,
                                const std::unordered_map<std::string, std::string>& parents,
                                const std::unordered_map<std::string, Strs>& children) {                  // plan.go:723-734
    Strs incNodes = findLeaves(findAncestor(node, parents, inc), children);
    Strs excNodes = findLeaves(findAncestor(node, parents, exc), children);
    return StringsRemoveStrings(incNodes, excNodes);
}
static Strs includeExcludeNodesIntersect(const Strs& nodes, int inc, int exc,
                                         const std::unordered_map<std::string, std::string>& parents,
                                         const std::unordered_map<std::string, Strs>& children) {         // plan.go:738-753
    Strs rv;
    for (auto& n : nodes) {
        if (rv.empty()) { rv = includeExcludeNodes(n, inc, exc, parents, children); continue; }
        rv = StringsIntersectStrings(rv, includeExcludeNodes(n, inc, exc, parents, children));
    }
    return rv;
}

struct Sorter {                                                                        // nodeSorter, plan.go:598-689
    const std::string* stateName;
    const Partition* partition;
    int numPartitions;
    const std::string* topPriorityNode;
    const std::unordered_map<std::string, Counts>* stateNodeCounts;
    const std::unordered_map<std::string, Counts>* nodeToNodeCounts;
    const Counts* nodePartitionCounts;
    const Counts* nodePositions;
    const Counts* nodeWeights;
    double stickiness;
    double Score(const std::string& node) const {
        double lowerPriorityBalanceFactor = 0.0;
        if (numPartitions > 0) {
            auto m = nodeToNodeCounts->find(*topPriorityNode);
            if (m != nodeToNodeCounts->end()) {
                auto c = m->second.find(node);
                if (c != m->second.end()) lowerPriorityBalanceFactor = (double)c->second / (double)numPartitions;
            }
        }
        double filledFactor = 0.0;
        if (numPartitions > 0) {
            auto c = nodePartitionCounts->find(node);
            if (c != nodePartitionCounts->end()) filledFactor = (0.001 * (double)c->second) / (double)numPartitions;
        }
        double currentFactor = 0.0;
        auto own = partition->nbs.find(*stateName);
        if (own != partition->nbs.end())
            for (auto& n : own->second) if (n == node) currentFactor = stickiness;
        double r = 0.0;
        auto sc = stateNodeCounts->find(*stateName);
        if (sc != stateNodeCounts->end()) {
            auto c = sc->second.find(node);
            if (c != sc->second.end()) r = (double)c->second;
        }
        r = r + lowerPriorityBalanceFactor;
        r = r + filledFactor;
        if (nodeWeights) {
            auto w = nodeWeights->find(node);
            if (w != nodeWeights->end() && w->second > 0) r = r / (double)w->second;
        }
        r = r - currentFactor;
        return r;
    }
    bool Less(const std::string& a, const std::string& b) const {                      // plan.go:617-628
        double si = Score(a), sj = Score(b);
        if (si < sj) return true;
        if (si > sj) return false;
        return nodePositions->at(a) < nodePositions->at(b);
    }
};

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: naive_proxy <config 2|3> <partitions> <nodes> [quiet]\n"); return 2; }
    const int cfg = atoi(argv[1]), P = atoi(argv[2]), N = atoi(argv[3]);
    const bool quiet = argc > 4;
    char buf[64];
    Strs nodesAll;
    for (int i = 0; i < N; i++) { snprintf(buf, sizeof buf, cfg == 2 ? "n%03d" : "n%04d", i); nodesAll.push_back(buf); }
    std::map<std::string, ModelState> model;
    model["primary"] = ModelState{0, 1};
    model["replica"] = ModelState{1, cfg == 2 ? 1 : 2};
    std::unordered_map<std::string, std::string> nodeHierarchy;
    std::unordered_map<std::string, std::vector<Rule>> hierarchyRules;
    const bool rules = cfg == 3;
    if (rules) {
        const int n_racks = (N + 15) / 16, n_zones = (n_racks + 7) / 8;
        for (int i = 0; i < N; i++) { snprintf(buf, sizeof buf, "r%03d", i / 16); nodeHierarchy[nodesAll[i]] = buf; }
        for (int r = 0; r < n_racks; r++) { char z[32]; snprintf(buf, sizeof buf, "r%03d", r); snprintf(z, sizeof z, "z%02d", r / 8); nodeHierarchy[buf] = z; }
        for (int z = 0; z < n_zones; z++) { char d[32]; snprintf(buf, sizeof buf, "z%02d", z); snprintf(d, sizeof d, "d%d", z / 8); nodeHierarchy[buf] = d; }
        hierarchyRules["replica"].push_back(Rule{2, 1});
    }
    std::map<std::string, Partition> prevMap, partitionsToAssign;                     // Go maps; iteration order never matters below
    for (int i = 0; i < P; i++) { Partition p; p.Name = std::to_string(i); partitionsToAssign[p.Name] = p; }
    Strs nodesToRemove, nodesToAdd = nodesAll;
    Strs nodes = nodesAll;

    const auto t0 = std::chrono::steady_clock::now();
    long long calls = 0;
    std::vector<Partition> result;
    int iterations = 0;
    for (int it = 0; it < 10; it++) {                                                  // planNextMapEx, plan.go:32-57
        iterations++;
        // ---- planNextMapInnerEx, plan.go:60-331
        Counts nodePositions;
        for (int i = 0; i < (int)nodes.size(); i++) nodePositions[nodes[i]] = i;
        Strs nodesNext = StringsRemoveStrings(nodes, nodesToRemove);
        std::unordered_map<std::string, Strs> hierarchyChildren;                      // mapParentsToMapChildren, plan.go:703-717
        {
            Strs kids;
            for (auto& kv : nodeHierarchy) kids.push_back(kv.first);
            std::sort(kids.begin(), kids.end());
            for (auto& c : kids) hierarchyChildren[nodeHierarchy[c]].push_back(c);
        }
        std::vector<Partition> nextPartitions;
        for (auto& kv : partitionsToAssign) {
            Partition p = kv.second;
            for (auto& sl : p.nbs) sl.second = StringsRemoveStrings(sl.second, nodesToRemove);
            nextPartitions.push_back(p);
        }
        auto name_key = [](const std::string& n) {                                     // plan.go:519-540
            char b[32];
            char* end = nullptr;
            long v = strtol(n.c_str(), &end, 10);
            if (*end == 0 && !n.empty() && v >= 0) { snprintf(b, sizeof b, "%10ld", v); return std::string(b); }
            return n;
        };
        std::sort(nextPartitions.begin(), nextPartitions.end(), [&](const Partition& a, const Partition& b) {
            std::string ka = name_key(a.Name), kb = name_key(b.Name);
            if (ka != kb) return ka < kb;
            return a.Name < b.Name;
        });
        std::unordered_map<std::string, Counts> stateNodeCounts;                      // countStateNodes, plan.go:374-399
        for (auto& kv : prevMap)
            for (auto& sl : kv.second.nbs)
                for (auto& n : sl.second) stateNodeCounts[sl.first][n] += 1;
        const int numPartitions = (int)prevMap.size();
        std::vector<std::string> stateNames = {"primary", "replica"};                  // sortStateNames
        for (auto& stateName : stateNames) {
            const int constraints = model[stateName].Constraints;
            // partitionSorter (plan.go:481-562): with nodesToRemove empty and every partition either holding no
            // node yet or nodesToAdd empty-but-non-nil, all partitions share a category; the order is the name order
            std::unordered_map<std::string, Counts> nodeToNodeCounts;                 // plan.go:266
            for (auto& partition : nextPartitions) {
                calls++;
                // ---- findBestNodes, plan.go:98-248
                const double stickiness = 1.5;
                Counts nodePartitionCounts;                                             // plan.go:118-124
                for (auto& sc : stateNodeCounts)
                    for (auto& nc : sc.second) nodePartitionCounts[nc.first] += nc.second;
                std::string topPriorityNode;
                {
                    auto itp = partition.nbs.find("primary");
                    if (itp != partition.nbs.end() && !itp->second.empty()) topPriorityNode = itp->second[0];
                }
                Strs excludeHigher;                                                     // plan.go:142-156
                for (auto& sl : partition.nbs)
                    if (model[sl.first].Priority < model[stateName].Priority)
                        excludeHigher.insert(excludeHigher.end(), sl.second.begin(), sl.second.end());
                Strs candidateNodes = StringsRemoveStrings(nodesNext, excludeHigher);
                Sorter sorter{&stateName, &partition, numPartitions, &topPriorityNode, &stateNodeCounts, &nodeToNodeCounts,
                              &nodePartitionCounts, &nodePositions, nullptr, stickiness};
                std::sort(candidateNodes.begin(), candidateNodes.end(),
                          [&](const std::string& a, const std::string& b) { return sorter.Less(a, b); });
                if (rules) {                                                            // plan.go:174-226
                    Strs hierarchyNodes;
                    auto hr = hierarchyRules.find(stateName);
                    if (hr != hierarchyRules.end())
                        for (auto& rule : hr->second) {
                            std::string h = topPriorityNode;
                            if (h.empty() && !hierarchyNodes.empty()) h = hierarchyNodes[0];
                            for (int i = 0; i < constraints; i++) {
                                Strs anchors{h};
                                anchors.insert(anchors.end(), hierarchyNodes.begin(), hierarchyNodes.end());
                                Strs hc = includeExcludeNodesIntersect(anchors, rule.IncludeLevel, rule.ExcludeLevel,
                                                                       nodeHierarchy, hierarchyChildren);
                                hc = StringsIntersectStrings(hc, nodesNext);
                                hc = StringsRemoveStrings(hc, excludeHigher);
                                std::sort(hc.begin(), hc.end(), [&](const std::string& a, const std::string& b) { return sorter.Less(a, b); });
                                if (!hc.empty()) hierarchyNodes.push_back(hc[0]);
                                else if (!candidateNodes.empty()) hierarchyNodes.push_back(candidateNodes[0]);
                            }
                        }
                    Strs joined = hierarchyNodes;
                    joined.insert(joined.end(), candidateNodes.begin(), candidateNodes.end());
                    candidateNodes = stringsDeduplicate(joined);
                }
                if ((int)candidateNodes.size() >= constraints) candidateNodes.resize(constraints);
                for (auto& c : candidateNodes) nodeToNodeCounts[topPriorityNode][c] += 1;   // plan.go:238-245
                // ---- assignStateToPartitions, plan.go:287-301
                Strs old = partition.nbs[stateName];
                for (auto& sl : partition.nbs) {
                    for (auto& n : StringsIntersectStrings(sl.second, old)) stateNodeCounts[sl.first][n] -= 1;
                    sl.second = StringsRemoveStrings(sl.second, old);
                }
                for (auto& sl : partition.nbs) {
                    for (auto& n : StringsIntersectStrings(sl.second, candidateNodes)) stateNodeCounts[sl.first][n] -= 1;
                    sl.second = StringsRemoveStrings(sl.second, candidateNodes);
                }
                partition.nbs[stateName] = candidateNodes;
                for (auto& n : candidateNodes) stateNodeCounts[stateName][n] += 1;
            }
        }
        // ---- convergence, plan.go:36-57
        bool same = true;
        for (auto& p : nextPartitions) {
            auto ip = prevMap.find(p.Name);
            if (ip == prevMap.end() || ip->second.nbs != p.nbs) { same = false; break; }
        }
        result = nextPartitions;
        if (same) break;
        for (auto& p : nextPartitions) { prevMap[p.Name] = p; partitionsToAssign[p.Name] = p; }
        nodes = StringsRemoveStrings(nodes, nodesToRemove);
        nodesToRemove.clear();
        nodesToAdd.clear();
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!quiet) {
        Counts pos;
        for (int i = 0; i < N; i++) pos[nodesAll[i]] = i;
        for (auto& p : result) {
            printf("%s|", p.Name.c_str());
            const Strs& a = p.nbs["primary"];
            for (size_t i = 0; i < a.size(); i++) printf("%s%d", i ? "," : "", pos[a[i]]);
            printf("|");
            const Strs& b = p.nbs["replica"];
            for (size_t i = 0; i < b.size(); i++) printf("%s%d", i ? "," : "", pos[b[i]]);
            printf("\n");
        }
    }
    fprintf(stderr, "{\"config\": %d, \"partitions\": %d, \"nodes\": %d, \"sweeps\": %d, \"find_best_nodes_calls\": %lld, "
            "\"seconds\": %.3f, \"assignments_per_s\": %.2f}\n", cfg, P, N, iterations, calls, secs,
            (double)P * (cfg == 2 ? 2 : 3) / secs);
    return 0;
}
