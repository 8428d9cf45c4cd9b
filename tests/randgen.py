"""Seeded random PlanNextMapEx() inputs (API level: dicts of strings) used to
cross-check the oracles and the HIP path.  Covers the reference's feature set:
add/remove nodes, partition weights, node weights (incl. negative + booster),
stickiness, multi-primary, 0-constraint states, hierarchy rules on ragged
trees, prevMap != partitionsToAssign, nodes outside nodesAll."""
import random


def random_case(seed, max_nodes=12, max_parts=24):
    rng = random.Random(seed)
    n_nodes = rng.randint(1, max_nodes)
    nodes = ["n%02d" % i for i in range(n_nodes)]
    rng.shuffle(nodes)
    n_parts = rng.randint(0, max_parts)
    numeric = rng.random() < 0.6
    pnames = [str(i) if numeric else "p%03d" % i for i in range(n_parts)]
    if numeric and rng.random() < 0.3 and n_parts > 2:
        pnames[1] = "x" + pnames[1]           # mixed numeric / non numeric names

    model_kind = rng.choice(["p", "pr", "pr", "pr2", "p2r", "prx"])
    if model_kind == "p":
        model = {"primary": {"priority": 0, "constraints": 1}}
    elif model_kind == "pr":
        model = {"primary": {"priority": 0, "constraints": 1},
                 "replica": {"priority": 1, "constraints": rng.choice([0, 1, 1, 2])}}
    elif model_kind == "pr2":
        model = {"primary": {"priority": 0, "constraints": 1},
                 "replica": {"priority": 1, "constraints": 2}}
    elif model_kind == "p2r":
        model = {"primary": {"priority": 0, "constraints": 2},
                 "replica": {"priority": 1, "constraints": 1}}
    else:
        model = {"primary": {"priority": 0, "constraints": 1},
                 "replica": {"priority": 1, "constraints": 1},
                 "zreadonly": {"priority": 2, "constraints": rng.choice([0, 1])}}
    states = list(model.keys())

    # an existing layout for some partitions
    extra_nodes = ["gone1", "gone2"] if rng.random() < 0.2 else []
    pool = nodes + extra_nodes

    def random_nbs(full):
        nbs = {}
        avail = pool[:]
        rng.shuffle(avail)
        for s in states:
            if not full and rng.random() < 0.3:
                continue
            k = max(model[s]["constraints"], 0)
            cnt = rng.choice([0, k, k, max(k - 1, 0), k + 1]) if full else rng.choice([0, k])
            lst = [avail.pop() for _ in range(min(cnt, len(avail)))]
            r = rng.random()
            if not lst and r < 0.15:
                nbs[s] = None
            else:
                nbs[s] = lst
        return nbs

    mode = rng.choice(["fresh", "aliased", "aliased", "separate"])
    prev, assign = {}, {}
    if mode == "fresh":
        for p in pnames:
            assign[p] = {"name": p, "nodesByState": {}}
    elif mode == "aliased":
        for p in pnames:
            prev[p] = {"name": p, "nodesByState": random_nbs(True)}
        assign = None
    else:
        for p in pnames:
            if rng.random() < 0.8:
                prev[p] = {"name": p, "nodesByState": random_nbs(True)}
            assign[p] = {"name": p, "nodesByState": random_nbs(rng.random() < 0.7)}
        for j in range(rng.randint(0, 3)):      # partitions only in prevMap
            q = "only%d" % j
            nbs = random_nbs(True)
            if rng.random() < 0.3:
                nbs["dead"] = [rng.choice(pool)]
            prev[q] = {"name": q, "nodesByState": nbs}

    n_rm = rng.choice([0, 0, 1, 2]) if n_nodes > 2 else 0
    to_remove = rng.sample(nodes, min(n_rm, n_nodes))
    rest = [n for n in nodes if n not in to_remove]
    n_add = rng.choice([0, 0, 1, 2, len(rest)])
    to_add = rng.sample(rest, min(n_add, len(rest)))
    r = rng.random()
    nodes_to_add = None if r < 0.15 else to_add
    nodes_to_remove = None if (not to_remove and rng.random() < 0.2) else to_remove
    if to_remove and mode == "separate":
        # reference panics when a partition to assign is missing from prevMap
        for p in pnames:
            prev.setdefault(p, {"name": p, "nodesByState": random_nbs(True)})

    opts = {"modelStateConstraints": None, "partitionWeights": None, "stateStickiness": None,
            "nodeWeights": None, "nodeHierarchy": None, "hierarchyRules": None}
    if rng.random() < 0.2:
        s = rng.choice(states)
        opts["modelStateConstraints"] = {s: rng.choice([0, 1, 2, 3])}
    if rng.random() < 0.4:
        opts["partitionWeights"] = {p: rng.choice([1, 2, 3, 10, 100])
                                    for p in pnames if rng.random() < 0.5}
    if rng.random() < 0.3:
        opts["stateStickiness"] = {s: rng.choice([0, 1, 5, 1000]) for s in states if rng.random() < 0.7}
    booster = None
    if rng.random() < 0.4:
        neg = rng.random() < 0.3
        opts["nodeWeights"] = {n: (rng.choice([-3, -2, -1, 0, 1, 2, 3]) if neg else rng.choice([1, 2, 3, 4]))
                               for n in nodes if rng.random() < 0.6}
        if neg and rng.random() < 0.7:
            booster = "cbgt"
    if rng.random() < 0.5:
        # ragged tree: racks of random size, some racks in zones, some nodes unparented
        hier = {}
        racks = ["r%d" % i for i in range(rng.randint(1, 4))]
        zones = ["z%d" % i for i in range(rng.randint(1, 2))]
        for n in nodes + (["ghost"] if rng.random() < 0.2 else []):
            if rng.random() < 0.9:
                hier[n] = rng.choice(racks)
        for rk in racks:
            if rng.random() < 0.8:
                hier[rk] = rng.choice(zones)
        if rng.random() < 0.2:
            hier["emptyrack"] = zones[0]
        opts["nodeHierarchy"] = hier
        rules = {}
        for s in states[1:]:
            if rng.random() < 0.8:
                rl = []
                for _ in range(rng.choice([1, 1, 1, 2])):
                    inc = rng.choice([1, 2, 2, 3])
                    exc = rng.choice([0, 1, 1, 2])
                    rl.append({"includeLevel": inc, "excludeLevel": exc})
                rules[s] = rl
        if rng.random() < 0.15:
            rules[states[0]] = [{"includeLevel": 1, "excludeLevel": 0}]
        opts["hierarchyRules"] = rules
    elif rng.random() < 0.1:
        opts["hierarchyRules"] = {}
    case = {"prevMap": prev, "partitionsToAssign": assign, "aliased": assign is None,
            "nodesAll": nodes, "nodesToRemove": nodes_to_remove, "nodesToAdd": nodes_to_add,
            "model": model, "booster": booster, "seed": seed}
    case.update(opts)
    return case


def random_regular_case(seed, max_parts=60):
    """Inputs whose hierarchy rule cuts the cluster into regions (uniform
    rack/zone trees), the shape the region-chain kernel accepts -- with enough
    variety (existing layouts, weights, removals, tiny regions) to also hit
    its escape path."""
    rng = random.Random(seed)
    rack = rng.choice([1, 2, 3, 4])
    racks_per_zone = rng.choice([2, 2, 3, 4])
    n_zones = rng.choice([2, 2, 3, 5])
    n_nodes = rack * racks_per_zone * n_zones - rng.choice([0, 0, 0, 1, 2])
    nodes = ["n%03d" % i for i in range(n_nodes)]
    hier = {}
    for i, n in enumerate(nodes):
        hier[n] = "r%02d" % (i // rack)
    for r in range((n_nodes + rack - 1) // rack):
        hier["r%02d" % r] = "z%d" % (r // racks_per_zone)
    if rng.random() < 0.5:
        for z in range(n_zones):
            hier["z%d" % z] = "dc"
    k_rep = rng.choice([1, 2, 2, 3])
    model = {"primary": {"priority": 0, "constraints": 1},
             "replica": {"priority": 1, "constraints": k_rep}}
    rule = rng.choice([(2, 1), (2, 1), (2, 0), (1, 0), (3, 1), (2, 2)])
    rules = {"replica": [{"includeLevel": rule[0], "excludeLevel": rule[1]}]}
    if rng.random() < 0.15:
        rules["primary"] = [{"includeLevel": 1, "excludeLevel": 0}]
    n_parts = rng.randint(1, max_parts)
    pnames = [str(i) for i in range(n_parts)]
    mode = rng.choice(["fresh", "aliased", "aliased", "stale"])
    prev, assign = {}, {}
    if mode == "fresh":
        for p in pnames:
            assign[p] = {"name": p, "nodesByState": {}}
    else:
        for p in pnames:
            pick = rng.sample(nodes, min(len(nodes), 1 + k_rep))
            if mode == "aliased" and rng.random() < 0.8:
                # a layout the rule itself could have produced: replicas in the primary's zone
                z = [n for n in nodes if hier[hier[n]] == hier[hier[pick[0]]] and n != pick[0]]
                rng.shuffle(z)
                pick = [pick[0]] + z[:k_rep]
            prev[p] = {"name": p, "nodesByState": {"primary": pick[:1], "replica": pick[1:]}}
        if mode == "stale":
            for p in pnames:
                assign[p] = {"name": p, "nodesByState": {"primary": list(prev[p]["nodesByState"]["primary"])}}
        else:
            assign = None
    to_remove = rng.sample(nodes, rng.choice([0, 0, 1, 2])) if mode != "fresh" or True else []
    if mode == "fresh":
        to_remove = []
    rest = [n for n in nodes if n not in to_remove]
    to_add = rng.sample(rest, rng.choice([0, 0, 2, len(rest)]) if len(rest) > 2 else 0)
    opts = {"modelStateConstraints": None, "partitionWeights": None, "stateStickiness": None,
            "nodeWeights": None, "nodeHierarchy": hier, "hierarchyRules": rules}
    if rng.random() < 0.3:
        opts["partitionWeights"] = {p: rng.choice([1, 2, 3, 10]) for p in pnames if rng.random() < 0.5}
    if rng.random() < 0.3:
        opts["nodeWeights"] = {n: rng.choice([1, 2, 3]) for n in nodes if rng.random() < 0.5}
    case = {"prevMap": prev, "partitionsToAssign": assign, "aliased": assign is None,
            "nodesAll": nodes, "nodesToRemove": to_remove, "nodesToAdd": to_add,
            "model": model, "booster": None, "seed": seed}
    case.update(opts)
    return case


def random_flat_wide_case(seed, k=None):
    """random_case() with a flat model of 3 or 4 copies in the second state (the shapes k_pass_tree<4> takes): no
    hierarchy rules, enough nodes for the copies most of the time."""
    rng = random.Random(seed * 7919 + 13)
    c = random_case(seed, max_nodes=16, max_parts=40)
    k = k or rng.choice([3, 4])
    kind = rng.choice(["pr", "pr", "p", "prx"])
    if kind == "p":
        c["model"] = {"primary": {"priority": 0, "constraints": k}}
    elif kind == "pr":
        c["model"] = {"primary": {"priority": 0, "constraints": 1}, "replica": {"priority": 1, "constraints": k}}
    else:
        c["model"] = {"primary": {"priority": 0, "constraints": rng.choice([1, 2])},
                      "replica": {"priority": 1, "constraints": k}, "zreadonly": {"priority": 2, "constraints": rng.choice([0, 1, 3])}}
    c["hierarchyRules"] = None
    c["modelStateConstraints"] = None
    if c.get("stateStickiness"):
        c["stateStickiness"] = {s: v for s, v in c["stateStickiness"].items() if s in c["model"]}
    states = set(c["model"])

    def trim(m):
        for p in (m or {}).values():
            p["nodesByState"] = {s: l for s, l in p["nodesByState"].items() if s in states or s == "dead"}
    trim(c["prevMap"])
    if not c["aliased"]:
        trim(c["partitionsToAssign"])
    return c
