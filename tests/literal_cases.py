"""API-level cases of a few thousand partitions whose expected results come from the LITERAL string-keyed oracle
(oracle/blance_ref.py: no interning, no ids) -- the one route to a result that shares no code with blance_amd/problem.py.
The id-based C oracle and the device both sit behind problem.py's interning, so an interning bug would be common to them at
sizes the literal oracle is never run on; tests/golden/literal_oracle_cases.json (made by
tests/tools/make_literal_fixtures.py, minutes of pure Python) pins them at 4,096 partitions."""
import hashlib
import json

from blance_amd import synth

P_LITERAL, N_LITERAL = 4096, 256


def canonical_sha(plan_map):
    """sha256 of a result map {name: {"name", "nodesByState": {state: [nodes]}}} as canonical JSON."""
    return hashlib.sha256(json.dumps(plan_map, sort_keys=True, separators=(",", ":")).encode()).hexdigest()


def warnings_sha(w):
    return hashlib.sha256(json.dumps(w or {}, sort_keys=True, separators=(",", ":")).encode()).hexdigest()


def node_weights_1124(N, width=4):
    """NodeWeights in {1, 1, 2, 4} by a fixed hash of the node index (config 5's law on config 3's tree)."""
    pick = [1, 1, 2, 4]
    return {("n%0" + str(width) + "d") % i: pick[(i * 2654435761 >> 7) % 4] for i in range(N)}


def case_named_weighted(P=P_LITERAL, N=N_LITERAL):
    """Regime (b) of bench.py's general_regime: config 3's model, tree and rule; scrambled non-numeric partition names
    (plan.go:525-528: Atoi fails, the raw name is the key) and Zipf partition weights."""
    return synth.config3_named_weighted_case(P, N)


def case_node_weights(P=P_LITERAL, N=N_LITERAL):
    """Config 3 with NodeWeights in {1, 1, 2, 4} (plan.go:675-679: the score divided by the weight)."""
    c = synth.config_case(3, P=P, N=N)
    c["nodeWeights"] = node_weights_1124(N)
    return c


def case_rebalance(plan_map, P=P_LITERAL, N=N_LITERAL, every=10, which=3):
    """Regime (a): prevMap = partitionsToAssign = a plan of config 3, every tenth node leaves, nodesToAdd = nil."""
    c = synth.config_case(3, P=P, N=N)
    c["prevMap"] = plan_map
    c["partitionsToAssign"] = plan_map
    c["aliased"] = True
    c["nodesToRemove"] = [n for i, n in enumerate(c["nodesAll"]) if i % every == which]
    c["nodesToAdd"] = None
    return c
