"""ORACLE (test infrastructure only) -- literal CPU restatement of couchbase/blance's
CalcPartitionMoves (moves.go:41-136), the consumer of the planner's output
(SURVEY.md section 8 f-1).  Follows the reference line by line on Python lists of
strings; pinned by the reference's own tables (tests/golden/moves_cases.json,
transcribed from moves_test.go by tools/extract_moves_golden.py)."""


def flatten_nodes_by_state(a):                    # plan.go:425-431 (order irrelevant: used as a set)
    rv = []
    for b in (a or {}).values():
        rv.extend(b or [])
    return rv


def strings_remove_strings(arr, remove):          # misc.go:27-36
    rm = set(remove or [])
    return [s for s in (arr or []) if s not in rm]


def strings_intersect_strings(a, b):              # misc.go:40-51
    bm = set(b or [])
    seen, rv = set(), []
    for s in (a or []):
        if s in bm and s not in seen:
            seen.add(s)
            rv.append(s)
    return rv


def find_state_changes(beg_idx, end_idx, state, states, beg, end):   # moves.go:121-136
    rv = []
    for node in (end or {}).get(state) or []:
        for i in range(beg_idx, end_idx):
            for n in (beg or {}).get(states[i]) or []:
                if n == node:
                    rv.append(node)
    return rv


def calc_partition_moves(states, beg, end, favor_min_nodes):         # moves.go:41-119
    """Returns [(node, state, op)]."""
    moves, seen = [], set()
    beg, end = beg or {}, end or {}

    def add_moves(nodes, state, op):              # moves.go:51-58
        for node in nodes:
            if node not in seen:
                seen.add(node)
                moves.append((node, state, op))

    beg_nodes = flatten_nodes_by_state(beg)
    end_nodes = flatten_nodes_by_state(end)
    adds = strings_remove_strings(end_nodes, beg_nodes)
    dels = strings_remove_strings(beg_nodes, end_nodes)
    n = len(states)
    if not favor_min_nodes:
        for si, state in enumerate(states):
            add_moves(find_state_changes(si + 1, n, state, states, beg, end), state, "promote")
            add_moves(find_state_changes(0, si, state, states, beg, end), state, "demote")
            add_moves(strings_intersect_strings(strings_remove_strings(end.get(state), beg.get(state)), adds),
                      state, "add")
            add_moves(strings_intersect_strings(strings_remove_strings(beg.get(state), end.get(state)), dels),
                      "", "del")
    else:
        for si in range(n - 1, -1, -1):
            state = states[si]
            add_moves(strings_intersect_strings(strings_remove_strings(beg.get(state), end.get(state)), dels),
                      "", "del")
            add_moves(find_state_changes(0, si, state, states, beg, end), state, "demote")
            add_moves(find_state_changes(si + 1, n, state, states, beg, end), state, "promote")
            add_moves(strings_intersect_strings(strings_remove_strings(end.get(state), beg.get(state)), adds),
                      state, "add")
    return moves
