"""go/blance/intern.go: staticOrder (the static part of partitionSorter's key, plan.go:519-540) cannot be compiled in this image
(no Go toolchain).  This is a line-by-line Python port of BOTH of its branches -- the integer radix path for keys in
[0, 9999999999] with its tie-run re-sort by Name, and the literal string path -- checked against the reference's comparator
taken literally (strings of "%10d" renderings compared as strings) on random names and weights: signs, leading zeros,
negative weights, weights beyond 999999999, twelve-digit names, non-numeric names (round 3's advisor)."""
import random
import re

_INT = re.compile(r"^[+-]?[0-9]+$")


def go_atoi(s):
    """strconv.Atoi: optional sign, decimal digits, nothing else; out of int64 range is an error."""
    if not _INT.match(s):
        return None
    v = int(s)
    return v if -(1 << 63) <= v < (1 << 63) else None


def literal_order(names, weight):
    """plan.go:519-540 as written: (wkey, nkey, Name) compared as strings."""
    def key(i):
        v = go_atoi(names[i])
        nkey = "%10d" % v if (v is not None and v >= 0) else names[i]
        return ("%10d" % (999999999 - weight[i]), nkey, names[i])
    return sorted(range(len(names)), key=key)


def radix_order(idx, keys, max_key):
    """radixOrder of intern.go: a stable LSD counting sort of idx by keys, 11 bits a pass, while (max >> shift) != 0."""
    shift = 0
    while shift < 64 and (max_key >> shift) != 0:
        cnt = [0] * 2049
        for i in idx:
            cnt[((keys[i] >> shift) & 2047) + 1] += 1
        for d in range(2048):
            cnt[d + 1] += cnt[d]
        tmp = [0] * len(idx)
        for i in idx:
            d = (keys[i] >> shift) & 2047
            tmp[cnt[d]] = i
            cnt[d] += 1
        idx = tmp
        shift += 11
    return idx


def go_static_order(names, weight):
    """staticOrder of go/blance/intern.go, branch for branch."""
    P = len(names)
    idx = list(range(P))
    num, wk = [0] * P, [0] * P
    max_num = max_wk = 0
    simple = True
    for i, name in enumerate(names):
        v = go_atoi(name)
        k = 999999999 - weight[i]
        if v is None or v < 0 or v > 9999999999 or k < 0 or k > 9999999999:
            simple = False
            break
        num[i], wk[i] = v, k
        max_num, max_wk = max(max_num, v), max(max_wk, k)
    if simple:
        idx = radix_order(idx, num, max_num)      # name key first, weight key second: the weight key decides
        idx = radix_order(idx, wk, max_wk)
        a = 0
        while a < P:                              # equal keys ("7" and "007"): by Name
            b = a + 1
            while b < P and num[idx[b]] == num[idx[a]] and wk[idx[b]] == wk[idx[a]]:
                b += 1
            if b - a > 1:
                idx[a:b] = sorted(idx[a:b], key=lambda i: names[i])
            a = b
        return idx
    keys = []
    for i, name in enumerate(names):
        v = go_atoi(name)
        nkey = "%10d" % v if (v is not None and v >= 0) else name
        keys.append(("%10d" % (999999999 - weight[i]), nkey))
    return sorted(idx, key=lambda i: (keys[i][0], keys[i][1], names[i]))


def _names(rnd, n, numeric_only):
    pool = set()
    while len(pool) < n:
        r = rnd.random()
        if r < 0.55:
            s = str(rnd.randrange(0, rnd.choice([20, 1000, 10 ** 6, 10 ** 10])))
        elif r < 0.7:
            s = "0" * rnd.randrange(1, 4) + str(rnd.randrange(0, 50))          # "007"
        elif r < 0.8:
            s = "+" + str(rnd.randrange(0, 50))                                  # "+5": Atoi takes the sign
        elif numeric_only:
            s = str(rnd.randrange(0, 9999999999 + 1))
        elif r < 0.85:
            s = rnd.choice(["-0", "-3", "-12"])                                   # Atoi ok, -0 >= 0, -3 < 0 -> raw name
        elif r < 0.9:
            s = str(rnd.randrange(10 ** 10, 10 ** 12))                           # beyond the integer path
        else:
            s = rnd.choice(["p", "part-", "vb_", " ", "1e3", "0x1", "7 ", "٣"]) + str(rnd.randrange(0, 30))
        pool.add(s)
    out = list(pool)
    rnd.shuffle(out)
    return out


def test_static_order_port_equals_the_literal_comparator():
    rnd = random.Random(20260924)
    simple_runs = fallback_runs = 0
    for trial in range(400):
        numeric_only = trial % 2 == 0
        names = _names(rnd, rnd.randrange(1, 60), numeric_only)
        if numeric_only:
            weight = [rnd.choice([1, 1, 1, 2, 10, 1000, 999999999, 0]) for _ in names]
        else:
            weight = [rnd.choice([1, 1, 5, 1000, 0, -1, -250, 999999999, 1000000000, 2 ** 31 - 1]) for _ in names]
        want = literal_order(names, weight)
        got = go_static_order(names, weight)
        assert got == want, (names, weight)
        is_simple = all((go_atoi(n) is not None and 0 <= go_atoi(n) <= 9999999999) for n in names) and \
            all(0 <= 999999999 - w <= 9999999999 for w in weight)
        simple_runs += is_simple
        fallback_runs += not is_simple
    assert simple_runs > 50 and fallback_runs > 50


def test_go_text_still_reads_like_the_port():
    """The port mirrors these lines of go/blance/intern.go; if they change, the port has to follow."""
    import os
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "go", "blance", "intern.go")).read()
    for line in ("if err != nil || v < 0 || int64(v) > 9999999999 || k < 0 || k > 9999999999 {",
                 "idx = radixOrder(idx, num, maxNum) // name key first, weight key second: the weight key decides",
                 "idx = radixOrder(idx, wk, maxWk)",
                 "for shift := uint(0); shift < 64 && (max>>shift) != 0; shift += 11 {",
                 "cnt[((key[i]>>shift)&2047)+1]++",
                 "for b < P && num[idx[b]] == num[idx[a]] && wk[idx[b]] == wk[idx[a]] {",
                 'keys[i] = key{fmt.Sprintf("%10d", 999999999-int(weight[i])), nkey}'):
        assert line in src, line
