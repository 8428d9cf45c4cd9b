#!/usr/bin/env python3
"""Transcribe the reference's planner golden tables into JSON fixtures.

Reads (never copies) the Go test sources of couchbase/blance under
/root/reference and emits machine-readable fixtures under tests/golden/:

  planner_cases.json   the 69 end-to-end PlanNextMap cases
                       (plan_test.go:392-2863, control_test.go:18-416)
  helper_cases.json    helper-function tables (plan_test.go:21-390,
                       misc_test.go:18-89)

The Go literals are a small regular subset of the language, parsed here by a
tiny tokenizer + composite-literal parser.  The "Vis" picture DSL
(plan_test.go:1663-1715) is decoded here, so the fixtures carry plain
PartitionMaps.  This script only runs in the build container (the reference
is not present on the GPU box); its output is committed.

Usage: python tools/extract_golden.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import json
import os
import re
import sys

# --------------------------------------------------------------------------
# Tokenizer
# --------------------------------------------------------------------------

TOKEN_RE = re.compile(
    r"""
    (?P<ws>\s+)
  | (?P<lcomment>//[^\n]*)
  | (?P<bcomment>/\*.*?\*/)
  | (?P<string>"(?:\\.|[^"\\])*")
  | (?P<raw>`[^`]*`)
  | (?P<char>'(?:\\.|[^'\\])')
  | (?P<number>-?\d+(?:\.\d+)?)
  | (?P<ident>[A-Za-z_][A-Za-z0-9_]*)
  | (?P<op>:=|==|!=|<=|>=|&&|\|\||\+\+|--|\+=|\.\.\.|[{}()\[\],:;.*&=<>!+\-/%|])
    """,
    re.X | re.S,
)


class Tok:
    __slots__ = ("kind", "val", "line")

    def __init__(self, kind, val, line):
        self.kind, self.val, self.line = kind, val, line

    def __repr__(self):
        return "Tok(%s,%r,@%d)" % (self.kind, self.val, self.line)


def tokenize(src):
    toks = []
    pos, line = 0, 1
    while pos < len(src):
        m = TOKEN_RE.match(src, pos)
        if not m:
            raise SyntaxError("cannot tokenize at line %d: %r" % (line, src[pos:pos + 30]))
        kind = m.lastgroup
        text = m.group()
        if kind == "string":
            toks.append(Tok("string", json.loads(text), line))
        elif kind == "raw":
            toks.append(Tok("string", text[1:-1], line))
        elif kind in ("number", "ident", "op", "char"):
            toks.append(Tok(kind, text, line))
        line += text.count("\n")
        pos = m.end()
    return toks


# --------------------------------------------------------------------------
# Types (just enough to resolve elided composite literal types)
# --------------------------------------------------------------------------

STRING = ("basic", "string")
INT = ("basic", "int")
BOOL = ("basic", "bool")


def T_map(k, v):
    return ("map", k, v)


def T_slice(t):
    return ("slice", t)


def T_ptr(t):
    return ("ptr", t)


NAMED = {
    "string": STRING,
    "int": INT,
    "bool": BOOL,
    "Partition": ("struct", {"Name": STRING, "NodesByState": T_map(STRING, T_slice(STRING))}),
    "PartitionModelState": ("struct", {"Priority": INT, "Constraints": INT}),
    "HierarchyRule": ("struct", {"IncludeLevel": INT, "ExcludeLevel": INT}),
}
NAMED["PartitionMap"] = T_map(STRING, T_ptr(("named", "Partition")))
NAMED["PartitionModel"] = T_map(STRING, T_ptr(("named", "PartitionModelState")))
NAMED["HierarchyRules"] = T_map(STRING, T_slice(T_ptr(("named", "HierarchyRule"))))
NAMED["PlanNextMapOptions"] = ("struct", {
    "ModelStateConstraints": T_map(STRING, INT),
    "PartitionWeights": T_map(STRING, INT),
    "StateStickiness": T_map(STRING, INT),
    "NodeWeights": T_map(STRING, INT),
    "NodeHierarchy": T_map(STRING, STRING),
    "HierarchyRules": ("named", "HierarchyRules"),
})
NAMED["VisTestCase"] = ("struct", {
    "Ignore": BOOL, "About": STRING, "FromTo": T_slice(T_slice(STRING)),
    "FromToPriority": BOOL, "Nodes": T_slice(STRING), "NodesToRemove": T_slice(STRING),
    "NodesToAdd": T_slice(STRING), "Model": ("named", "PartitionModel"),
    "ModelStateConstraints": T_map(STRING, INT), "PartitionWeights": T_map(STRING, INT),
    "StateStickiness": T_map(STRING, INT), "NodeWeights": T_map(STRING, INT),
    "NodeHierarchy": T_map(STRING, STRING), "HierarchyRules": ("named", "HierarchyRules"),
    "expNumWarnings": INT,
})


def resolve(t):
    while t is not None and t[0] == "named":
        t = NAMED[t[1]]
    return t


class Parser:
    def __init__(self, toks, env=None):
        self.toks = toks
        self.i = 0
        self.env = env if env is not None else {}

    def peek(self, k=0):
        return self.toks[self.i + k] if self.i + k < len(self.toks) else Tok("eof", None, -1)

    def next(self):
        t = self.peek()
        self.i += 1
        return t

    def expect(self, val):
        t = self.next()
        if t.val != val:
            raise SyntaxError("expected %r got %r at line %d" % (val, t.val, t.line))
        return t

    # ---- types
    def parse_type(self):
        t = self.peek()
        if t.val == "[":
            self.next()
            self.expect("]")
            return T_slice(self.parse_type())
        if t.val == "*":
            self.next()
            return T_ptr(self.parse_type())
        if t.val == "map":
            self.next()
            self.expect("[")
            k = self.parse_type()
            self.expect("]")
            return T_map(k, self.parse_type())
        if t.val == "struct":
            self.next()
            self.expect("{")
            fields = {}
            while self.peek().val != "}":
                names = [self.next().val]
                while self.peek().val == ",":
                    self.next()
                    names.append(self.next().val)
                ft = self.parse_type()
                for n in names:
                    fields[n] = ft
                if self.peek().val == ";":
                    self.next()
            self.expect("}")
            return ("struct", fields)
        if t.kind == "ident":
            self.next()
            if t.val in NAMED:
                return ("named", t.val)
            raise SyntaxError("unknown type %r at line %d" % (t.val, t.line))
        raise SyntaxError("bad type at line %d: %r" % (t.line, t.val))

    def looks_like_type(self):
        t = self.peek()
        if t.val in ("[", "map", "struct", "*"):
            # '[' could start an index expr, but never at expression start here
            return True
        return t.kind == "ident" and t.val in NAMED and self.peek(1).val == "{"

    # ---- expressions
    def parse_expr(self, want=None):
        t = self.peek()
        if t.val == "&":
            self.next()
            return self.parse_expr(want[1] if want and want[0] == "ptr" else want)
        if t.val == "{":  # elided type
            et = want
            if et is not None and et[0] == "ptr":
                et = et[1]
            return self.parse_composite(et)
        if self.looks_like_type():
            ty = self.parse_type()
            if self.peek().val == "{":
                return self.parse_composite(ty)
            if self.peek().val == "(":  # conversion e.g. []string(nil)
                self.next()
                v = self.parse_expr(ty)
                self.expect(")")
                return v
            raise SyntaxError("type without literal at line %d" % t.line)
        if t.kind == "string":
            self.next()
            return t.val
        if t.kind == "number":
            self.next()
            return int(t.val)
        if t.val == "-" and self.peek(1).kind == "number":
            self.next()
            return -int(self.next().val)
        if t.kind == "ident":
            self.next()
            if t.val == "nil":
                return None
            if t.val == "true":
                return True
            if t.val == "false":
                return False
            if t.val in self.env:
                return json.loads(json.dumps(self.env[t.val]))  # deep copy
            raise SyntaxError("unknown identifier %r at line %d" % (t.val, t.line))
        raise SyntaxError("bad expression at line %d: %r" % (t.line, t.val))

    def parse_composite(self, ty):
        rt = resolve(ty)
        start = self.expect("{")
        if rt is None:
            raise SyntaxError("composite literal without type at line %d" % start.line)
        kind = rt[0]
        if kind == "slice":
            out = []
            while self.peek().val != "}":
                out.append(self.parse_expr(rt[1]))
                if self.peek().val == ",":
                    self.next()
            self.expect("}")
            return out
        if kind == "map":
            out = {}
            while self.peek().val != "}":
                k = self.parse_expr(rt[1])
                self.expect(":")
                out[k] = self.parse_expr(rt[2])
                if self.peek().val == ",":
                    self.next()
            self.expect("}")
            return out
        if kind == "struct":
            out = {"__line__": start.line}
            keyed = self.peek().kind == "ident" and self.peek(1).val == ":"
            if not keyed:  # positional struct literal: fields in declaration order
                for name, ft in rt[1].items():
                    if self.peek().val == "}":
                        break
                    out[name] = self.parse_expr(ft)
                    if self.peek().val == ",":
                        self.next()
                self.expect("}")
                return out
            while self.peek().val != "}":
                name = self.next().val
                self.expect(":")
                if name not in rt[1]:
                    raise SyntaxError("unknown field %r at line %d" % (name, start.line))
                out[name] = self.parse_expr(rt[1][name])
                if self.peek().val == ",":
                    self.next()
            self.expect("}")
            return out
        raise SyntaxError("cannot build literal of %r" % (rt,))


# --------------------------------------------------------------------------
# Locating things in a test file
# --------------------------------------------------------------------------

def split_funcs(toks):
    """Return {funcName: (startIdx, endIdx)} over top-level `func Name(`."""
    out = {}
    i = 0
    while i < len(toks):
        if toks[i].val == "func" and toks[i + 1].kind == "ident" and toks[i + 2].val == "(":
            name = toks[i + 1].val
            j = i
            while toks[j].val != "{":
                j += 1
            depth, k = 0, j
            while True:
                if toks[k].val == "{":
                    depth += 1
                elif toks[k].val == "}":
                    depth -= 1
                    if depth == 0:
                        break
                k += 1
            out[name] = (j + 1, k)
            i = k + 1
        else:
            i += 1
    return out


def parse_assignments(toks, lo, hi, stop_names=()):
    """Parse leading `name := <literal>` statements of a function body into an
    env; returns env.  Statements that are not literal short-var-decls are
    skipped by brace matching."""
    env = {}
    p = Parser(toks, env)
    p.i = lo
    while p.i < hi:
        t = p.peek()
        if t.kind == "ident" and p.peek(1).val == ":=":
            name = t.val
            save = p.i
            p.i += 2
            try:
                if p.looks_like_type() or p.peek().val in ("&",):
                    env[name] = p.parse_expr()
                    continue
            except SyntaxError:
                pass
            p.i = save + 1
        else:
            p.i += 1
    return env


def strip_lines(v):
    if isinstance(v, dict):
        return {k: strip_lines(x) for k, x in v.items() if k != "__line__"}
    if isinstance(v, list):
        return [strip_lines(x) for x in v]
    return v


def norm_pmap(pm):
    """Go PartitionMap literal -> {"name": {"name":.., "nodesByState": {...}}}."""
    if pm is None:
        return None
    out = {}
    for k, p in pm.items():
        out[k] = {"name": p.get("Name", ""), "nodesByState": p.get("NodesByState")}
    return out


def norm_model(m):
    if m is None:
        return None
    return {k: {"priority": v.get("Priority", 0), "constraints": v.get("Constraints", 0)}
            for k, v in m.items()}


def norm_rules(r):
    if r is None:
        return None
    return {k: [{"includeLevel": x.get("IncludeLevel", 0), "excludeLevel": x.get("ExcludeLevel", 0)}
                for x in v] for k, v in r.items()}


def case_common(c):
    return {
        "nodesAll": c.get("Nodes"),
        "nodesToRemove": c.get("NodesToRemove"),
        "nodesToAdd": c.get("NodesToAdd"),
        "model": norm_model(c.get("Model")),
        "modelStateConstraints": c.get("ModelStateConstraints"),
        "partitionWeights": c.get("PartitionWeights"),
        "stateStickiness": c.get("StateStickiness"),
        "nodeWeights": c.get("NodeWeights"),
        "nodeHierarchy": c.get("NodeHierarchy"),
        "hierarchyRules": norm_rules(c.get("HierarchyRules")),
        "booster": None,
    }


def decode_vis(c):
    """plan_test.go:1663-1715 — decode the from/to pictures.  Go's sort.Sort
    on these <=12-element rows is an insertion sort, i.e. stable."""
    state_names = {"m": "primary", "s": "replica"}
    cell = 2 if c.get("FromToPriority") else 1
    prev, exp = {}, {}
    for i, (frm, to) in enumerate(c["FromTo"]):
        name = "%03d" % i
        for pic, dst in ((frm, prev), (to, exp)):
            row = [(pic[j:j + cell], chr(97 + j // cell)) for j in range(0, len(pic), cell)]
            row.sort(key=lambda e: e[0])  # stable
            nbs = {}
            for entry, node in row:
                st = state_names.get(entry[0:1], "")
                if st:
                    nbs.setdefault(st, []).append(node)
            dst[name] = {"name": name, "nodesByState": nbs}
    return prev, exp


def extract_plan_tests(path):
    src = open(path).read()
    toks = tokenize(src)
    funcs = split_funcs(toks)
    cases = []

    # ---- TestPlanNextMap: anonymous struct table
    lo, hi = funcs["TestPlanNextMap"]
    p = Parser(toks)
    p.i = lo
    assert p.next().val == "tests" and p.next().val == ":="
    ty = p.parse_type()
    table = p.parse_composite(ty)
    for idx, c in enumerate(table):
        d = {"suite": "TestPlanNextMap", "index": idx, "about": c.get("About", ""),
             "source": "plan_test.go:%d" % c["__line__"]}
        d["prevMap"] = norm_pmap(strip_lines(c.get("PrevMap")))
        d["partitionsToAssign"] = norm_pmap(strip_lines(c.get("PartitionsToAssign")))
        d["aliased"] = False
        d.update(case_common(strip_lines(c)))
        d["exp"] = norm_pmap(strip_lines(c.get("exp")))
        d["expNumWarnings"] = c.get("expNumWarnings", 0)
        d["warningCountMode"] = "messages"      # plan_test.go:1599-1608
        cases.append(d)

    # ---- Vis suites
    for fn in ("TestPlanNextMapVis", "TestPlanNextMapHierarchy", "TestMultiPrimary",
               "Test2Replicas", "TestPlanNextMapHierarchyMultiRackFailureCases"):
        lo, hi = funcs[fn]
        # env: leading `x := literal` up to `tests :=`
        j = lo
        while not (toks[j].val == "tests" and toks[j + 1].val == ":="):
            j += 1
        env = parse_assignments(toks, lo, j)
        p = Parser(toks, env)
        p.i = j + 2
        ty = p.parse_type()
        table = p.parse_composite(ty)
        for idx, c in enumerate(table):
            line = c["__line__"]
            c = strip_lines(c)
            prev, exp = decode_vis(c)
            d = {"suite": fn, "index": idx, "about": c.get("About", ""),
                 "source": "plan_test.go:%d" % line,
                 "ignored": bool(c.get("Ignore", False))}
            d["prevMap"] = prev
            d["partitionsToAssign"] = None
            d["aliased"] = True                 # plan_test.go:1716-1718
            d.update(case_common(c))
            d["exp"] = exp
            d["expNumWarnings"] = c.get("expNumWarnings", 0)
            d["warningCountMode"] = "partitions"  # plan_test.go:1738
            cases.append(d)
    return cases


def extract_control_tests(path):
    src = open(path).read()
    toks = tokenize(src)
    funcs = split_funcs(toks)
    cases = []
    for n in (1, 2, 3, 4):
        fn = "TestControlCase%d" % n
        lo, hi = funcs[fn]
        env = parse_assignments(toks, lo, hi)
        # find the PlanNextMapEx( call
        j = lo
        while not (toks[j].val == "PlanNextMapEx" and toks[j + 1].val == "("):
            j += 1
        line = toks[j].line
        p = Parser(toks, env)
        p.i = j + 2
        args = []
        wants = [("named", "PartitionMap"), ("named", "PartitionMap"), T_slice(STRING),
                 T_slice(STRING), T_slice(STRING), ("named", "PartitionModel"),
                 ("named", "PlanNextMapOptions")]
        for w in wants:
            args.append(strip_lines(p.parse_expr(w)))
            if p.peek().val == ",":
                p.next()
        p.expect(")")
        opts = args[6] or {}
        c = {"Nodes": args[2], "NodesToRemove": args[3], "NodesToAdd": args[4], "Model": args[5]}
        c.update(opts)
        d = {"suite": "TestControlCase", "index": n - 1, "about": fn,
             "source": "control_test.go:%d" % line}
        d["prevMap"] = norm_pmap(args[0])
        d["partitionsToAssign"] = norm_pmap(args[1])
        d["aliased"] = False
        d.update(case_common(c))
        d["booster"] = "cbgt"                   # control_test.go:19-26
        d["exp"] = norm_pmap(strip_lines(env["expect"]))
        d["expNumWarnings"] = 0                 # `if len(warnings) > 0 { t.Errorf`
        d["warningCountMode"] = "partitions"
        cases.append(d)
    return cases


# --------------------------------------------------------------------------
# Helper tables (anonymous struct tables with ad-hoc field types)
# --------------------------------------------------------------------------

def extract_table(toks, funcs, fn):
    lo, hi = funcs[fn]
    j = lo
    while j < hi and not (toks[j].val == "tests" and toks[j + 1].val == ":="):
        j += 1
    if j >= hi:
        raise SyntaxError("%s is not table driven" % fn)
    env = parse_assignments(toks, lo, j)
    p = Parser(toks, env)
    p.i = j + 2
    ty = p.parse_type()
    return [strip_lines(c) for c in p.parse_composite(ty)]


def extract_helpers(ref):
    out = {}
    toks = tokenize(open(os.path.join(ref, "plan_test.go")).read())
    funcs = split_funcs(toks)
    for fn in ("TestFlattenNodesByState", "TestRemoveNodesFromNodesByState", "TestStateNameSorter",
               "TestCountStateNodes", "TestPartitionMapToArrayCopy", "TestFindAncestor",
               "TestFindLeaves", "TestMapParentsToMapChildren"):
        out[fn] = extract_table(toks, funcs, fn)
    toks = tokenize(open(os.path.join(ref, "misc_test.go")).read())
    funcs = split_funcs(toks)
    for fn in ("TestStringsRemoveStrings", "TestStringsIntersectStrings"):
        out[fn] = extract_table(toks, funcs, fn)
    # misc_test.go:18-32 is three inline assertions, not a table
    out["TestStringsToMap"] = [
        {"s": [], "exp": {}, "source": "misc_test.go:19-23"},
        {"s": ["a"], "exp": {"a": True}, "source": "misc_test.go:24-27"},
        {"s": ["a", "b", "a"], "exp": {"a": True, "b": True}, "source": "misc_test.go:28-31"},
    ]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    cases = extract_plan_tests(os.path.join(a.ref, "plan_test.go"))
    cases += extract_control_tests(os.path.join(a.ref, "control_test.go"))
    with open(os.path.join(a.out, "planner_cases.json"), "w") as f:
        json.dump({"generator": "tools/extract_golden.py",
                   "reference": "couchbase/blance plan_test.go, control_test.go",
                   "cases": cases}, f, indent=1, sort_keys=True)
    helpers = extract_helpers(a.ref)
    with open(os.path.join(a.out, "helper_cases.json"), "w") as f:
        json.dump({"generator": "tools/extract_golden.py", "tables": helpers}, f, indent=1,
                  sort_keys=True)
    active = [c for c in cases if not c.get("ignored")]
    print("planner cases: %d total, %d active" % (len(cases), len(active)))
    for k, v in helpers.items():
        print("helper table %s: %d rows" % (k, len(v)))


if __name__ == "__main__":
    sys.exit(main())
