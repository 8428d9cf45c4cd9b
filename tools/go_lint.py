#!/usr/bin/env python3
"""A stand-in for `go vet` where no Go toolchain exists: a tokenizer for Go and the compile errors that blind edits of
go/blance/*.go are most likely to introduce --

  * brackets that do not nest ((), [], {} paired by a stack, outside strings and comments);
  * an imported package that is never used, a standard package used without its import;
  * a local variable that is declared (`:=`, `var`, range / type-switch clauses) and never mentioned again
    ("declared and not used");
  * `x := ...` with a single name that the same block has already declared ("no new variables on left side of :=").

It does not type-check.  `python tools/go_lint.py [files...]`, exit status 1 when something is found."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STD = {"fmt", "sort", "strconv", "strings", "errors", "unsafe", "sync", "math", "reflect", "testing", "runtime", "os",
       "bytes", "time", "json"}
KEYWORDS = {"break", "case", "chan", "const", "continue", "default", "defer", "else", "fallthrough", "for", "func", "go",
            "goto", "if", "import", "interface", "map", "package", "range", "return", "select", "struct", "switch", "type",
            "var"}
TOKEN = re.compile(r"""
    (?P<ws>\s+)
  | (?P<lc>//[^\n]*)
  | (?P<bc>/\*.*?\*/)
  | (?P<raw>`[^`]*`)
  | (?P<str>"(?:[^"\\\n]|\\.)*")
  | (?P<rune>'(?:[^'\\\n]|\\.)+')
  | (?P<id>[A-Za-z_][A-Za-z_0-9]*)
  | (?P<num>\d[\dA-Za-z_.]*)
  | (?P<op>:=|\.\.\.|<<=|>>=|&\^=|&\^|<-|\+\+|--|&&|\|\||[=!<>+\-*/%&|^]=|<<|>>|[(){}\[\];,.:=+\-*/%&|^<>!~])
""", re.S | re.X)


def tokenize(text, name):
    toks, pos, line = [], 0, 1
    while pos < len(text):
        m = TOKEN.match(text, pos)
        if not m:
            raise SyntaxError("%s:%d: cannot tokenize %r" % (name, line, text[pos:pos + 20]))
        kind = m.lastgroup
        s = m.group()
        if kind not in ("ws", "lc", "bc"):
            toks.append((kind, s, line))
        # Go's automatic semicolons: a newline after an identifier, literal, `)`, `]`, `}`, ++, --, return ... ends a statement
        if kind in ("ws", "lc", "bc") and "\n" in s and toks:
            k, v, _ = toks[-1]
            if k in ("id", "num", "str", "raw", "rune") or v in (")", "]", "}", "++", "--"):
                if not (k == "id" and v in KEYWORDS - {"break", "continue", "fallthrough", "return"}):
                    toks.append(("op", ";", line))
        line += s.count("\n")
        pos = m.end()
    return toks


def check_brackets(toks, name, out):
    pairs = {")": "(", "]": "[", "}": "{"}
    stack = []
    for kind, s, line in toks:
        if kind != "op":
            continue
        if s in "([{":
            stack.append((s, line))
        elif s in pairs:
            if not stack or stack[-1][0] != pairs[s]:
                out.append("%s:%d: %r does not close %s" % (name, line, s, ("%r of line %d" % stack[-1]) if stack else "anything"))
                return
            stack.pop()
    for s, line in stack:
        out.append("%s:%d: %r is never closed" % (name, line, s))


def check_imports(toks, name, out):
    imported, i = {}, 0
    while i < len(toks):
        if toks[i][1] == "import":
            j = i + 1
            if toks[j][1] == "(":
                j += 1
                while toks[j][1] != ")":
                    if toks[j][0] == "str":
                        imported[toks[j][1].strip('"').split("/")[-1]] = toks[j][2]
                    j += 1
            elif toks[j][0] == "str":
                imported[toks[j][1].strip('"').split("/")[-1]] = toks[j][2]
            i = j
        i += 1
    used = set()
    for a, b in zip(toks, toks[1:]):
        if a[0] == "id" and b[1] == "." and (a[1] in STD or a[1] in imported or a[1] == "C"):
            used.add(a[1])
    # a use is `pkg.` where pkg is not itself a selector (x.fmt.y) and not a local name; locals named like packages do not occur here
    for pkg, line in imported.items():
        if pkg not in used:
            out.append("%s:%d: %r imported and not used" % (name, line, pkg))
    for pkg in used:
        if pkg not in imported:
            out.append("%s: package %r used without import" % (name, pkg))


def functions(toks):
    """(name, index of the body's `{`, index of its `}`) of every top-level func with a body."""
    depth, i, res = 0, 0, []
    while i < len(toks):
        s = toks[i][1]
        if s in "{([" and toks[i][0] == "op":
            depth += 1
        elif s in "})]" and toks[i][0] == "op":
            depth -= 1
        elif s == "func" and depth == 0:
            j, d = i + 1, 0
            # skip receiver, name, parameters, results up to the body's `{` at bracket depth 0
            while j < len(toks):
                t = toks[j][1]
                if toks[j][0] == "op" and t in "([":
                    d += 1
                elif toks[j][0] == "op" and t in ")]":
                    d -= 1
                elif t == "{" and d == 0:
                    # `interface{}` / `struct{` in the signature: their braces follow those keywords
                    if toks[j - 1][1] in ("interface", "struct"):
                        k, dd = j, 0
                        while True:
                            if toks[k][1] == "{":
                                dd += 1
                            elif toks[k][1] == "}":
                                dd -= 1
                                if dd == 0:
                                    break
                            k += 1
                        j = k + 1
                        continue
                    break
                elif t == ";" and d == 0:
                    j = None
                    break
                j += 1
            if j is None or j >= len(toks):
                i += 1
                continue
            k, dd = j, 0
            while True:
                if toks[k][0] == "op" and toks[k][1] == "{":
                    dd += 1
                elif toks[k][0] == "op" and toks[k][1] == "}":
                    dd -= 1
                    if dd == 0:
                        break
                k += 1
            fname = next((t[1] for t in toks[i + 1:j] if t[0] == "id" and t[1] not in KEYWORDS), "func")
            res.append((fname, j, k))
            i = k
        i += 1
    return res


def declared_names(toks, lo, hi):
    """(name, token index, is_single_short_decl) for local declarations between lo and hi."""
    decls = []
    i = lo
    while i < hi:
        kind, s, _ = toks[i]
        if s == ":=":
            # names to the left, back to the start of the statement / clause
            j, names = i - 1, []
            while j >= lo:
                if toks[j][0] == "id" and toks[j][1] not in KEYWORDS:
                    names.append((toks[j][1], j))
                    j -= 1
                    if j >= lo and toks[j][1] == ",":
                        j -= 1
                        continue
                break
            single = len(names) == 1
            for n, at in names:
                if n != "_":
                    decls.append((n, at, single))
        elif s == "var" and kind == "id":
            j = i + 1
            while j < hi and toks[j][0] == "id" and toks[j][1] not in KEYWORDS:
                if toks[j][1] != "_":
                    decls.append((toks[j][1], j, False))
                j += 1
                if j < hi and toks[j][1] == ",":
                    j += 1
                    continue
                break
        i += 1
    return decls


def check_function(toks, name, fname, lo, hi, out):
    decls = declared_names(toks, lo, hi)
    decl_at = {at for _, at, _ in decls}
    for n, at, _ in decls:
        used = False
        for j in range(lo, hi):
            if j in decl_at and toks[j][1] == n and j != at:
                continue                       # a later redeclaration in another scope is not a use
            if j != at and toks[j][0] == "id" and toks[j][1] == n:
                if toks[j - 1][1] == "." and toks[j - 1][0] == "op":
                    continue                   # a field or method of that name
                if toks[j + 1][1] == ":" and toks[j - 1][1] in ("{", ","):
                    continue                   # a key of a composite literal
                used = True
                break
        if not used:
            out.append("%s:%d: %s: %r declared and not used" % (name, toks[at][2], fname, n))
    # `x := ...` twice for the same single name directly in the same block
    scopes = [set()]
    single_at = {at: n for n, at, single in decls if single}
    for j in range(lo, hi + 1):
        s = toks[j][1]
        if toks[j][0] == "op" and s == "{":
            scopes.append(set())
        elif toks[j][0] == "op" and s == "}":
            scopes.pop()
        elif j in single_at:
            # the statement's own block: headers of if / for / switch open an implicit scope -- skip those
            k = j - 1
            header = False
            while k >= lo and toks[k][1] not in (";", "{", "}"):
                if toks[k][1] in ("if", "for", "switch", "case", "select"):
                    header = True
                k -= 1
            if header:
                continue
            n = single_at[j]
            if n in scopes[-1]:
                out.append("%s:%d: %s: no new variables on left side of := (%r)" % (name, toks[j][2], fname, n))
            scopes[-1].add(n)


def lint(path):
    name = os.path.relpath(path, ROOT)
    out = []
    toks = tokenize(open(path).read(), name)
    check_brackets(toks, name, out)
    if out:
        return out
    check_imports(toks, name, out)
    for fname, lo, hi in functions(toks):
        check_function(toks, name, fname, lo, hi, out)
    return out


def main():
    files = sys.argv[1:] or sorted(os.path.join(ROOT, "go", "blance", f) for f in os.listdir(os.path.join(ROOT, "go", "blance"))
                                    if f.endswith(".go"))
    bad = []
    for f in files:
        bad += lint(f)
    for line in bad:
        print(line)
    print("go lint: %d file(s), %s" % (len(files), "clean" if not bad else "%d finding(s)" % len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
