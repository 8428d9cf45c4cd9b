"""CPU restatement of encoding/json over the reference's PartitionMap
(api.go:24-36: map[string]*Partition, struct tags `name` / `nodesByState`).
TEST INFRASTRUCTURE ONLY -- imported by tests/ and tools/wire_bench.py, never by the
product (blance_amd/wire.py binds the C++ codec and has no fallback).

PARITY UNPINNED: the reference holds no JSON vectors for this type (it only
json.Marshal's values into test failure messages, plan_test.go:1591-1593) and no Go
toolchain exists here, so this file restates encoding/json's documented rules
(Go >= 1.22) and the C++ codec is checked against it:
  Marshal   -- map keys sorted bytewise; struct fields in declaration order; nil
               pointer / map / slice -> null; strings: HTML-safe escaping (< > &),
               U+2028/9 escaped, control characters \\b \\f \\n \\r \\t or \\u00XX,
               each invalid UTF-8 byte -> \\ufffd;
  Unmarshal -- exact or case-insensitive field match (ASCII folding here), unknown
               fields ignored, repeated map keys replace, a repeated nodesByState
               field merges, null leaves zero values, wrong JSON types are errors,
               unpaired surrogate escapes and invalid UTF-8 become U+FFFD.
Values: None = nil; {key: None | {"name": str, "nodesByState": None | {state: None | [str]}}}."""
import json


class WireTypeError(ValueError):
    pass


class WireSyntaxError(ValueError):
    pass


def _utf8_len(b, i):
    """utf8.DecodeRune's accepted sequence length at b[i:], 0 if invalid."""
    c = b[i]
    n = len(b)
    if c < 0x80:
        return 1
    if c < 0xC2:
        return 0

    def cont(j):
        return j < n and (b[j] & 0xC0) == 0x80
    if c < 0xE0:
        return 2 if cont(i + 1) else 0
    if c < 0xF0:
        if not (cont(i + 1) and cont(i + 2)):
            return 0
        if c == 0xE0 and b[i + 1] < 0xA0:
            return 0
        if c == 0xED and b[i + 1] > 0x9F:
            return 0
        return 3
    if c < 0xF5:
        if not (cont(i + 1) and cont(i + 2) and cont(i + 3)):
            return 0
        if c == 0xF0 and b[i + 1] < 0x90:
            return 0
        if c == 0xF4 and b[i + 1] > 0x8F:
            return 0
        return 4
    return 0


def go_string(b):
    """encodeState.string with escapeHTML = true, over raw bytes."""
    if isinstance(b, str):
        b = b.encode("utf-8", "surrogatepass")
    out = bytearray(b'"')
    i = 0
    while i < len(b):
        c = b[i]
        if c < 0x80:
            ch = bytes([c])
            if c >= 0x20 and ch not in b'"\\<>&':
                out += ch
            elif ch == b'"':
                out += b'\\"'
            elif ch == b"\\":
                out += b"\\\\"
            elif c == 8:
                out += b"\\b"
            elif c == 12:
                out += b"\\f"
            elif c == 10:
                out += b"\\n"
            elif c == 13:
                out += b"\\r"
            elif c == 9:
                out += b"\\t"
            else:
                out += b"\\u00%02x" % c
            i += 1
            continue
        n = _utf8_len(b, i)
        if n == 0:
            out += b"\\ufffd"
            i += 1
            continue
        seq = b[i:i + n]
        if seq in (b"\xe2\x80\xa8", b"\xe2\x80\xa9"):
            out += b"\\u202" + (b"8" if seq[2] == 0xA8 else b"9")
        else:
            out += seq
        i += n
    out += b'"'
    return bytes(out)


def _key_bytes(s):
    return s if isinstance(s, bytes) else s.encode("utf-8", "surrogatepass")


def marshal(pmap):
    if pmap is None:
        return b"null"
    parts = []
    for key in sorted(pmap, key=_key_bytes):
        p = pmap[key]
        if p is None:
            parts.append(go_string(key) + b":null")
            continue
        nbs = p.get("nodesByState")
        if nbs is None:
            body = b"null"
        else:
            ents = []
            for state in sorted(nbs, key=_key_bytes):
                lst = nbs[state]
                ents.append(go_string(state) + b":" +
                            (b"null" if lst is None else b"[" + b",".join(go_string(n) for n in lst) + b"]"))
            body = b"{" + b",".join(ents) + b"}"
        parts.append(go_string(key) + b':{"name":' + go_string(p.get("name", "")) + b',"nodesByState":' + body + b"}")
    return b"{" + b",".join(parts) + b"}"


def _sanitize(data):
    """Invalid UTF-8 bytes -> U+FFFD each (what the decoder does inside strings; outside of
    strings such bytes are a syntax error either way)."""
    out = bytearray()
    i = 0
    while i < len(data):
        n = _utf8_len(data, i)
        if n == 0:
            out += "�".encode("utf-8")
            i += 1
        else:
            out += data[i:i + n]
            i += n
    return bytes(out)


def _fix(s):
    return "".join("�" if 0xD800 <= ord(ch) < 0xE000 else ch for ch in s)


class _Pairs(list):
    pass


def unmarshal(data):
    if isinstance(data, str):
        data = data.encode("utf-8", "surrogatepass")

    def bad_const(name):
        raise WireSyntaxError("invalid literal %s" % name)
    try:
        doc = json.loads(_sanitize(data).decode("utf-8"), object_pairs_hook=_Pairs, parse_constant=bad_const)
    except WireSyntaxError:
        raise
    except ValueError as e:
        raise WireSyntaxError(str(e))
    if doc is None:
        return None
    if not isinstance(doc, _Pairs):
        raise WireTypeError("PartitionMap must be an object")
    out = {}
    for key, val in doc:
        key = _fix(key)
        if val is None:
            out[key] = None
            continue
        if not isinstance(val, _Pairs):
            raise WireTypeError("partition must be an object")
        part = {"name": "", "nodesByState": None}
        for f, fv in val:
            lf = "".join(chr(ord(c) + 32) if "A" <= c <= "Z" else c for c in f)
            if lf == "name":
                if fv is None:
                    continue
                if not isinstance(fv, str):
                    raise WireTypeError("name must be a string")
                part["name"] = _fix(fv)
            elif lf == "nodesbystate":
                if fv is None:
                    part["nodesByState"] = None
                    continue
                if not isinstance(fv, _Pairs):
                    raise WireTypeError("nodesByState must be an object")
                nbs = part["nodesByState"] if part["nodesByState"] is not None else {}
                for state, lst in fv:
                    state = _fix(state)
                    if lst is None:
                        nbs[state] = None
                        continue
                    if isinstance(lst, _Pairs) or not isinstance(lst, list):
                        raise WireTypeError("state list must be an array")
                    vals = []
                    for n in lst:
                        if n is None:
                            vals.append("")
                        elif isinstance(n, str):
                            vals.append(_fix(n))
                        else:
                            raise WireTypeError("node name must be a string")
                    nbs[state] = vals
                part["nodesByState"] = nbs
        out[key] = part
    return out
