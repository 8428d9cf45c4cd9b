#!/usr/bin/env python3
"""tests/golden/wire_cases.json: JSON vectors for the PartitionMap wire format (api.go:24-36), DERIVED BY HAND from the rules
encoding/json documents (package documentation of Marshal / Unmarshal, quoted per vector) -- not produced by running any
codec of this repository.  The bytes are written out literally below; this script only packs them as hex (some vectors hold
bytes that are not valid UTF-8 and cannot sit in a JSON string).  The reference holds no JSON vectors for this type and no Go
toolchain exists in the build image, so these rule-derived vectors are what pins oracle/wire_ref.py and the C++ codec.
Deliberately absent: \\b and \\f (their escape changed between Go releases: \\u0008 / \\u000c before Go 1.22, \\b / \\f since)."""
import json
import os

M = "encoding/json Marshal doc: "
U = "encoding/json Unmarshal doc: "
P = lambda name, nbs: {"name": name, "nodesByState": nbs}          # noqa: E731

MARSHAL = [
    # (id, rule, value, expected bytes)
    ("nil-map", M + "'Map values encode as JSON objects' ... a nil map encodes as the null JSON value",
     None, b"null"),
    ("empty-map", M + "'Map values encode as JSON objects'", {}, b"{}"),
    ("nil-pointer", M + "'Pointer values encode as the value pointed to. A nil pointer encodes as the null JSON value.'",
     {"p": None}, b'{"p":null}'),
    ("field-names-and-order", M + "struct fields encode under their tag names (api.go:30 `json:\"name\"`, api.go:35 `json:\"nodesByState\"`), in "
     "declaration order; no omitempty: zero values are written",
     {"p": P("", {})}, b'{"p":{"name":"","nodesByState":{}}}'),
    ("nil-inner-map-and-slice", M + "'a nil slice encodes as the null JSON value'; the same for a nil map",
     {"p": P("p", None), "q": P("q", {"primary": None, "replica": []})},
     b'{"p":{"name":"p","nodesByState":null},"q":{"name":"q","nodesByState":{"primary":null,"replica":[]}}}'),
    ("keys-sorted", M + "'The map keys are sorted' (as strings, i.e. byte-wise: \"10\" < \"9\", \"B\" < \"a\")",
     {"9": P("9", {"b": ["x"], "a": ["y"], "B": []}), "10": P("10", {}), "a": None, "B": None},
     b'{"10":{"name":"10","nodesByState":{}},"9":{"name":"9","nodesByState":{"B":[],"a":["y"],"b":["x"]}},"B":null,"a":null}'),
    ("html-escape", M + "'String values encode as JSON strings ... \"<\", \">\", \"&\", U+2028, and U+2029 are escaped to \"\\u003c\",\"\\u003e\", "
     "\"\\u0026\", \"\\u2028\", and \"\\u2029\"' (map keys are strings too)",
     {"<k&>": P("a<b>&c", {"s": ["\u2028", "x\u2029y"]})},
     b'{"\\u003ck\\u0026\\u003e":{"name":"a\\u003cb\\u003e\\u0026c","nodesByState":{"s":["\\u2028","x\\u2029y"]}}}'),
    ("quote-backslash-controls", "RFC 8259 section 7 as encoding/json writes it: quotation mark and reverse solidus are escaped with a "
     "backslash, \\n \\r \\t by their short forms, other control characters as \\u00XX (lower-case hex); DEL (0x7f) and non-ASCII "
     "runes are written as they are",
     {"p": P('q"b\\s', {"s": ["a\nb\rc\td", "\x01\x1f", "\x7f", "\u00e9\u4e16\U0001F600"]})},
     b'{"p":{"name":"q\\"b\\\\s","nodesByState":{"s":["a\\nb\\rc\\td","\\u0001\\u001f","\x7f","\xc3\xa9\xe4\xb8\x96\xf0\x9f\x98\x80"]}}}'),
]

# strings with invalid UTF-8 come as raw bytes: (id, rule, key bytes, name bytes, expected)
MARSHAL_RAW = [
    ("invalid-utf8", M + "'String values encode as JSON strings coerced to valid UTF-8, replacing invalid bytes with the Unicode replacement "
     "rune' (one \\ufffd per invalid byte: a lone continuation byte, a truncated sequence, an overlong form, a surrogate)",
     b"k", b"a\x80b\xc3(\xe2\x82\xc0\xaf\xed\xa0\x80z",
     b'{"k":{"name":"a\\ufffdb\\ufffd(\\ufffd\\ufffd\\ufffd\\ufffd\\ufffd\\ufffd\\ufffdz","nodesByState":null}}'),
]

UNMARSHAL = [
    # (id, rule, bytes, expected value)
    ("null-document", U + "'The JSON null value unmarshals into an interface, map, pointer, or slice by setting that Go value to nil.'",
     b" null ", None),
    ("whitespace", "RFC 8259 section 2: insignificant whitespace (space, \\t, \\n, \\r) is allowed around the structural characters",
     b' {\t"p" :\n{ "name":\r"x" , "nodesByState" : { "s" : [ "a" , "b" ] } } } ', {"p": P("x", {"s": ["a", "b"]})}),
    ("case-insensitive-fields", U + "'preferring an exact match but also accepting a case-insensitive match'",
     b'{"p":{"NAME":"x","NodesByState":{"s":["a"]}}}', {"p": P("x", {"s": ["a"]})}),
    ("unknown-fields-ignored", U + "'By default, object keys which don't have a corresponding struct field are ignored'",
     b'{"p":{"extra":[1,{"x":null}],"name":"x","more":true,"nodesByState":null}}', {"p": P("x", None)}),
    ("missing-fields-zero", U + "a field that the object does not mention keeps its zero value (\"\" and a nil map)",
     b'{"p":{}}', {"p": P("", None)}),
    ("null-members", U + "'The JSON null value unmarshals into ... map, pointer, or slice by setting that Go value to nil. ... Otherwise, the JSON "
     "null value has no effect' (a null name stays \"\")",
     b'{"p":null,"q":{"name":null,"nodesByState":{"s":null,"t":[]}}}', {"p": None, "q": P("", {"s": None, "t": []})}),
    ("escapes", "RFC 8259 section 7: \\\" \\\\ \\/ \\b \\f \\n \\r \\t and \\uXXXX, a surrogate pair \\uD83D\\uDE00 is one code point",
     b'{"k\\u0041":{"name":"\\"\\\\\\/\\n\\r\\t\\u00e9\\uD83D\\uDE00","nodesByState":{}}}', {"kA": P('"\\/\n\r\t\u00e9\U0001F600', {})}),
    ("repeated-map-key", U + "unmarshaling into a map stores key-value pairs one after the other: a repeated key replaces the earlier entry",
     b'{"p":{"name":"first","nodesByState":{}},"p":{"name":"second","nodesByState":null}}', {"p": P("second", None)}),
    ("array-resets", U + "'To unmarshal a JSON array into a slice, Unmarshal resets the slice length to zero and then appends each element' "
     "(a repeated state key replaces the list)",
     b'{"p":{"name":"p","nodesByState":{"s":["a","b"],"s":["c"]}}}', {"p": P("p", {"s": ["c"]})}),
]

UNMARSHAL_RAW = [
    # (id, rule, bytes, expected key bytes, expected name bytes)
    ("invalid-utf8-in", U + "'When unmarshaling quoted strings, invalid UTF-8 or invalid UTF-16 surrogate pairs are not treated as an error. "
     "Instead, they are replaced by the Unicode replacement character U+FFFD.'",
     b'{"k":{"name":"a\x80b\\ud800c\\udc00","nodesByState":null}}', b"k", "a\ufffdb\ufffdc\ufffd".encode("utf-8")),
]

ERRORS = [
    # (id, rule, bytes)
    ("wrong-type-name", U + "'If a JSON value is not appropriate for a given target type ... Unmarshal ... returns an UnmarshalTypeError' "
     "(a number where the string `name` is expected)", b'{"p":{"name":5,"nodesByState":null}}'),
    ("wrong-type-list", U + "UnmarshalTypeError: an object where the []string of a state is expected", b'{"p":{"name":"x","nodesByState":{"s":{}}}}'),
    ("wrong-type-document", U + "UnmarshalTypeError: an array where the map is expected", b'[]'),
    ("trailing-data", "json.Unmarshal: 'invalid character ... after top-level value' (a SyntaxError)", b'{} {}'),
    ("truncated", "json.Unmarshal: 'unexpected end of JSON input' (a SyntaxError)", b'{"p":{"name":"x"'),
    ("control-in-string", "RFC 8259 section 7: control characters must be escaped; encoding/json: 'invalid character ... in string literal'",
     b'{"p":{"name":"a\nb","nodesByState":null}}'),
    ("bad-escape", "encoding/json: 'invalid character ... in string escape code'", b'{"p":{"name":"\\x","nodesByState":null}}'),
    ("trailing-comma", "RFC 8259: no trailing comma in an object", b'{"p":null,}'),
]


def main():
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = {"generator": "tests/tools/make_wire_vectors.py (vectors written by hand from the cited rules; hex packing only)",
           "marshal": [{"id": i, "rule": r, "value": v, "hex": b.hex()} for i, r, v, b in MARSHAL],
           "marshal_raw": [{"id": i, "rule": r, "key_hex": k.hex(), "name_hex": n.hex(), "hex": b.hex()} for i, r, k, n, b in MARSHAL_RAW],
           "unmarshal": [{"id": i, "rule": r, "hex": b.hex(), "value": v} for i, r, b, v in UNMARSHAL],
           "unmarshal_raw": [{"id": i, "rule": r, "hex": b.hex(), "key_hex": k.hex(), "name_hex": n.hex()} for i, r, b, k, n in UNMARSHAL_RAW],
           "errors": [{"id": i, "rule": r, "hex": b.hex()} for i, r, b in ERRORS]}
    with open(os.path.join(root, "tests", "golden", "wire_cases.json"), "w") as f:
        json.dump(out, f, indent=1)
    print({k: len(v) for k, v in out.items() if isinstance(v, list)})


if __name__ == "__main__":
    main()
