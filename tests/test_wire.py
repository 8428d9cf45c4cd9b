"""PartitionMap JSON wire format: the C++ streaming codec of
include/blance_wire.h against the restatement of encoding/json in oracle/wire_ref.py.
Host-only code: everything here runs without a GPU."""
import ctypes
import json
import os
import random
import re

import pytest
from hypothesis import given, settings, strategies as st

import __graft_entry__ as entry
from blance_amd import wire
from oracle import wire_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built():
    entry.build_wire()


def _both_ways(pmap):
    """marshal -> unmarshal through both implementations; all four artefacts must agree."""
    want = wire_ref.marshal(pmap)
    got = wire.encode(pmap)
    assert got == want
    m = wire.decode(got)
    back = m.to_dict()
    assert back == wire_ref.unmarshal(want)
    assert m.encode() == want                    # arrays -> bytes without passing through Python objects
    m.close()
    return back


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(wire.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "blance_wire.h")).read()
    declared = set(re.findall(r"\b(blance_wire_[a-z_]+)\s*\(", header))
    assert declared == set(wire.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_golden_planner_maps_round_trip(golden_cases):
    """Every map the reference's planner tests hold (prevMap, partitionsToAssign, expected
    result; plan_test.go / control_test.go via tests/golden/planner_cases.json)."""
    n = 0
    for c in golden_cases:
        for k in ("prevMap", "partitionsToAssign", "exp"):
            pmap = c.get(k)
            if pmap is None:
                continue
            assert _both_ways(pmap) == pmap
            n += 1
    assert n >= 150


def test_struct_and_nil_forms():
    pmap = {"p0": None,
            "p1": {"name": "", "nodesByState": None},
            "p2": {"name": "other-name", "nodesByState": {}},
            "p3": {"name": "p3", "nodesByState": {"primary": None, "replica": [], "a": ["n1", "", "n1"]}},
            "": {"name": "empty key", "nodesByState": {"": [""]}}}
    assert _both_ways(pmap) == pmap
    assert wire.encode(None) == b"null" == wire_ref.marshal(None)
    m = wire.decode(b" null ")
    assert m.map_is_nil and m.to_dict() is None
    assert wire.decode(b"{}").to_dict() == {}
    assert wire.encode({}) == b"{}"


def test_key_order_is_bytewise():
    keys = ["10", "9", "a", "B", "é", "z", "", "a\x00", "a b", "\U0001F600", "￿"]
    pmap = {k: {"name": k, "nodesByState": {s: [k] for s in keys}} for k in keys}
    out = wire.encode(pmap)
    assert out == wire_ref.marshal(pmap)
    assert list(json.loads(out).keys()) == sorted(keys, key=lambda s: s.encode("utf-8"))


def test_string_escaping_matches_go():
    samples = ['plain', 'quote"back\\slash', '<script>&amp;</script>', "tab\tnl\ncr\rbs\bff\f", "\x00\x01\x1f\x7f",
               "line\u2028sep\u2029para", "é漢字\U0001F600", "a/b"]
    for s in samples:
        pmap = {s: {"name": s, "nodesByState": {s: [s]}}}
        assert _both_ways(pmap) == pmap
    assert wire.encode({"<": None}) == b'{"\\u003c":null}'
    assert wire.encode({"\x7f\x1f": None}) == b'{"\x7f\\u001f":null}'
    # invalid UTF-8 goes out as one � per bad byte
    bad = {b"a\xffb\xe2\x82": {"name": b"\xc0\xaf", "nodesByState": {b"s": [b"\xed\xa0\x80"]}}}
    assert wire.encode(bad) == wire_ref.marshal(bad)
    assert wire.encode(bad).count(b"\\ufffd") == 1 + 2 + 2 + 3


def test_decoder_follows_unmarshal_rules():
    docs = [
        b'{"a":{"Name":"x","NODESBYSTATE":{"s":["n"]},"nodesByStates":{"t":[]},"other":{"deep":[1,2,{"x":null}]}}}',
        b'{"a":{"name":"first"},"b":null,"a":{"nodesByState":{"s":["later wins"]}}}',            # repeated map key
        b'{"a":{"nodesByState":{"s":["1"],"t":["2"]},"name":"n","nodesByState":{"s":["3"],"u":null}}}',   # merge
        b'{"a":{"nodesByState":{"s":["1"]},"nodesByState":null}}',
        b'{"a":{"nodesByState":{"s":["1"],"s":null,"s":["2","3"]}}}',
        b'{"a":{"name":null,"nodesByState":{"s":[null,"x",null]}}}',
        b'{"\\u0061\\u00e9\\ud83d\\ude00":{"name":"\\ud800 lone \\udc00 \\ud800\\u0041","nodesByState":{"\\/":["\\b\\f\\n\\r\\t\\"\\\\"]}}}',
        b' \t\r\n{ "a" : { "name" : "x" , "nodesByState" : { "s" : [ "n" , "m" ] } } } \n',
        b'{"a\xff":{"name":"\xe2\x82","nodesByState":{"\xf0\x9f\x98\x80":["\xed\xa0\x80"]}}}',
        b'{"a":{"name":"x","nodesByState":{"s":["n"]}},"n":123e-5,"t":true,"f":false}'.replace(b',"n":123e-5,"t":true,"f":false', b''),
    ]
    for d in docs:
        m = wire.decode(d)
        assert m.to_dict() == wire_ref.unmarshal(d), d
        assert m.encode() == wire_ref.marshal(wire_ref.unmarshal(d)), d
        m.close()


@pytest.mark.parametrize("doc,status", [
    (b'', wire_ref.WireSyntaxError), (b'{', wire_ref.WireSyntaxError), (b'{"a":}', wire_ref.WireSyntaxError),
    (b'{"a":null,}', wire_ref.WireSyntaxError), (b'{"a":null} x', wire_ref.WireSyntaxError),
    (b'{"a":{"name":"x\ny"}}', wire_ref.WireSyntaxError), (b'{"a":{"name":"\\x"}}', wire_ref.WireSyntaxError),
    (b'{"a":{"name":"\\u12g4"}}', wire_ref.WireSyntaxError), (b'{"a":nul}', wire_ref.WireSyntaxError),
    (b'{"a":{"other":01}}', wire_ref.WireSyntaxError), (b'{"a":{"other":[1,]}}', wire_ref.WireSyntaxError),
    (b'{"a":{"other":NaN}}', wire_ref.WireSyntaxError), (b"{'a':null}", wire_ref.WireSyntaxError),
    (b'[]', wire_ref.WireTypeError), (b'"x"', wire_ref.WireTypeError), (b'{"a":[]}', wire_ref.WireTypeError),
    (b'{"a":{"name":5}}', wire_ref.WireTypeError), (b'{"a":{"nodesByState":[]}}', wire_ref.WireTypeError),
    (b'{"a":{"nodesByState":{"s":"n"}}}', wire_ref.WireTypeError),
    (b'{"a":{"nodesByState":{"s":[1]}}}', wire_ref.WireTypeError),
    (b'{"a":{"nodesByState":{"s":{}}}}', wire_ref.WireTypeError),
])
def test_decoder_errors(doc, status):
    with pytest.raises(status):
        wire_ref.unmarshal(doc)
    with pytest.raises(wire.WireError) as e:
        wire.decode(doc)
    assert e.value.status == (-1 if status is wire_ref.WireSyntaxError else -2)


_names = st.text(alphabet=st.characters(blacklist_categories=("Cs",)), max_size=6) | st.sampled_from(
    ["primary", "replica", "n0", "n1", "<&>", "  ", "\x00"])
_lists = st.none() | st.lists(_names, max_size=4)
_parts = st.none() | st.fixed_dictionaries({"name": _names, "nodesByState": st.none() | st.dictionaries(_names, _lists, max_size=4)})


@settings(max_examples=300, deadline=None)
@given(st.dictionaries(_names, _parts, max_size=6))
def test_random_maps_round_trip(pmap):
    assert _both_ways(pmap) == pmap


@settings(max_examples=200, deadline=None)
@given(st.binary(max_size=40))
def test_random_bytes_never_disagree_on_acceptance(data):
    """Arbitrary bytes: both decoders accept (same value) or both refuse."""
    try:
        want = wire_ref.unmarshal(data)
    except (wire_ref.WireSyntaxError, wire_ref.WireTypeError):
        with pytest.raises(wire.WireError):
            wire.decode(data)
        return
    assert wire.decode(data).to_dict() == want


def test_mutated_documents_agree():
    """Single-byte edits of a valid document: accept/refuse decisions and values agree."""
    base = wire_ref.marshal({"p%d" % i: {"name": "p%d" % i, "nodesByState": {"primary": ["n%d" % (i % 3)], "replica": None}}
                             for i in range(4)})
    rng = random.Random(7)
    for _ in range(1500):
        b = bytearray(base)
        for _ in range(rng.choice([1, 1, 2])):
            i = rng.randrange(len(b))
            op = rng.random()
            if op < 0.4:
                b[i] = rng.choice(b'{}[]":,\\ntu0 \x00\xff\xe2')
            elif op < 0.7:
                del b[i]
            else:
                b.insert(i, rng.choice(b'{}[]":,\\ntu0 '))
        data = bytes(b)
        try:
            want = wire_ref.unmarshal(data)
        except wire_ref.WireSyntaxError:
            with pytest.raises(wire.WireError):
                wire.decode(data)
            continue
        except wire_ref.WireTypeError:
            with pytest.raises(wire.WireError) as e:
                wire.decode(data)
            assert e.value.status == -2, data
            continue
        assert wire.decode(data).to_dict() == want, data


def test_large_map_round_trip():
    """A config-2-sized map (65,536 partitions): planner output shape, byte-identical both ways."""
    P, N = 65536, 256
    pmap = {str(i): {"name": str(i), "nodesByState": {"primary": ["n%03d" % (i % N)], "replica": ["n%03d" % ((i % N) ^ 1)]}}
            for i in range(P)}
    want = wire_ref.marshal(pmap)
    got = wire.encode(pmap)
    assert got == want
    m = wire.decode(got)
    assert (m.n_parts, m.n_states, m.n_nodes) == (P, 2, N)
    assert m.encode() == want
    assert json.loads(got) == pmap
    m.close()


def test_caller_owned_forms_agree_with_the_handles(golden_cases):
    """blance_wire_decode_into / blance_wire_encode_into (nothing of the library's crosses the boundary): same arrays
    and same bytes as the handle-based calls; a too small buffer is answered with the sizes and the second call fits."""
    docs = [b"null", b"{}", b'{"a":null}', b'{"0":{"name":"0","nodesByState":{"primary":["a","b"],"replica":null}}}']
    for c in golden_cases[:40]:
        for k in ("prevMap", "exp"):
            if c.get(k) is not None:
                docs.append(wire.encode(c[k]))
    for doc in docs:
        m = wire.decode(doc)
        v, arrays, calls = wire.decode_into_numpy(doc)
        assert calls <= 2
        for f in ("map_is_nil", "n_parts", "n_states", "n_nodes", "n_entries", "n_node_refs"):
            assert getattr(v, f) == getattr(m.view, f), f
        assert list(arrays["part_kind"][:v.n_parts]) == list(m.part_kind)
        assert list(arrays["entry_nodes"][:v.n_node_refs]) == list(m.entry_nodes)
        assert list(arrays["entry_off"][:v.n_entries + 1]) == list(m.entry_off)
        want = m.encode()
        got, calls = wire.encode_into_numpy(v)                # from the caller's arrays into the caller's bytes
        assert got == want and calls <= 2
        got, calls = wire.encode_into_numpy(m.view, cap=len(want))
        assert got == want and calls == 1
        m.close()
    with pytest.raises(wire.WireError) as e:
        wire.decode_into_numpy(b'{"a":')
    assert e.value.status == -1


def _wire_vectors():
    with open(os.path.join(ROOT, "tests", "golden", "wire_cases.json")) as f:
        return json.load(f)


def test_rule_derived_vectors_marshal():
    """tests/golden/wire_cases.json: bytes written out BY HAND from the rules encoding/json documents (each vector cites its
    rule) -- they pin both the restatement oracle/wire_ref.py and the C++ codec."""
    v = _wire_vectors()
    for c in v["marshal"]:
        want = bytes.fromhex(c["hex"])
        assert wire_ref.marshal(c["value"]) == want, (c["id"], c["rule"])
        assert wire.encode(c["value"]) == want, (c["id"], c["rule"])
    for c in v["marshal_raw"]:
        want = bytes.fromhex(c["hex"])
        pmap = {bytes.fromhex(c["key_hex"]): {"name": bytes.fromhex(c["name_hex"]), "nodesByState": None}}
        assert wire_ref.marshal(pmap) == want, (c["id"], c["rule"])
        assert wire.encode(pmap) == want, (c["id"], c["rule"])
    assert len(v["marshal"]) + len(v["marshal_raw"]) >= 9


def test_rule_derived_vectors_unmarshal():
    v = _wire_vectors()
    for c in v["unmarshal"]:
        doc = bytes.fromhex(c["hex"])
        assert wire_ref.unmarshal(doc) == c["value"], (c["id"], c["rule"])
        m = wire.decode(doc)
        assert m.to_dict() == c["value"], (c["id"], c["rule"])
        m.close()
    for c in v["unmarshal_raw"]:
        doc = bytes.fromhex(c["hex"])
        want = {bytes.fromhex(c["key_hex"]).decode("utf-8"): {"name": bytes.fromhex(c["name_hex"]).decode("utf-8"), "nodesByState": None}}
        assert wire_ref.unmarshal(doc) == want, (c["id"], c["rule"])
        m = wire.decode(doc)
        assert m.to_dict() == want, (c["id"], c["rule"])
        m.close()
    for c in v["errors"]:
        doc = bytes.fromhex(c["hex"])
        with pytest.raises((wire_ref.WireSyntaxError, wire_ref.WireTypeError)):
            wire_ref.unmarshal(doc)
        with pytest.raises(wire.WireError):
            wire.decode(doc)
    assert len(v["unmarshal"]) + len(v["unmarshal_raw"]) + len(v["errors"]) >= 18
