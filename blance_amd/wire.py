"""ctypes binding of the PartitionMap JSON codec (include/blance_wire.h,
blance_amd/csrc/host/blance_wire.cpp) -- SURVEY.md section 8(f) rank 2.

decode(bytes) -> WireMap (numpy views of the interned struct of arrays);
WireMap.to_dict() rebuilds the Go value as Python objects (None = nil) for tests;
encode(dict | WireMap) -> bytes, byte-identical to json.Marshal(PartitionMap).
There is no Python fallback: a missing library raises ImportError."""
import ctypes as C
import os

import numpy as np

ABSENT, NIL, LIST = 0, 1, 2
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libblance_wire.so")
EXPORTS = ("blance_wire_decode", "blance_wire_view_of", "blance_wire_free", "blance_wire_encode",
           "blance_wire_free_bytes", "blance_wire_last_error", "blance_wire_abi_version",
           "blance_wire_decode_into", "blance_wire_encode_into")
ERR_SPACE = -4


class View(C.Structure):
    _fields_ = [("map_is_nil", C.c_int32),
                ("n_parts", C.c_int64), ("n_states", C.c_int64), ("n_nodes", C.c_int64),
                ("n_entries", C.c_int64), ("n_node_refs", C.c_int64),
                ("key_bytes", C.c_void_p), ("key_off", C.c_void_p),
                ("name_bytes", C.c_void_p), ("name_off", C.c_void_p),
                ("part_kind", C.c_void_p), ("part_off", C.c_void_p),
                ("state_bytes", C.c_void_p), ("state_off", C.c_void_p),
                ("node_bytes", C.c_void_p), ("node_off", C.c_void_p),
                ("entry_state", C.c_void_p), ("entry_kind", C.c_void_p),
                ("entry_off", C.c_void_p), ("entry_nodes", C.c_void_p)]


class Buffers(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("cap_parts", "cap_states", "cap_nodes", "cap_entries", "cap_node_refs",
                                          "cap_key_bytes", "cap_name_bytes", "cap_state_bytes", "cap_node_bytes")] + \
               [(n, C.c_void_p) for n in ("key_bytes", "key_off", "name_bytes", "name_off", "part_kind", "part_off",
                                          "state_bytes", "state_off", "node_bytes", "node_off",
                                          "entry_state", "entry_kind", "entry_off", "entry_nodes")]


class WireError(ValueError):
    def __init__(self, status, msg):
        ValueError.__init__(self, "blance_wire status %d: %s" % (status, msg))
        self.status = status


_lib = None


def load_library(path=None):
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise ImportError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`" % path)
    lib = C.CDLL(path)
    lib.blance_wire_decode.restype = C.c_int
    lib.blance_wire_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]
    lib.blance_wire_view_of.restype = C.c_int
    lib.blance_wire_view_of.argtypes = [C.c_void_p, C.POINTER(View)]
    lib.blance_wire_free.restype = None
    lib.blance_wire_free.argtypes = [C.c_void_p]
    lib.blance_wire_encode.restype = C.c_int
    lib.blance_wire_encode.argtypes = [C.POINTER(View), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    lib.blance_wire_free_bytes.restype = None
    lib.blance_wire_free_bytes.argtypes = [C.c_void_p]
    lib.blance_wire_last_error.restype = C.c_char_p
    lib.blance_wire_decode_into.restype = C.c_int
    lib.blance_wire_decode_into.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Buffers), C.POINTER(View)]
    lib.blance_wire_encode_into.restype = C.c_int
    lib.blance_wire_encode_into.argtypes = [C.POINTER(View), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.blance_wire_abi_version.restype = C.c_int
    if lib.blance_wire_abi_version() != 2:
        raise ImportError("blance_wire ABI version mismatch")
    _lib = lib
    return lib


def _arr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(n,))


def _strings(bytes_ptr, off):
    total = int(off[-1]) if len(off) else 0
    blob = C.string_at(bytes_ptr, total) if total else b""
    return [blob[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]


class WireMap:
    """A decoded PartitionMap; arrays are views into library memory until close()."""

    def __init__(self, lib, handle):
        self._lib, self._h = lib, handle
        v = View()
        st = lib.blance_wire_view_of(handle, C.byref(v))
        if st:
            raise WireError(st, lib.blance_wire_last_error().decode())
        self.view = v
        self.map_is_nil = bool(v.map_is_nil)
        P, E = int(v.n_parts), int(v.n_entries)
        self.n_parts, self.n_states, self.n_nodes = P, int(v.n_states), int(v.n_nodes)
        self.key_off = _arr(v.key_off, P + 1, np.int64)
        self.name_off = _arr(v.name_off, P + 1, np.int64)
        self.part_kind = _arr(v.part_kind, P, np.uint8)
        self.part_off = _arr(v.part_off, P + 1, np.int64)
        self.state_off = _arr(v.state_off, self.n_states + 1, np.int64)
        self.node_off = _arr(v.node_off, self.n_nodes + 1, np.int64)
        self.entry_state = _arr(v.entry_state, E, np.int32)
        self.entry_kind = _arr(v.entry_kind, E, np.uint8)
        self.entry_off = _arr(v.entry_off, E + 1, np.int64)
        self.entry_nodes = _arr(v.entry_nodes, int(v.n_node_refs), np.int32)

    def keys(self):
        return _strings(self.view.key_bytes, self.key_off)

    def names(self):
        return _strings(self.view.name_bytes, self.name_off)

    def states(self):
        return _strings(self.view.state_bytes, self.state_off)

    def nodes(self):
        return _strings(self.view.node_bytes, self.node_off)

    def to_dict(self):
        """{key: None | {"name": str, "nodesByState": None | {state: None | [node, ...]}}}; strings as str
        (UTF-8; the decoder only emits valid UTF-8)."""
        if self.map_is_nil:
            return None
        keys, names = self.keys(), self.names()
        states = [s.decode("utf-8") for s in self.states()]
        nodes = [s.decode("utf-8") for s in self.nodes()]
        out = {}
        for i in range(self.n_parts):
            kind = int(self.part_kind[i])
            if kind == ABSENT:
                out[keys[i].decode("utf-8")] = None
                continue
            nbs = None
            if kind == LIST:
                nbs = {}
                for e in range(int(self.part_off[i]), int(self.part_off[i + 1])):
                    if int(self.entry_kind[e]) == LIST:
                        nbs[states[int(self.entry_state[e])]] = [
                            nodes[int(x)] for x in self.entry_nodes[int(self.entry_off[e]):int(self.entry_off[e + 1])]]
                    else:
                        nbs[states[int(self.entry_state[e])]] = None
            out[keys[i].decode("utf-8")] = {"name": names[i].decode("utf-8"), "nodesByState": nbs}
        return out

    def encode(self):
        return _encode_view(self._lib, self.view)

    def close(self):
        if self._h:
            self._lib.blance_wire_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


_BUF_FIELDS = (("key_bytes", "cap_key_bytes", np.uint8, 0), ("key_off", "cap_parts", np.int64, 1),
               ("name_bytes", "cap_name_bytes", np.uint8, 0), ("name_off", "cap_parts", np.int64, 1),
               ("part_kind", "cap_parts", np.uint8, 0), ("part_off", "cap_parts", np.int64, 1),
               ("state_bytes", "cap_state_bytes", np.uint8, 0), ("state_off", "cap_states", np.int64, 1),
               ("node_bytes", "cap_node_bytes", np.uint8, 0), ("node_off", "cap_nodes", np.int64, 1),
               ("entry_state", "cap_entries", np.int32, 0), ("entry_kind", "cap_entries", np.uint8, 0),
               ("entry_off", "cap_entries", np.int64, 1), ("entry_nodes", "cap_node_refs", np.int32, 0))


def decode_into_numpy(data, caps=None, lib_path=None):
    """The caller-owned form: numpy arrays of the caller (this function) are filled by blance_wire_decode_into.
    caps: dict of capacities (default: small, so that the size report and the second call are exercised).
    Returns (View, arrays, calls)."""
    lib = load_library(lib_path)
    if isinstance(data, str):
        data = data.encode("utf-8")
    b = Buffers()
    for f, _ in Buffers._fields_[:9]:
        setattr(b, f, (caps or {}).get(f, 0))
    calls = 0
    while True:
        arrays = {}
        for name, cap, dt, extra in _BUF_FIELDS:
            arrays[name] = np.zeros(int(getattr(b, cap)) + extra, dtype=dt)
            setattr(b, name, arrays[name].ctypes.data)
        v = View()
        st = lib.blance_wire_decode_into(data, len(data), C.byref(b), C.byref(v))
        calls += 1
        if st == ERR_SPACE and calls < 3:
            continue                             # cap_* now hold what the document needs
        if st:
            raise WireError(st, lib.blance_wire_last_error().decode())
        return v, arrays, calls


def encode_into_numpy(view, cap=0, lib_path=None):
    """The caller-owned form of encode: returns (bytes, calls)."""
    lib = load_library(lib_path)
    need = C.c_size_t()
    calls = 0
    while True:
        buf = np.zeros(max(cap, 1), dtype=np.uint8)
        st = lib.blance_wire_encode_into(C.byref(view), buf.ctypes.data, cap, C.byref(need))
        calls += 1
        if st == ERR_SPACE and calls < 3:
            cap = need.value
            continue
        if st:
            raise WireError(st, lib.blance_wire_last_error().decode())
        return bytes(buf[:need.value]), calls


def decode(data, lib_path=None):
    lib = load_library(lib_path)
    if isinstance(data, str):
        data = data.encode("utf-8")
    h = C.c_void_p()
    st = lib.blance_wire_decode(data, len(data), C.byref(h))
    if st:
        raise WireError(st, lib.blance_wire_last_error().decode())
    return WireMap(lib, h)


def _encode_view(lib, view):
    out, n = C.c_void_p(), C.c_size_t()
    st = lib.blance_wire_encode(C.byref(view), C.byref(out), C.byref(n))
    if st:
        raise WireError(st, lib.blance_wire_last_error().decode())
    try:
        return C.string_at(out, n.value)
    finally:
        lib.blance_wire_free_bytes(out)


def _b(s):
    return s if isinstance(s, bytes) else s.encode("utf-8", "surrogatepass")


def encode(pmap, lib_path=None):
    """json.Marshal of a PartitionMap given as Python objects (see WireMap.to_dict; strings may be
    str or bytes -- bytes let tests feed invalid UTF-8).  Interning here is plain Python: this entry
    is the tested mirror of what the cgo side does; bulk callers hand over arrays via encode_arrays."""
    lib = load_library(lib_path)
    if pmap is None:
        v = View()
        v.map_is_nil = 1
        return _encode_view(lib, v)
    keys, names, part_kind, part_off = [], [], [], [0]
    states, nodes = {}, {}
    entry_state, entry_kind, entry_off, entry_nodes = [], [], [0], []
    for key, part in pmap.items():
        keys.append(_b(key))
        if part is None:
            names.append(b"")
            part_kind.append(ABSENT)
            part_off.append(len(entry_state))
            continue
        names.append(_b(part.get("name", "")))
        nbs = part.get("nodesByState")
        part_kind.append(NIL if nbs is None else LIST)
        for state, lst in (nbs or {}).items():
            entry_state.append(states.setdefault(_b(state), len(states)))
            entry_kind.append(NIL if lst is None else LIST)
            for n in (lst or []):
                entry_nodes.append(nodes.setdefault(_b(n), len(nodes)))
            entry_off.append(len(entry_nodes))
        part_off.append(len(entry_state))
    return encode_arrays(keys, names, part_kind, part_off, list(states), list(nodes), entry_state, entry_kind,
                         entry_off, entry_nodes, lib_path=lib_path)


def _blob(strs):
    off = np.zeros(len(strs) + 1, dtype=np.int64)
    if strs:
        off[1:] = np.cumsum([len(s) for s in strs])
    return b"".join(strs), off


def encode_arrays(keys, names, part_kind, part_off, states, nodes, entry_state, entry_kind, entry_off, entry_nodes,
                  lib_path=None):
    lib = load_library(lib_path)
    keep = []

    def arr(x, dt):
        a = np.ascontiguousarray(np.asarray(x, dtype=dt))
        keep.append(a)
        return a.ctypes.data

    def blob(strs):
        b, off = _blob(strs)
        buf = C.create_string_buffer(b, len(b) + 1)
        keep.append(buf)
        keep.append(off)
        return C.cast(buf, C.c_void_p).value, off.ctypes.data

    v = View()
    v.n_parts, v.n_states, v.n_nodes = len(keys), len(states), len(nodes)
    v.n_entries, v.n_node_refs = len(entry_state), len(entry_nodes)
    v.key_bytes, v.key_off = blob(keys)
    v.name_bytes, v.name_off = blob(names)
    v.state_bytes, v.state_off = blob(states)
    v.node_bytes, v.node_off = blob(nodes)
    v.part_kind = arr(part_kind, np.uint8)
    v.part_off = arr(part_off, np.int64)
    v.entry_state = arr(entry_state, np.int32)
    v.entry_kind = arr(entry_kind, np.uint8)
    v.entry_off = arr(entry_off, np.int64)
    v.entry_nodes = arr(entry_nodes, np.int32)
    return _encode_view(lib, v)
