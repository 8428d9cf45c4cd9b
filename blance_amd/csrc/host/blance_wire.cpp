// Streaming codec for the PartitionMap JSON wire format (include/blance_wire.h).
// Host only.  One pass over the bytes, strings interned through open-addressing
// tables keyed by the decoded bytes; no DOM, no per-string allocation.
//
// Reference behaviour followed: encoding/json over map[string]*Partition with the
// struct tags of api.go:28-36 (`name`, `nodesByState`); see the header for the rules.
#include "blance_wire.h"

#include <cstdlib>
#include <cstring>
#include <new>
#include <memory>
#include <string>
#include <vector>

namespace {

constexpr int kAbiVersion = 2;
thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

// ---------------------------------------------------------------- string tables
struct Blob {                               // concatenated strings + offsets
    std::string bytes;
    std::vector<int64_t> off{0};
    int64_t size() const { return (int64_t)off.size() - 1; }
    void push(const char* p, size_t n) {
        bytes.append(p, n);
        off.push_back((int64_t)bytes.size());
    }
    const char* at(int64_t i, size_t* n) const {
        *n = (size_t)(off[i + 1] - off[i]);
        return bytes.data() + off[i];
    }
};

inline uint64_t hash_bytes(const char* p, size_t n) {          // FNV-1a, 64 bit, finalised
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= (unsigned char)p[i]; h *= 1099511628211ull; }
    h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 32;
    return h;
}

struct Interner {                           // bytes -> dense id, first seen order
    Blob blob;
    std::vector<int32_t> slot;              // -1 = empty
    std::vector<uint64_t> hashes;           // per id
    Interner() : slot(64, -1) {}
    void grow() {
        std::vector<int32_t> ns(slot.size() * 2, -1);
        const size_t mask = ns.size() - 1;
        for (int32_t id = 0; id < (int32_t)hashes.size(); id++) {
            size_t i = hashes[id] & mask;
            while (ns[i] >= 0) i = (i + 1) & mask;
            ns[i] = id;
        }
        slot.swap(ns);
    }
    int32_t intern(const char* p, size_t n) {
        const uint64_t h = hash_bytes(p, n);
        size_t mask = slot.size() - 1, i = h & mask;
        while (slot[i] >= 0) {
            const int32_t id = slot[i];
            if (hashes[id] == h) {
                size_t m;
                const char* q = blob.at(id, &m);
                if (m == n && memcmp(p, q, n) == 0) return id;
            }
            i = (i + 1) & mask;
        }
        const int32_t id = (int32_t)hashes.size();
        slot[i] = id;
        hashes.push_back(h);
        blob.push(p, n);
        if (hashes.size() * 2 > slot.size()) grow();
        return id;
    }
};

}  // namespace

struct blance_wire_map {
    int32_t map_is_nil = 0;
    Blob keys, names;
    std::vector<uint8_t> part_kind;
    std::vector<int64_t> part_off{0};
    Interner states, nodes;
    std::vector<int32_t> entry_state;
    std::vector<uint8_t> entry_kind;
    std::vector<int64_t> entry_off{0};
    std::vector<int32_t> entry_nodes;
};

namespace {

// ---------------------------------------------------------------- decoder
struct Parser {
    const char* b;
    const char* p;
    const char* e;
    std::string scratch;                    // decoded bytes of the current string
    std::string msg;
    int code = 0;

    bool err(int c, const char* what) {
        if (!code) {
            code = c;
            msg = std::string(what) + " at byte " + std::to_string((long long)(p - b));
        }
        return false;
    }
    void ws() {
        while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++;
    }
    bool lit(const char* s, size_t n) {
        if ((size_t)(e - p) < n || memcmp(p, s, n) != 0) return err(BLANCE_WIRE_ERR_SYNTAX, "invalid literal");
        p += n;
        return true;
    }
    static void put_utf8(std::string& o, uint32_t c) {
        if (c < 0x80) o.push_back((char)c);
        else if (c < 0x800) { o.push_back((char)(0xC0 | (c >> 6))); o.push_back((char)(0x80 | (c & 0x3F))); }
        else if (c < 0x10000) {
            o.push_back((char)(0xE0 | (c >> 12))); o.push_back((char)(0x80 | ((c >> 6) & 0x3F)));
            o.push_back((char)(0x80 | (c & 0x3F)));
        } else {
            o.push_back((char)(0xF0 | (c >> 18))); o.push_back((char)(0x80 | ((c >> 12) & 0x3F)));
            o.push_back((char)(0x80 | ((c >> 6) & 0x3F))); o.push_back((char)(0x80 | (c & 0x3F)));
        }
    }
    int hex4(const char* q) {
        int v = 0;
        for (int i = 0; i < 4; i++) {
            char c = q[i];
            int d = c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1;
            if (d < 0) return -1;
            v = v * 16 + d;
        }
        return v;
    }
    // length of the valid UTF-8 sequence at q (0 if invalid), Go's utf8.DecodeRune rules
    static int utf8_len(const unsigned char* q, const unsigned char* end) {
        unsigned char c = q[0];
        if (c < 0x80) return 1;
        if (c < 0xC2) return 0;
        if (c < 0xE0) return (end - q >= 2 && (q[1] & 0xC0) == 0x80) ? 2 : 0;
        if (c < 0xF0) {
            if (end - q < 3 || (q[1] & 0xC0) != 0x80 || (q[2] & 0xC0) != 0x80) return 0;
            if (c == 0xE0 && q[1] < 0xA0) return 0;
            if (c == 0xED && q[1] > 0x9F) return 0;          // surrogates
            return 3;
        }
        if (c < 0xF5) {
            if (end - q < 4 || (q[1] & 0xC0) != 0x80 || (q[2] & 0xC0) != 0x80 || (q[3] & 0xC0) != 0x80) return 0;
            if (c == 0xF0 && q[1] < 0x90) return 0;
            if (c == 0xF4 && q[1] > 0x8F) return 0;
            return 4;
        }
        return 0;
    }
    // Parses a string; on return [*out, *out + *n) are its decoded bytes: a slice of the
    // input when nothing had to be rewritten (the common case), else of `scratch`.
    bool str(const char** out, size_t* n) {
        if (p >= e || *p != '"') return err(BLANCE_WIRE_ERR_SYNTAX, "expected string");
        p++;
        const char* s = p;
        // fast path: printable ASCII without escapes
        while (p < e) {
            unsigned char c = (unsigned char)*p;
            if (c == '"') { *out = s; *n = (size_t)(p - s); p++; return true; }
            if (c == '\\' || c < 0x20 || c >= 0x80) break;
            p++;
        }
        scratch.assign(s, (size_t)(p - s));
        while (p < e) {
            unsigned char c = (unsigned char)*p;
            if (c == '"') { p++; *out = scratch.data(); *n = scratch.size(); return true; }
            if (c < 0x20) return err(BLANCE_WIRE_ERR_SYNTAX, "control character in string");
            if (c == '\\') {
                if (p + 1 >= e) break;
                char x = p[1];
                p += 2;
                switch (x) {
                    case '"': scratch.push_back('"'); break;
                    case '\\': scratch.push_back('\\'); break;
                    case '/': scratch.push_back('/'); break;
                    case 'b': scratch.push_back('\b'); break;
                    case 'f': scratch.push_back('\f'); break;
                    case 'n': scratch.push_back('\n'); break;
                    case 'r': scratch.push_back('\r'); break;
                    case 't': scratch.push_back('\t'); break;
                    case 'u': {
                        if (e - p < 4) return err(BLANCE_WIRE_ERR_SYNTAX, "short \\u escape");
                        int v = hex4(p);
                        if (v < 0) return err(BLANCE_WIRE_ERR_SYNTAX, "bad \\u escape");
                        p += 4;
                        uint32_t cp = (uint32_t)v;
                        if (cp >= 0xD800 && cp < 0xDC00) {            // high surrogate: needs a low one
                            int lo = -1;
                            if (e - p >= 6 && p[0] == '\\' && p[1] == 'u') lo = hex4(p + 2);
                            if (lo >= 0xDC00 && lo < 0xE000) {
                                cp = 0x10000 + ((cp - 0xD800) << 10) + ((uint32_t)lo - 0xDC00);
                                p += 6;
                            } else {
                                cp = 0xFFFD;
                            }
                        } else if (cp >= 0xDC00 && cp < 0xE000) {
                            cp = 0xFFFD;
                        }
                        put_utf8(scratch, cp);
                        break;
                    }
                    default: return err(BLANCE_WIRE_ERR_SYNTAX, "bad escape");
                }
                continue;
            }
            if (c < 0x80) { scratch.push_back((char)c); p++; continue; }
            int l = utf8_len((const unsigned char*)p, (const unsigned char*)e);
            if (l == 0) { put_utf8(scratch, 0xFFFD); p++; }          // one bad byte -> U+FFFD
            else { scratch.append(p, (size_t)l); p += l; }
        }
        return err(BLANCE_WIRE_ERR_SYNTAX, "unterminated string");
    }
    // skips any JSON value (unknown struct fields)
    bool skip(int depth = 0) {
        if (depth > 10000) return err(BLANCE_WIRE_ERR_SYNTAX, "nesting too deep");
        ws();
        if (p >= e) return err(BLANCE_WIRE_ERR_SYNTAX, "unexpected end");
        char c = *p;
        if (c == '"') { const char* s; size_t n; return str(&s, &n); }
        if (c == '{' || c == '[') {
            const char close = c == '{' ? '}' : ']';
            p++;
            ws();
            if (p < e && *p == close) { p++; return true; }
            for (;;) {
                if (c == '{') {
                    ws();
                    const char* s; size_t n;
                    if (!str(&s, &n)) return false;
                    ws();
                    if (p >= e || *p != ':') return err(BLANCE_WIRE_ERR_SYNTAX, "expected ':'");
                    p++;
                }
                if (!skip(depth + 1)) return false;
                ws();
                if (p >= e) return err(BLANCE_WIRE_ERR_SYNTAX, "unexpected end");
                if (*p == ',') { p++; continue; }
                if (*p == close) { p++; return true; }
                return err(BLANCE_WIRE_ERR_SYNTAX, "expected ',' or close");
            }
        }
        if (c == 't') return lit("true", 4);
        if (c == 'f') return lit("false", 5);
        if (c == 'n') return lit("null", 4);
        if (c == '-' || (c >= '0' && c <= '9')) return number();
        return err(BLANCE_WIRE_ERR_SYNTAX, "unexpected character");
    }
    bool number() {
        if (p < e && *p == '-') p++;
        if (p >= e) return err(BLANCE_WIRE_ERR_SYNTAX, "bad number");
        if (*p == '0') p++;
        else if (*p >= '1' && *p <= '9') { while (p < e && *p >= '0' && *p <= '9') p++; }
        else return err(BLANCE_WIRE_ERR_SYNTAX, "bad number");
        if (p < e && *p == '.') {
            p++;
            if (p >= e || *p < '0' || *p > '9') return err(BLANCE_WIRE_ERR_SYNTAX, "bad number");
            while (p < e && *p >= '0' && *p <= '9') p++;
        }
        if (p < e && (*p == 'e' || *p == 'E')) {
            p++;
            if (p < e && (*p == '+' || *p == '-')) p++;
            if (p >= e || *p < '0' || *p > '9') return err(BLANCE_WIRE_ERR_SYNTAX, "bad number");
            while (p < e && *p >= '0' && *p <= '9') p++;
        }
        return true;
    }
    // a value that should have been of another type: syntax-check it, then report a type error
    bool wrong_type(const char* what) {
        const char* at = p;
        if (!skip()) return false;
        p = at;
        return err(BLANCE_WIRE_ERR_TYPE, what);
    }
    bool is_null() {
        return (e - p) >= 4 && memcmp(p, "null", 4) == 0;
    }
};

bool ieq(const char* s, size_t n, const char* lower) {     // ASCII case-insensitive field match
    size_t m = strlen(lower);
    if (n != m) return false;
    for (size_t i = 0; i < n; i++) {
        char c = s[i];
        if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
        if (c != lower[i]) return false;
    }
    return true;
}

// One partition's state entries while its object is being read (a repeated state name
// replaces the earlier entry, a repeated "nodesByState" merges): kept in a scratch list
// and flushed when the partition object closes.
struct PendingEntry {
    int32_t state;
    uint8_t kind;
    size_t lo, hi;                           // into pending_nodes
};

struct Decoder {
    Parser ps;
    blance_wire_map* m;
    std::vector<PendingEntry> pend;
    std::vector<int32_t> pend_nodes;

    bool list(PendingEntry& en) {            // []string or null
        ps.ws();
        if (ps.p < ps.e && *ps.p == 'n') {
            if (!ps.lit("null", 4)) return false;
            en.kind = BLANCE_WIRE_NIL;
            en.lo = en.hi = pend_nodes.size();
            return true;
        }
        if (ps.p >= ps.e || *ps.p != '[') return ps.wrong_type("state list must be an array of strings or null");
        ps.p++;
        en.kind = BLANCE_WIRE_LIST;
        en.lo = pend_nodes.size();
        ps.ws();
        if (ps.p < ps.e && *ps.p == ']') { ps.p++; en.hi = en.lo; return true; }
        for (;;) {
            ps.ws();
            if (ps.p < ps.e && *ps.p == 'n') {                 // null element leaves the zero value ""
                if (!ps.lit("null", 4)) return false;
                pend_nodes.push_back(m->nodes.intern("", 0));
            } else if (ps.p < ps.e && *ps.p == '"') {
                const char* s; size_t n;
                if (!ps.str(&s, &n)) return false;
                pend_nodes.push_back(m->nodes.intern(s, n));
            } else {
                return ps.wrong_type("node name must be a string");
            }
            ps.ws();
            if (ps.p >= ps.e) return ps.err(BLANCE_WIRE_ERR_SYNTAX, "unexpected end");
            if (*ps.p == ',') { ps.p++; continue; }
            if (*ps.p == ']') { ps.p++; break; }
            return ps.err(BLANCE_WIRE_ERR_SYNTAX, "expected ',' or ']'");
        }
        en.hi = pend_nodes.size();
        return true;
    }

    bool nodes_by_state(uint8_t* kind) {     // map[string][]string or null
        ps.ws();
        if (ps.p < ps.e && *ps.p == 'n') {   // null: the map becomes nil (entries decoded so far are dropped)
            if (!ps.lit("null", 4)) return false;
            *kind = BLANCE_WIRE_NIL;
            pend.clear();
            return true;
        }
        if (ps.p >= ps.e || *ps.p != '{') return ps.wrong_type("nodesByState must be an object or null");
        ps.p++;
        *kind = BLANCE_WIRE_LIST;
        ps.ws();
        if (ps.p < ps.e && *ps.p == '}') { ps.p++; return true; }
        for (;;) {
            ps.ws();
            const char* s; size_t n;
            if (!ps.str(&s, &n)) return false;
            const int32_t st = m->states.intern(s, n);
            ps.ws();
            if (ps.p >= ps.e || *ps.p != ':') return ps.err(BLANCE_WIRE_ERR_SYNTAX, "expected ':'");
            ps.p++;
            PendingEntry en{st, BLANCE_WIRE_NIL, 0, 0};
            if (!list(en)) return false;
            bool replaced = false;
            for (auto& q : pend) if (q.state == st) { q = en; replaced = true; break; }
            if (!replaced) pend.push_back(en);
            ps.ws();
            if (ps.p >= ps.e) return ps.err(BLANCE_WIRE_ERR_SYNTAX, "unexpected end");
            if (*ps.p == ',') { ps.p++; continue; }
            if (*ps.p == '}') { ps.p++; return true; }
            return ps.err(BLANCE_WIRE_ERR_SYNTAX, "expected ',' or '}'");
        }
    }

    // *Partition: object or null.  Appends the partition (key already known) to the map.
    bool partition(std::string& name, uint8_t* kind) {
        ps.ws();
        name.clear();
        pend.clear();
        pend_nodes.clear();
        if (ps.p < ps.e && *ps.p == 'n') {
            if (!ps.lit("null", 4)) return false;
            *kind = BLANCE_WIRE_ABSENT;
            return true;
        }
        if (ps.p >= ps.e || *ps.p != '{') return ps.wrong_type("partition must be an object or null");
        ps.p++;
        *kind = BLANCE_WIRE_NIL;             // NodesByState nil until the field shows up
        ps.ws();
        if (ps.p < ps.e && *ps.p == '}') { ps.p++; return true; }
        for (;;) {
            ps.ws();
            const char* s; size_t n;
            if (!ps.str(&s, &n)) return false;
            const bool is_name = ieq(s, n, "name"), is_nbs = !is_name && ieq(s, n, "nodesbystate");
            ps.ws();
            if (ps.p >= ps.e || *ps.p != ':') return ps.err(BLANCE_WIRE_ERR_SYNTAX, "expected ':'");
            ps.p++;
            ps.ws();
            if (is_name) {
                if (ps.p < ps.e && *ps.p == 'n') {
                    if (!ps.lit("null", 4)) return false;           // null leaves the field as it is
                } else if (ps.p < ps.e && *ps.p == '"') {
                    const char* v; size_t vn;
                    if (!ps.str(&v, &vn)) return false;
                    name.assign(v, vn);
                } else {
                    return ps.wrong_type("name must be a string");
                }
            } else if (is_nbs) {
                if (!nodes_by_state(kind)) return false;
            } else {
                if (!ps.skip()) return false;
            }
            ps.ws();
            if (ps.p >= ps.e) return ps.err(BLANCE_WIRE_ERR_SYNTAX, "unexpected end");
            if (*ps.p == ',') { ps.p++; continue; }
            if (*ps.p == '}') { ps.p++; return true; }
            return ps.err(BLANCE_WIRE_ERR_SYNTAX, "expected ',' or '}'");
        }
    }

    // Partitions are appended in document order; a repeated key replaces the earlier
    // partition (kept at its first position).  Replacement rewrites the tail arrays, so
    // it is handled by decoding into per-partition records first when a duplicate shows up.
    struct PartRec {
        std::string name;
        uint8_t kind;
        std::vector<PendingEntry> entries;
        std::vector<int32_t> nodes;
    };

    bool document() {
        ps.ws();
        if (ps.p < ps.e && *ps.p == 'n') {
            if (!ps.lit("null", 4)) return false;
            m->map_is_nil = 1;
            return tail();
        }
        if (ps.p >= ps.e || *ps.p != '{') return ps.wrong_type("PartitionMap must be an object or null");
        ps.p++;
        Interner keys;                          // key -> partition index
        std::vector<PartRec> replaced;          // only for repeated keys
        std::vector<int64_t> replaced_at;
        ps.ws();
        if (ps.p < ps.e && *ps.p == '}') { ps.p++; return tail(); }
        std::string name;
        for (;;) {
            ps.ws();
            const char* s; size_t n;
            if (!ps.str(&s, &n)) return false;
            const int32_t before = (int32_t)keys.hashes.size();
            const int32_t idx = keys.intern(s, n);
            ps.ws();
            if (ps.p >= ps.e || *ps.p != ':') return ps.err(BLANCE_WIRE_ERR_SYNTAX, "expected ':'");
            ps.p++;
            uint8_t kind;
            if (!partition(name, &kind)) return false;
            if (idx == before) {                 // new key: append
                m->names.push(name.data(), name.size());
                m->part_kind.push_back(kind);
                for (const auto& en : pend) {
                    m->entry_state.push_back(en.state);
                    m->entry_kind.push_back(en.kind);
                    m->entry_nodes.insert(m->entry_nodes.end(), pend_nodes.begin() + (long)en.lo, pend_nodes.begin() + (long)en.hi);
                    m->entry_off.push_back((int64_t)m->entry_nodes.size());
                }
                m->part_off.push_back((int64_t)m->entry_state.size());
            } else {                             // repeated key: remember, patch at the end
                PartRec r;
                r.name = name; r.kind = kind; r.entries = pend; r.nodes = pend_nodes;
                bool found = false;
                for (size_t i = 0; i < replaced_at.size(); i++)
                    if (replaced_at[i] == idx) { replaced[i] = std::move(r); found = true; break; }
                if (!found) { replaced.push_back(std::move(r)); replaced_at.push_back(idx); }
            }
            ps.ws();
            if (ps.p >= ps.e) return ps.err(BLANCE_WIRE_ERR_SYNTAX, "unexpected end");
            if (*ps.p == ',') { ps.p++; continue; }
            if (*ps.p == '}') { ps.p++; break; }
            return ps.err(BLANCE_WIRE_ERR_SYNTAX, "expected ',' or '}'");
        }
        m->keys = std::move(keys.blob);
        if (!replaced.empty()) rebuild(replaced, replaced_at);
        return tail();
    }

    void rebuild(const std::vector<PartRec>& rep, const std::vector<int64_t>& at) {
        blance_wire_map o;
        const int64_t P = m->keys.size();
        std::vector<int> which((size_t)P, -1);
        for (size_t i = 0; i < at.size(); i++) which[(size_t)at[i]] = (int)i;
        for (int64_t i = 0; i < P; i++) {
            if (which[(size_t)i] < 0) {
                size_t n;
                const char* s = m->names.at(i, &n);
                o.names.push(s, n);
                o.part_kind.push_back(m->part_kind[(size_t)i]);
                for (int64_t en = m->part_off[(size_t)i]; en < m->part_off[(size_t)i + 1]; en++) {
                    o.entry_state.push_back(m->entry_state[(size_t)en]);
                    o.entry_kind.push_back(m->entry_kind[(size_t)en]);
                    o.entry_nodes.insert(o.entry_nodes.end(), m->entry_nodes.begin() + m->entry_off[(size_t)en],
                                         m->entry_nodes.begin() + m->entry_off[(size_t)en + 1]);
                    o.entry_off.push_back((int64_t)o.entry_nodes.size());
                }
            } else {
                const PartRec& r = rep[(size_t)which[(size_t)i]];
                o.names.push(r.name.data(), r.name.size());
                o.part_kind.push_back(r.kind);
                for (const auto& en : r.entries) {
                    o.entry_state.push_back(en.state);
                    o.entry_kind.push_back(en.kind);
                    o.entry_nodes.insert(o.entry_nodes.end(), r.nodes.begin() + (long)en.lo, r.nodes.begin() + (long)en.hi);
                    o.entry_off.push_back((int64_t)o.entry_nodes.size());
                }
            }
            o.part_off.push_back((int64_t)o.entry_state.size());
        }
        m->names = std::move(o.names);
        m->part_kind = std::move(o.part_kind);
        m->part_off = std::move(o.part_off);
        m->entry_state = std::move(o.entry_state);
        m->entry_kind = std::move(o.entry_kind);
        m->entry_off = std::move(o.entry_off);
        m->entry_nodes = std::move(o.entry_nodes);
    }

    bool tail() {
        ps.ws();
        if (ps.p != ps.e) return ps.err(BLANCE_WIRE_ERR_SYNTAX, "invalid character after top-level value");
        return true;
    }
};

// ---------------------------------------------------------------- encoder
const char kHex[] = "0123456789abcdef";

void put_string(std::string& o, const char* s, size_t n) {     // encodeState.string, escapeHTML = true
    o.push_back('"');
    size_t start = 0, i = 0;
    const unsigned char* u = (const unsigned char*)s;
    while (i < n) {
        unsigned char c = u[i];
        if (c < 0x80) {
            if (c >= 0x20 && c != '"' && c != '\\' && c != '<' && c != '>' && c != '&') { i++; continue; }
            o.append(s + start, i - start);
            switch (c) {
                case '"': o += "\\\""; break;
                case '\\': o += "\\\\"; break;
                case '\b': o += "\\b"; break;
                case '\f': o += "\\f"; break;
                case '\n': o += "\\n"; break;
                case '\r': o += "\\r"; break;
                case '\t': o += "\\t"; break;
                default:
                    o += "\\u00";
                    o.push_back(kHex[c >> 4]);
                    o.push_back(kHex[c & 0xF]);
            }
            i++;
            start = i;
            continue;
        }
        int l = Parser::utf8_len(u + i, u + n);
        if (l == 0) {
            o.append(s + start, i - start);
            o += "\\ufffd";
            i++;
            start = i;
            continue;
        }
        if (l == 3 && u[i] == 0xE2 && u[i + 1] == 0x80 && (u[i + 2] == 0xA8 || u[i + 2] == 0xA9)) {   // U+2028 / U+2029
            o.append(s + start, i - start);
            o += "\\u202";
            o.push_back(kHex[u[i + 2] & 0xF]);
            i += 3;
            start = i;
            continue;
        }
        i += (size_t)l;
    }
    o.append(s + start, n - start);
    o.push_back('"');
}

struct View {                                  // string i of a blob (blance_wire_encode checks the offsets first)
    const blance_wire_view* v;
    const char* str(const char* bytes, const int64_t* off, int64_t i, size_t* n) const {
        *n = (size_t)(off[i + 1] - off[i]);
        return bytes + off[i];
    }
};

int less_bytes(const char* a, size_t an, const char* b, size_t bn) {
    int c = memcmp(a, b, an < bn ? an : bn);
    if (c) return c;
    return an < bn ? -1 : (an > bn ? 1 : 0);
}

template <class Less>
void sort_idx(std::vector<int64_t>& idx, Less less) {          // small and dependency-free: merge sort
    std::vector<int64_t> tmp(idx.size());
    for (size_t w = 1; w < idx.size(); w *= 2) {
        for (size_t lo = 0; lo < idx.size(); lo += 2 * w) {
            size_t mid = lo + w < idx.size() ? lo + w : idx.size(), hi = lo + 2 * w < idx.size() ? lo + 2 * w : idx.size();
            size_t a = lo, b = mid, k = lo;
            while (a < mid && b < hi) tmp[k++] = less(idx[b], idx[a]) ? idx[b++] : idx[a++];
            while (a < mid) tmp[k++] = idx[a++];
            while (b < hi) tmp[k++] = idx[b++];
        }
        idx.swap(tmp);
    }
}

}  // namespace

extern "C" {

int blance_wire_abi_version(void) { return kAbiVersion; }
const char* blance_wire_last_error(void) { return g_err.c_str(); }

int blance_wire_decode(const char* json, size_t len, blance_wire_map** out) try {
    if (!out || (!json && len)) return fail(BLANCE_WIRE_ERR_ARG, "null argument");
    *out = nullptr;
    std::unique_ptr<blance_wire_map> m(new blance_wire_map());      // (an exception below must not leak a half-built map)
    Decoder d;
    d.m = m.get();
    d.ps.b = d.ps.p = json;
    d.ps.e = json + len;
    if (!d.document()) return fail(d.ps.code ? d.ps.code : BLANCE_WIRE_ERR_SYNTAX, d.ps.msg);
    *out = m.release();
    return BLANCE_WIRE_OK;
} catch (const std::bad_alloc&) {
    return fail(BLANCE_WIRE_ERR_ARG, "out of memory");           // no exception crosses the C boundary
} catch (...) {
    return fail(BLANCE_WIRE_ERR_ARG, "unexpected exception");
}

int blance_wire_view_of(const blance_wire_map* m, blance_wire_view* v) {
    if (!m || !v) return fail(BLANCE_WIRE_ERR_ARG, "null argument");
    memset(v, 0, sizeof *v);
    v->map_is_nil = m->map_is_nil;
    v->n_parts = m->keys.size();
    v->n_states = m->states.blob.size();
    v->n_nodes = m->nodes.blob.size();
    v->n_entries = (int64_t)m->entry_state.size();
    v->n_node_refs = (int64_t)m->entry_nodes.size();
    v->key_bytes = m->keys.bytes.data();   v->key_off = m->keys.off.data();
    v->name_bytes = m->names.bytes.data(); v->name_off = m->names.off.data();
    v->part_kind = m->part_kind.data();
    v->part_off = m->part_off.data();
    v->state_bytes = m->states.blob.bytes.data(); v->state_off = m->states.blob.off.data();
    v->node_bytes = m->nodes.blob.bytes.data();   v->node_off = m->nodes.blob.off.data();
    v->entry_state = m->entry_state.data();
    v->entry_kind = m->entry_kind.data();
    v->entry_off = m->entry_off.data();
    v->entry_nodes = m->entry_nodes.data();
    return BLANCE_WIRE_OK;
}

void blance_wire_free(blance_wire_map* m) { delete m; }
void blance_wire_free_bytes(char* p) { free(p); }

}  // extern "C"

namespace {

// json.Marshal of the view into o (the view is checked first; nothing is read through a bad offset)
int encode_to(const blance_wire_view* v, std::string& o) {
    if (v->map_is_nil) {
        o = "null";
    } else {
        const int64_t P = v->n_parts;
        // the view is the caller's: sizes, pointers and offset arrays are checked before anything is read through them
        if (P < 0 || v->n_states < 0 || v->n_nodes < 0 || v->n_entries < 0 || v->n_node_refs < 0)
            return fail(BLANCE_WIRE_ERR_ARG, "negative size");
        if (!v->key_off || !v->name_off || !v->part_off || !v->state_off || !v->node_off || !v->entry_off ||
            (P > 0 && (!v->key_bytes || !v->name_bytes || !v->part_kind)) ||
            (v->n_entries > 0 && (!v->entry_state || !v->entry_kind)) || (v->n_node_refs > 0 && !v->entry_nodes) ||
            (v->n_states > 0 && !v->state_bytes) || (v->n_nodes > 0 && !v->node_bytes))
            return fail(BLANCE_WIRE_ERR_ARG, "null array in the view");
        auto monotone = [](const int64_t* off, int64_t n) {
            if (off[0] != 0) return false;
            for (int64_t i = 0; i < n; i++) if (off[i + 1] < off[i]) return false;
            return true;
        };
        if (!monotone(v->key_off, P) || !monotone(v->name_off, P) || !monotone(v->part_off, P) ||
            !monotone(v->state_off, v->n_states) || !monotone(v->node_off, v->n_nodes) || !monotone(v->entry_off, v->n_entries))
            return fail(BLANCE_WIRE_ERR_ARG, "offsets do not start at 0 or are not monotone");
        if (v->part_off[P] > v->n_entries || v->entry_off[v->n_entries] > v->n_node_refs)
            return fail(BLANCE_WIRE_ERR_ARG, "offsets run past the arrays they index");
        for (int64_t e = 0; e < v->n_entries; e++)
            if (v->entry_state[e] < 0 || v->entry_state[e] >= v->n_states) return fail(BLANCE_WIRE_ERR_ARG, "state id out of range");
        for (int64_t r = 0; r < v->n_node_refs; r++)
            if (v->entry_nodes[r] < 0 || v->entry_nodes[r] >= v->n_nodes) return fail(BLANCE_WIRE_ERR_ARG, "node id out of range");
        o.reserve((size_t)(v->key_off[P] + v->name_off[P]) * 2 + (size_t)v->n_node_refs * 10 + (size_t)P * 40 + 16);
        View vw{v};
        std::vector<int64_t> order((size_t)P);
        for (int64_t i = 0; i < P; i++) order[(size_t)i] = i;
        bool sorted = true;                      // already in key order? (the planner's output is)
        for (int64_t i = 1; i < P && sorted; i++) {
            size_t an, bn;
            const char* a = vw.str(v->key_bytes, v->key_off, i - 1, &an);
            const char* b = vw.str(v->key_bytes, v->key_off, i, &bn);
            if (less_bytes(a, an, b, bn) >= 0) sorted = false;
        }
        if (!sorted)
            sort_idx(order, [&](int64_t x, int64_t y) {
                size_t an, bn;
                const char* a = vw.str(v->key_bytes, v->key_off, x, &an);
                const char* b = vw.str(v->key_bytes, v->key_off, y, &bn);
                return less_bytes(a, an, b, bn) < 0;
            });
        // rank of every state name, so a partition's few entries sort by an integer
        std::vector<int64_t> sorder((size_t)v->n_states), srank((size_t)v->n_states);
        for (int64_t i = 0; i < v->n_states; i++) sorder[(size_t)i] = i;
        sort_idx(sorder, [&](int64_t x, int64_t y) {
            size_t an, bn;
            const char* a = vw.str(v->state_bytes, v->state_off, x, &an);
            const char* b = vw.str(v->state_bytes, v->state_off, y, &bn);
            return less_bytes(a, an, b, bn) < 0;
        });
        for (int64_t i = 0; i < v->n_states; i++) srank[(size_t)sorder[(size_t)i]] = i;
        std::vector<int64_t> ents;
        o.push_back('{');
        for (int64_t oi = 0; oi < P; oi++) {
            const int64_t i = order[(size_t)oi];
            if (oi) o.push_back(',');
            size_t n;
            const char* s = vw.str(v->key_bytes, v->key_off, i, &n);
            put_string(o, s, n);
            o.push_back(':');
            if (v->part_kind[i] == BLANCE_WIRE_ABSENT) { o += "null"; continue; }
            o += "{\"name\":";
            s = vw.str(v->name_bytes, v->name_off, i, &n);
            put_string(o, s, n);
            o += ",\"nodesByState\":";
            if (v->part_kind[i] == BLANCE_WIRE_NIL) {
                o += "null";
            } else {
                ents.clear();
                for (int64_t e = v->part_off[i]; e < v->part_off[i + 1]; e++) ents.push_back(e);
                for (size_t a = 1; a < ents.size(); a++) {          // insertion sort by state name rank
                    int64_t x = ents[a];
                    size_t b = a;
                    while (b > 0 && srank[(size_t)v->entry_state[ents[b - 1]]] > srank[(size_t)v->entry_state[x]]) {
                        ents[b] = ents[b - 1];
                        b--;
                    }
                    ents[b] = x;
                }
                o.push_back('{');
                for (size_t a = 0; a < ents.size(); a++) {
                    const int64_t e = ents[a];
                    if (a) o.push_back(',');
                    s = vw.str(v->state_bytes, v->state_off, v->entry_state[e], &n);
                    put_string(o, s, n);
                    o.push_back(':');
                    if (v->entry_kind[e] != BLANCE_WIRE_LIST) { o += "null"; continue; }
                    o.push_back('[');
                    for (int64_t r = v->entry_off[e]; r < v->entry_off[e + 1]; r++) {
                        if (r > v->entry_off[e]) o.push_back(',');
                        s = vw.str(v->node_bytes, v->node_off, v->entry_nodes[r], &n);
                        put_string(o, s, n);
                    }
                    o.push_back(']');
                }
                o.push_back('}');
            }
            o.push_back('}');
        }
        o.push_back('}');
    }
    return BLANCE_WIRE_OK;
}

}  // namespace

extern "C" {

int blance_wire_encode(const blance_wire_view* v, char** out_json, size_t* out_len) try {
    if (!v || !out_json || !out_len) return fail(BLANCE_WIRE_ERR_ARG, "null argument");
    std::string o;
    const int st = encode_to(v, o);
    if (st) return st;
    char* buf = (char*)malloc(o.size() + 1);
    if (!buf) return fail(BLANCE_WIRE_ERR_ARG, "out of memory");
    memcpy(buf, o.data(), o.size());
    buf[o.size()] = 0;
    *out_json = buf;
    *out_len = o.size();
    return BLANCE_WIRE_OK;
} catch (const std::bad_alloc&) {
    return fail(BLANCE_WIRE_ERR_ARG, "out of memory");           // no exception crosses the C boundary
} catch (...) {
    return fail(BLANCE_WIRE_ERR_ARG, "unexpected exception");
}

int blance_wire_encode_into(const blance_wire_view* v, char* buf, size_t cap, size_t* need) try {
    if (!v || !need || (!buf && cap)) return fail(BLANCE_WIRE_ERR_ARG, "null argument");
    std::string o;
    const int st = encode_to(v, o);
    if (st) return st;
    *need = o.size();
    if (o.size() > cap) return fail(BLANCE_WIRE_ERR_SPACE, "the caller's buffer is too small (see *need)");
    memcpy(buf, o.data(), o.size());
    return BLANCE_WIRE_OK;
} catch (const std::bad_alloc&) {
    return fail(BLANCE_WIRE_ERR_ARG, "out of memory");           // no exception crosses the C boundary
} catch (...) {
    return fail(BLANCE_WIRE_ERR_ARG, "unexpected exception");
}

int blance_wire_decode_into(const char* json, size_t len, blance_wire_buffers* b, blance_wire_view* view) try {
    if (!b || !view) return fail(BLANCE_WIRE_ERR_ARG, "null argument");
    blance_wire_map* m_raw = nullptr;
    int st = blance_wire_decode(json, len, &m_raw);
    if (st) return st;
    std::unique_ptr<blance_wire_map> m_owner(m_raw);                // freed on every way out, exceptions included
    blance_wire_map* m = m_raw;
    blance_wire_view v;
    blance_wire_view_of(m, &v);
    const int64_t kb = v.n_parts ? v.key_off[v.n_parts] : 0, nb = v.n_parts ? v.name_off[v.n_parts] : 0;
    const int64_t sb = v.n_states ? v.state_off[v.n_states] : 0, db = v.n_nodes ? v.node_off[v.n_nodes] : 0;
    const bool fits = b->cap_parts >= v.n_parts && b->cap_states >= v.n_states && b->cap_nodes >= v.n_nodes &&
                      b->cap_entries >= v.n_entries && b->cap_node_refs >= v.n_node_refs && b->cap_key_bytes >= kb &&
                      b->cap_name_bytes >= nb && b->cap_state_bytes >= sb && b->cap_node_bytes >= db;
    const bool have = b->key_off && b->name_off && b->part_off && b->state_off && b->node_off && b->entry_off &&
                      (!v.n_parts || (b->part_kind && (!kb || b->key_bytes) && (!nb || b->name_bytes))) &&
                      (!sb || b->state_bytes) && (!db || b->node_bytes) &&
                      (!v.n_entries || (b->entry_state && b->entry_kind)) && (!v.n_node_refs || b->entry_nodes);
    // what the document needs, whether or not it fits
    b->cap_parts = v.n_parts; b->cap_states = v.n_states; b->cap_nodes = v.n_nodes; b->cap_entries = v.n_entries;
    b->cap_node_refs = v.n_node_refs; b->cap_key_bytes = kb; b->cap_name_bytes = nb; b->cap_state_bytes = sb; b->cap_node_bytes = db;
    if (!fits) return fail(BLANCE_WIRE_ERR_SPACE, "the caller's arrays are too small (sizes written to cap_*)");
    if (!have) return fail(BLANCE_WIRE_ERR_ARG, "null array in the buffers");
    auto put = [](void* dst, const void* src, size_t n) { if (n) memcpy(dst, src, n); };
    put(b->key_bytes, v.key_bytes, (size_t)kb);     put(b->key_off, v.key_off, sizeof(int64_t) * (size_t)(v.n_parts + 1));
    put(b->name_bytes, v.name_bytes, (size_t)nb);   put(b->name_off, v.name_off, sizeof(int64_t) * (size_t)(v.n_parts + 1));
    put(b->part_kind, v.part_kind, (size_t)v.n_parts);
    put(b->part_off, v.part_off, sizeof(int64_t) * (size_t)(v.n_parts + 1));
    put(b->state_bytes, v.state_bytes, (size_t)sb); put(b->state_off, v.state_off, sizeof(int64_t) * (size_t)(v.n_states + 1));
    put(b->node_bytes, v.node_bytes, (size_t)db);   put(b->node_off, v.node_off, sizeof(int64_t) * (size_t)(v.n_nodes + 1));
    put(b->entry_state, v.entry_state, sizeof(int32_t) * (size_t)v.n_entries);
    put(b->entry_kind, v.entry_kind, (size_t)v.n_entries);
    put(b->entry_off, v.entry_off, sizeof(int64_t) * (size_t)(v.n_entries + 1));
    put(b->entry_nodes, v.entry_nodes, sizeof(int32_t) * (size_t)v.n_node_refs);
    *view = v;
    view->key_bytes = b->key_bytes;     view->key_off = b->key_off;
    view->name_bytes = b->name_bytes;   view->name_off = b->name_off;
    view->part_kind = b->part_kind;     view->part_off = b->part_off;
    view->state_bytes = b->state_bytes; view->state_off = b->state_off;
    view->node_bytes = b->node_bytes;   view->node_off = b->node_off;
    view->entry_state = b->entry_state; view->entry_kind = b->entry_kind;
    view->entry_off = b->entry_off;     view->entry_nodes = b->entry_nodes;
    return BLANCE_WIRE_OK;
} catch (const std::bad_alloc&) {
    return fail(BLANCE_WIRE_ERR_ARG, "out of memory");           // no exception crosses the C boundary
} catch (...) {
    return fail(BLANCE_WIRE_ERR_ARG, "unexpected exception");
}

}  // extern "C"
