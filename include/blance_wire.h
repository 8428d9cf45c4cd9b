/* blance_wire.h -- streaming codec for the PartitionMap JSON wire format
 * (SURVEY.md section 8(f) rank 2).  Host-side C ABI, no device code.
 *
 * Replaces, on the caller's side of PlanNextMap, encoding/json's reflection walk
 * over  map[string]*Partition  (reference api.go:24-36: struct tags `name`,
 * `nodesByState`): one pass over the bytes straight into the interned form the
 * planner's cgo shim needs (dense node / state ids, CSR lists), and the inverse.
 *
 * Document shape:  { "<key>": null | { "name": "<str>", "nodesByState": null |
 *                    { "<state>": null | [ "<node>", ... ], ... } }, ... }
 *
 * Decoding follows encoding/json.Unmarshal into a nil PartitionMap:
 *   - struct fields match exactly or ASCII-case-insensitively, unknown fields are skipped;
 *   - a repeated map key (partition key, state name) replaces the earlier value;
 *     a repeated "nodesByState" field merges into the map decoded so far;
 *   - null: partition -> nil pointer, nodesByState -> nil map, state list -> nil slice,
 *     list element -> "" , name -> left as it is;
 *   - strings: \uXXXX escapes incl. surrogate pairs, unpaired surrogates and invalid
 *     UTF-8 become U+FFFD;
 *   - a value of the wrong JSON type, or malformed JSON, is an error (no partial result).
 * Encoding follows encoding/json.Marshal (Go >= 1.22): map keys sorted bytewise, struct
 * fields in declaration order, compact, HTML-safe escapes (< > & as \u003c \u003e \u0026),
 * U+2028 / U+2029 as \u2028 / \u2029, control characters as \b \f \n \r \t or \u00XX,
 * invalid UTF-8 bytes as \ufffd.
 *
 * Two forms of every direction.  Caller-owned (the planner ABI's rule, blance_hip.h): blance_wire_decode_into
 * fills arrays the caller brought, blance_wire_encode_into a byte buffer of the caller's; a too small buffer is
 * reported with the sizes needed and nothing of the library's crosses the boundary.  Handle-based (one pass, no
 * size guess): blance_wire_decode returns a blance_wire_map whose arrays live until blance_wire_free(),
 * blance_wire_encode bytes to release with blance_wire_free_bytes().  Lists are kept per (partition, state entry)
 * in document order.
 */
#ifndef BLANCE_WIRE_H
#define BLANCE_WIRE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BLANCE_WIRE_OK 0
#define BLANCE_WIRE_ERR_SYNTAX (-1)   /* malformed JSON */
#define BLANCE_WIRE_ERR_TYPE (-2)     /* a value of the wrong JSON type (UnmarshalTypeError) */
#define BLANCE_WIRE_ERR_ARG (-3)
#define BLANCE_WIRE_ERR_SPACE (-4)    /* a caller's buffer is too small; the sizes needed are reported */

#define BLANCE_WIRE_ABSENT 0          /* same values as BLANCE_LIST_* of blance_hip.h */
#define BLANCE_WIRE_NIL 1
#define BLANCE_WIRE_LIST 2

typedef struct blance_wire_map blance_wire_map;

/* A decoded (or to-be-encoded) PartitionMap, struct of arrays.  String tables are byte
 * blobs with n+1 offsets.  Partition i owns state entries [part_off[i], part_off[i+1]);
 * entry e is state entry_state[e] with kind entry_kind[e] (NIL or LIST) and the node ids
 * entry_nodes[entry_off[e] .. entry_off[e+1]). */
typedef struct blance_wire_view {
    int32_t map_is_nil;               /* the document was `null` */
    int64_t n_parts, n_states, n_nodes, n_entries, n_node_refs;
    const char* key_bytes;   const int64_t* key_off;     /* [n_parts + 1] map keys, document order */
    const char* name_bytes;  const int64_t* name_off;    /* [n_parts + 1] Partition.Name */
    const uint8_t* part_kind;                            /* [n_parts] ABSENT = nil *Partition, NIL = nil NodesByState, LIST = map */
    const int64_t* part_off;                             /* [n_parts + 1] */
    const char* state_bytes; const int64_t* state_off;   /* [n_states + 1] interned state names, first seen order */
    const char* node_bytes;  const int64_t* node_off;    /* [n_nodes + 1] interned node names, first seen order */
    const int32_t* entry_state;                          /* [n_entries] */
    const uint8_t* entry_kind;                           /* [n_entries] NIL or LIST */
    const int64_t* entry_off;                            /* [n_entries + 1] */
    const int32_t* entry_nodes;                          /* [n_node_refs] */
} blance_wire_view;

/* json.Unmarshal(data, &PartitionMap{}): replaces reflect-driven decoding + a per-string
 * allocation with one pass and two hash tables.  *out must be freed with blance_wire_free. */
int blance_wire_decode(const char* json, size_t len, blance_wire_map** out);
int blance_wire_view_of(const blance_wire_map* m, blance_wire_view* view);
void blance_wire_free(blance_wire_map* m);

/* json.Marshal(PartitionMap): *out_json is malloc'ed by the library, release it with
 * blance_wire_free_bytes.  The view's arrays are the caller's. */
int blance_wire_encode(const blance_wire_view* view, char** out_json, size_t* out_len);
void blance_wire_free_bytes(char* p);

/* Caller-owned storage for a decoded map: capacities (elements / bytes) in, the document's sizes out -- also when
 * the call fails with BLANCE_WIRE_ERR_SPACE, so the second call fits.  Offset arrays hold one element more than
 * their capacity says (n + 1 offsets).  A document of len bytes never needs more than len of anything. */
typedef struct blance_wire_buffers {
    int64_t cap_parts, cap_states, cap_nodes, cap_entries, cap_node_refs;
    int64_t cap_key_bytes, cap_name_bytes, cap_state_bytes, cap_node_bytes;
    char* key_bytes;   int64_t* key_off;
    char* name_bytes;  int64_t* name_off;
    uint8_t* part_kind;
    int64_t* part_off;
    char* state_bytes; int64_t* state_off;
    char* node_bytes;  int64_t* node_off;
    int32_t* entry_state;
    uint8_t* entry_kind;
    int64_t* entry_off;
    int32_t* entry_nodes;
} blance_wire_buffers;

/* json.Unmarshal into the caller's arrays; *view then points into them.  Cost: the document is parsed into a map
 * of the library's first and copied over (peak memory: both), and a BLANCE_WIRE_ERR_SPACE answer means a second
 * parse after the caller has grown its arrays to the sizes reported -- callers that can take a handle
 * (blance_wire_decode) pay for one parse and no copy.  No exception crosses any entry point of this header. */
int blance_wire_decode_into(const char* json, size_t len, blance_wire_buffers* buffers, blance_wire_view* view);
/* json.Marshal into the caller's buffer; *need = the document's length, also with BLANCE_WIRE_ERR_SPACE. */
int blance_wire_encode_into(const blance_wire_view* view, char* buf, size_t cap, size_t* need);

const char* blance_wire_last_error(void);   /* thread local, with the byte offset of a syntax error */
int blance_wire_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
