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
 * All arrays returned by the accessors are owned by the blance_wire_map and live until
 * blance_wire_free().  Lists are kept per (partition, state entry) in document order.
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

const char* blance_wire_last_error(void);   /* thread local, with the byte offset of a syntax error */
int blance_wire_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
