/*
 * blance_hip.h -- C ABI of the MI355X-native PlanNextMap planner.
 *
 * This is the drop-in boundary for couchbase/blance's planner path.  The
 * reference has no FFI; the seam this ABI replaces is the call
 *
 *     PlanNextMapEx(...) -> planNextMapEx(...)        api.go:147-157, plan.go:23-58
 *
 * i.e. the whole convergence loop (plan.go:32-56) with every greedy sweep
 * (planNextMapInnerEx, plan.go:60-331) runs behind ONE call of blance_plan().
 * The host side (Go over cgo, or the C++/Python mirrors in this repo) interns
 * strings to dense ids, flattens the maps to the int32 SoA arrays below, calls
 * blance_plan(), un-interns the result and replays the caller-visible
 * mutations of plan.go:49-55.  See INTEGRATION.md for the cgo stub.
 *
 * ABI rules: plain C, caller owns every buffer, nothing is retained after a
 * call returns (cgo pointer rule), no exceptions cross the boundary, every
 * entry point returns an int status (0 = ok, <0 = error).  The planning entry
 * points take no callbacks; the one function-pointer table of this header is
 * the struct blance_comm: an embedder's own collectives for a plan sharded over several
 * ranks; a production multi-GPU caller uses blance_comm_init_rccl instead
 * and passes none.
 *
 * Id spaces
 *   node id      0..n_nodes-1 = position in nodesAll (plan.go:72-75; names must
 *                be unique).  n_nodes..n_nodes_ext-1 = names that occur only in
 *                partition lists / nodesToRemove / nodesToAdd / NodeWeights:
 *                they are counted and filtered but are never candidates
 *                (candidates come from nodesAll, plan.go:77,:142).
 *   state id     0..n_states-1 = the model's states in sortStateNames() order
 *                (plan.go:437-474), which is the pass order of plan.go:307.
 *                Pseudo state id n_states ("other") carries loads of state
 *                names that are not in the model; they only feed
 *                nodePartitionCounts (plan.go:118-124).
 *   partition id 0..n_parts-1 = the partitions of partitionsToAssign, any order.
 *   vertex id    hierarchy vertices: 0..n_nodes_ext-1 are the nodes, the rest
 *                are interior names of NodeHierarchy plus the "" vertex that
 *                findAncestor() yields past a root (plan.go:755-762).
 */
#ifndef BLANCE_HIP_H
#define BLANCE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BLANCE_ABI_VERSION 6

/* status codes */
#define BLANCE_OK                0
#define BLANCE_ERR_BAD_ARG      -1  /* NULL / inconsistent sizes / id out of range      */
#define BLANCE_ERR_UNSUPPORTED  -2  /* input outside the supported envelope             */
#define BLANCE_ERR_CAPACITY     -3  /* caller's output buffers too small                */
#define BLANCE_ERR_DEVICE       -4  /* HIP runtime failure (see blance_last_error)      */
#define BLANCE_ERR_NO_DEVICE    -5  /* no gfx950 device visible                         */
#define BLANCE_ERR_COMM         -6  /* RCCL / caller collective failure                 */

/* list kinds: Go distinguishes a missing map key, a nil slice and an empty
 * non-nil slice under reflect.DeepEqual (plan.go:38). */
#define BLANCE_LIST_ABSENT 0
#define BLANCE_LIST_NIL    1
#define BLANCE_LIST_SET    2

/* NodeScoreBooster (plan.go:691-697) cannot be an arbitrary Go callback on the
 * device; the one booster known in the wild (couchbase/cbgt, restated in
 * control_test.go:19-26: max(float64(-w), stickiness)) is a built-in. */
#define BLANCE_BOOSTER_NONE 0
#define BLANCE_BOOSTER_CBGT 1

/* engines (blance_options.engine) */
#define BLANCE_ENGINE_AUTO        0 /* fastest exact schedule available                 */
#define BLANCE_ENGINE_SEQUENTIAL  1 /* one in-order step at a time (always exact)       */

typedef struct blance_problem {
    /* ---- sizes -------------------------------------------------------- */
    int32_t n_nodes;          /* N  = len(nodesAll)                                      */
    int32_t n_nodes_ext;      /* NX >= N                                                 */
    int32_t n_states;         /* M  = len(model)                                         */
    int32_t n_parts;          /* P  = len(partitionsToAssign)                            */
    int32_t n_prev;           /* len(prevMap): NumPartitions of sweep 1 (plan.go:161)    */
    int32_t n_loads;          /* entries in load_*                                       */
    int32_t n_rules;          /* entries in rule_inc/rule_exc                            */
    int32_t n_vertices;       /* VX (0 when hierarchy_rules_nil)                         */
    int32_t max_iterations;   /* MaxIterationsPerPlan (plan.go:21)                       */

    /* ---- flags -------------------------------------------------------- */
    int32_t partition_weights_nil;  /* opts.PartitionWeights == nil (plan.go:105,:270)   */
    int32_t nodes_to_add_nil;       /* nodesToAdd == nil (plan.go:554)                   */
    int32_t hierarchy_rules_nil;    /* opts.HierarchyRules == nil (plan.go:174)          */
    int32_t booster_kind;           /* BLANCE_BOOSTER_*                                  */
    int32_t top_state;              /* state id of minimum Priority (plan.go:126-132)    */

    /* ---- model, by state id ------------------------------------------ */
    const int32_t* state_priority;        /* [M] PartitionModelState.Priority            */
    const int32_t* state_constraints;     /* [M] effective k: ModelStateConstraints
                                                 override applied (plan.go:308-319)      */
    const int32_t* state_stickiness;      /* [M] opts.StateStickiness[state]             */
    const uint8_t* state_has_stickiness;  /* [M] key present (0 everywhere if map nil)   */

    /* ---- nodes, by node id ------------------------------------------- */
    const uint8_t* node_removed;      /* [NX] in nodesToRemove                           */
    const uint8_t* node_added;        /* [NX] in nodesToAdd                              */
    const int32_t* node_weight;       /* [NX] opts.NodeWeights[node]                     */
    const uint8_t* node_has_weight;   /* [NX] key present (0 everywhere if map nil)      */

    /* ---- partitions of partitionsToAssign, by partition id ------------ */
    const int32_t* part_order;        /* [P] partition ids ordered by the static part of
                                             partitionSorter's key (weight desc, numeric
                                             name, name), plan.go:519-540; the per-pass
                                             category "0/1/2" (plan.go:542-561) is
                                             applied on the device as a stable 3-way
                                             partition of this order                     */
    const int32_t* part_weight;       /* [P] opts.PartitionWeights[name]                 */
    const uint8_t* part_has_weight;   /* [P]                                             */
    const uint8_t* part_in_prev;      /* [P] name is a key of prevMap                    */
    const uint8_t* part_prev_never_equal; /* [P] prevMap[name] can never DeepEqual a
                                             result (nil NodesByState map, non-model
                                             state keys, Name != key)                    */
    /* partitionsToAssign[p].NodesByState[state], CSR over (p * M + state) */
    const int32_t* assign_off;        /* [P*M + 1]                                       */
    const int32_t* assign_nodes;      /* [assign_off[P*M]] node ids, list order kept     */
    const uint8_t* assign_kind;       /* [P*M] BLANCE_LIST_*                             */
    /* prevMap[name].NodesByState[state] for the same partitions (empty when
     * !part_in_prev) */
    const int32_t* prev_off;          /* [P*M + 1]                                       */
    const int32_t* prev_nodes;
    const uint8_t* prev_kind;         /* [P*M]                                           */

    /* ---- extra prevMap loads (countStateNodes, plan.go:374-399) -------- */
    /* One entry per (partition, state, node) occurrence that the CSR above does
     * not carry: partitions that are only in prevMap, and non-model states. */
    const int32_t* load_state;        /* [n_loads] 0..M (M = "other")                    */
    const int32_t* load_node;         /* [n_loads] node id                               */
    const int32_t* load_weight;       /* [n_loads] the partition's weight                */
    const uint8_t* load_first_sweep_only; /* [n_loads] entry belongs to a partition that
                                             is also in partitionsToAssign, so sweep 1's
                                             write-back (plan.go:49-52) replaces it      */

    /* ---- hierarchy (plan.go:703-774) ---------------------------------- */
    const int32_t* rule_off;          /* [M + 1] rules of state s: rule_off[s]..[s+1]    */
    const int32_t* rule_inc;          /* [n_rules] HierarchyRule.IncludeLevel            */
    const int32_t* rule_exc;          /* [n_rules] HierarchyRule.ExcludeLevel            */
    int32_t        vertex_empty;      /* vertex id of ""                                 */
    const int32_t* vertex_parent;     /* [VX] NodeHierarchy[v], vertex_empty if missing  */
    const int32_t* vertex_leaf_lo;    /* [VX] leaves(v) = leaf positions [lo, hi) in a   */
    const int32_t* vertex_leaf_hi;    /* [VX] DFS over children sorted by name           */
    const int32_t* node_leaf_pos;     /* [NX] leaf position of the node, -1 if the node
                                             has children (it is then never a leaf)      */
} blance_problem;

typedef struct blance_result {
    /* nextMap[p].NodesByState[state], CSR over (p * M + state); caller allocates */
    int32_t* out_off;         /* [P*M + 1]                                               */
    int32_t* out_nodes;       /* [out_capacity]                                          */
    uint8_t* out_kind;        /* [P*M]                                                   */
    int64_t  out_capacity;    /* in: capacity of out_nodes (blance_result_capacity())    */
    /* warnings of the last sweep (plan.go:70,:231-235): one (partition, state)
     * pair per message, grouped by pass, in processing order */
    int32_t* warn_part;       /* [warn_capacity]                                         */
    int32_t* warn_state;      /* [warn_capacity]                                         */
    int64_t  warn_capacity;   /* in: P*M always suffices                                 */
    int64_t  n_warnings;      /* out                                                     */
    int32_t  iterations;      /* out: sweeps run (1..max_iterations)                     */
    int32_t  converged;       /* out: loop left through plan.go:43-45                    */
    /* out: timings of this call */
    double   device_ms;       /* first kernel -> last kernel, hipEvents                  */
    double   total_ms;        /* incl. H2D / D2H of problem and result                   */
    /* out: engine statistics (steps = findBestNodes calls) */
    int64_t  steps_total;
    int64_t  steps_sequential;    /* steps resolved one at a time                        */
    int64_t  steps_batched;       /* steps resolved by an exact parallel schedule        */
    int64_t  kernel_launches;
    /* out: the dominant kernel -- the state-pass kernel (k_pass_chain, or
     * k_pass_seq over a whole pass), one launch per hierarchy-rule state pass --
     * timed with hipEvents on the planner's stream: sum of its launch durations
     * and launch count; and the same for passes run by the flat bulk driver
     * (several small kernels + host round trips per pass) */
    double   pass_kernel_ms;
    int64_t  pass_kernel_launches;
    double   flat_pass_ms;
    int64_t  flat_passes;
    /* out: of pass_kernel_ms / pass_kernel_launches, the part of the all-blank chain kernel
     * (k_pass_chain_planes / k_pass_chain_blank: the first hierarchy pass of a fresh plan) */
    double   blank_pass_ms;
    int64_t  blank_pass_launches;
    /* out (ABI 5): of pass_kernel_ms / pass_kernel_launches, the part of chain passes that k_stay_by_top verified as stays
     * (one thread per top priority node, incl. the grouping by top node in front of it) -- what is left after this and the
     * all-blank part is k_pass_chain (hierarchy-rule states) or k_pass_queue / k_pass_tree / k_pass_seq (flat states) */
    double   stay_pass_ms;
    int64_t  stay_pass_launches;
    /* out (ABI 6): stream synchronisations the host made inside this plan (each one reads a few flag words that decide what
     * is launched next; the device idles for the round trip) */
    int64_t  host_syncs;
} blance_result;

typedef struct blance_options {
    int32_t engine;           /* BLANCE_ENGINE_*                                         */
    int32_t device_id;        /* HIP device ordinal                                      */
    int32_t reserved[6];      /* zero in production.  Test knobs: [0] workgroup size of k_pass_seq (64 / 256 / 512 /
                               * 1024), [1] smallest pass handed to the bulk engines, [2] & 1 = k_pass_seq without
                               * verified-stay speculation (further bits of [2]: blance_amd/hip.py).
                               * & 256 = the all-blank chain pass WITHOUT its periodic form (csrc/k_period.h, on by
                               * default since round 4; the environment's BLANCE_PERIODIC=0 does the same);
                               * & 16384 = a communicator of ONE rank takes the sharded branch of every chain pass:
                               * both collectives of "one plan on several GPUs" below execute (how ncclAllReduce /
                               * ncclAllGather are exercised and timed on a one-GPU machine)                      */
} blance_options;

typedef struct blance_ctx blance_ctx;   /* opaque: device buffers, stream, events */

/* Upper bound on out_nodes entries for a problem: sum over (p, state) of
 * max(constraints, len(assign list)). */
int64_t blance_result_capacity(const blance_problem* pb);

/* Create / destroy a planner context bound to one gfx950 device. */
int blance_ctx_create(const blance_options* opt, blance_ctx** out);
void blance_ctx_destroy(blance_ctx* ctx);

/* Run planNextMapEx (plan.go:23-58) for one problem.  Host buffers in, host
 * buffers out.  Re-entrant per context (calls on one ctx are serialised). */
int blance_plan(blance_ctx* ctx, const blance_problem* pb, blance_result* res);

/* Device-resident variant used by the benchmark: upload once, plan many times,
 * download once.  blance_plan() == upload + plan_resident + download. */
int blance_upload(blance_ctx* ctx, const blance_problem* pb);
int blance_plan_resident(blance_ctx* ctx, blance_result* res /* timings+stats only */);
int blance_download(blance_ctx* ctx, blance_result* res);

/* ---- host buffers the device reaches at link speed (ABI 5) ------------------------------------
 * the caller owns every buffer.  blance_host_alloc() hands out page-locked host
 * memory that stays the caller's (free it with blance_host_free(); freed blocks are kept for the
 * next allocation, pinning costs milliseconds).  The arrays of a blance_problem / blance_result
 * that lie in such memory -- or in memory the caller registered with the HIP runtime itself -- are
 * copied by DMA where they lie; every other (pageable) array passes through a page-locked buffer
 * of the context, moved by a few host threads.  Either way nothing of the caller's is touched
 * after the call returns.  NULL when the runtime cannot provide the memory (use malloc then). */
void* blance_host_alloc(size_t bytes);
void blance_host_free(void* p);
/* (ABI 6) Freed blocks are cached for the next blance_host_alloc -- up to 2 GiB of page-locked, unswappable memory per
 * process.  blance_host_trim() returns every cached block to the system now; the destruction of a process's last context
 * does the same.  Blocks the caller still holds are not touched. */
void blance_host_trim(void);

/* ---- CalcPartitionMoves for every partition at once (moves.go:41-136) ---------
 * The planner's consumer (orchestrate.go:273-287 calls it per partition): the
 * ordered node-by-node state transitions from begMap[p] to endMap[p].  Per
 * partition independent.  State ids 0..n_states-1 follow the `states` argument
 * (superior first); pseudo state n_states collects the nodes of map keys that
 * are not in `states` (they only feed flattenNodesByState, plan.go:425-431). */
#define BLANCE_OP_ADD      0
#define BLANCE_OP_DEL      1
#define BLANCE_OP_PROMOTE  2
#define BLANCE_OP_DEMOTE   3

typedef struct blance_moves_problem {
    int32_t n_parts;
    int32_t n_states;             /* M = len(states)                                      */
    int32_t favor_min_nodes;      /* favorMinNodes                                        */
    const int32_t* beg_off;       /* [P*(M+1) + 1] CSR of begNodesByState over p*(M+1)+s  */
    const int32_t* beg_nodes;     /* node ids (any non-negative numbering)                */
    const int32_t* end_off;       /* [P*(M+1) + 1] the same for endNodesByState           */
    const int32_t* end_nodes;
} blance_moves_problem;

typedef struct blance_moves_result {
    int32_t* op_off;              /* [P + 1] moves of partition p: op_off[p]..op_off[p+1] */
    int32_t* op_node;             /* [capacity] NodeStateOp.Node                          */
    int32_t* op_state;            /* [capacity] NodeStateOp.State as a state id, -1 = ""  */
    int32_t* op_kind;             /* [capacity] BLANCE_OP_*                               */
    int64_t  capacity;            /* in: >= beg_off[last] + end_off[last] always suffices */
    double   device_ms;           /* out                                                  */
} blance_moves_result;

int blance_calc_moves(blance_ctx* ctx, const blance_moves_problem* pb, blance_moves_result* res);

/* ---- plan quality: what a caller would otherwise get by re-walking the
 * result map -- countStateNodes (plan.go:374-399) applied to the map the last blance_plan /
 * blance_plan_resident produced, reduced per state over the nodes of nodesNext (plan.go:77).
 * Arrays are the caller's, n_states entries each (state id order). */
typedef struct blance_plan_stats {
    int32_t n_states;        /* in: capacity of the arrays below (>= the problem's n_states) */
    int32_t n_nodes_next;    /* out: nodes counted (nodesAll minus nodesToRemove) */
    int64_t* load_min;       /* out: smallest weighted load of a node in this state */
    int64_t* load_max;
    int64_t* load_sum;       /* out: sum of the loads = sum over partitions of weight * len(list) on live nodes */
    int64_t* load_sumsq;     /* out: sum of squares (variance = sumsq / n - (sum / n)^2) */
    int32_t* nodes_used;     /* out: nodes with load > 0 */
    int64_t* unmet_slots;    /* out: sum over partitions of max(0, constraints - len(list)); warnings of plan.go:231-234 */
    int64_t* rule_violations; /* out (ABI 4; may be NULL): (partition, slot) pairs of this state whose node breaks one of the
                              * state's hierarchy rules against the partition's top priority node or an EARLIER node of the
                              * same list -- outside leaves(findAncestor(a, IncludeLevel)) or inside
                              * leaves(findAncestor(a, ExcludeLevel)) of such an anchor a (plan.go:723-753); what the
                              * fallback of plan.go:216-218 produces when racks disappear.  0 for states without rules */
} blance_plan_stats;

int blance_plan_stats_get(blance_ctx* ctx, blance_plan_stats* stats);

/* ---- one plan on several GPUs (BASELINE.json config 4) ------------------------------------
 * The steps of a state pass that runs as region chains (one chain per hierarchy region)
 * shard over the ranks by region: every rank holds the whole problem (upload the same problem on
 * every rank) and runs the chains of its contiguous slice of the regions.  Per such pass the ranks
 * make exactly TWO collectives -- pass boundaries are the only exact exchange points of
 * plan.go:253-303:
 *   A. one int32 sum all-reduce of [8 pass flags | change of the per-node load vector
 *      (stateNodeCounts, plan.go:94)]: (n_states + 1) * n_nodes_ext + 8 words;
 *   B. one all-gather of the pass outputs: a rank's steps are contiguous in chain order, every
 *      rank contributes its slice (padded to the longest slice).
 * Everything else (flat passes, ordering, convergence test) is computed by every rank identically,
 * so all ranks return the same result, bit-identical to a single-rank plan.
 * Collectives: the library's own RCCL communicator (blance_comm_init_rccl; one process per GPU,
 * the 128-byte id of blance_comm_unique_id() made on rank 0 and handed to the others by the host),
 * or an embedder's (blance_comm_set; used by the tests: gloo ranks over the SIMT emulator, and G
 * contexts on ONE MI355X whose buffers the hook sums / gathers).
 * Every rank must make the same sequence of plan calls.  A rank that fails before collective A
 * still takes part in it with a poison flag, so that all ranks return BLANCE_ERR_COMM together;
 * after any failed sharded call the communicator is invalid (destroy the contexts). */
typedef struct blance_comm {
    int32_t rank, n_ranks;
    /* in-place sum over the ranks of `count` int32 values at `device_buf` (device memory of this
     * context); called between kernels with the stream idle; 0 = ok */
    int (*allreduce_sum_i32)(void* user, int32_t* device_buf, int64_t count);
    void* user;
    /* in-place all-gather: `device_buf` holds n_ranks blocks of `count_per_rank` int32 values, this
     * rank's block (at rank * count_per_rank) is filled in; on return every block is.  May be NULL:
     * the outputs are then summed with allreduce_sum_i32 (every rank's foreign slices zeroed). */
    int (*allgather_i32)(void* user, int32_t* device_buf, int64_t count_per_rank);
} blance_comm;

int blance_comm_unique_id(void* id_out_128 /* 128 bytes */);
int blance_comm_init_rccl(blance_ctx* ctx, int32_t n_ranks, int32_t rank, const void* id_128);
int blance_comm_set(blance_ctx* ctx, const blance_comm* comm /* NULL: back to a single rank */);
/* collectives made / int32 words moved by this context's sharded plans so far */
int blance_comm_stats(blance_ctx* ctx, int64_t* calls, int64_t* words);
/* (ABI 5) device time, in ms, this context's plans have spent inside RCCL collectives so far: an event on either side of
 * every ncclAllReduce / ncclAllGather on the planner's stream (0 for an embedder's own collectives, which run on the host) */
int blance_comm_time_ms(blance_ctx* ctx, double* ms);

/* 1 if this library is the GPU-less SIMT emulator build of the tests (its "device" memory is host
 * memory), 0 for the gfx950 product. */
int blance_is_emulated(void);

/* Validate a problem without touching a device (sizes, id ranges, supported
 * envelope).  Same status codes as blance_plan. */
int blance_validate(const blance_problem* pb);

/* Text of the last error on this thread ("" if none). */
const char* blance_last_error(void);

int blance_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* BLANCE_HIP_H */
