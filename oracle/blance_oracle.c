/*
 * ORACLE (test infrastructure only) -- id-based CPU restatement of
 * couchbase/blance's PlanNextMap path over the flat problem of
 * include/blance_hip.h.
 *
 * This file is the checker, never the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The HIP
 * library must never link or call it.
 *
 * It follows the reference's algorithm step by step (citations are to
 * /root/reference/plan.go unless noted) on dense integer tables: maps keyed by
 * node/state/partition names become arrays indexed by the interned ids, the
 * comparison sort of candidate nodes (plan.go:171-172, strict total order
 * (score, position), plan.go:617-628) becomes successive lexicographic argmins,
 * and the string-set hierarchy algebra (plan.go:723-774) becomes interval
 * algebra over DFS leaf positions.  Scores are IEEE fp64 evaluated in the
 * reference's operation order (plan.go:634-689); build with
 * -ffp-contract=off and without -ffast-math.
 *
 * Parity pinning: blance has no Go toolchain here, so this restatement is
 * pinned (a) by the reference's own 69 golden planner cases
 * (tests/golden/planner_cases.json <- plan_test.go, control_test.go) and
 * (b) against the literal string-keyed restatement oracle/blance_ref.py on
 * random instances (tests/test_oracle.py).  Beyond 16x4 / 8x9 the reference
 * pins nothing; parity at benchmark scale is transitive through this file.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/blance_hip.h"

#define MAX_INTERVALS 256

typedef struct {
    int n;
    int32_t lo[MAX_INTERVALS], hi[MAX_INTERVALS];
} iset;

typedef struct {
    const blance_problem* pb;
    int N, NX, M, P, L;      /* L = list stride                                   */
    int NP;                  /* len(prevMap) of the current sweep (plan.go:161)   */
    int any_removed;         /* len(nodesToRemove) > 0 in this sweep              */
    int add_nil;             /* nodesToAdd == nil in this sweep                   */
    const uint8_t* removed;  /* per sweep views (all-zero after sweep 1)          */
    const uint8_t* added;
    uint8_t* zeros;          /* [NX] */
    uint8_t* alive;          /* [NX] nodesNext membership (plan.go:77)            */
    int n_alive;
    /* live NodesByState of nextPartitions (plan.go:83-88) */
    int32_t* live;  int32_t* live_len;  uint8_t* live_kind;
    /* current prevMap view of the same partitions */
    int32_t* prv;   int32_t* prv_len;   uint8_t* prv_kind;
    uint8_t* in_prev; uint8_t* never_equal;
    int64_t* cnt;            /* [(M+1) * NX] stateNodeCounts (plan.go:94)         */
    int64_t* tot;            /* [NX] nodePartitionCounts (plan.go:118-124)        */
    int32_t* ntn;            /* [(NX+1) * N] nodeToNodeCounts (plan.go:266)       */
    int32_t* ntn_rows_used;  int n_ntn_rows_used; uint8_t* ntn_row_flag;
    int32_t* hmark;          /* [NX] stamp: node is in a higher priority state    */
    int32_t* omark;          /* [NX] stamp: node holds this state for p           */
    int32_t* cmark;          /* [NX] stamp: node already in the output list       */
    int32_t stamp;
    int32_t* order;          /* [P] pass order                                    */
    /* warnings of the current sweep */
    int32_t* warn_part; int32_t* warn_state; int64_t n_warn;
    int64_t steps, step_limit;
} octx;

static double booster_eval(int kind, int w, double stickiness) {
    if (kind == BLANCE_BOOSTER_CBGT) {           /* control_test.go:19-26 */
        double score = (double)(-w);
        if (score < stickiness) score = stickiness;
        return score;
    }
    return 0.0;
}

/* nodeSorter.Score, plan.go:634-689, operation order kept. */
static double node_score(const octx* c, int m, int n, int row, double stick) {
    const blance_problem* pb = c->pb;
    double lp = 0.0;
    if (c->NP > 0)                                /* plan.go:638-644 */
        lp = (double)c->ntn[(size_t)row * c->N + n] / (double)c->NP;
    double ff = 0.0;
    if (c->NP > 0)                                /* plan.go:647-652 */
        ff = (0.001 * (double)c->tot[n]) / (double)c->NP;
    double cf = 0.0;
    if (c->omark[n] == c->stamp) cf = stick;      /* plan.go:654-662 */
    double r = (double)c->cnt[(size_t)m * c->NX + n];   /* plan.go:664-670 */
    r = r + lp;
    r = r + ff;
    if (pb->node_has_weight[n]) {                 /* plan.go:675-684 */
        int w = pb->node_weight[n];
        if (w > 0) r = r / (double)w;
        else if (w < 0 && pb->booster_kind != BLANCE_BOOSTER_NONE)
            r += booster_eval(pb->booster_kind, w, cf);
    }
    r = r - cf;
    return r;
}

static void adjust(octx* c, int state, int n, int64_t amt) {   /* plan.go:353-363 */
    c->cnt[(size_t)state * c->NX + n] += amt;
    c->tot[n] += amt;
}

/* ---- hierarchy as interval algebra (plan.go:723-774) -------------------- */

static int ancestor(const blance_problem* pb, int v, int level) {   /* plan.go:755-762 */
    while (level > 0) { v = pb->vertex_parent[v]; level--; }
    return v;
}

static void iset_push(iset* s, int lo, int hi) {
    if (lo < hi && s->n < MAX_INTERVALS) { s->lo[s->n] = lo; s->hi[s->n] = hi; s->n++; }
}

/* includeExcludeNodes, plan.go:723-734: leaves(anc(a,inc)) - leaves(anc(a,exc)) */
static void set_of_anchor(const blance_problem* pb, int a, int inc, int exc, iset* out) {
    int vi = ancestor(pb, a, inc), ve = ancestor(pb, a, exc);
    int alo = pb->vertex_leaf_lo[vi], ahi = pb->vertex_leaf_hi[vi];
    int blo = pb->vertex_leaf_lo[ve], bhi = pb->vertex_leaf_hi[ve];
    out->n = 0;
    iset_push(out, alo, ahi < blo ? ahi : blo);
    iset_push(out, alo > bhi ? alo : bhi, ahi);
}

static void iset_intersect(const iset* a, const iset* b, iset* out) {
    int i = 0, j = 0;
    out->n = 0;
    while (i < a->n && j < b->n) {
        int lo = a->lo[i] > b->lo[j] ? a->lo[i] : b->lo[j];
        int hi = a->hi[i] < b->hi[j] ? a->hi[i] : b->hi[j];
        iset_push(out, lo, hi);
        if (a->hi[i] < b->hi[j]) i++; else j++;
    }
}

/* includeExcludeNodesIntersect, plan.go:738-753, with the reset-on-empty of :746 */
static void fold_anchors(const blance_problem* pb, const int* anchors, int n_anchors,
                         int inc, int exc, iset* rv) {
    iset res, tmp;
    rv->n = 0;
    for (int i = 0; i < n_anchors; i++) {
        set_of_anchor(pb, anchors[i], inc, exc, &res);
        if (rv->n == 0) { *rv = res; continue; }
        iset_intersect(rv, &res, &tmp);
        *rv = tmp;
    }
}

static int iset_contains(const iset* s, int pos) {
    for (int i = 0; i < s->n; i++)
        if (pos >= s->lo[i] && pos < s->hi[i]) return 1;
    return 0;
}

/* ---- findBestNodes, plan.go:98-248 -------------------------------------- */

/* Best candidate by (score, position) among nodesNext minus higher-priority
 * nodes, optionally restricted to a leaf-interval set and optionally skipping
 * nodes already emitted.  Returns -1 if none. */
static int argmin_nodes(octx* c, int m, int row, double stick, const iset* mask, int skip_emitted) {
    const blance_problem* pb = c->pb;
    int best = -1;
    double bs = 0.0;
    for (int n = 0; n < c->N; n++) {
        if (!c->alive[n] || c->hmark[n] == c->stamp) continue;
        if (skip_emitted && c->cmark[n] == c->stamp) continue;
        if (mask) {
            int lp = pb->node_leaf_pos[n];
            if (lp < 0 || !iset_contains(mask, lp)) continue;
        }
        double s = node_score(c, m, n, row, stick);
        if (best < 0 || s < bs) { best = n; bs = s; }   /* ties: lowest position wins */
    }
    return best;
}

static int find_best_nodes(octx* c, int p, int m, int k, int32_t* chosen, int cap, int* is_nil) {
    const blance_problem* pb = c->pb;
    const int M = c->M, L = c->L;
    c->stamp++;
    /* stickiness, plan.go:104-115 */
    double stick = 1.5;
    if (!pb->partition_weights_nil) {
        if (pb->part_has_weight[p]) stick = (double)pb->part_weight[p];
        else if (pb->state_has_stickiness[m]) stick = (double)pb->state_stickiness[m];
    }
    /* topPriorityNode, plan.go:134-138 */
    int top = -1;
    {
        int idx = p * M + pb->top_state;
        if (c->live_kind[idx] != BLANCE_LIST_ABSENT && c->live_len[idx] > 0)
            top = c->live[(size_t)idx * L];
    }
    int row = top < 0 ? c->NX : top;
    /* excludeHigherPriorityNodes, plan.go:146-154 */
    int any_higher_key = 0;
    for (int t = 0; t < M; t++) {
        int idx = p * M + t;
        if (c->live_kind[idx] == BLANCE_LIST_ABSENT) continue;
        if (pb->state_priority[t] < pb->state_priority[m]) {
            any_higher_key = 1;
            for (int i = 0; i < c->live_len[idx]; i++) c->hmark[c->live[(size_t)idx * L + i]] = c->stamp;
        }
    }
    /* currentFactor membership, plan.go:654-662 */
    {
        int idx = p * M + m;
        if (c->live_kind[idx] != BLANCE_LIST_ABSENT)
            for (int i = 0; i < c->live_len[idx]; i++) c->omark[c->live[(size_t)idx * L + i]] = c->stamp;
    }
    int n_out = 0;
    int n_hn = 0;
    int32_t* hn = NULL;
    if (!pb->hierarchy_rules_nil) {               /* plan.go:174-226 */
        int n_rules = pb->rule_off[m + 1] - pb->rule_off[m];
        int hn_cap = n_rules * k + 1;
        hn = (int32_t*)malloc(sizeof(int32_t) * (size_t)hn_cap);
        int* anchors = (int*)malloc(sizeof(int) * (size_t)(hn_cap + 1));
        int cand0 = -2;                           /* candidateNodes[0], computed lazily */
        for (int r = pb->rule_off[m]; r < pb->rule_off[m + 1]; r++) {
            int h = top < 0 ? pb->vertex_empty : top;
            if (top < 0 && n_hn > 0) h = hn[0];
            for (int i = 0; i < k; i++) {
                iset fold;
                anchors[0] = h;
                for (int j = 0; j < n_hn; j++) anchors[1 + j] = hn[j];
                fold_anchors(pb, anchors, 1 + n_hn, pb->rule_inc[r], pb->rule_exc[r], &fold);
                int best = argmin_nodes(c, m, row, stick, &fold, 0);
                if (best >= 0) hn[n_hn++] = best;
                else {
                    if (cand0 == -2) cand0 = argmin_nodes(c, m, row, stick, NULL, 0);
                    if (cand0 >= 0) hn[n_hn++] = cand0;
                }
            }
        }
        free(anchors);
        /* candidateNodes = dedupe(hierarchyNodes ++ candidateNodes), plan.go:224-225 */
        for (int j = 0; j < n_hn && n_out < k && n_out < cap; j++) {
            if (c->cmark[hn[j]] == c->stamp) continue;
            c->cmark[hn[j]] = c->stamp;
            chosen[n_out++] = hn[j];
        }
        free(hn);
    }
    /* the sorted candidate list, consumed lazily: plan.go:171-172, :228-235 */
    while (n_out < k && n_out < cap) {
        int best = argmin_nodes(c, m, row, stick, NULL, 1);
        if (best < 0) break;
        c->cmark[best] = c->stamp;
        chosen[n_out++] = best;
    }
    if (n_out < k) {                              /* plan.go:230-235 */
        c->warn_part[c->n_warn] = p;
        c->warn_state[c->n_warn] = m;
        c->n_warn++;
    }
    *is_nil = (n_out == 0 && c->n_alive == 0 && !any_higher_key && pb->hierarchy_rules_nil);
    /* nodeToNodeCounts, plan.go:238-245 (only ever read when NP > 0) */
    if (c->NP > 0 && n_out > 0) {
        if (!c->ntn_row_flag[row]) { c->ntn_row_flag[row] = 1; c->ntn_rows_used[c->n_ntn_rows_used++] = row; }
        for (int i = 0; i < n_out; i++) c->ntn[(size_t)row * c->N + chosen[i]]++;
    }
    c->steps++;
    return n_out;
}

/* removeNodesFromNodesByState with the dec callback, plan.go:290-297, :408-421 */
static void remove_from_all_states(octx* c, int p, const int32_t* rm, int n_rm, int64_t w) {
    const int M = c->M, L = c->L;
    for (int t = 0; t < M; t++) {
        int idx = p * M + t;
        if (c->live_kind[idx] == BLANCE_LIST_ABSENT) continue;
        int32_t* lst = c->live + (size_t)idx * L;
        int len = c->live_len[idx], out = 0;
        for (int i = 0; i < len; i++) {
            int n = lst[i], hit = 0, dup = 0;
            for (int j = 0; j < n_rm; j++) if (rm[j] == n) { hit = 1; break; }
            if (hit) {
                for (int j = 0; j < i; j++) if (lst[j] == n) { dup = 1; break; }   /* misc.go:45 */
                if (!dup) adjust(c, t, n, -w);
            }
        }
        for (int i = 0; i < len; i++) {
            int n = lst[i], hit = 0;
            for (int j = 0; j < n_rm; j++) if (rm[j] == n) { hit = 1; break; }
            if (!hit) lst[out++] = n;
        }
        c->live_len[idx] = out;
        c->live_kind[idx] = BLANCE_LIST_SET;
    }
}

/* Note: the first loop above tests duplicates against the not-yet-compacted
 * list, which is why removal is a separate second loop. */

/* assignStateToPartitions, plan.go:253-303 */
static int state_pass(octx* c, int m, int k) {
    const blance_problem* pb = c->pb;
    const int M = c->M, L = c->L, P = c->P;
    /* partitionSorter category (plan.go:542-561), stable 3-way split of the static order */
    uint8_t* cat = (uint8_t*)malloc((size_t)P + 1);
    for (int p = 0; p < P; p++) {
        int cv = 2;
        int is0 = 0;
        if (c->any_removed && c->in_prev[p]) {
            int idx = p * M + m;
            if (c->prv_kind[idx] == BLANCE_LIST_SET)
                for (int i = 0; i < c->prv_len[idx]; i++)
                    if (c->removed[c->prv[(size_t)idx * L + i]]) { is0 = 1; break; }
        }
        if (is0) cv = 0;
        else if (!c->add_nil) {
            int hit = 0;
            for (int t = 0; t < M && !hit; t++) {
                int idx = p * M + t;
                if (c->live_kind[idx] == BLANCE_LIST_ABSENT) continue;
                for (int i = 0; i < c->live_len[idx]; i++)
                    if (c->added[c->live[(size_t)idx * L + i]]) { hit = 1; break; }
            }
            if (!hit) cv = 1;
        }
        cat[p] = (uint8_t)cv;
    }
    int pos = 0;
    for (int cv = 0; cv < 3; cv++)
        for (int i = 0; i < P; i++) {
            int p = pb->part_order[i];
            if (cat[p] == cv) c->order[pos++] = p;
        }
    free(cat);
    /* nodeToNodeCounts := fresh, plan.go:266 */
    for (int i = 0; i < c->n_ntn_rows_used; i++) {
        int row = c->ntn_rows_used[i];
        memset(c->ntn + (size_t)row * c->N, 0, sizeof(int32_t) * (size_t)c->N);
        c->ntn_row_flag[row] = 0;
    }
    c->n_ntn_rows_used = 0;

    int32_t* chosen = (int32_t*)malloc(sizeof(int32_t) * (size_t)(L + 1));
    int32_t* old = (int32_t*)malloc(sizeof(int32_t) * (size_t)(L + 1));
    for (int oi = 0; oi < P; oi++) {
        if (c->step_limit > 0 && c->steps >= c->step_limit) { free(chosen); free(old); return 1; }
        int p = c->order[oi];
        int64_t w = 1;                            /* plan.go:269-275 */
        if (!pb->partition_weights_nil && pb->part_has_weight[p]) w = pb->part_weight[p];
        int is_nil = 0;
        int n_ch = find_best_nodes(c, p, m, k, chosen, L, &is_nil);
        int idx = p * M + m;
        int n_old = 0;
        if (c->live_kind[idx] != BLANCE_LIST_ABSENT) {
            n_old = c->live_len[idx];
            memcpy(old, c->live + (size_t)idx * L, sizeof(int32_t) * (size_t)n_old);
        }
        remove_from_all_states(c, p, old, n_old, w);        /* plan.go:290-293 */
        remove_from_all_states(c, p, chosen, n_ch, w);      /* plan.go:294-297 */
        memcpy(c->live + (size_t)idx * L, chosen, sizeof(int32_t) * (size_t)n_ch);   /* :299 */
        c->live_len[idx] = n_ch;
        c->live_kind[idx] = is_nil ? BLANCE_LIST_NIL : BLANCE_LIST_SET;
        for (int i = 0; i < n_ch; i++) adjust(c, m, chosen[i], w);                  /* :301 */
    }
    free(chosen);
    free(old);
    return 0;
}

/* planNextMapInnerEx, plan.go:60-331.  Returns 1 if the step limit stopped it. */
static int sweep(octx* c, int iteration) {
    const blance_problem* pb = c->pb;
    const int M = c->M, L = c->L, P = c->P, NX = c->NX;
    c->n_warn = 0;
    if (iteration == 0) {
        c->removed = pb->node_removed; c->added = pb->node_added;
        c->add_nil = pb->nodes_to_add_nil; c->NP = pb->n_prev;
    } else {                                      /* plan.go:53-55 */
        c->removed = c->zeros; c->added = c->zeros; c->add_nil = 0;
        int fresh = 0;
        for (int p = 0; p < P; p++) if (!pb->part_in_prev[p]) fresh++;
        c->NP = pb->n_prev + fresh;
    }
    c->any_removed = 0;
    for (int n = 0; n < NX; n++) if (c->removed[n]) c->any_removed = 1;
    /* nextPartitions = copy of partitionsToAssign minus nodesToRemove, plan.go:83-88 */
    if (iteration == 0) {
        for (int idx = 0; idx < P * M; idx++) {
            int len = 0;
            for (int i = pb->assign_off[idx]; i < pb->assign_off[idx + 1]; i++) {
                int n = pb->assign_nodes[i];
                if (!c->removed[n]) c->live[(size_t)idx * L + len++] = n;
            }
            c->live_len[idx] = len;
            c->live_kind[idx] = pb->assign_kind[idx] == BLANCE_LIST_ABSENT ? BLANCE_LIST_ABSENT
                                                                          : BLANCE_LIST_SET;
        }
    } else {
        /* partitionsToAssign[name] = last result (plan.go:51); live already holds it,
         * and every present key becomes a non-nil slice again (plan.go:418). */
        for (int idx = 0; idx < P * M; idx++)
            if (c->live_kind[idx] != BLANCE_LIST_ABSENT) c->live_kind[idx] = BLANCE_LIST_SET;
    }
    /* stateNodeCounts = countStateNodes(prevMap), plan.go:94, :374-399 */
    memset(c->cnt, 0, sizeof(int64_t) * (size_t)(M + 1) * NX);
    memset(c->tot, 0, sizeof(int64_t) * (size_t)NX);
    for (int i = 0; i < pb->n_loads; i++) {
        if (iteration > 0 && pb->load_first_sweep_only[i]) continue;
        adjust(c, pb->load_state[i], pb->load_node[i], pb->load_weight[i]);
    }
    for (int p = 0; p < P; p++) {
        if (!c->in_prev[p]) continue;
        int64_t w = 1;
        if (!pb->partition_weights_nil && pb->part_has_weight[p]) w = pb->part_weight[p];
        for (int m = 0; m < M; m++) {
            int idx = p * M + m;
            for (int i = 0; i < c->prv_len[idx]; i++) adjust(c, m, c->prv[(size_t)idx * L + i], w);
        }
    }
    /* state passes, plan.go:307-324 */
    for (int m = 0; m < M; m++) {
        int k = pb->state_constraints[m];
        if (k > 0 && state_pass(c, m, k)) return 1;
    }
    return 0;
}

int64_t blance_oracle_result_capacity(const blance_problem* pb) {
    int64_t cap = 0;
    for (int idx = 0; idx < pb->n_parts * pb->n_states; idx++) {
        int len = pb->assign_off[idx + 1] - pb->assign_off[idx];
        int k = pb->state_constraints[idx % pb->n_states];
        cap += len > k ? len : k;
    }
    return cap;
}

/* planNextMapEx, plan.go:23-58.  step_limit > 0 stops after that many
 * findBestNodes calls (for bounded CPU-baseline samples): returns 1 and fills
 * *steps_done / *seconds only. */
int blance_oracle_plan_ex(const blance_problem* pb, blance_result* res, int64_t step_limit,
                          int64_t* steps_done, double* seconds) {
    octx c;
    memset(&c, 0, sizeof(c));
    c.pb = pb;
    const int N = c.N = pb->n_nodes, NX = c.NX = pb->n_nodes_ext, M = c.M = pb->n_states,
              P = c.P = pb->n_parts;
    c.step_limit = step_limit;
    int L = 1;
    for (int m = 0; m < M; m++) if (pb->state_constraints[m] > L) L = pb->state_constraints[m];
    for (int idx = 0; idx < P * M; idx++) {
        int a = pb->assign_off[idx + 1] - pb->assign_off[idx];
        int b = pb->prev_off[idx + 1] - pb->prev_off[idx];
        if (a > L) L = a;
        if (b > L) L = b;
    }
    c.L = L;
    size_t PM = (size_t)P * M;
    c.zeros = (uint8_t*)calloc((size_t)NX + 1, 1);
    c.alive = (uint8_t*)calloc((size_t)NX + 1, 1);
    for (int n = 0; n < N; n++) if (!pb->node_removed[n]) { c.alive[n] = 1; c.n_alive++; }
    c.live = (int32_t*)malloc(sizeof(int32_t) * (PM * L + 1));
    c.live_len = (int32_t*)calloc(PM + 1, sizeof(int32_t));
    c.live_kind = (uint8_t*)calloc(PM + 1, 1);
    c.prv = (int32_t*)malloc(sizeof(int32_t) * (PM * L + 1));
    c.prv_len = (int32_t*)calloc(PM + 1, sizeof(int32_t));
    c.prv_kind = (uint8_t*)calloc(PM + 1, 1);
    c.in_prev = (uint8_t*)malloc((size_t)P + 1);
    c.never_equal = (uint8_t*)malloc((size_t)P + 1);
    memcpy(c.in_prev, pb->part_in_prev, (size_t)P);
    memcpy(c.never_equal, pb->part_prev_never_equal, (size_t)P);
    for (size_t idx = 0; idx < PM; idx++) {
        int len = 0;
        for (int i = pb->prev_off[idx]; i < pb->prev_off[idx + 1]; i++)
            c.prv[idx * L + len++] = pb->prev_nodes[i];
        c.prv_len[idx] = len;
        c.prv_kind[idx] = pb->prev_kind[idx];
    }
    c.cnt = (int64_t*)calloc((size_t)(M + 1) * NX + 1, sizeof(int64_t));
    c.tot = (int64_t*)calloc((size_t)NX + 1, sizeof(int64_t));
    c.ntn = (int32_t*)calloc((size_t)(NX + 1) * (N > 0 ? N : 1), sizeof(int32_t));
    c.ntn_rows_used = (int32_t*)malloc(sizeof(int32_t) * (size_t)(NX + 2));
    c.ntn_row_flag = (uint8_t*)calloc((size_t)NX + 2, 1);
    c.hmark = (int32_t*)calloc((size_t)NX + 1, sizeof(int32_t));
    c.omark = (int32_t*)calloc((size_t)NX + 1, sizeof(int32_t));
    c.cmark = (int32_t*)calloc((size_t)NX + 1, sizeof(int32_t));
    c.order = (int32_t*)malloc(sizeof(int32_t) * ((size_t)P + 1));
    c.warn_part = (int32_t*)malloc(sizeof(int32_t) * (PM + 1));
    c.warn_state = (int32_t*)malloc(sizeof(int32_t) * (PM + 1));

    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    int iterations = 0, converged = 0, stopped = 0;
    for (int it = 0; it < pb->max_iterations; it++) {           /* plan.go:32 */
        stopped = sweep(&c, it);
        if (stopped) break;
        iterations++;
        /* convergence: every result partition DeepEquals prevMap[name], plan.go:36-45 */
        int not_match = 0;
        for (int p = 0; p < P && !not_match; p++) {
            if (!c.in_prev[p] || c.never_equal[p]) { not_match = 1; break; }
            for (int m = 0; m < M; m++) {
                size_t idx = (size_t)p * M + m;
                if (c.live_kind[idx] != c.prv_kind[idx] || c.live_len[idx] != c.prv_len[idx] ||
                    memcmp(c.live + idx * L, c.prv + idx * L, sizeof(int32_t) * (size_t)c.live_len[idx])) {
                    not_match = 1;
                    break;
                }
            }
        }
        if (!not_match) { converged = 1; break; }
        /* prevMap[name] = partitionsToAssign[name] = result, plan.go:49-52 */
        memcpy(c.prv, c.live, sizeof(int32_t) * PM * L);
        memcpy(c.prv_len, c.live_len, sizeof(int32_t) * PM);
        memcpy(c.prv_kind, c.live_kind, PM);
        memset(c.in_prev, 1, (size_t)P);
        memset(c.never_equal, 0, (size_t)P);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double secs = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    if (steps_done) *steps_done = c.steps;
    if (seconds) *seconds = secs;

    int status = BLANCE_OK;
    if (stopped) status = 1;
    else if (res) {
        int64_t off = 0;
        for (size_t idx = 0; idx < PM; idx++) {
            res->out_off[idx] = (int32_t)off;
            res->out_kind[idx] = c.live_kind[idx];
            if (off + c.live_len[idx] > res->out_capacity) { status = BLANCE_ERR_CAPACITY; break; }
            memcpy(res->out_nodes + off, c.live + idx * L, sizeof(int32_t) * (size_t)c.live_len[idx]);
            off += c.live_len[idx];
        }
        if (status == BLANCE_OK) {
            res->out_off[PM] = (int32_t)off;
            if (c.n_warn > res->warn_capacity) status = BLANCE_ERR_CAPACITY;
            else {
                memcpy(res->warn_part, c.warn_part, sizeof(int32_t) * (size_t)c.n_warn);
                memcpy(res->warn_state, c.warn_state, sizeof(int32_t) * (size_t)c.n_warn);
                res->n_warnings = c.n_warn;
            }
            res->iterations = iterations;
            res->converged = converged;
            res->device_ms = 0.0;
            res->total_ms = secs * 1e3;
            res->steps_total = c.steps;
            res->steps_sequential = c.steps;
            res->steps_batched = 0;
            res->kernel_launches = 0;
            res->pass_kernel_ms = 0.0;
            res->pass_kernel_launches = 0;
            res->flat_pass_ms = 0.0;
            res->flat_passes = 0;
        }
    }
    free(c.zeros); free(c.alive); free(c.live); free(c.live_len); free(c.live_kind);
    free(c.prv); free(c.prv_len); free(c.prv_kind); free(c.in_prev); free(c.never_equal);
    free(c.cnt); free(c.tot); free(c.ntn); free(c.ntn_rows_used); free(c.ntn_row_flag);
    free(c.hmark); free(c.omark); free(c.cmark); free(c.order); free(c.warn_part); free(c.warn_state);
    return status;
}

int blance_oracle_plan(const blance_problem* pb, blance_result* res) {
    return blance_oracle_plan_ex(pb, res, 0, NULL, NULL);
}
