/*
 * rad_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 * See rad_oracle.h for what each function restates and where parity is pinned.
 */
#define _GNU_SOURCE
#include "rad_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ====================================================================== */
/* A1  Tanimoto on packed bits                                            */
/* ====================================================================== */

void orc_tanimoto_counts(const uint8_t *a, const uint8_t *b, size_t nbytes,
                         uint32_t *and_out, uint32_t *or_out) {
    uint32_t ca = 0, co = 0;
    size_t i = 0;
    for (; i + 8 <= nbytes; i += 8) {
        uint64_t x, y;
        memcpy(&x, a + i, 8);
        memcpy(&y, b + i, 8);
        ca += (uint32_t)__builtin_popcountll(x & y);
        co += (uint32_t)__builtin_popcountll(x | y);
    }
    for (; i < nbytes; ++i) {
        ca += (uint32_t)__builtin_popcount((unsigned)(a[i] & b[i]));
        co += (uint32_t)__builtin_popcount((unsigned)(a[i] | b[i]));
    }
    *and_out = ca;
    *or_out = co;
}

/* float edge convention: one division, one subtraction, round-to-nearest;
 * both all-zero -> 0.0f (identical vectors) */
float orc_distance_f32(uint32_t and_cnt, uint32_t or_cnt) {
    if (or_cnt == 0) return 0.0f;
    volatile float q = (float)and_cnt / (float)or_cnt; /* no fused/extended eval */
    return 1.0f - q;
}

void orc_scan(const uint8_t *corpus, uint64_t n, size_t row_bytes,
              const uint8_t *query, uint32_t *and_out, uint32_t *or_out) {
    for (uint64_t i = 0; i < n; ++i)
        orc_tanimoto_counts(query, corpus + i * row_bytes, row_bytes,
                            &and_out[i], &or_out[i]);
}

void orc_gather(const uint8_t *corpus, size_t row_bytes, const uint8_t *query,
                const uint32_t *slots, uint64_t n_slots, uint32_t *and_out,
                uint32_t *or_out) {
    for (uint64_t i = 0; i < n_slots; ++i)
        orc_tanimoto_counts(query, corpus + (uint64_t)slots[i] * row_bytes,
                            row_bytes, &and_out[i], &or_out[i]);
}

/* ====================================================================== */
/* graph accessors (A3 / A4)                                              */
/* ====================================================================== */

static const uint32_t *graph_row(const orc_graph_t *g, uint32_t slot, int level,
                                 uint32_t *cap) {
    if (slot >= g->n || level < 0 || level > g->levels[slot]) return NULL;
    if (level == 0) {
        *cap = g->cap0;
        return g->adj0 + (uint64_t)slot * g->cap0;
    }
    *cap = g->capU;
    return g->adjU + ((uint64_t)g->upper_row[slot] + (uint64_t)(level - 1)) * g->capU;
}

int orc_graph_neighbors(const orc_graph_t *g, uint32_t slot, int level,
                        uint32_t *out, uint32_t out_cap) {
    uint32_t cap = 0;
    const uint32_t *row = graph_row(g, slot, level, &cap);
    if (!row) return -1;
    uint32_t k = 0;
    for (uint32_t j = 0; j < cap && row[j] != ORC_NO_SLOT; ++j)
        if (k < out_cap) out[k++] = row[j];
    return (int)k;
}

uint64_t orc_graph_top_level(const orc_graph_t *g, uint32_t *out, uint64_t cap) {
    uint64_t k = 0;
    for (uint64_t i = 0; i < g->n; ++i)
        if (g->levels[i] == g->max_level) {
            if (k < cap) out[k] = (uint32_t)i;
            ++k;
        }
    return k;
}

/* ====================================================================== */
/* RAD traversal with Tanimoto scoring                                    */
/* ====================================================================== */

/* --- priority queue: rad/priority_queue.py:22-42 ----------------------- */
/* ZSET order: ascending score (double), ties by bytewise order of the
 * member string "{node_id}:{level}".                                     */
typedef struct {
    float dist;     /* the score the "scoring_fn" returned                */
    uint32_t slot;
    uint32_t and_cnt, or_cnt;
    uint8_t level;
} pq_item_t;

/* "{node_id}:{level}" written out by hand (what snprintf("%u:%d") produces; levels are 0..15) */
static int member_str(char *m, uint32_t s, int l) {
    char d[12];
    int k = 0, n = 0;
    do { d[k++] = (char)('0' + s % 10u); s /= 10u; } while (s);
    while (k) m[n++] = d[--k];
    m[n++] = ':';
    if (l >= 10) m[n++] = (char)('0' + l / 10);
    m[n++] = (char)('0' + l % 10);
    m[n] = 0;
    return n;
}
static int member_cmp(uint32_t sa, int la, uint32_t sb, int lb) {
    char ma[32], mb[32];
    member_str(ma, sa, la);
    member_str(mb, sb, lb);
    return strcmp(ma, mb); /* ASCII: identical to bytewise compare */
}

static int pq_less(const pq_item_t *a, const pq_item_t *b) {
    double da = (double)a->dist, db = (double)b->dist;
    if (da < db) return 1;
    if (da > db) return 0;
    return member_cmp(a->slot, a->level, b->slot, b->level) < 0;
}

typedef struct {
    pq_item_t *v;
    uint64_t n, cap;
} pq_t;

static void pq_push(pq_t *q, pq_item_t it) {
    if (q->n == q->cap) {
        q->cap = q->cap ? q->cap * 2 : 1024;
        q->v = (pq_item_t *)realloc(q->v, q->cap * sizeof(pq_item_t));
    }
    uint64_t i = q->n++;
    while (i > 0) {
        uint64_t p = (i - 1) / 2;
        if (!pq_less(&it, &q->v[p])) break;
        q->v[i] = q->v[p];
        i = p;
    }
    q->v[i] = it;
}

static int pq_pop(pq_t *q, pq_item_t *out) {
    if (q->n == 0) return 0;
    *out = q->v[0];
    pq_item_t last = q->v[--q->n];
    uint64_t i = 0;
    for (;;) {
        uint64_t l = 2 * i + 1, r = l + 1, m;
        if (l >= q->n) break;
        m = (r < q->n && pq_less(&q->v[r], &q->v[l])) ? r : l;
        if (!pq_less(&q->v[m], &last)) break;
        q->v[i] = q->v[m];
        i = m;
    }
    if (q->n) q->v[i] = last;
    return 1;
}

/* --- visited: rad/visited.py:17-29, key = (node_id, level) ------------- */
/* --- scored : rad/scored.py:37-61, key = node_id, insertion order ------ */
typedef struct {
    uint64_t *keys; /* 0 = empty; stored key+1 */
    uint32_t *vals;
    uint64_t mask, count;
} hset_t;

static uint64_t h64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

static void hset_init(hset_t *s, uint64_t expect, int with_vals) {
    uint64_t cap = 1024;
    while (cap < expect * 2 + 16) cap <<= 1;
    s->keys = (uint64_t *)calloc(cap, sizeof(uint64_t));
    s->vals = with_vals ? (uint32_t *)calloc(cap, sizeof(uint32_t)) : NULL;
    s->mask = cap - 1;
    s->count = 0;
}

static void hset_free(hset_t *s) { free(s->keys); free(s->vals); }

static void hset_grow(hset_t *s);

/* returns 1 if key was already present; otherwise inserts (with val) */
static int hset_test_and_set(hset_t *s, uint64_t key, uint32_t val, uint32_t *found_val) {
    if ((s->count + 1) * 2 > s->mask + 1) hset_grow(s);
    uint64_t k1 = key + 1, i = h64(key) & s->mask;
    for (;;) {
        if (s->keys[i] == 0) {
            s->keys[i] = k1;
            if (s->vals) s->vals[i] = val;
            s->count++;
            return 0;
        }
        if (s->keys[i] == k1) {
            if (found_val && s->vals) *found_val = s->vals[i];
            return 1;
        }
        i = (i + 1) & s->mask;
    }
}

static int hset_find(const hset_t *s, uint64_t key, uint32_t *found_val) {
    uint64_t k1 = key + 1, i = h64(key) & s->mask;
    for (;;) {
        if (s->keys[i] == 0) return 0;
        if (s->keys[i] == k1) {
            if (found_val && s->vals) *found_val = s->vals[i];
            return 1;
        }
        i = (i + 1) & s->mask;
    }
}

static void hset_grow(hset_t *s) {
    hset_t old = *s;
    uint64_t cap = (old.mask + 1) * 2;
    s->keys = (uint64_t *)calloc(cap, sizeof(uint64_t));
    s->vals = old.vals ? (uint32_t *)calloc(cap, sizeof(uint32_t)) : NULL;
    s->mask = cap - 1;
    s->count = 0;
    for (uint64_t i = 0; i <= old.mask; ++i)
        if (old.keys[i]) hset_test_and_set(s, old.keys[i] - 1, old.vals ? old.vals[i] : 0, NULL);
    free(old.keys);
    free(old.vals);
}

int orc_rad_traverse(const orc_graph_t *g, const uint8_t *corpus,
                     size_t row_bytes, const uint8_t *query,
                     uint64_t n_to_score, uint64_t max_pops,
                     uint32_t *out_slots, uint32_t *out_and, uint32_t *out_or,
                     uint64_t out_cap, uint32_t *pop_nodes, uint8_t *pop_levels,
                     uint64_t pop_cap, orc_trav_stats_t *stats) {
    pq_t pq = {0};
    hset_t visited, scored;
    hset_init(&visited, n_to_score + 64, 0);
    hset_init(&scored, n_to_score + 64, 1);
    uint64_t n_scored = 0, n_pops = 0, n_nbr = 0;
    int rc = 0;

    /* scored-set insert-if-absent, returns index in insertion order */
#define SCORE_NODE(slot_, idx_out)                                             \
    do {                                                                       \
        uint32_t fv_ = 0;                                                      \
        if (hset_find(&scored, (slot_), &fv_)) {                               \
            (idx_out) = fv_;                                                   \
        } else {                                                               \
            if (n_scored >= out_cap) { rc = -2; goto done; }                   \
            uint32_t a_, o_;                                                   \
            orc_tanimoto_counts(query, corpus + (uint64_t)(slot_) * row_bytes, \
                                row_bytes, &a_, &o_);                          \
            out_slots[n_scored] = (slot_);                                     \
            out_and[n_scored] = a_;                                            \
            out_or[n_scored] = o_;                                             \
            hset_test_and_set(&scored, (slot_), (uint32_t)n_scored, NULL);     \
            (idx_out) = (uint32_t)n_scored;                                    \
            n_scored++;                                                        \
        }                                                                      \
    } while (0)

    /* prime: rad/traverser.py:141-170 */
    int start_level = g->max_level - 1;
    if (start_level < 0) start_level = 0;
    for (uint64_t i = 0; i < g->n; ++i) {
        if (g->levels[i] != g->max_level) continue;
        uint32_t idx;
        SCORE_NODE((uint32_t)i, idx);
        hset_test_and_set(&visited, ((uint64_t)i << 8) | (uint64_t)start_level, 0, NULL);
        pq_item_t it = {orc_distance_f32(out_and[idx], out_or[idx]), (uint32_t)i,
                        out_and[idx], out_or[idx], (uint8_t)start_level};
        pq_push(&pq, it);
    }

    for (;;) {
        /* idealised sequential termination: checked before every pop
         * (rad/coordination_service.py:434-437) */
        if (n_scored >= n_to_score) break;
        if (max_pops && n_pops >= max_pops) break;
        pq_item_t cur;
        if (!pq_pop(&pq, &cur)) break;            /* request_work :308-310 */
        if (pop_nodes && n_pops < pop_cap) {
            pop_nodes[n_pops] = cur.slot;
            if (pop_levels) pop_levels[n_pops] = cur.level;
        }
        n_pops++;
        uint32_t cap = 0;
        const uint32_t *row = graph_row(g, cur.slot, cur.level, &cap);
        if (!row) { rc = -3; goto done; }
        /* the rows the loop below may read: software prefetch, as a tuned CPU implementation would */
        for (uint32_t j = 0; j < cap && row[j] != ORC_NO_SLOT; ++j) {
            const uint8_t *r_ = corpus + (uint64_t)row[j] * row_bytes;
            __builtin_prefetch(r_);
            if (row_bytes > 64) __builtin_prefetch(r_ + 64);
        }
        /* submit_work_results :369-389 (scores computed in
         * distributed_worker.py:296-305 only for nodes not yet scored) */
        for (uint32_t j = 0; j < cap && row[j] != ORC_NO_SLOT; ++j) {
            uint32_t nb = row[j];
            n_nbr++;
            if (hset_test_and_set(&visited, ((uint64_t)nb << 8) | cur.level, 0, NULL))
                continue;
            uint32_t idx;
            SCORE_NODE(nb, idx);
            pq_item_t it = {orc_distance_f32(out_and[idx], out_or[idx]), nb,
                            out_and[idx], out_or[idx], cur.level};
            pq_push(&pq, it);
        }
        /* descend: :391-395 (an empty row still descends — stated deviation
         * from distributed_worker.py:286-288) */
        if (cur.level > 0) {
            int nl = cur.level - 1;
            if (!hset_test_and_set(&visited, ((uint64_t)cur.slot << 8) | (uint64_t)nl, 0, NULL)) {
                pq_item_t it = cur;
                it.level = (uint8_t)nl;
                pq_push(&pq, it);
            }
        }
    }
done:
    if (stats) {
        stats->n_scored = n_scored;
        stats->n_pops = n_pops;
        stats->n_evals = n_scored;
        stats->n_nbr = n_nbr;
        stats->f_valid = pq.n > 0;
        stats->f_and = pq.n ? pq.v[0].and_cnt : 0;
        stats->f_or = pq.n ? pq.v[0].or_cnt : 0;
        stats->f_slot = pq.n ? pq.v[0].slot : 0;
        stats->f_level = pq.n ? pq.v[0].level : 0;
        stats->f_pad = 0;
    }
    free(pq.v);
    hset_free(&visited);
    hset_free(&scored);
    return rc;
#undef SCORE_NODE
}

/* ====================================================================== */
/* the same traversal as a resumable stepper: scores come from outside    */
/* ====================================================================== */
/* What the row-sharded multi-GPU mode needs: the control flow of orc_rad_traverse cut at the point
 * where a fingerprint is read.  One step = apply the scores of the pending requests (scored insert +
 * queue insert, rad/coordination_service.py:379-389), then prime the next batch of top-level nodes or
 * pop / expand until some neighbour needs a score (rad/distributed_worker.py:296-305: only nodes not
 * yet in the scored set are scored) or the traversal ends.  The stepper never touches the corpus: the
 * caller evaluates the requested slots wherever their rows live.  Driven with a local evaluator it
 * reproduces orc_rad_traverse exactly (tests/test_oracle_golden.py). */
struct orc_stepper {
    const orc_graph_t *g;
    uint64_t n_to_score;
    pq_t pq;
    hset_t visited, scored;
    uint32_t *out_slots, *out_and, *out_or;
    uint64_t out_cap, n_scored, n_pops, n_nbr;
    uint32_t *pop_nodes; uint8_t *pop_levels; uint64_t pop_cap;
    uint32_t *top; uint64_t n_top, prime_at;
    int start_level;
    uint32_t pend[64]; uint32_t n_pend; int pend_level; int pend_prime;
    int status;   /* 0 running, 1 done (n_to_score), 2 done (queue empty), <0 error */
};

orc_stepper_t *orc_stepper_create(const orc_graph_t *g, uint64_t n_to_score, uint64_t pop_cap) {
    orc_stepper_t *s = (orc_stepper_t *)calloc(1, sizeof *s);
    s->g = g; s->n_to_score = n_to_score;
    hset_init(&s->visited, n_to_score + 64, 0);
    hset_init(&s->scored, n_to_score + 64, 1);
    s->n_top = 0;
    for (uint64_t i = 0; i < g->n; ++i) if (g->levels[i] == g->max_level) s->n_top++;
    s->top = (uint32_t *)malloc((s->n_top ? s->n_top : 1) * 4);
    uint64_t k = 0;
    for (uint64_t i = 0; i < g->n; ++i) if (g->levels[i] == g->max_level) s->top[k++] = (uint32_t)i;
    s->out_cap = n_to_score + 64 + s->n_top;
    s->out_slots = (uint32_t *)malloc(s->out_cap * 4);
    s->out_and = (uint32_t *)malloc(s->out_cap * 4);
    s->out_or = (uint32_t *)malloc(s->out_cap * 4);
    s->pop_cap = pop_cap;
    s->pop_nodes = pop_cap ? (uint32_t *)malloc(pop_cap * 4) : NULL;
    s->pop_levels = pop_cap ? (uint8_t *)malloc(pop_cap) : NULL;
    s->start_level = g->max_level > 0 ? g->max_level - 1 : 0;
    return s;
}

void orc_stepper_destroy(orc_stepper_t *s) {
    if (!s) return;
    free(s->pq.v); hset_free(&s->visited); hset_free(&s->scored);
    free(s->out_slots); free(s->out_and); free(s->out_or); free(s->pop_nodes); free(s->pop_levels); free(s->top);
    free(s);
}

static void stepper_push(orc_stepper_t *s, uint32_t slot, uint32_t a, uint32_t o, int level) {
    pq_item_t it = {orc_distance_f32(a, o), slot, a, o, (uint8_t)level};
    pq_push(&s->pq, it);
}

/* and_in/or_in: counts of the slots returned by the previous call, same order.  Returns the number of
 * slots written to req_out (their scores are due at the next call), 0 when the traversal has ended. */
int orc_stepper_step(orc_stepper_t *s, const uint32_t *and_in, const uint32_t *or_in, uint32_t *req_out,
                     uint32_t req_cap) {
    if (s->status != 0) return 0;
    /* ---- finish: the pending nodes are scored now */
    for (uint32_t i = 0; i < s->n_pend; ++i) {
        if (s->n_scored >= s->out_cap) { s->status = -2; return 0; }
        const uint32_t slot = s->pend[i];
        s->out_slots[s->n_scored] = slot; s->out_and[s->n_scored] = and_in[i]; s->out_or[s->n_scored] = or_in[i];
        hset_test_and_set(&s->scored, slot, (uint32_t)s->n_scored, NULL);
        s->n_scored++;
        stepper_push(s, slot, and_in[i], or_in[i], s->pend_level);
    }
    s->n_pend = 0;
    if (req_cap > 64) req_cap = 64;
    /* ---- prime: rad/traverser.py:141-170, req_cap top-level nodes at a time */
    if (s->prime_at < s->n_top) {
        while (s->prime_at < s->n_top && s->n_pend < req_cap) {
            const uint32_t slot = s->top[s->prime_at++];
            hset_test_and_set(&s->visited, ((uint64_t)slot << 8) | (uint64_t)s->start_level, 0, NULL);
            s->pend[s->n_pend++] = slot;      /* top-level nodes are distinct and nothing is scored yet */
        }
        s->pend_level = s->start_level;
        memcpy(req_out, s->pend, s->n_pend * 4);
        return (int)s->n_pend;
    }
    /* ---- pop / expand until a neighbour needs a score */
    for (;;) {
        if (s->n_scored >= s->n_to_score) { s->status = 1; return 0; }
        pq_item_t cur;
        if (!pq_pop(&s->pq, &cur)) { s->status = 2; return 0; }
        if (s->pop_nodes && s->n_pops < s->pop_cap) { s->pop_nodes[s->n_pops] = cur.slot; s->pop_levels[s->n_pops] = cur.level; }
        s->n_pops++;
        uint32_t cap = 0;
        const uint32_t *row = graph_row(s->g, cur.slot, cur.level, &cap);
        if (!row) { s->status = -3; return 0; }
        for (uint32_t j = 0; j < cap && row[j] != ORC_NO_SLOT; ++j) {
            const uint32_t nb = row[j];
            s->n_nbr++;
            if (hset_test_and_set(&s->visited, ((uint64_t)nb << 8) | cur.level, 0, NULL)) continue;
            uint32_t idx = 0;
            if (hset_find(&s->scored, nb, &idx)) stepper_push(s, nb, s->out_and[idx], s->out_or[idx], cur.level);
            else s->pend[s->n_pend++] = nb;
        }
        s->pend_level = cur.level;
        if (cur.level > 0) {
            const int nl = cur.level - 1;
            if (!hset_test_and_set(&s->visited, ((uint64_t)cur.slot << 8) | (uint64_t)nl, 0, NULL))
                stepper_push(s, cur.slot, cur.and_cnt, cur.or_cnt, nl);
        }
        if (s->n_pend) {
            if (s->n_pend > req_cap) { s->status = -4; return 0; }
            memcpy(req_out, s->pend, s->n_pend * 4);
            return (int)s->n_pend;
        }
    }
}

int orc_stepper_status(const orc_stepper_t *s) { return s->status; }
void orc_stepper_stats(const orc_stepper_t *s, orc_trav_stats_t *st) {
    memset(st, 0, sizeof *st);
    st->n_scored = s->n_scored; st->n_pops = s->n_pops; st->n_evals = s->n_scored; st->n_nbr = s->n_nbr;
}
uint64_t orc_stepper_results(const orc_stepper_t *s, uint32_t *slots, uint32_t *and_cnt, uint32_t *or_cnt, uint64_t cap) {
    const uint64_t n = s->n_scored < cap ? s->n_scored : cap;
    memcpy(slots, s->out_slots, n * 4); memcpy(and_cnt, s->out_and, n * 4); memcpy(or_cnt, s->out_or, n * 4);
    return s->n_scored;
}
uint64_t orc_stepper_pop_log(const orc_stepper_t *s, uint32_t *nodes, uint8_t *levels, uint64_t cap) {
    const uint64_t m = s->n_pops < s->pop_cap ? s->n_pops : s->pop_cap;
    const uint64_t n = m < cap ? m : cap;
    if (n) { memcpy(nodes, s->pop_nodes, n * 4); memcpy(levels, s->pop_levels, n); }
    return m;
}

typedef struct {
    const orc_graph_t *g;
    const uint8_t *corpus;
    size_t row_bytes;
    const uint8_t *queries;
    uint32_t nq;
    uint64_t n_to_score;
    orc_trav_stats_t *stats;
    volatile uint32_t *next;
    int rc;
    uint64_t *hashes;   /* optional: order-sensitive hash of every traversal's scored list */
} many_ctx_t;

/* order-sensitive 64-bit hash of a scored list (position, slot, and | or << 16): a wrap-around sum of mixed terms,
 * so it can be formed in any order — the product computes the same value on the device
 * (radhip_traversal_result_hashes) and bench.py compares the two for its parity sample */
static uint64_t hash_mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
uint64_t orc_result_hash(const uint32_t *slots, const uint32_t *and_cnt, const uint32_t *or_cnt, uint64_t n) {
    uint64_t h = 0;
    for (uint64_t i = 0; i < n; ++i)
        h += hash_mix64(((i << 32) | (uint64_t)slots[i]) + hash_mix64((uint64_t)(and_cnt[i] | (or_cnt[i] << 16))));
    return h;
}

static void *many_worker(void *p) {
    many_ctx_t *c = (many_ctx_t *)p;
    uint64_t cap = c->n_to_score + c->g->cap0 + 64;
    uint32_t *s = (uint32_t *)malloc(cap * 4), *a = (uint32_t *)malloc(cap * 4),
             *o = (uint32_t *)malloc(cap * 4);
    for (;;) {
        uint32_t q = __sync_fetch_and_add(c->next, 1);
        if (q >= c->nq) break;
        int rc = orc_rad_traverse(c->g, c->corpus, c->row_bytes,
                                  c->queries + (uint64_t)q * c->row_bytes,
                                  c->n_to_score, 0, s, a, o, cap, NULL, NULL, 0,
                                  &c->stats[q]);
        if (rc) c->rc = rc;
        if (c->hashes) c->hashes[q] = orc_result_hash(s, a, o, c->stats[q].n_scored < cap ? c->stats[q].n_scored : cap);
    }
    free(s); free(a); free(o);
    return NULL;
}

int orc_rad_traverse_many(const orc_graph_t *g, const uint8_t *corpus,
                          size_t row_bytes, const uint8_t *queries, uint32_t nq,
                          uint64_t n_to_score, int n_threads,
                          orc_trav_stats_t *stats_out) {
    return orc_rad_traverse_many_h(g, corpus, row_bytes, queries, nq, n_to_score, n_threads, stats_out, NULL);
}

int orc_rad_traverse_many_h(const orc_graph_t *g, const uint8_t *corpus,
                            size_t row_bytes, const uint8_t *queries, uint32_t nq,
                            uint64_t n_to_score, int n_threads,
                            orc_trav_stats_t *stats_out, uint64_t *hashes_out) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    volatile uint32_t next = 0;
    many_ctx_t ctx[256];
    pthread_t th[256];
    for (int t = 0; t < n_threads; ++t) {
        many_ctx_t c = {g, corpus, row_bytes, queries, nq, n_to_score, stats_out, &next, 0, hashes_out};
        ctx[t] = c;
        pthread_create(&th[t], NULL, many_worker, &ctx[t]);
    }
    int rc = 0;
    for (int t = 0; t < n_threads; ++t) {
        pthread_join(th[t], NULL);
        if (ctx[t].rc) rc = ctx[t].rc;
    }
    return rc;
}

/* ====================================================================== */
/* usearch-shaped HNSW (parity unpinned vs usearch)       */
/* ====================================================================== */

/* total order of candidates: exact rational distance (1 - and/or), ties by
 * slot.  d1 < d2  <=>  and1*or2 > and2*or1  (or==0 counts as distance 0). */
typedef struct {
    uint32_t and_cnt, or_cnt, slot;
} cand_t;

static int dist_cmp(uint32_t a1, uint32_t o1, uint32_t a2, uint32_t o2) {
    /* returns <0 if d1<d2, 0 if equal, >0 if d1>d2 */
    uint64_t l, r;
    if (o1 == 0 && o2 == 0) return 0;
    if (o1 == 0) return (a2 == o2) ? 0 : -1; /* d1 = 0 */
    if (o2 == 0) return (a1 == o1) ? 0 : 1;
    l = (uint64_t)a1 * o2;
    r = (uint64_t)a2 * o1;
    return (l > r) ? -1 : (l < r) ? 1 : 0;
}

static int cand_less(const cand_t *x, const cand_t *y) {
    int c = dist_cmp(x->and_cnt, x->or_cnt, y->and_cnt, y->or_cnt);
    if (c) return c < 0;
    return x->slot < y->slot;
}

struct orc_hnsw {
    uint32_t ndim_bits, M, cap0, ef_add;
    uint64_t seed;
    size_t row_bytes;
    uint64_t n, cap_nodes;
    uint8_t *rows;
    int8_t *levels;
    uint32_t *adj0;
    uint32_t *upper_row;
    uint32_t *adjU;
    uint64_t n_upper_rows, cap_upper_rows;
    int32_t max_level;
    uint32_t entry;
    uint32_t *stamp; /* visited epochs */
    uint32_t epoch;
};

int orc_hnsw_level_of(uint64_t seed, uint64_t slot, uint32_t connectivity) {
    uint64_t h = h64(seed ^ (slot * 0x9E3779B97F4A7C15ULL + 0x632BE59BD9B4E019ULL));
    int l = 0;
    while (l < 15 && (h % connectivity) == 0) {
        h /= connectivity;
        l++;
    }
    return l;
}

orc_hnsw_t *orc_hnsw_create(uint32_t ndim_bits, uint32_t connectivity,
                            uint32_t connectivity_base, uint32_t expansion_add,
                            uint64_t seed) {
    orc_hnsw_t *h = (orc_hnsw_t *)calloc(1, sizeof(*h));
    h->ndim_bits = ndim_bits;
    h->M = connectivity;
    h->cap0 = connectivity_base ? connectivity_base : 2 * connectivity;
    h->ef_add = expansion_add;
    h->seed = seed;
    h->row_bytes = (ndim_bits + 7) / 8;
    h->max_level = -1;
    h->entry = ORC_NO_SLOT;
    return h;
}

void orc_hnsw_destroy(orc_hnsw_t *h) {
    if (!h) return;
    free(h->rows); free(h->levels); free(h->adj0); free(h->upper_row);
    free(h->adjU); free(h->stamp); free(h);
}

uint64_t orc_hnsw_size(const orc_hnsw_t *h) { return h->n; }
const uint8_t *orc_hnsw_rows(const orc_hnsw_t *h) { return h->rows; }

void orc_hnsw_graph(const orc_hnsw_t *h, orc_graph_t *out) {
    out->n = h->n; out->cap0 = h->cap0; out->capU = h->M;
    out->max_level = h->max_level; out->entry = h->entry;
    out->levels = h->levels; out->adj0 = h->adj0; out->upper_row = h->upper_row;
    out->adjU = h->adjU; out->n_upper_rows = h->n_upper_rows;
}

static void hnsw_reserve(orc_hnsw_t *h, uint64_t n_total) {
    if (n_total <= h->cap_nodes) return;
    uint64_t nc = h->cap_nodes ? h->cap_nodes : 1024;
    while (nc < n_total) nc *= 2;
    h->rows = (uint8_t *)realloc(h->rows, nc * h->row_bytes);
    h->levels = (int8_t *)realloc(h->levels, nc);
    h->adj0 = (uint32_t *)realloc(h->adj0, nc * h->cap0 * 4);
    h->upper_row = (uint32_t *)realloc(h->upper_row, nc * 4);
    h->stamp = (uint32_t *)realloc(h->stamp, nc * 4);
    memset(h->stamp + h->cap_nodes, 0, (nc - h->cap_nodes) * 4);
    h->cap_nodes = nc;
}

static uint32_t *hnsw_row(const orc_hnsw_t *h, uint32_t slot, int level, uint32_t *cap) {
    if (level == 0) { *cap = h->cap0; return h->adj0 + (uint64_t)slot * h->cap0; }
    *cap = h->M;
    return h->adjU + ((uint64_t)h->upper_row[slot] + (uint64_t)(level - 1)) * h->M;
}

static cand_t eval_cand(const uint8_t *rows, size_t rb, const uint8_t *q, uint32_t slot) {
    cand_t c; c.slot = slot;
    orc_tanimoto_counts(q, rows + (uint64_t)slot * rb, rb, &c.and_cnt, &c.or_cnt);
    return c;
}

/* generic best-first layer search over abstract row accessor ------------- */
typedef struct {
    const uint8_t *rows; size_t rb;
    /* graph access */
    const orc_hnsw_t *h; const orc_graph_t *g;
    uint32_t *stamp; uint32_t epoch;
    uint64_t n_evals, n_pops;
} search_ctx_t;

static const uint32_t *ctx_row(const search_ctx_t *c, uint32_t slot, int level, uint32_t *cap) {
    if (c->h) return hnsw_row(c->h, slot, level, cap);
    return graph_row(c->g, slot, level, cap);
}

/* sorted ascending array "top" of at most ef entries, with expanded flags.
 * Equivalent to usearch's (next-candidates min-heap + bounded top buffer):
 * a candidate that is not among the ef best once the buffer is full can only
 * terminate the loop, never change the result. */
static uint32_t search_layer(search_ctx_t *c, const uint8_t *q, const cand_t *entries,
                             uint32_t n_entries, uint32_t ef, int level,
                             cand_t *top /* [ef+1] */) {
    uint8_t *expanded = (uint8_t *)calloc(ef + 1, 1);
    uint32_t n_top = 0;
    for (uint32_t e = 0; e < n_entries; ++e) {
        c->stamp[entries[e].slot] = c->epoch;
        uint32_t p = n_top;
        while (p > 0 && cand_less(&entries[e], &top[p - 1])) { top[p] = top[p - 1]; expanded[p] = expanded[p - 1]; p--; }
        top[p] = entries[e]; expanded[p] = 0;
        if (n_top < ef) n_top++;
    }
    for (;;) {
        uint32_t i = 0;
        while (i < n_top && expanded[i]) i++;
        if (i == n_top) break;
        expanded[i] = 1;
        cand_t cur = top[i];
        c->n_pops++;
        uint32_t cap = 0;
        const uint32_t *row = ctx_row(c, cur.slot, level, &cap);
        for (uint32_t j = 0; j < cap && row[j] != ORC_NO_SLOT; ++j) {
            uint32_t nb = row[j];
            if (c->stamp[nb] == c->epoch) continue;
            c->stamp[nb] = c->epoch;
            cand_t x = eval_cand(c->rows, c->rb, q, nb);
            c->n_evals++;
            if (n_top == ef && !cand_less(&x, &top[n_top - 1])) continue;
            uint32_t p = (n_top < ef) ? n_top : n_top - 1;
            while (p > 0 && cand_less(&x, &top[p - 1])) { top[p] = top[p - 1]; expanded[p] = expanded[p - 1]; p--; }
            top[p] = x; expanded[p] = 0;
            if (n_top < ef) n_top++;
        }
    }
    free(expanded);
    return n_top;
}

/* greedy 1-best descent on one level (usearch search_for_one, [RECALLED]) */
static cand_t greedy_level(search_ctx_t *c, const uint8_t *q, cand_t cur, int level) {
    int changed = 1;
    while (changed) {
        changed = 0;
        c->n_pops++;
        uint32_t cap = 0;
        const uint32_t *row = ctx_row(c, cur.slot, level, &cap);
        cand_t best = cur;
        for (uint32_t j = 0; j < cap && row[j] != ORC_NO_SLOT; ++j) {
            cand_t x = eval_cand(c->rows, c->rb, q, row[j]);
            c->n_evals++;
            if (cand_less(&x, &best)) { best = x; changed = 1; }
        }
        cur = best;
    }
    return cur;
}

/* neighbour-selection heuristic (hnswlib getNeighborsByHeuristic2 shape):
 * scan candidates in ascending distance to base; keep c unless an already kept
 * a is strictly closer to c than base is. */
static uint32_t select_heuristic(const uint8_t *rows, size_t rb, const cand_t *cands,
                                 uint32_t n, uint32_t m, uint32_t *out) {
    uint32_t k = 0;
    for (uint32_t i = 0; i < n && k < m; ++i) {
        int good = 1;
        for (uint32_t a = 0; a < k; ++a) {
            uint32_t aa, oo;
            orc_tanimoto_counts(rows + (uint64_t)out[a] * rb, rows + (uint64_t)cands[i].slot * rb, rb, &aa, &oo);
            if (dist_cmp(aa, oo, cands[i].and_cnt, cands[i].or_cnt) < 0) { good = 0; break; }
        }
        if (good) out[k++] = cands[i].slot;
    }
    return k;
}

typedef struct { uint32_t target, source; int8_t level; } rev_req_t;

static int rev_cmp(const void *a, const void *b) {
    const rev_req_t *x = (const rev_req_t *)a, *y = (const rev_req_t *)b;
    if (x->target != y->target) return x->target < y->target ? -1 : 1;
    if (x->level != y->level) return x->level < y->level ? -1 : 1;
    if (x->source != y->source) return x->source < y->source ? -1 : 1;
    return 0;
}

static int cand_qsort(const void *a, const void *b) {
    const cand_t *x = (const cand_t *)a, *y = (const cand_t *)b;
    if (cand_less(x, y)) return -1;
    if (cand_less(y, x)) return 1;
    return 0;
}

/* batch schedule shared with the product: size = clamp(start/16, 1, max_batch) */
static uint64_t batch_size_at(uint64_t start, uint32_t max_batch) {
    uint64_t s = start / 16;
    if (s < 1) s = 1;
    if (s > max_batch) s = max_batch;
    return s;
}

int orc_hnsw_add(orc_hnsw_t *h, const uint8_t *rows_in, uint64_t count, uint32_t max_batch) {
    if (max_batch < 1) max_batch = 1;
    uint64_t first = h->n, total = h->n + count;
    hnsw_reserve(h, total);
    memcpy(h->rows + first * h->row_bytes, rows_in, count * h->row_bytes);
    const size_t rb = h->row_bytes;
    cand_t *top = (cand_t *)malloc((h->ef_add + 2) * sizeof(cand_t));
    uint32_t maxcap = h->cap0 > h->M ? h->cap0 : h->M;
    cand_t *tmp = (cand_t *)malloc((maxcap + 2) * sizeof(cand_t));
    uint32_t *sel = (uint32_t *)malloc((maxcap + 2) * 4);

    uint64_t start = first;
    while (start < total) {
        uint64_t bs = batch_size_at(start, max_batch);
        if (start + bs > total) bs = total - start;
        uint64_t end = start + bs;
        /* levels + row allocation for the batch (phase 0) */
        for (uint64_t i = start; i < end; ++i) {
            int lv = orc_hnsw_level_of(h->seed, i, h->M);
            h->levels[i] = (int8_t)lv;
            for (uint32_t j = 0; j < h->cap0; ++j) h->adj0[i * h->cap0 + j] = ORC_NO_SLOT;
            if (lv > 0) {
                if (h->n_upper_rows + (uint64_t)lv > h->cap_upper_rows) {
                    uint64_t nc = h->cap_upper_rows ? h->cap_upper_rows * 2 : 1024;
                    while (nc < h->n_upper_rows + (uint64_t)lv) nc *= 2;
                    h->adjU = (uint32_t *)realloc(h->adjU, nc * h->M * 4);
                    h->cap_upper_rows = nc;
                }
                h->upper_row[i] = (uint32_t)h->n_upper_rows;
                for (uint64_t r = 0; r < (uint64_t)lv * h->M; ++r)
                    h->adjU[h->n_upper_rows * h->M + r] = ORC_NO_SLOT;
                h->n_upper_rows += (uint64_t)lv;
            } else {
                h->upper_row[i] = ORC_NO_SLOT;
            }
        }
        /* phase A+B: every new node searches the pre-batch snapshot and writes
         * its own rows; reverse-edge requests are collected */
        uint64_t req_cap = bs * 16 * (uint64_t)maxcap + 16, n_req = 0;
        rev_req_t *req = (rev_req_t *)malloc(req_cap * sizeof(rev_req_t));
        const int32_t snap_max_level = h->max_level;
        const uint32_t snap_entry = h->entry;
        for (uint64_t i = start; i < end; ++i) {
            if (snap_entry == ORC_NO_SLOT) continue; /* very first node */
            const uint8_t *q = h->rows + i * rb;
            search_ctx_t c = {h->rows, rb, h, NULL, h->stamp, 0, 0, 0};
            cand_t cur = eval_cand(h->rows, rb, q, snap_entry);
            int lv = h->levels[i];
            for (int l = snap_max_level; l > lv; --l) cur = greedy_level(&c, q, cur, l);
            for (int l = (lv < snap_max_level ? lv : snap_max_level); l >= 0; --l) {
                if (++h->epoch == 0) { memset(h->stamp, 0, h->cap_nodes * 4); h->epoch = 1; }
                c.epoch = h->epoch;
                uint32_t nt = search_layer(&c, q, &cur, 1, h->ef_add, l, top);
                uint32_t cap_l = l == 0 ? h->cap0 : h->M;
                uint32_t ns = select_heuristic(h->rows, rb, top, nt, cap_l, sel);
                uint32_t capr; uint32_t *row = hnsw_row(h, (uint32_t)i, l, &capr);
                for (uint32_t j = 0; j < ns; ++j) {
                    row[j] = sel[j];
                    rev_req_t r = {sel[j], (uint32_t)i, (int8_t)l};
                    req[n_req++] = r;
                }
                cur = top[0];
            }
        }
        /* phase C: reverse edges, deterministic order (target, level, source) */
        qsort(req, n_req, sizeof(rev_req_t), rev_cmp);
        for (uint64_t r = 0; r < n_req; ++r) {
            uint32_t t = req[r].target, s = req[r].source; int l = req[r].level;
            uint32_t capr; uint32_t *row = hnsw_row(h, t, l, &capr);
            uint32_t cnt = 0;
            while (cnt < capr && row[cnt] != ORC_NO_SLOT) cnt++;
            if (cnt < capr) { row[cnt] = s; continue; }
            const uint8_t *base = h->rows + (uint64_t)t * rb;
            for (uint32_t j = 0; j < cnt; ++j) tmp[j] = eval_cand(h->rows, rb, base, row[j]);
            tmp[cnt] = eval_cand(h->rows, rb, base, s);
            qsort(tmp, cnt + 1, sizeof(cand_t), cand_qsort);
            uint32_t ns = select_heuristic(h->rows, rb, tmp, cnt + 1, capr, sel);
            for (uint32_t j = 0; j < capr; ++j) row[j] = j < ns ? sel[j] : ORC_NO_SLOT;
        }
        free(req);
        /* phase D: entry point / max level */
        for (uint64_t i = start; i < end; ++i)
            if (h->levels[i] > h->max_level) { h->max_level = h->levels[i]; h->entry = (uint32_t)i; }
        h->n = end;
        start = end;
    }
    free(top); free(tmp); free(sel);
    return 0;
}

static uint32_t do_search(search_ctx_t *c, const uint8_t *query, uint32_t entry, int max_level,
                          uint32_t k, uint32_t ef, uint32_t *out_slots, uint32_t *out_and,
                          uint32_t *out_or) {
    if (entry == ORC_NO_SLOT) return 0;
    if (ef < k) ef = k;
    cand_t *top = (cand_t *)malloc((ef + 2) * sizeof(cand_t));
    cand_t cur = eval_cand(c->rows, c->rb, query, entry);
    c->n_evals++;
    for (int l = max_level; l > 0; --l) cur = greedy_level(c, query, cur, l);
    uint32_t nt = search_layer(c, query, &cur, 1, ef, 0, top);
    if (nt > k) nt = k;
    for (uint32_t i = 0; i < nt; ++i) {
        out_slots[i] = top[i].slot;
        if (out_and) out_and[i] = top[i].and_cnt;
        if (out_or) out_or[i] = top[i].or_cnt;
    }
    free(top);
    return nt;
}

uint32_t orc_hnsw_search(const orc_hnsw_t *h, const uint8_t *query, uint32_t k,
                         uint32_t ef, uint32_t *out_slots, uint32_t *out_and,
                         uint32_t *out_or, uint64_t *n_evals, uint64_t *n_pops) {
    orc_hnsw_t *hm = (orc_hnsw_t *)h;
    if (++hm->epoch == 0) { memset(hm->stamp, 0, hm->cap_nodes * 4); hm->epoch = 1; }
    search_ctx_t c = {h->rows, h->row_bytes, h, NULL, hm->stamp, hm->epoch, 0, 0};
    uint32_t n = do_search(&c, query, h->entry, h->max_level, k, ef, out_slots, out_and, out_or);
    if (n_evals) *n_evals = c.n_evals;
    if (n_pops) *n_pops = c.n_pops;
    return n;
}

uint32_t orc_graph_search(const orc_graph_t *g, const uint8_t *corpus,
                          size_t row_bytes, const uint8_t *query, uint32_t k,
                          uint32_t ef, uint32_t *out_slots, uint32_t *out_and,
                          uint32_t *out_or, uint64_t *n_evals, uint64_t *n_pops) {
    uint32_t *stamp = (uint32_t *)calloc(g->n, 4);
    search_ctx_t c = {corpus, row_bytes, NULL, g, stamp, 1, 0, 0};
    uint32_t n = do_search(&c, query, g->entry, g->max_level, k, ef, out_slots, out_and, out_or);
    if (n_evals) *n_evals = c.n_evals;
    if (n_pops) *n_pops = c.n_pops;
    free(stamp);
    return n;
}

/* ====================================================================== */
/* synthetic corpus / graph (closed form)                                 */
/* ====================================================================== */

static inline uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
static inline uint64_t h3(uint64_t seed, uint64_t a, uint64_t b) {
    return mix64(mix64(seed ^ (a * 0xD6E8FEB86659FD93ULL)) + b);
}
/* AND of k hashed words: each bit set with probability 2^-k */
static inline uint64_t sparse_word(uint64_t seed, uint64_t a, uint64_t b, int k) {
    uint64_t w = ~0ULL;
    for (int t = 0; t < k; ++t) w &= h3(seed + (uint64_t)t * 0x100000001B3ULL, a, b);
    return w;
}

#define SYN_CS 32u /* rows per cluster (target)        */
#define SYN_SC 64u /* clusters per super-cluster       */
#define TAG_S  0x5355504552ULL
#define TAG_CD 0x434C5544ULL
#define TAG_CA 0x434C5541ULL
#define TAG_RD 0x524F5744ULL
#define TAG_RA 0x524F5741ULL
#define TAG_G0 0x4752415048ULL

static inline uint64_t syn_nc(uint64_t n) { uint64_t nc = n / SYN_CS; return nc ? nc : 1; }

#define TAG_H0 0x48494552ULL
#define TAG_HD 0x48494544ULL
#define TAG_HA 0x48494541ULL
#define SYN_HMUL 0x9E3779B1ULL

/* mode 2: a 4-ary hierarchy with neighbourhood structure at every scale.  Row r sits on leaf
 * (r * odd) mod 4^D of a tree of depth D = ceil(log4 n_total) (a bijection: rows that are close in the
 * tree are scattered over the slots); every tree node keeps 15/16 of its parent's set bits and gains
 * 1/256 of the clear ones (density stays ~1/17), so two rows whose lowest common ancestor is h levels
 * up share ~(15/16)^(2h) of their bits: similarity falls smoothly with h instead of in two steps. */
static int syn_depth(uint64_t n_total) {
    int d = 1;
    while (d < 31 && (1ULL << (2 * d)) < n_total) d++;
    return d;
}
static uint64_t synth_word_h(uint64_t seed, uint64_t row, uint64_t n_total, uint32_t w) {
    const int D = syn_depth(n_total);
    const uint64_t leaf = (row * SYN_HMUL) & ((1ULL << (2 * D)) - 1ULL);
    uint64_t x = sparse_word(seed ^ TAG_H0, 0, w, 4);
    for (int d = 1; d <= D; ++d) {
        const uint64_t a = leaf >> (2 * (D - d));
        const uint64_t lv = (uint64_t)d << 40;
        x = (x & ~sparse_word((seed ^ TAG_HD) + lv, a, w, 4)) | sparse_word((seed ^ TAG_HA) + lv, a, w, 8);
    }
    return x;
}

static uint64_t synth_word(uint64_t seed, uint64_t row, uint64_t n_total, uint32_t w, int mode) {
    if (mode == 0) return h3(seed, row, w);
    if (mode == 2) return synth_word_h(seed, row, n_total, w);
    uint64_t nc = syn_nc(n_total);
    uint64_t c = row % nc, s = c / SYN_SC;
    uint64_t sb = sparse_word(seed ^ TAG_S, s, w, 4);
    uint64_t cb = (sb & ~sparse_word(seed ^ TAG_CD, c, w, 2)) | sparse_word(seed ^ TAG_CA, c, w, 6);
    return (cb & ~sparse_word(seed ^ TAG_RD, row, w, 3)) | sparse_word(seed ^ TAG_RA, row, w, 6);
}

void orc_synth_rows(uint8_t *out, uint64_t first_row, uint64_t n_rows,
                    uint64_t n_total, uint32_t ndim_bits, uint64_t seed, int mode) {
    uint32_t nbytes = (ndim_bits + 7) / 8;
    uint32_t nw = (nbytes + 7) / 8;
    for (uint64_t r = 0; r < n_rows; ++r) {
        uint8_t *dst = out + r * nbytes;
        for (uint32_t w = 0; w < nw; ++w) {
            uint64_t x = synth_word(seed, first_row + r, n_total, w, mode);
            uint32_t take = nbytes - w * 8 < 8 ? nbytes - w * 8 : 8;
            memcpy(dst + w * 8, &x, take); /* little-endian byte order */
        }
        /* clear padding bits beyond ndim in the last byte (MSB-first packbits
         * convention is irrelevant to popcount; we zero the high bits) */
        if (ndim_bits % 8) dst[nbytes - 1] &= (uint8_t)((1u << (ndim_bits % 8)) - 1u);
    }
}

int32_t orc_synth_max_level(uint64_t n, uint32_t M) {
    int32_t l = 0;
    uint64_t p = 1;
    while (l < 15 && p * M < n) { p *= M; l++; }
    return l; /* largest l with M^l < n */
}

static uint64_t ipow(uint64_t b, int e) { uint64_t r = 1; while (e-- > 0) r *= b; return r; }

static int synth_level(uint64_t r, uint32_t M, int32_t L) {
    if (r == 0) return L;
    int l = 0;
    while (l < L && r % M == 0) { r /= M; l++; }
    return l;
}

uint64_t orc_synth_upper_rows(uint64_t n, uint32_t M) {
    int32_t L = orc_synth_max_level(n, M);
    uint64_t tot = 0;
    for (int l = 1; l <= L; ++l) { uint64_t p = ipow(M, l); tot += (n + p - 1) / p; }
    return tot;
}

void orc_synth_graph(uint64_t n, uint32_t M, uint32_t cap0, uint64_t seed,
                     int8_t *levels, uint32_t *adj0, uint32_t *upper_row, uint32_t *adjU) {
    const int32_t L = orc_synth_max_level(n, M);
    const uint64_t nc = syn_nc(n);
    const uint32_t H = cap0 / 2, LK = cap0 - H, LN = LK / 2, LF = LK - LN;
    const uint64_t gs = seed ^ TAG_G0;
    for (uint64_t r = 0; r < n; ++r) {
        int lv = synth_level(r, M, L);
        levels[r] = (int8_t)lv;
        /* ---- level 0 ---- */
        uint32_t *row = adj0 + r * cap0;
        uint32_t k = 0;
        uint64_t c = r % nc, m = r / nc;
        uint64_t cs = (n - c + nc - 1) / nc; /* rows in cluster c */
        for (uint32_t j = 0; j < H && j + 1 < cs; ++j) {
            uint64_t m2 = (m + 1 + j) % cs;
            row[k++] = (uint32_t)(c + m2 * nc);
        }
        uint64_t s0 = (c / SYN_SC) * SYN_SC;
        uint64_t ss = (s0 + SYN_SC <= nc) ? SYN_SC : nc - s0; /* clusters in this super-cluster */
        for (uint32_t jj = 0; jj < LN; ++jj) { /* near links: same super-cluster */
            uint64_t off;
            uint64_t hh = h3(gs, r, jj);
            if (ss - 1 >= LN) { uint64_t bw = (ss - 1) / LN; off = 1 + jj * bw + hh % bw; }
            else if (jj + 1 < ss) off = 1 + jj;
            else continue;
            uint64_t c2 = s0 + ((c - s0) + off) % ss;
            uint64_t cs2 = (n - c2 + nc - 1) / nc;
            uint64_t m2 = (hh >> 32) % cs2;
            row[k++] = (uint32_t)(c2 + m2 * nc);
        }
        if (nc >= 4 * (uint64_t)SYN_SC) { /* far links: cyclic cluster distance >= SC */
            uint64_t span = nc - 2 * (uint64_t)SYN_SC + 1; /* offsets SC .. nc-SC */
            for (uint32_t jj = 0; jj < LF; ++jj) {
                uint64_t hh = h3(gs, r, 1000 + jj);
                uint64_t bw = span / LF;
                uint64_t off = SYN_SC + jj * bw + hh % bw;
                uint64_t c2 = (c + off) % nc;
                uint64_t cs2 = (n - c2 + nc - 1) / nc;
                uint64_t m2 = (hh >> 32) % cs2;
                row[k++] = (uint32_t)(c2 + m2 * nc);
            }
        }
        while (k < cap0) row[k++] = ORC_NO_SLOT;
        /* ---- upper levels ---- */
        if (lv == 0) { upper_row[r] = ORC_NO_SLOT; continue; }
        uint64_t base = 0;
        for (int l = 1; l <= L; ++l) { uint64_t p = ipow(M, l); base += (r + p - 1) / p; }
        upper_row[r] = (uint32_t)base;
        for (int l = 1; l <= lv; ++l) {
            uint32_t *ur = adjU + (base + (uint64_t)(l - 1)) * M;
            uint64_t p = ipow(M, l), kk = r / p, nl = (n + p - 1) / p;
            uint32_t u = 0;
            for (uint32_t j = 0; j < M; ++j) {
                uint64_t off;
                if (nl - 1 >= M) { uint64_t bw = (nl - 1) / M; off = 1 + j * bw + h3(gs, r, 2000 + 64 * (uint64_t)l + j) % bw; }
                else if (j + 1 < nl) off = 1 + j;
                else break;
                ur[u++] = (uint32_t)(((kk + off) % nl) * p);
            }
            while (u < M) ur[u++] = ORC_NO_SLOT;
        }
    }
}
