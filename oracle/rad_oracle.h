/*
 * rad_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the algorithms on RAD's HNSW neighbor-expansion hot
 * path, used only as the checker by tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py.  Nothing under rad_amd/ may include, link or
 * call this file.
 *
 * What each function restates (paths relative to the reference tree):
 *   orc_tanimoto_counts / orc_distance_f32
 *       usearch metric='tanimoto' on dtype='b1' selected at README.md:49-50,
 *       examples/DUDEZ_example.ipynb:185-186, tests/test_hnsw_service.py:20-21.
 *       The usearch source is NOT in the reference tree (empty submodule,
 *       .gitmodules:1-3, no pinned version): the integer (and, or) popcounts are
 *       mathematically unambiguous; the float edge value is this build's stated
 *       convention  d = 1.0f - (float)and / (float)or , d = 0.0f when or == 0.
 *   orc_rad_traverse
 *       the sequential RAD control flow  rad/traverser.py:128-176 (prime),
 *       rad/coordination_service.py:290-347 (request_work),
 *       rad/distributed_worker.py:272-333 (_process_work_item),
 *       rad/coordination_service.py:349-413 (submit_work_results), over
 *       rad/priority_queue.py:22-42 (pop-min, ties by bytewise member order of
 *       "{node_id}:{level}"), rad/visited.py:17-29 (test-and-set on
 *       (node_id, level)), rad/scored.py:37-61 (insert-if-absent, insertion
 *       order), with scoring_fn = Tanimoto distance to a query fingerprint.
 *   orc_hnsw_*
 *       usearch-shaped HNSW insert / search ([RECALLED] upstream algorithm,
 *       SURVEY.md §3.5) — PARITY UNPINNED against usearch itself: no usearch
 *       binary, source or golden vector exists in the reference tree.
 *   orc_synth_*
 *       closed-form synthetic corpus / graph generators (data, not the path).
 */
#ifndef RAD_ORACLE_H
#define RAD_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NO_SLOT 0xFFFFFFFFu

/* ---- A1: Tanimoto on packed bits ------------------------------------- */
void orc_tanimoto_counts(const uint8_t *a, const uint8_t *b, size_t nbytes,
                         uint32_t *and_out, uint32_t *or_out);
float orc_distance_f32(uint32_t and_cnt, uint32_t or_cnt);
/* one query against n rows (row stride = row_bytes) */
void orc_scan(const uint8_t *corpus, uint64_t n, size_t row_bytes,
              const uint8_t *query, uint32_t *and_out, uint32_t *or_out);
/* candidate list */
void orc_gather(const uint8_t *corpus, size_t row_bytes, const uint8_t *query,
                const uint32_t *slots, uint64_t n_slots, uint32_t *and_out,
                uint32_t *or_out);

/* ---- graph store (same layout the product's load_graph takes) -------- */
typedef struct {
    uint64_t n;              /* nodes                                        */
    uint32_t cap0;           /* level-0 row width (connectivity_base)        */
    uint32_t capU;           /* upper-level row width (connectivity)         */
    int32_t max_level;       /* 0-based top level                            */
    uint32_t entry;          /* entry slot                                   */
    const int8_t *levels;    /* [n] level of each node                       */
    const uint32_t *adj0;    /* [n * cap0], ORC_NO_SLOT padded               */
    const uint32_t *upper_row; /* [n] first upper row of node, or NO_SLOT    */
    const uint32_t *adjU;    /* [n_upper_rows * capU], ORC_NO_SLOT padded    */
    uint64_t n_upper_rows;
} orc_graph_t;

/* neighbors of (slot, level) in stored order; returns count, -1 if the node
 * does not exist on that level (A3: index.get_neighbors, rad/hnsw_service.py:222) */
int orc_graph_neighbors(const orc_graph_t *g, uint32_t slot, int level,
                        uint32_t *out, uint32_t out_cap);
/* A4: index.get_top_level_nodes (rad/hnsw_service.py:229): slots with
 * level == max_level, ascending */
uint64_t orc_graph_top_level(const orc_graph_t *g, uint32_t *out, uint64_t cap);

/* ---- RAD traversal, Tanimoto-scored ---------------------------------- */
typedef struct {
    uint64_t n_scored;   /* entries written to out_* (traversal order)        */
    uint64_t n_pops;     /* node expansions performed                         */
    uint64_t n_evals;    /* Tanimoto evaluations (== n_scored)                */
    uint64_t n_nbr;      /* adjacency entries examined                        */
    /* best item left in the queue when the traversal stopped (frontier candidate) */
    uint32_t f_valid, f_and, f_or, f_slot, f_level, f_pad;
} orc_trav_stats_t;

/* Runs prime + best-first traversal until n_scored >= n_to_score, the queue
 * is empty, or max_pops expansions were done (0 = unlimited).
 * out_slots/out_and/out_or: scored set in insertion order (cap >= n_to_score + cap0).
 * pop_nodes/pop_levels (optional, may be NULL): expansion log. */
int orc_rad_traverse(const orc_graph_t *g, const uint8_t *corpus,
                     size_t row_bytes, const uint8_t *query,
                     uint64_t n_to_score, uint64_t max_pops,
                     uint32_t *out_slots, uint32_t *out_and, uint32_t *out_or,
                     uint64_t out_cap, uint32_t *pop_nodes, uint8_t *pop_levels,
                     uint64_t pop_cap, orc_trav_stats_t *stats);

/* nq independent traversals on n_threads pthreads (cpu_baseline leg).
 * queries: nq rows of row_bytes.  Only stats are kept. */
int orc_rad_traverse_many(const orc_graph_t *g, const uint8_t *corpus,
                          size_t row_bytes, const uint8_t *queries, uint32_t nq,
                          uint64_t n_to_score, int n_threads,
                          orc_trav_stats_t *stats_out /* [nq] */);
/* the same, also returning orc_result_hash of every traversal's scored list (hashes_out[nq], may be NULL) */
int orc_rad_traverse_many_h(const orc_graph_t *g, const uint8_t *corpus,
                            size_t row_bytes, const uint8_t *queries, uint32_t nq,
                            uint64_t n_to_score, int n_threads,
                            orc_trav_stats_t *stats_out, uint64_t *hashes_out);
uint64_t orc_result_hash(const uint32_t *slots, const uint32_t *and_cnt, const uint32_t *or_cnt, uint64_t n);

/* ---- the same traversal cut at the fingerprint read (row-sharded multi-GPU mode) ----
 * The control flow of orc_rad_traverse as a stepper that never touches the corpus: one call applies the
 * (and, or) counts of the slots it asked for last time, then primes / pops / expands until some
 * neighbour needs a score, and returns those slots.  See rad_oracle.c for the reference lines. */
typedef struct orc_stepper orc_stepper_t;
orc_stepper_t *orc_stepper_create(const orc_graph_t *g, uint64_t n_to_score, uint64_t pop_cap);
void orc_stepper_destroy(orc_stepper_t *s);
int orc_stepper_step(orc_stepper_t *s, const uint32_t *and_in, const uint32_t *or_in, uint32_t *req_out,
                     uint32_t req_cap);
int orc_stepper_status(const orc_stepper_t *s);   /* 0 running, 1 n_to_score reached, 2 queue empty */
void orc_stepper_stats(const orc_stepper_t *s, orc_trav_stats_t *st);
uint64_t orc_stepper_results(const orc_stepper_t *s, uint32_t *slots, uint32_t *and_cnt, uint32_t *or_cnt, uint64_t cap);
uint64_t orc_stepper_pop_log(const orc_stepper_t *s, uint32_t *nodes, uint8_t *levels, uint64_t cap);

/* ---- usearch-shaped HNSW (parity unpinned vs usearch) ---------------- */
typedef struct orc_hnsw orc_hnsw_t;
orc_hnsw_t *orc_hnsw_create(uint32_t ndim_bits, uint32_t connectivity,
                            uint32_t connectivity_base, uint32_t expansion_add,
                            uint64_t seed);
void orc_hnsw_destroy(orc_hnsw_t *h);
/* level of the node that will occupy `slot` (integer-only geometric draw) */
int orc_hnsw_level_of(uint64_t seed, uint64_t slot, uint32_t connectivity);
/* insert rows [first, first+count) of `rows` in batches following the
 * deterministic batch schedule (max_batch = 1 -> classical sequential insert) */
int orc_hnsw_add(orc_hnsw_t *h, const uint8_t *rows, uint64_t count,
                 uint32_t max_batch);
uint64_t orc_hnsw_size(const orc_hnsw_t *h);
/* export the graph in orc_graph_t layout; pointers stay owned by h */
void orc_hnsw_graph(const orc_hnsw_t *h, orc_graph_t *out);
const uint8_t *orc_hnsw_rows(const orc_hnsw_t *h);
/* k nearest by best-first search with expansion ef; returns count */
uint32_t orc_hnsw_search(const orc_hnsw_t *h, const uint8_t *query, uint32_t k,
                         uint32_t ef, uint32_t *out_slots, uint32_t *out_and,
                         uint32_t *out_or, uint64_t *n_evals, uint64_t *n_pops);
/* same, over an externally supplied graph + corpus */
uint32_t orc_graph_search(const orc_graph_t *g, const uint8_t *corpus,
                          size_t row_bytes, const uint8_t *query, uint32_t k,
                          uint32_t ef, uint32_t *out_slots, uint32_t *out_and,
                          uint32_t *out_or, uint64_t *n_evals, uint64_t *n_pops);

/* ---- synthetic data (closed form; the product has its own HIP copy) --- */
/* mode 0: every bit Bernoulli(0.5); mode 1: clustered sparse (ECFP-like) */
void orc_synth_rows(uint8_t *out, uint64_t first_row, uint64_t n_rows,
                    uint64_t n_total, uint32_t ndim_bits, uint64_t seed,
                    int mode);
int32_t orc_synth_max_level(uint64_t n, uint32_t connectivity);
uint64_t orc_synth_upper_rows(uint64_t n, uint32_t connectivity);
/* fills levels[n], adj0[n*cap0], upper_row[n], adjU[n_upper_rows*capU] */
void orc_synth_graph(uint64_t n, uint32_t connectivity, uint32_t cap0,
                     uint64_t seed, int8_t *levels, uint32_t *adj0,
                     uint32_t *upper_row, uint32_t *adjU);

#ifdef __cplusplus
}
#endif
#endif
