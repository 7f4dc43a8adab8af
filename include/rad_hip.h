/*
 * rad_hip.h — C ABI of librad_hip.so: the MI355X (gfx950) implementation of
 * RAD's HNSW neighbor-expansion hot path.
 *
 * Host code (the Python files of rad_amd, or any other FFI) binds exactly these symbols; no
 * torch / C++ types cross the boundary.  The reference (keiserlab/rad) has no
 * FFI of its own for this path: it calls the pybind11 object `usearch.index.Index`
 * (un-vendored fork, .gitmodules:1-3).  Each entry point below names the
 * reference call it replaces (file:line relative to the reference tree).
 *
 * Conventions
 *   - every function returns int: 0 = ok, negative = RADHIP_E_*; a thread-local
 *     message is readable through radhip_last_error().
 *   - the caller owns every buffer it passes; the library copies what it keeps.
 *   - no Python callbacks, no global state besides the per-process HIP context
 *     that is created lazily by the first call that needs the device (so an
 *     index object can be created in a parent and used in a forked child, as
 *     rad/hnsw_service.py:129-134 does, provided the parent made no device call).
 *   - there is NO CPU fallback: device entry points fail with RADHIP_E_NO_DEVICE
 *     when no gfx950 device is visible.
 *   - fingerprints are packed bits, ceil(ndim/8) bytes per row, any bit order
 *     (np.packbits output, README.md:61); ndim <= 2048.
 *   - "slot" is the 0-based insertion index of a node (RAD's node_id); slots
 *     must be < 1e9 for the RAD traversal (its queue key packs the decimal
 *     member-string order of rad/priority_queue.py:42 into 30+4 bits).
 */
#ifndef RAD_HIP_H
#define RAD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RADHIP_ABI_VERSION 1
#define RADHIP_NO_SLOT 0xFFFFFFFFu

enum {
    RADHIP_OK = 0,
    RADHIP_E_INVALID = -1,    /* bad argument                                  */
    RADHIP_E_NO_DEVICE = -2,  /* no gfx950 device / HIP runtime unavailable    */
    RADHIP_E_HIP = -3,        /* a HIP runtime call failed                     */
    RADHIP_E_NOMEM = -4,      /* host or device allocation failed              */
    RADHIP_E_STATE = -5,      /* call order violated (e.g. no graph loaded)    */
    RADHIP_E_CAPACITY = -6,   /* a fixed-capacity device structure overflowed  */
    RADHIP_E_RANGE = -7,      /* slot / level out of range                     */
    RADHIP_E_COMM = -8        /* RCCL failure                                  */
};

/* ---- library ----------------------------------------------------------- */
const char *radhip_last_error(void);
const char *radhip_backend_name(void); /* "hip:gfx950" */
int radhip_abi_version(void);
/* 16 hex digits: a hash of every source file the library was built from (kernels included) */
const char *radhip_build_id(void);
/* the same for the traversal kernels alone (traverse.hip, traverse4.inc, common.h): the key of measured counters kept beside the code */
const char *radhip_traverse_build_id(void);
int radhip_device_count(int *out_count);
/* Keep a copy of `line`; while armed, SIGABRT / SIGSEGV / SIGBUS / SIGTERM write it to stdout and end the process with status 0
 * (a multi-GPU bench whose measured leg is done arms this before side legs that a GPU fault or a launcher's SIGTERM could end:
 * the line the caller composed says so).  NULL disarms and restores the previous handlers.  The reference's counterpart of surviving a dead worker is the re-queueing of its stale work
 * assignment (rad/coordination_service.py:554-574); a benchmark process has nobody to re-queue for it. */
int radhip_arm_last_words(const char *line);

/* ---- index: corpus + layered adjacency resident in HBM ----------------- */
typedef struct radhip_index radhip_index_t;

typedef struct {
    uint64_t n;              /* nodes                                        */
    uint32_t ndim_bits;
    uint32_t row_bytes;      /* ceil(ndim/8): what the host passes            */
    uint32_t row_stride;     /* bytes per row in HBM (16-B lanes, pow2 count) */
    uint32_t connectivity;   /* upper-level row width                         */
    uint32_t connectivity_base; /* level-0 row width                          */
    uint32_t expansion_add;
    int32_t max_level;       /* 0-based top level, -1 when empty              */
    uint32_t entry;          /* entry slot                                    */
    uint64_t n_upper_rows;
    uint64_t device_bytes;   /* HBM held by this index                        */
    int32_t device;
    int32_t has_vectors;
    int32_t has_graph;
    /* row-sharded index (one rank's rows of a larger corpus): n above is the size of the WHOLE corpus /
     * graph, the rows resident on this device are slots [shard_first, shard_first + shard_rows) */
    int32_t sharded;
    uint64_t shard_first;
    uint64_t shard_rows;
} radhip_index_info_t;

/* replaces usearch Index(ndim=, dtype='b1', metric='tanimoto', connectivity=,
 * expansion_add=) — README.md:47-53, scripts/start_hnsw_server.py:44-50.
 * connectivity_base = 0 selects 2*connectivity.  No device call is made. */
int radhip_index_create(uint32_t ndim_bits, uint32_t connectivity,
                        uint32_t connectivity_base, uint32_t expansion_add,
                        int device, radhip_index_t **out);
int radhip_index_destroy(radhip_index_t *idx);
int radhip_index_info(const radhip_index_t *idx, radhip_index_info_t *out);

/* corpus upload (host rows of row_bytes each) — the vector half of
 * usearch Index.add(keys, fps), README.md:58 */
int radhip_index_load_vectors(radhip_index_t *idx, const uint8_t *rows, uint64_t n);
/* closed-form synthetic corpus generated on the device (bench configs 2-4:
 * avoids a 12.8 GB host transfer).  rows [first_row, first_row+n) of a
 * logical corpus of n_total rows; mode 0 = Bernoulli(0.5) bits, 1 = clustered
 * sparse (ECFP-like, ~7 % density; two cluster levels, random beyond), 2 =
 * hierarchical sparse (~6 % density; a 4-ary tree of depth ceil(log4 n_total),
 * rows scattered over its leaves: neighbourhood structure at every scale). */
int radhip_index_synth_vectors(radhip_index_t *idx, uint64_t n, uint64_t first_row,
                               uint64_t n_total, uint64_t seed, int mode);
int radhip_index_read_vectors(const radhip_index_t *idx, uint64_t first, uint64_t count,
                              uint8_t *out_rows);
/* ONE RANK'S ROWS ONLY (row-sharded multi-GPU mode, BASELINE.json north_star "partition the fingerprint
 * corpus across the 8 GPUs"): rows = the `count` host rows of slots [first, first+count) of a corpus of
 * n_total rows / the same range of the closed-form corpus generated on the device.  The index never
 * allocates, stages or generates the other ranks' rows; it is `sharded` from the start (only the sharded
 * traversal, radhip_shard_*, reads its fingerprints) and takes the graph over all n_total nodes from
 * radhip_index_load_graph, radhip_index_synth_graph or radhip_index_broadcast_graph. */
int radhip_index_load_vectors_shard(radhip_index_t *idx, const uint8_t *rows, uint64_t first, uint64_t count,
                                    uint64_t n_total);
int radhip_index_synth_vectors_shard(radhip_index_t *idx, uint64_t count, uint64_t first_row,
                                     uint64_t n_total, uint64_t seed, int mode);

/* adjacency upload: levels[n] (int8), adj0[n*connectivity_base] and
 * adjU[n_upper_rows*connectivity] padded with RADHIP_NO_SLOT, upper_row[n] =
 * first upper row of a node (its level-l list is row upper_row+l-1) or
 * RADHIP_NO_SLOT.  Rows must not contain duplicates or the node itself.
 * This is what loading a saved graph (usearch Index(path=, view=True,
 * exclude_vectors=True), scripts/start_hnsw_server.py:69) feeds. */
int radhip_index_load_graph(radhip_index_t *idx, uint64_t n, int32_t max_level,
                            uint32_t entry, const int8_t *levels, const uint32_t *adj0,
                            const uint32_t *upper_row, const uint32_t *adjU,
                            uint64_t n_upper_rows);
/* closed-form synthetic layered graph over the n rows already in the index
 * (bench configs 3-4; labelled synthetic everywhere it is reported) */
int radhip_index_synth_graph(radhip_index_t *idx, uint64_t seed);
/* sizes for the read-back buffers are in radhip_index_info */
int radhip_index_read_graph(const radhip_index_t *idx, int8_t *levels, uint32_t *adj0,
                            uint32_t *upper_row, uint32_t *adjU);

/* ---- graph-locality layout (optional; performance only) ------------------- */
/* Renumbers the slots so that graph neighbours get nearby layout ids (greedy graph growing over the
 * level-0 adjacency, host threads; n_threads = 0 picks the host's share).  The traversal kernels then
 * keep their visited / scored set (rad/visited.py:17-29, rad/scored.py:37-47) as 2 bits per node in
 * groups of 384 ids per 128-B line instead of one hash entry per node: the probes of one adjacency row
 * touch as many lines as the row spans groups.  No result depends on the layout — queue keys, scored
 * lists and pop logs use slots; any change of the graph or corpus discards it. */
typedef struct {
    int32_t valid;          /* a layout is installed                                   */
    uint32_t group;         /* ids per 128-B line of the grouped table (384)           */
    uint64_t id_limit;      /* every layout id is below this                           */
    double groups_per_row;  /* mean groups spanned by a level-0 adjacency row (sampled) */
    double degree;          /* mean valid entries of a level-0 row (sampled)            */
    double seconds;         /* host time of the last optimize call                      */
} radhip_layout_info_t;
int radhip_index_optimize_layout(radhip_index_t *idx, uint32_t n_threads);
int radhip_index_layout_info(const radhip_index_t *idx, radhip_layout_info_t *out);
/* test / tooling hooks: install a given injective layout, read the current one (u32[n]) */
int radhip_index_set_layout(radhip_index_t *idx, const uint32_t *lid);
int radhip_index_read_layout(const radhip_index_t *idx, uint32_t *out_lid);

/* replaces index.get_neighbors(node_id, level) — rad/hnsw_service.py:222,
 * rad/hnsw_server.py:483: writes the neighbor slots of (slot, level) in stored
 * order; RADHIP_E_RANGE if the node does not exist on that level. */
int radhip_get_neighbors(const radhip_index_t *idx, uint32_t slot, int32_t level,
                         uint32_t *out_slots, uint32_t cap, uint32_t *out_n);
/* replaces index.get_top_level_nodes() — rad/hnsw_service.py:229,
 * rad/hnsw_server.py:196: slots with level == max_level, ascending. */
int radhip_get_top_level_nodes(const radhip_index_t *idx, uint32_t *out_slots,
                               uint64_t cap, uint64_t *out_n);

/* ---- key <-> slot map (host memory only; no device call) ---------------- */
/* the key half of usearch Index.add(keys, fps) — README.md:58: keys of slots
 * [first_slot, first_slot+n).  Slots never given a key keep the identity key. */
int radhip_index_set_keys(radhip_index_t *idx, uint64_t first_slot, const uint64_t *keys, uint64_t n);
int radhip_index_read_keys(const radhip_index_t *idx, uint64_t first, uint64_t count, uint64_t *out_keys);
int radhip_keys_from_slots(const radhip_index_t *idx, const uint32_t *slots, uint64_t n, uint64_t *out_keys);
/* replaces index.get_node_ids_from_keys(keys) — examples/DUDEZ_example.ipynb:408 (and serves the
 * key join of rad/hnsw_service.py:271 in the other direction): out_slots[i] = slot of keys[i], or
 * RADHIP_NO_SLOT when the key is unknown (counted in *out_missing, may be NULL).  Binary search over
 * the slots ordered by key (built on the first lookup after a change); duplicate keys resolve to the
 * lowest slot. */
int radhip_slots_from_keys(const radhip_index_t *idx, const uint64_t *keys, uint64_t n, uint32_t *out_slots,
                           uint64_t *out_missing);
/* index.get_neighbors(node_id, level) in the reference's own list shape — rad/hnsw_service.py:222,
 * mock tests/test_redis_auth.py:37-39: out_pairs = [slot, key, slot, key, ...], *out_n = neighbours */
int radhip_get_neighbors_keyed(const radhip_index_t *idx, uint32_t slot, int32_t level, uint64_t *out_pairs,
                               uint32_t cap_pairs, uint32_t *out_n);

/* ---- A1: Tanimoto kernels (usearch metric='tanimoto', dtype='b1') ------ */
/* K1: nq queries x rows [first, first+count): and_out/or_out are [nq*count]
 * host arrays, query-major. */
int radhip_tanimoto_scan(radhip_index_t *idx, const uint8_t *queries, uint32_t nq,
                         uint64_t first, uint64_t count, uint32_t *and_out,
                         uint32_t *or_out);
/* K1 reduced on the chip: the k nearest rows of [first, first+count) for each query, in
 * (distance, slot) order — the order of a brute-force scan; replaces usearch's exact search
 * (Index.search(..., exact=True)), which RAD uses for recall figures only.  out_* are [nq*k]
 * (rows shorter than k are padded with RADHIP_NO_SLOT / 0), out_counts[nq]; 1 <= k <= 1984. */
int radhip_tanimoto_topk(radhip_index_t *idx, const uint8_t *queries, uint32_t nq, uint32_t k,
                         uint64_t first, uint64_t count, uint32_t *out_slots, uint32_t *out_and,
                         uint32_t *out_or, uint32_t *out_counts);
/* K2: candidate lists.  cand_offsets[nq+1] delimits each query's slots. */
int radhip_tanimoto_gather(radhip_index_t *idx, const uint8_t *queries, uint32_t nq,
                           const uint32_t *cand_slots, const uint64_t *cand_offsets,
                           uint32_t *and_out, uint32_t *or_out);
/* device time (HIP events, this thread's last radhip_tanimoto_scan / _gather call), kernels only */
double radhip_last_kernel_ms(void);
/* float edge value: 1.0f - (float)and/(float)or, 0.0f when or == 0 */
float radhip_distance_f32(uint32_t and_cnt, uint32_t or_cnt);

/* ---- A2: usearch-shaped HNSW insert + search on the device -------------- */
/* replaces usearch Index.add(keys, fps) — README.md:58,
 * examples/DUDEZ_example.ipynb:192: appends `count` rows (slots n .. n+count-1) to the corpus
 * and links them into the layered graph: integer geometric level draw from (seed, slot),
 * greedy descent, best-first layer search with expansion_add, heuristic neighbour selection,
 * reverse edges with re-selection.  Inserts run in deterministic batches of size
 * clamp(start/16, 1, max_batch) that search the pre-batch graph; max_batch = 1 is the
 * classical sequential insert.  (usearch's own source is not in the reference tree: the
 * algorithm is restated in oracle/rad_oracle.c orc_hnsw_add, parity unpinned vs usearch.) */
int radhip_index_add(radhip_index_t *idx, const uint8_t *rows, uint64_t count, uint64_t seed,
                     uint32_t max_batch);
/* replaces usearch Index.search(vectors, count) (never called by RAD itself; the same inner
 * loop as add): k nearest of each query by best-first search with expansion ef >= k.
 * out_* are [nq*k]; out_counts[nq] (a query whose search reaches fewer than k nodes gets a
 * shorter row, padded with RADHIP_NO_SLOT / 0); out_evals/out_pops (optional) count Tanimoto
 * evaluations / node expansions per query. */
int radhip_search(radhip_index_t *idx, const uint8_t *queries, uint32_t nq, uint32_t k, uint32_t ef,
                  uint32_t *out_slots, uint32_t *out_and, uint32_t *out_or, uint32_t *out_counts,
                  uint64_t *out_evals, uint64_t *out_pops);
/* the graph half of usearch Index.add for rows that are ALREADY resident (radhip_index_load_vectors /
 * radhip_index_synth_vectors): links rows [nodes of the graph so far, rows of the corpus) exactly as
 * radhip_index_add would have, without a host copy of the corpus going through the call */
int radhip_index_link_resident(radhip_index_t *idx, uint64_t seed, uint32_t max_batch);
/* level a node inserted at `slot` gets (host arithmetic, integer only) */
int radhip_level_of(uint64_t seed, uint64_t slot, uint32_t connectivity);

/* ---- A5-A10: RAD best-first traversal, Tanimoto-scored, on the device --- */
/* One independent traversal per query: prime (rad/traverser.py:128-176) from
 * the top-level nodes at level max(0, max_level-1), then pop-min / expand /
 * visited test-and-set / score-if-unscored / insert / descend
 * (rad/coordination_service.py:290-413, rad/distributed_worker.py:272-333)
 * with scoring_fn = Tanimoto distance to the query, until n_to_score nodes are
 * scored (checked before every pop) or the queue is empty. */
typedef struct radhip_traversal radhip_traversal_t;

typedef struct {
    uint64_t n_scored;   /* nodes in the scored set (== Tanimoto evaluations) */
    uint64_t n_pops;     /* node expansions                                   */
    uint64_t n_nbr;      /* adjacency entries examined                        */
    uint64_t n_repivot;  /* queue maintenance: pivot raises (far -> registers)    */
    uint64_t n_flush;    /* queue maintenance: staging flushes (sorted runs)     */
    int32_t status;      /* 0 running, 1 done(n_to_score), 2 done(queue empty),
                            3 parked at an intermediate target,
                            negative RADHIP_E_* on a device-side failure       */
    int32_t n_remid;     /* queue maintenance: mid-level refills (the only pass over every far run) */
    uint64_t n_upper;    /* (node, level >= 1) pairs in the upper-level visited set (what its size is chosen for) */
} radhip_trav_stats_t;

#define RADHIP_TRAV_LOG_POPS 1u  /* keep the (node, level) expansion log      */
/* Heavy state per RESIDENT ROW of the four-per-wavefront kernel instead of per traversal (round 4).  The visited / scored
 * tables, the queue's key pool, run table and mid run of a traversal (2.5 MB at n_to_score = 100k) are what a resident
 * row of the kernel works in; what a traversal owns by its number is its query, its header and its scored list (the
 * output: rad/scored.py:63-85).  A row that is done with a traversal takes the next one of the batch and reuses its tables
 * under a new epoch (the reference's per-run Redis keys, rad/traverser.py:93-102, restated as a tag in every entry).  A
 * batch of 65536 traversals then needs 40 GB + 52 GB instead of 217 GB.  Such a batch runs to completion: max_pops and
 * radhip_traversal_set_targets are refused.  Ignored where the four-per-wavefront kernel does not run or the batch fits
 * one resident round (radhip_traversal_slots tells). */
#define RADHIP_TRAV_SLOTS 4u
/* The object gets a HIP stream of its own, so that radhip_traversal_start on one object overlaps the tail of another's launch */
#define RADHIP_TRAV_OWN_STREAM 8u

int radhip_traversal_create(radhip_index_t *idx, const uint8_t *queries, uint32_t nq,
                            uint64_t n_to_score, uint32_t flags,
                            radhip_traversal_t **out);
int radhip_traversal_destroy(radhip_traversal_t *t);
/* re-arm the same state for a new set of nq queries (bench steps) */
int radhip_traversal_reset(radhip_traversal_t *t, const uint8_t *queries);
/* ... or its first `count` (<= nq) traversals only; the others rest */
int radhip_traversal_reset_count(radhip_traversal_t *t, const uint8_t *queries, uint32_t count);
/* CHAINED BATCHES (round 4).  A launch ends with its longest traversals running alone (~150 ms whatever its size), so several
 * batches go into ONE launch: one object of nq = batches x traversals, per-row state (RADHIP_TRAV_SLOTS is implied), and a RING
 * of `list_ring` scored lists (rounded up to a power of two, at least two resident rounds of the kernel): traversal i writes
 * list i mod ring once traversal i - ring is done with it.  Every traversal's counts stay in the statistics; the scored lists
 * that can still be read (radhip_traversal_results / _result_hashes) are those of the last `ring` traversals — what a
 * consumer that drains results as traversals complete would have freed (rad/scored.py:63-85; request_work has no batch
 * boundary: rad/coordination_service.py:290).  radhip_traversal_list_ring: the ring in use (0 = one list per traversal). */
int radhip_traversal_create_ring(radhip_index_t *idx, const uint8_t *queries, uint32_t nq, uint64_t n_to_score,
                                 uint32_t flags, uint32_t list_ring, radhip_traversal_t **out);
uint32_t radhip_traversal_list_ring(const radhip_traversal_t *t);
/* advance every unfinished traversal by at most max_pops expansions
 * (0 = run to completion); returns the number still running. */
int radhip_traversal_run(radhip_traversal_t *t, uint64_t max_pops, uint32_t *out_running);
/* The same as radhip_traversal_run(t, 0, ...) in two halves: start enqueues the launch of the re-armed batch and returns,
 * finish waits for it and reports (capacity fallbacks included).  With two objects on two streams (RADHIP_TRAV_OWN_STREAM)
 * the second batch's wavefronts start while the first batch's longest traversals are still running: the ~130 ms tail of a
 * launch is paid once per run, not once per batch (request_work has no batch boundary either: rad/coordination_service.py:290). */
int radhip_traversal_start(radhip_traversal_t *t);
int radhip_traversal_finish(radhip_traversal_t *t, uint32_t *out_running);
/* milliseconds on the device's clock from the start of `from`'s last launch to the end of `to`'s last launch */
int radhip_traversal_elapsed_between(const radhip_traversal_t *from, const radhip_traversal_t *to, double *out_ms);
/* start and end of the object's last finished launch, in ms on the device's clock since a fixed point of the process:
 * the device's busy time over overlapping launches of several objects is the union of these intervals */
int radhip_traversal_launch_interval(const radhip_traversal_t *t, double *out_start_ms, double *out_end_ms);
/* resident rows whose state the batch shares (RADHIP_TRAV_SLOTS); 0 = state per traversal */
uint32_t radhip_traversal_slots(const radhip_traversal_t *t);
int radhip_traversal_stats(const radhip_traversal_t *t, radhip_trav_stats_t *out /* [nq] */);
/* scored set of traversal q in insertion (traversal) order —
 * rad/scored.py:63-85 get_molecules; scores as integer (and, or) counts */
int radhip_traversal_results(const radhip_traversal_t *t, uint32_t q, uint32_t *out_slots,
                             uint32_t *out_and, uint32_t *out_or, uint64_t cap,
                             uint64_t *out_n);
/* order-sensitive 64-bit hash of the scored lists of traversals [first, first+count), formed on the device:
 * sum over positions i of mix64((i << 32 | slot_i) + mix64(and_i | or_i << 16)), mix64 = the splitmix64 finaliser —
 * a whole-list parity check that moves 8 bytes per traversal to the host (rad/scored.py:63-85 order) */
int radhip_traversal_result_hashes(const radhip_traversal_t *t, uint32_t first, uint32_t count, uint64_t *out);
int radhip_traversal_pop_log(const radhip_traversal_t *t, uint32_t q, uint32_t *out_nodes,
                             uint8_t *out_levels, uint64_t cap, uint64_t *out_n);
/* device time of the traversal kernel launches since create/reset, measured
 * with HIP events on the library's stream */
int radhip_traversal_kernel_time(const radhip_traversal_t *t, double *out_ms,
                                 uint64_t *out_launches);
uint64_t radhip_traversal_state_bytes(const radhip_traversal_t *t);
/* traversals resident on the device at once (one wavefront each): batch sizes that are a
 * multiple of it avoid a partially filled last round */
int radhip_traversal_resident_capacity(radhip_index_t *idx, uint32_t *out);
/* Which kernel a traversal object was bound to at create time: 4 = four traversals per wavefront
 * (batches larger than two resident rounds of the one-per-wavefront kernel, rows <= 16 wide), 1 = one per
 * wavefront with speculative fingerprint gathers (small batches, wide rows).  Same results. */
int radhip_traversal_kernel(const radhip_traversal_t *t);
/* Which visited/scored table the object uses: 2 = bucket table (16-B buckets of four 4-byte entries, ONE request
 * per probe; the default of the four-per-wavefront kernel), 0 = one 8-byte hash entry per probe (the one-per-
 * wavefront kernel and the sharded wave engine; RADHIP_TABLE=hash forces it for A/B runs), 1 = grouped (2 bits
 * per node in 16-B chunks of 48 layout ids, 8 chunks to a 128-B line; opt-in with RADHIP_TABLE=group in the
 * environment, needs radhip_index_optimize_layout — fewer HBM lines per expansion, measured slower end to end
 * on the round-2 workloads, see profiles/r02).  Same results. */
int radhip_traversal_table(const radhip_traversal_t *t);

/* per-traversal stop targets (each clamped to n_to_score): a traversal parks (status 3) once
 * n_scored >= its target and resumes when the target is raised — the hook the sharded
 * multi-GPU traversal uses to split a global n_to_score over shards round by round. */
int radhip_traversal_set_targets(radhip_traversal_t *t, const uint64_t *targets /* [nq] */);
/* frontier candidate scores: the best key left in each traversal's queue when the kernel
 * last returned (UINT64_MAX = queue empty; bits 61..38 hold the 24-bit distance q), and the
 * scored counts — what the ranks all-gather between rounds. */
int radhip_traversal_frontier(const radhip_traversal_t *t, uint64_t *out_keys, uint64_t *out_scored);

/* ---- row-sharded multi-GPU traversal (BASELINE.json north_star) --------------
 * One process per GPU.  The corpus is partitioned by contiguous slot range, the layered graph is ONE
 * graph over all rows (adjacency replicated), and the traversals of a batch are partitioned over the
 * ranks.  Per frontier step a rank advances each of its traversals to the point where a fingerprint
 * would be read (the control flow of rad/coordination_service.py:290-413 cut after the visited /
 * scored tests), the candidate slots of all ranks are all-gathered (RCCL over xGMI), every rank scores
 * the candidates whose rows it owns, and the scores return to the asking rank (reduce-scatter of
 * disjoint contributions).  Strict best-first per traversal: results are bit-identical to the
 * single-GPU traversal of the same corpus and graph for any number of ranks. */
/* keep rows [first, first+count) of the resident corpus, free the rest (graph, levels, keys stay whole).  For a
 * rank that did hold everything (the one that built the graph); the others are created with their rows only
 * (radhip_index_*_shard).  When the device has no room for the shard beside the corpus the shard leaves through
 * host memory and the corpus is freed first. */
int radhip_index_keep_rows(radhip_index_t *idx, uint64_t first, uint64_t count);
/* ---- peer-mapped corpus (round 4): BASELINE's row partitioning without a lock step ---------------------------------
 * Every rank of a node maps the row shards of ALL ranks into one contiguous virtual address range (HIP virtual memory
 * management; the peers' shards are their exported dmabuf descriptors, read over xGMI), seals it, and from then on holds
 * the whole corpus as far as every kernel is concerned: radhip_traversal_* run unchanged, a gather of a remote row is a
 * 128-B read over the fabric.  No collective, results bit-identical to one GPU by construction.  Replaces nothing in the
 * reference (its corpus lives in one usearch index in one process: README.md:45-58); it is how a corpus that does not fit
 * one GPU is served at the single-GPU kernel's rate instead of the lock-step loop's.
 *   peer_create -> peer_fill_synth | peer_fill_rows -> peer_export (one fd, handed to every peer) -> peer_import (each
 *   peer's fd) -> peer_seal.   Rows per shard = ceil(n_total / world) rounded up to the allocation granule (2 MiB). */
int radhip_index_peer_create(radhip_index_t *idx, int rank, int world, uint64_t n_total, uint64_t *out_rows_per_shard);
int radhip_index_peer_fill_synth(radhip_index_t *idx, uint64_t seed, int mode);
int radhip_index_peer_fill_rows(radhip_index_t *idx, const uint8_t *rows, uint64_t count);
int radhip_index_peer_export(radhip_index_t *idx, int *out_fd);
int radhip_index_peer_import(radhip_index_t *idx, int peer_rank, int fd);
int radhip_index_peer_seal(radhip_index_t *idx);
/* the graph of `src` copied device to device into `dst` (same device, same process, same row widths) */
int radhip_index_copy_graph_from(radhip_index_t *dst, radhip_index_t *src);
typedef struct radhip_shard radhip_shard_t;
typedef struct radhip_comm radhip_comm_t;
/* queries_all: world * nq query rows, rank-major — traversal (r, q) belongs to rank r; every rank passes
 * the same array (it scores other ranks' candidates against their queries).  The index must hold rows
 * [row_first, row_first+row_count): the whole corpus, or exactly that range after radhip_index_keep_rows. */
#define RADHIP_SHARD_OWN_STREAM 2u  /* radhip_shard_create flag: the shard gets a stream of its own (second group of a pair) */
int radhip_shard_create(radhip_index_t *idx, int rank, int world, uint64_t row_first, uint64_t row_count,
                        const uint8_t *queries_all, uint32_t nq, uint64_t n_to_score, uint32_t flags,
                        radhip_shard_t **out);
int radhip_shard_destroy(radhip_shard_t *s);
/* re-arm the same state for a new batch of world * nq queries */
int radhip_shard_reset(radhip_shard_t *s, const uint8_t *queries_all);
/* the product loop: step kernel, ncclAllGather of the candidates, evaluation kernel, ncclReduceScatter of
 * the scores — device buffers end to end, one stream — until no rank has a live traversal (the live counts
 * travel behind the candidates, so all ranks stop at the same step; the host looks at them every fourth
 * step, so up to three empty steps may follow the last useful one) or max_steps (0 = none) */
int radhip_shard_run(radhip_shard_t *s, radhip_comm_t *comm, uint64_t max_steps, uint64_t *out_steps);
/* the same loop for TWO groups of traversals at once, each with its own state, stream and communicator (b created
 * with RADHIP_SHARD_OWN_STREAM, comm_b a second radhip_comm_create): a frontier step is latency-bound end to end,
 * so one group's step kernel runs while the other group's collectives are in flight.  Results per group are those
 * of radhip_shard_run.  Both groups stop together (up to a few empty steps for the one that finishes first). */
int radhip_shard_run_pair(radhip_shard_t *a, radhip_comm_t *comm_a, radhip_shard_t *b, radhip_comm_t *comm_b,
                          uint64_t max_steps, uint64_t *out_steps);
/* Failure behaviour of both loops: a traversal that fails ON THE DEVICE (a fixed-capacity structure overflowed)
 * raises bit 31 of its rank's live word, which travels behind the candidates — every rank sees it at the same step,
 * leaves the loop and returns RADHIP_E_CAPACITY.  A HOST-side failure of one rank (a HIP / RCCL call) makes that
 * rank drain its stream, abort its communicator (ncclCommAbort: the peers' collectives fail or their next look
 * times out after RADHIP_SHARD_TIMEOUT_S, default 300) and return the error; the communicator is unusable afterwards. */
/* speculation of the thread engine (RADHIP_SHARD_SPEC = 0 | 1 | 2 queue heads expanded speculatively per step, default
 * 0: measured on one GPU it halves the frontier steps and makes each step as much longer, profiles/r03): scores asked for speculatively, how many of them finished an expansion without another step, how many
 * expansions that were.  Committed state never depends on it (strict pop order). */
int radhip_shard_speculation(const radhip_shard_t *s, uint32_t *out_depth, uint64_t *out_requested, uint64_t *out_used,
                             uint64_t *out_hits);
/* the same step in host-staged pieces, for an exchange the host program owns (tests; rehearsing N ranks
 * on one GPU): step -> get_requests | exchange | set_requests_all -> evaluate -> get_scores_out |
 * exchange (sum over ranks of block `rank`) | set_scores_in -> step ...   W = radhip_shard_width() slots
 * per traversal and step (RADHIP_NO_SLOT padded; the widest adjacency row, plus one more row width for all speculative
 * candidates when speculation is on — RADHIP_SHARD_SPEC_ROWS=2 gives every speculative head its own); scores
 * are and | or << 16. */
uint32_t radhip_shard_width(const radhip_shard_t *s);
/* Slots of this shard: the queue and the sets of a traversal (3.6 MB at n_to_score = 100k) exist once per slot, and a slot
 * whose traversal is done takes the next one of the batch.  nq by default (one slot per traversal); RADHIP_SHARD_SLOTS=N at
 * create gives the row engine N < nq slots (radhip_shard_run only: the host-staged pieces below need one slot per
 * traversal).  Results, statistics and pop logs are by traversal number either way. */
uint32_t radhip_shard_slots(const radhip_shard_t *s);
/* which step kernel drives the local traversals (RADHIP_SHARD_ENGINE = row | thread | wave, read at create):
 * 2 = "row" (the default: sixteen lanes per traversal, a 16-ary heap whose levels are 128-B lines, neighbours probed
 * one per lane; state in HBM), 0 = "thread" (one thread per traversal, 8-ary heap; also the fallback for queues of
 * more than 2^24 entries), 1 = "wave" (the single-GPU traversal kernel cut at the fingerprint read; adjacency rows
 * of <= 16 slots; slowest per step, kept as a cross-check).  Same results from all three. */
int radhip_shard_engine(const radhip_shard_t *s);
int radhip_shard_step(radhip_shard_t *s, uint32_t *out_live);
int radhip_shard_get_requests(radhip_shard_t *s, uint32_t *host /* [nq * W] */);
int radhip_shard_set_requests_all(radhip_shard_t *s, const uint32_t *host_all /* [world * nq * W] */);
int radhip_shard_evaluate(radhip_shard_t *s);
int radhip_shard_get_scores_out(radhip_shard_t *s, uint32_t *host /* [world * nq * W] */);
int radhip_shard_set_scores_in(radhip_shard_t *s, const uint32_t *host /* [nq * W] */);
int radhip_shard_stats(const radhip_shard_t *s, radhip_trav_stats_t *out /* [nq] */);
int radhip_shard_results(const radhip_shard_t *s, uint32_t q, uint32_t *out_slots, uint32_t *out_and,
                         uint32_t *out_or, uint64_t cap, uint64_t *out_n);
int radhip_shard_pop_log(const radhip_shard_t *s, uint32_t q, uint32_t *out_nodes, uint8_t *out_levels,
                         uint64_t cap, uint64_t *out_n);
int radhip_shard_timing(const radhip_shard_t *s, double *out_step_ms, double *out_eval_ms, uint64_t *out_steps,
                        uint64_t *out_exchanged_bytes);
uint64_t radhip_shard_state_bytes(const radhip_shard_t *s);

/* ---- multi-GPU exchange: RCCL over xGMI, one process per GPU -------------------- */
int radhip_comm_unique_id(uint8_t *out128);          /* rank 0 creates, host hands to others */
int radhip_comm_create(int rank, int world, const uint8_t *id128, int device, radhip_comm_t **out);
int radhip_comm_destroy(radhip_comm_t *c);
/* ncclAllGather of `count` u64 words per rank (host buffers are staged through HBM) */
int radhip_comm_allgather_u64(radhip_comm_t *c, const uint64_t *send, uint64_t count, uint64_t *recv);
int radhip_comm_rank(const radhip_comm_t *c);
int radhip_comm_world(const radhip_comm_t *c);
/* what the communicator really spans, from RCCL itself (ncclCommCount / ncclCommUserRank / ncclGetVersion) and
 * the PCI bus id of the device this rank drives (hipDeviceGetPCIBusId): evidence for the N > 1 bench line */
typedef struct {
    int32_t rccl_version;     /* ncclGetVersion code */
    int32_t comm_count;       /* ncclCommCount */
    int32_t comm_rank;        /* ncclCommUserRank */
    int32_t device;           /* HIP device ordinal */
    char pci_bus_id[32];      /* e.g. "0000:05:00.0" */
} radhip_comm_info_t;
int radhip_comm_info(const radhip_comm_t *c, radhip_comm_info_t *out);
/* the adjacency of `root` (levels, level-0 rows, upper rows, top-level nodes) on every rank: ncclBroadcast of the
 * device arrays over xGMI, nothing goes through host memory.  The receiving index must have the same
 * connectivity / connectivity_base; its own graph (if any) is replaced. */
int radhip_index_broadcast_graph(radhip_index_t *idx, radhip_comm_t *comm, int root);

/* host restatement of the device queue key, exported so CPU tests can check
 * its order against the Redis ZSET order of rad/priority_queue.py:22-42
 * (ascending score, ties by bytes of "{node_id}:{level}") */
uint64_t radhip_rad_key(uint32_t and_cnt, uint32_t or_cnt, uint32_t slot, uint32_t level);
void radhip_rad_key_decode(uint64_t key, uint32_t *slot, uint32_t *level);
/* test hook: the same key evaluated by the device code path (float-reciprocal q, comparison-tree
 * digit count) for n (and, or, slot, level) tuples */
int radhip_debug_device_keys(radhip_index_t *idx, const uint32_t *and_cnt, const uint32_t *or_cnt,
                             const uint32_t *slot, const uint32_t *level, uint64_t n, uint64_t *out_keys);
/* test hook: the traversal kernel's staging sort (one wavefront, keys in registers) on `batches` buffers of 256
 * u64 slots each, counts[b] <= radhip_debug_staging_capacity() keys valid; out: the same layout, the first
 * capacity slots of every buffer sorted ascending with all-ones behind the valid keys */
int radhip_debug_sort_staging(radhip_index_t *idx, const uint64_t *keys, const uint32_t *counts, uint32_t batches,
                              uint64_t *out);
uint32_t radhip_debug_staging_capacity(void);

#ifdef __cplusplus
}
#endif
#endif
