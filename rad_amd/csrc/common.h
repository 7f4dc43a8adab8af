// common.h — shared host/device helpers of librad_hip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/rad_hip.h"

// ---------------------------------------------------------------- errors --
void radhip_set_error(const char *fmt, ...);

#define RH_FAIL(code, ...)            \
    do {                              \
        radhip_set_error(__VA_ARGS__); \
        return (code);                \
    } while (0)

#define RH_HIP(expr)                                                              \
    do {                                                                          \
        hipError_t e_ = (expr);                                                   \
        if (e_ != hipSuccess) {                                                   \
            radhip_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                             __FILE__, __LINE__);                                 \
            (void)hipGetLastError();   /* (the runtime keeps a failed call as its "last error": a later launch check would report it again) */ \
            return (e_ == hipErrorOutOfMemory) ? RADHIP_E_NOMEM : RADHIP_E_HIP;   \
        }                                                                         \
    } while (0)

#define RH_TRY(expr)            \
    do {                        \
        int rc_ = (expr);       \
        if (rc_ != RADHIP_OK) return rc_; \
    } while (0)

// ------------------------------------------------------------- the index --
struct RhPeerMap;
struct radhip_index {
    uint32_t ndim_bits = 0, row_bytes = 0, row_stride = 0, lpr = 0;  // lpr = 16-B lanes per row
    uint32_t M = 0, cap0 = 0, ef_add = 0;
    int device = 0;
    uint64_t n = 0;           // rows resident in d_fp
    bool sharded = false;
    uint64_t shard_first = 0; // slot of row 0 of d_fp: 0 unless this index holds one rank's rows only (radhip_index_*_shard, keep_rows)
    uint64_t n_total = 0;     // sharded: rows of the whole corpus (= nodes of the graph every rank holds); else unused
    bool poisoned = false;    // a failed add() left reverse links to rows that were never committed: every further use is refused
    // ---- host mirror (adjacency reads need no device: fork-safe) ----------
    uint64_t g_n = 0;         // nodes in the graph
    int32_t max_level = -1;
    uint32_t entry = RADHIP_NO_SLOT;
    uint64_t n_upper_rows = 0;
    std::vector<int8_t> h_levels;
    std::vector<uint32_t> h_adj0, h_upper_row, h_adjU, h_top;
    bool h_graph_valid = false;   // host mirror holds the graph
    // ---- key <-> slot map (usearch keys: README.md:58 add(keys, fps); get_node_ids_from_keys,
    // examples/DUDEZ_example.ipynb:408): host memory only, sorted view built on first lookup
    std::vector<uint64_t> h_keys;        // key of every slot (identity where never set)
    std::vector<uint32_t> h_key_order;   // slots ordered by (key, slot)
    bool key_order_valid = false;
    std::vector<uint8_t> h_rows;  // staged corpus awaiting upload (freed after)
    bool h_rows_pending = false;
    bool has_graph = false, has_vectors = false;
    // ---- device ------------------------------------------------------------
    bool dev_ready = false;
    hipStream_t stream = nullptr;
    uint4 *d_fp = nullptr;        // [n * lpr]
    int8_t *d_levels = nullptr;   // [g_n]
    uint32_t *d_adj0 = nullptr;   // [g_n * cap0]
    uint32_t *d_upper_row = nullptr;
    uint32_t *d_adjU = nullptr;   // [n_upper_rows * M]
    uint32_t *d_top = nullptr;    // top-level slots
    uint32_t n_top = 0;
    bool d_graph_valid = false;
    uint64_t fp_cap_rows = 0;
    // peer-mapped corpus (index.hip, round 4): d_fp points into a reserved virtual range that holds the row shards of all
    // ranks of a node, this rank's own allocation and the peers' imported ones (xGMI).  Such a corpus is read-only.
    struct RhPeerMap *peer = nullptr;
    uint64_t cap_nodes = 0, cap_upper = 0;   // allocated rows of d_levels / d_adj0 / d_upper_row and of d_adjU (add() grows them by half)
    uint64_t device_bytes = 0;
    // bumped by every call that changes the graph or the corpus (load_graph, synth_graph, add,
    // load/synth_vectors): traversal objects remember the generation they were sized for
    uint64_t graph_gen = 0;
    // ---- graph-locality layout (layout.hip): lid[slot] = position of the slot in an order that keeps
    // graph neighbours together; the grouped visited table of the traversal kernels is keyed by it.
    // Results never depend on it (queue keys and outputs use slots), only the table's HBM lines do.
    std::vector<uint32_t> h_lid;
    bool layout_valid = false;
    uint32_t *d_lid = nullptr;      // [g_n]
    uint2 *d_adjx0 = nullptr;       // [g_n * cap0] {slot, lid} pairs (NO_SLOT padded)
    uint2 *d_adjxU = nullptr;       // [n_upper_rows * M]
    uint2 *d_topx = nullptr;        // [n_top]
    uint64_t lid_limit = 0;         // every lid < lid_limit
    double layout_lines_per_row = 0.0, layout_degree = 0.0, layout_seconds = 0.0;
    std::mutex mu;
};

void rh_layout_invalidate(radhip_index *idx);      // layout.hip: the graph changed
int rh_optimize_layout_locked(radhip_index *idx, uint32_t n_threads);   // idx->mu held by the caller

int rh_ensure_device(radhip_index *idx);           // lazy HIP init + pending uploads
// entry points that address d_fp by slot refuse an index that keeps only its shard of the rows
#define RH_REQUIRE_FULL_CORPUS(idx)                                                                        \
    do {                                                                                                   \
        if ((idx)->sharded) RH_FAIL(RADHIP_E_STATE, "this index holds rows [%llu, %llu) of %llu only (one rank's shard): " \
                                    "use the sharded traversal", (unsigned long long)(idx)->shard_first,   \
                                    (unsigned long long)((idx)->shard_first + (idx)->n), (unsigned long long)(idx)->n_total); \
    } while (0)
#define RH_REQUIRE_SOUND(idx)                                                                              \
    do {                                                                                                   \
        if ((idx)->poisoned) RH_FAIL(RADHIP_E_STATE, "a failed add() left this index inconsistent (reverse links to rows " \
                                     "that were never committed): destroy it and build a new one");        \
    } while (0)
int rh_ensure_host_graph(radhip_index *idx);       // D2H mirror of a device-generated graph
int rh_alloc_graph_dev(radhip_index *idx);         // index.hip: (re)allocate the device graph arrays for g_n / n_upper_rows
int rh_upload_top(radhip_index *idx);              // index.hip: h_top -> d_top

// ------------------------------------------------- device-side primitives --
#define RH_WAVE 64

// LDS traffic of one wave is executed in issue order: cross-lane hand-offs through LDS inside a
// wave need only this compiler-level barrier, not __syncthreads()
#define RH_WAVE_SYNC()                                           \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)

__device__ __forceinline__ uint32_t rh_popc4(const uint4 v) {
    return __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
}
__device__ __forceinline__ uint32_t rh_popc4_and(const uint4 a, const uint4 b) {
    return __popc(a.x & b.x) + __popc(a.y & b.y) + __popc(a.z & b.z) + __popc(a.w & b.w);
}
// ---- DPP cross-lane helpers (no LDS traffic, unlike __shfl/ds_bpermute) ---------------
#define RH_DPP_QUAD_XOR1 0xB1     /* quad_perm:[1,0,3,2] */
#define RH_DPP_QUAD_XOR2 0x4E     /* quad_perm:[2,3,0,1] */
#define RH_DPP_ROW_HALF_MIRROR 0x141
#define RH_DPP_ROW_MIRROR 0x140
#define RH_DPP_ROW_SHR(n) (0x110 + (n))
#define RH_DPP_ROW_BCAST15 0x142
#define RH_DPP_ROW_BCAST31 0x143

template <int CTRL>
__device__ __forceinline__ uint32_t rh_dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
// sum over groups of LPR adjacent lanes (LPR = 1,2,4,8,16); every lane of the group gets
// the total.  Butterfly: xor1, xor2 inside quads, then half-row and row mirrors.
template <int LPR>
__device__ __forceinline__ uint32_t rh_group_sum(uint32_t v) {
    if (LPR >= 2) v += rh_dpp<RH_DPP_QUAD_XOR1>(v);
    if (LPR >= 4) v += rh_dpp<RH_DPP_QUAD_XOR2>(v);
    if (LPR >= 8) v += rh_dpp<RH_DPP_ROW_HALF_MIRROR>(v);
    if (LPR >= 16) v += rh_dpp<RH_DPP_ROW_MIRROR>(v);
    return v;
}
// Sum of p[u] over the 8 lanes of a row FOR EIGHT VALUES AT ONCE, transposed: lane c of the row ends up with the
// row total of p[c].  A butterfly that halves the values a lane carries at every step (partner lane c ^ 7, c ^ 3,
// c ^ 1: half-row mirror and quad permutes) costs 4 + 2 + 1 exchanges instead of the 8 x 3 of eight full
// reductions — the scan kernels are VALU-bound at 8 queries per pass (profiles/r02/README.md §7).
__device__ __forceinline__ uint32_t rh_transpose_sum8(const uint32_t (&p)[8], bool b2, bool b1, bool b0) {
    uint32_t t[4], s[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const uint32_t keep = b2 ? p[j + 4] : p[j], send = b2 ? p[j] : p[j + 4]; t[j] = keep + rh_dpp<RH_DPP_ROW_HALF_MIRROR>(send); }
#pragma unroll
    for (int j = 0; j < 2; ++j) { const uint32_t keep = b1 ? t[j + 2] : t[j], send = b1 ? t[j] : t[j + 2]; s[j] = keep + rh_dpp<0x1B>(send); }   // quad_perm:[3,2,1,0]
    const uint32_t keep = b0 ? s[1] : s[0], send = b0 ? s[0] : s[1];
    return keep + rh_dpp<RH_DPP_QUAD_XOR1>(send);
}

// wave-wide unsigned min, result uniform (read from lane 63 after a row scan + row broadcasts)
__device__ __forceinline__ uint32_t rh_wave_min_u32(uint32_t v) {
#define RH_MIN_STEP(CTRL, ROWMASK)                                                              \
    {                                                                                           \
        const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, CTRL, ROWMASK, 0xf, false); \
        v = o < v ? o : v;                                                                      \
    }
    RH_MIN_STEP(RH_DPP_ROW_SHR(1), 0xf)
    RH_MIN_STEP(RH_DPP_ROW_SHR(2), 0xf)
    RH_MIN_STEP(RH_DPP_ROW_SHR(4), 0xf)
    RH_MIN_STEP(RH_DPP_ROW_SHR(8), 0xf)
    RH_MIN_STEP(RH_DPP_ROW_BCAST15, 0xa)
    RH_MIN_STEP(RH_DPP_ROW_BCAST31, 0xc)
#undef RH_MIN_STEP
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// wave-wide min of u64 keys (two u32 passes: high words, then low words among the ties)
__device__ __forceinline__ unsigned long long rh_wave_min_u64(unsigned long long k) {
    const uint32_t hi = (uint32_t)(k >> 32), lo = (uint32_t)k;
    const uint32_t mhi = rh_wave_min_u32(hi);
    const uint32_t mlo = rh_wave_min_u32(hi == mhi ? lo : 0xFFFFFFFFu);
    return ((unsigned long long)mhi << 32) | mlo;
}

// Claim visited-table bucket h for the expansion in flight.  `tab` (N words of LDS, 0 = free) is
// an exact-match open-addressed set of the buckets already claimed by lanes of THIS expansion;
// it must outlive every claim round of the expansion: a lane that lost bucket h walks on to
// h+1, which still reads empty in HBM even when another lane won it earlier (entries are
// stored only after the fingerprints are scored).  Returns true when this lane now owns h (ci =
// its word, to be zeroed once the entry is stored), false when another lane does.  At most
// N/2 claims are live at once, so the walk over `tab` always ends.
template <uint32_t N>
__device__ __forceinline__ bool claim_bucket(uint32_t *tab, uint32_t h, uint32_t &ci) {
    const uint32_t tag = h + 1u;
    uint32_t i = h & (N - 1u);
    for (;;) {
        const uint32_t old = atomicCAS(&tab[i], 0u, tag);
        if (old == 0u) { ci = i; return true; }
        if (old == tag) return false;
        i = (i + 1u) & (N - 1u);
    }
}

// ---------------------------------------------------- RAD queue key (u64) --
// Order = Redis ZSET order of rad/priority_queue.py:22-42 for a Tanimoto score:
//   ascending score, ties by bytes of the member string "{node_id}:{level}".
// bits 61..38 q  = floor((or-and) * 2^23 / or): strictly monotone in the exact
//                  rational distance 1 - and/or for or <= 2048, hence in its
//                  correctly rounded float32 value (tests/test_key_order.py)
// bits 37..8  p  = (slot+1) * 10^(9-d) - 1   (d = decimal digits of slot):
//                  the digits right-padded with '9's
// bits  7..4  9-d : on equal p the longer decimal string sorts first (':' > '9')
// bits  3..0  rank of the level string in bytewise order ("10" < "2")
#define RH_KEY_INF 0xFFFFFFFFFFFFFFFFull

__host__ __device__ __forceinline__ uint32_t rh_q24(uint32_t a, uint32_t o) {
    if (o == 0) return 0;
    // exact: the true quotient is either an integer or >= 2^-11 away from one,
    // the double division error is < 2^-29
    return (uint32_t)((double)((uint64_t)(o - a) << 23) / (double)o);
}
__host__ __device__ __forceinline__ uint32_t rh_level_rank(uint32_t level) {
    return level < 2 ? level : (level >= 10 ? level - 8 : level + 6);
}
__host__ __device__ __forceinline__ uint32_t rh_rank_level(uint32_t r) {
    return r < 2 ? r : (r <= 7 ? r + 8 : r - 6);
}
__host__ __device__ __forceinline__ uint32_t rh_pow10(uint32_t e) {
    uint32_t p = 1;
    p = e >= 8 ? 100000000u : e == 7 ? 10000000u : e == 6 ? 1000000u : e == 5 ? 100000u
      : e == 4 ? 10000u : e == 3 ? 1000u : e == 2 ? 100u : e == 1 ? 10u : p;
    return p;
}
__host__ __device__ __forceinline__ uint64_t rh_make_key(uint32_t q24, uint32_t slot, uint32_t level) {
    uint32_t d = 1 + (slot >= 10u) + (slot >= 100u) + (slot >= 1000u) + (slot >= 10000u) +
                 (slot >= 100000u) + (slot >= 1000000u) + (slot >= 10000000u) + (slot >= 100000000u);
    uint32_t dl = 9 - d;
    uint32_t p = (slot + 1u) * rh_pow10(dl) - 1u;
    return ((uint64_t)q24 << 38) | ((uint64_t)p << 8) | ((uint64_t)dl << 4) | (uint64_t)rh_level_rank(level);
}
// ---- device-tuned restatements (same values; tests/test_gpu_kernels.py::test_device_keys) ----
// floor((o-a) * 2^23 / o) without double precision: float reciprocal estimate (within +-2),
// then an exact remainder fix-up in wrap-around 32-bit arithmetic (the true remainder is small).
__device__ __forceinline__ uint32_t rh_q24_dev(uint32_t a, uint32_t o) {
    const uint32_t x = o - a;
    uint32_t q = (uint32_t)((float)x * __frcp_rn((float)o) * 8388608.0f);
    int32_t r = (int32_t)((x << 23) - q * o);
    if (r < 0) { q -= 1u; r += (int32_t)o; }
    if (r < 0) { q -= 1u; r += (int32_t)o; }
    if (r >= (int32_t)o) { q += 1u; r -= (int32_t)o; }
    if (r >= (int32_t)o) { q += 1u; r -= (int32_t)o; }
    return o == 0u ? 0u : q;
}
// digit count and 10^(9-d) by a comparison tree instead of nine compares + a select chain
__device__ __forceinline__ uint64_t rh_make_key_dev(uint32_t q24, uint32_t slot, uint32_t level) {
    uint32_t dl, m;
    if (slot < 10000u) {
        if (slot < 100u) { const bool c = slot < 10u; dl = c ? 8u : 7u; m = c ? 100000000u : 10000000u; }
        else { const bool c = slot < 1000u; dl = c ? 6u : 5u; m = c ? 1000000u : 100000u; }
    } else if (slot < 100000000u) {
        if (slot < 1000000u) { const bool c = slot < 100000u; dl = c ? 4u : 3u; m = c ? 10000u : 1000u; }
        else { const bool c = slot < 10000000u; dl = c ? 2u : 1u; m = c ? 100u : 10u; }
    } else { dl = 0u; m = 1u; }
    const uint32_t p = (slot + 1u) * m - 1u;
    return ((uint64_t)q24 << 38) | ((uint64_t)p << 8) | ((uint64_t)dl << 4) | (uint64_t)rh_level_rank(level);
}
__host__ __device__ __forceinline__ void rh_decode_key(uint64_t key, uint32_t *slot, uint32_t *level) {
    uint32_t p = (uint32_t)(key >> 8) & 0x3FFFFFFFu;
    uint32_t dl = (uint32_t)(key >> 4) & 0xFu;
    *slot = (p + 1u) / rh_pow10(dl) - 1u;
    *level = rh_rank_level((uint32_t)key & 0xFu);
}
