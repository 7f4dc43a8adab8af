// build.hip — A2: usearch-shaped best-first layer search (search + HNSW insert) on
// gfx950, one wavefront per query / per inserted node.
//
// Algorithm (restated in oracle/rad_oracle.c orc_hnsw_add / orc_graph_search; usearch itself
// is not in the reference tree — [RECALLED] shape, SURVEY.md §3.5, parity unpinned vs usearch):
//   greedy 1-best descent through the levels above the target level, then per level a
//   best-first search keeping the `ef` closest (sorted buffer with "expanded" marks — the
//   same result as usearch's candidate heap + bounded top buffer), neighbour selection by
//   the hnswlib/usearch heuristic, reverse edges with re-selection when a row is full.
//   Order of candidates: exact rational Tanimoto distance (24-bit q, common.h), ties by slot.
//   Inserts run in deterministic batches: every node of a batch searches the pre-batch
//   graph; reverse-edge requests are applied in (target, level, source) order.
#include "common.h"

#include <rocprim/rocprim.hpp>   // device radix sort / select for the reverse-edge request list (a utility, not a kernel of the path)

#include <algorithm>
#include <new>

#define BK_INF 0xFFFFFFFFFFFFFFFFull
#define BK_REG_EF_CAP 512u   // up to this many entries the sorted top buffer of a search lives in registers (8 per lane)
#define VIS_EMPTY 0xFFFFFFFFu

__host__ __device__ __forceinline__ unsigned long long bk_key(uint32_t q24, uint32_t slot) {
    return ((unsigned long long)q24 << 33) | ((unsigned long long)slot << 1);
}
__device__ __forceinline__ uint32_t bk_slot(unsigned long long k) { return (uint32_t)(k >> 1); }
__device__ __forceinline__ uint32_t bk_q(unsigned long long k) { return (uint32_t)(k >> 33); }

#define WSYNC()                                                  \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)

// wave-wide minima by DPP (common.h): the shuffle forms went through LDS (ds_bpermute) six times per call
__device__ __forceinline__ unsigned long long bk_wave_min_u64(unsigned long long v) { return rh_wave_min_u64(v); }
__device__ __forceinline__ uint32_t bk_wave_min_u32(uint32_t v) { return rh_wave_min_u32(v); }

struct GraphView {
    const uint4 *fp;
    const int8_t *levels;
    uint32_t *adj0;
    const uint32_t *upper_row;
    uint32_t *adjU;
    uint32_t cap0, capU;
};

__device__ __forceinline__ uint32_t *gv_row(const GraphView &G, uint32_t slot, uint32_t level, uint32_t *cap) {
    if (level == 0) { *cap = G.cap0; return G.adj0 + (uint64_t)slot * G.cap0; }
    *cap = G.capU;
    return G.adjU + ((uint64_t)G.upper_row[slot] + (level - 1u)) * G.capU;
}

// per-wave scratch in LDS (dynamic): two top buffers of ef_cap keys + small arrays
struct WaveLds {
    unsigned long long *topA, *topB;  // [ef_cap]
    unsigned long long *newk;         // [128] (also the re-selection sort buffer)
    uint32_t *u32a;                   // [64]
    uint32_t *u32b;                   // [64]
    uint32_t *u32c;                   // [64]
    uint32_t *claim;                  // [128] buckets of the visited table claimed by the expansion in flight
};

__device__ __forceinline__ WaveLds carve_lds(unsigned char *base, uint32_t ef_cap) {
    WaveLds L;
    // one buffer of ef_cap keys when the search keeps its top buffer in registers (ef_cap <= 512: search_layer_reg moves the
    // entries through it between merges), two (ping-pong) for search_layer
    L.topA = reinterpret_cast<unsigned long long *>(base);
    L.topB = ef_cap <= BK_REG_EF_CAP ? L.topA : L.topA + ef_cap;
    L.newk = L.topB + ef_cap;
    L.u32a = reinterpret_cast<uint32_t *>(L.newk + 128);
    L.u32b = L.u32a + 64;
    L.u32c = L.u32b + 64;
    L.claim = L.u32c + 64;
    return L;
}
static size_t wave_lds_bytes(uint32_t ef_cap) { return (size_t)ef_cap * (ef_cap <= BK_REG_EF_CAP ? 8 : 16) + 128 * 8 + 3 * 64 * 4 + 128 * 4; }

// Tanimoto of the wave's query chunk `qv` against up to 64 rows whose slots are in
// L.u32a[0..n); results (and, or) to L.u32b / L.u32c.
template <int LPR>
__device__ __forceinline__ void eval_rows(const uint4 *fp, const uint4 qv, uint32_t qpop, const WaveLds &L,
                                          uint32_t n, uint32_t lane) {
    constexpr uint32_t RPP = 64 / LPR;
    const uint32_t chunk = lane % LPR;
    for (uint32_t base = 0; base < n; base += RPP) {
        const uint32_t ri = base + lane / LPR;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ri < n) v = fp[(uint64_t)L.u32a[ri] * LPR + chunk];
        const uint32_t rp = rh_group_sum<LPR>(rh_popc4(v));
        const uint32_t aa = rh_group_sum<LPR>(rh_popc4_and(v, qv));
        if (ri < n && chunk == 0) { L.u32b[ri] = aa; L.u32c[ri] = qpop + rp - aa; }
    }
    WSYNC();
}

// visited hash set in HBM (u32 slots, VIS_EMPTY = free); returns true if newly inserted
__device__ __forceinline__ bool vis_test_and_set(uint32_t *vis, uint32_t vmask, uint32_t vshift, uint32_t slot) {
    uint32_t h = (slot * 2654435769u) >> vshift;
    for (;;) {
        const uint32_t old = atomicCAS(&vis[h], VIS_EMPTY, slot);
        if (old == VIS_EMPTY) return true;
        if (old == slot) return false;
        h = (h + 1u) & vmask;
    }
}

struct SearchCounters { uint64_t evals, pops; int32_t status; };
// -DBK_PROFILE: where a search spends its time (s_memtime ticks per section of search_layer_reg, summed over all wavefronts and
// printed by radhip_index_link_resident): the build's counterpart of traverse4.inc's RH_PROFILE
#ifdef BK_PROFILE
__device__ unsigned long long bk_prof[12];   // [0..5] sections of search_layer_reg, [6] pops with new neighbours, [7] kept keys, [8] whole insert, [9] select_heuristic, [10] candidates it examined, [11] calls
#define BK_T(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pacc[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define BK_T(i) do { } while (0)
#endif

// best-first search on one level.  In: sorted top (topA) with n_top entries whose slots are
// already in `vis`.  Out: sorted top in L.topA (buffers swapped back), returns n_top.
template <int LPR>
__device__ uint32_t search_layer(const GraphView &G, const uint4 qv, uint32_t qpop, uint32_t level, uint32_t ef,
                                 uint32_t n_top, WaveLds &L, uint32_t *vis, uint32_t vlog2, uint32_t &vis_count,
                                 SearchCounters &C, uint32_t lane) {
    const uint32_t vmask = (1u << vlog2) - 1u, vshift = 32u - vlog2;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (;;) {
        // first unexpanded entry
        uint32_t first = 0xFFFFFFFFu;
        for (uint32_t i = lane; i < n_top; i += 64)
            if (!(L.topA[i] & 1ull)) { first = i; break; }
        const uint32_t pos = bk_wave_min_u32(first);
        if (pos == 0xFFFFFFFFu) break;
        const unsigned long long ck = L.topA[pos];
        if (lane == 0) L.topA[pos] = ck | 1ull;
        WSYNC();
        C.pops++;
        const uint32_t cur = bk_slot(ck);
        uint32_t cap;
        const uint32_t *row = gv_row(G, cur, level, &cap);
        uint32_t nb = RADHIP_NO_SLOT;
        if (lane < cap) nb = row[lane];
        // Visited test-and-set of the row's slots with plain (L2-served) loads: the table belongs to
        // this wavefront alone, so the only race is between lanes of this expansion wanting the same
        // empty bucket — settled by the LDS claim set, as in the traversal kernels (atomics to the
        // table execute memory-side and were the slowest part of an expansion).
        bool isnew = false;
        {
            bool pending = nb != RADHIP_NO_SLOT, cand = false;
            uint32_t h = (nb * 2654435769u) >> vshift, ci = 0;
            for (;;) {
                if (pending) {
                    for (;;) {
                        const uint32_t e = __hip_atomic_load(&vis[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (e == VIS_EMPTY) { cand = true; break; }
                        if (e == nb) break;
                        h = (h + 1u) & vmask;
                    }
                    pending = false;
                }
                if (cand) {
                    cand = false;
                    if (claim_bucket<128u>(L.claim, h, ci)) {
                        isnew = true;
                        __hip_atomic_store(&vis[h], nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else { pending = true; h = (h + 1u) & vmask; }
                }
                if (!__ballot(pending)) break;
            }
            if (isnew) L.claim[ci] = 0u;
        }
        const unsigned long long nbal = __ballot(isnew);
        const uint32_t nn = (uint32_t)__popcll(nbal);
        if (nn == 0) continue;
        vis_count += nn;
        if (vis_count > (1u << vlog2) / 2u) { C.status = RADHIP_E_CAPACITY; break; }
        const uint32_t rank = (uint32_t)__popcll(nbal & lt_mask);
        if (isnew) L.u32a[rank] = nb;
        WSYNC();
        eval_rows<LPR>(G.fp, qv, qpop, L, nn, lane);
        C.evals += nn;
        // keys of the new candidates, one per lane (lane < nn)
        unsigned long long key = BK_INF;
        if (lane < nn) key = bk_key(rh_q24_dev(L.u32b[lane], L.u32c[lane]), L.u32a[lane]);
        bool keep = lane < nn;
        if (n_top == ef) keep = keep && ((key >> 1) < (L.topA[ef - 1u] >> 1));
        const unsigned long long kb = __ballot(keep);
        const uint32_t m = (uint32_t)__popcll(kb);
        if (m == 0) continue;
        const uint32_t kr = (uint32_t)__popcll(kb & lt_mask);
        WSYNC();
        if (keep) L.newk[64 + kr] = key;  // unsorted staging
        WSYNC();
        // rank among the kept keys -> sorted newk[0..m)
        uint32_t r_new = 0;
        if (keep)
            for (uint32_t j = 0; j < m; ++j) r_new += (L.newk[64 + j] < key) ? 1u : 0u;
        WSYNC();
        if (keep) L.newk[r_new] = key;
        WSYNC();
        // merged position of every kept key: r_new + lower_bound(top, key)
        if (keep) {
            uint32_t lo = 0, hi = n_top;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if ((L.topA[mid] >> 1) < (key >> 1)) lo = mid + 1; else hi = mid;
            }
            const uint32_t p = lo + r_new;
            if (p < ef) L.topB[p] = key;
        }
        // merged position of every old entry: i + lower_bound(newk, top[i])
        for (uint32_t i = lane; i < n_top; i += 64) {
            const unsigned long long t = L.topA[i];
            uint32_t lo = 0, hi = m;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if ((L.newk[mid] >> 1) < (t >> 1)) lo = mid + 1; else hi = mid;
            }
            const uint32_t p = i + lo;
            if (p < ef) L.topB[p] = t;
        }
        n_top = n_top + m < ef ? n_top + m : ef;
        WSYNC();
        unsigned long long *tmp = L.topA; L.topA = L.topB; L.topB = tmp;
    }
    return n_top;
}

// ---- the same search with the sorted top buffer IN REGISTERS (round 4) ------------------------------------------------------
// Counters of the 100M-row build at expansion_add 400 (profiles/r04/README.md section 11): a pop takes 6.5 us of which 42 % are
// the scan for the first unexpanded entry and the merge of the new keys — chains of dependent LDS reads (a binary search per
// old entry, per new key).  Here entry i of the buffer lives in lane i % 64, register i / 64 (EPL registers: ef <= 64 * EPL):
// the scan is a compare per register and one DPP minimum; the merge counts, per new key (a scalar broadcast from the lane
// that holds it), the old entries below it with a ballot per register and bumps the shift of those above it — no search at
// all; the entries then move to their new lanes through ONE LDS buffer (independent writes and reads).  Same order of pops,
// same buffer contents at every step as search_layer (tests: adjacency bit-exact against the oracle).
__device__ __forceinline__ unsigned long long bk_readlane64(unsigned long long v, uint32_t ln) {   // ln wave-uniform
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)ln) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)ln);
}
template <int EPL>
__device__ __forceinline__ unsigned long long bk_top_get(const unsigned long long (&t)[EPL], uint32_t pos) {   // pos wave-uniform
    unsigned long long v = BK_INF;
    const uint32_t kk = pos >> 6, ln = pos & 63u;
#pragma unroll
    for (int k = 0; k < EPL; ++k)
        if ((uint32_t)k == kk) v = bk_readlane64(t[k], ln);
    return v;
}

template <int LPR, int EPL>
__device__ uint32_t search_layer_reg(const GraphView &G, const uint4 qv, uint32_t qpop, uint32_t level, uint32_t ef,
                                     uint32_t n_top, WaveLds &L, uint32_t *vis, uint32_t vlog2, uint32_t &vis_count,
                                     SearchCounters &C, uint32_t lane) {
    const uint32_t vmask = (1u << vlog2) - 1u, vshift = 32u - vlog2;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    unsigned long long t[EPL];
#pragma unroll
    for (int k = 0; k < EPL; ++k) { const uint32_t i = 64u * k + lane; t[k] = i < n_top ? L.topA[i] : BK_INF; }
#ifdef BK_PROFILE
    unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
    struct Flush { unsigned long long *p; uint32_t lane; __device__ ~Flush() { if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&bk_prof[i], p[i]); } } flush_{pacc, lane};
#endif
    for (;;) {
        BK_T(0);
        uint32_t first = 0xFFFFFFFFu;
#pragma unroll
        for (int k = EPL - 1; k >= 0; --k)
            if (!(t[k] & 1ull)) first = 64u * k + lane;      // (BK_INF reads as expanded)
        const uint32_t pos = rh_wave_min_u32(first);
        if (pos == 0xFFFFFFFFu) break;
        const unsigned long long ck = bk_top_get<EPL>(t, pos);
#pragma unroll
        for (int k = 0; k < EPL; ++k)
            if ((uint32_t)k == (pos >> 6) && lane == (pos & 63u)) t[k] |= 1ull;
        BK_T(1);
        C.pops++;
        const uint32_t cur = bk_slot(ck);
        uint32_t cap;
        const uint32_t *row = gv_row(G, cur, level, &cap);
        uint32_t nb = RADHIP_NO_SLOT;
        if (lane < cap) nb = row[lane];
        bool isnew = false;
        {   // visited test-and-set (see search_layer)
            bool pending = nb != RADHIP_NO_SLOT, cand = false;
            uint32_t h = (nb * 2654435769u) >> vshift, ci = 0;
            for (;;) {
                if (pending) {
                    for (;;) {
                        const uint32_t e = __hip_atomic_load(&vis[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (e == VIS_EMPTY) { cand = true; break; }
                        if (e == nb) break;
                        h = (h + 1u) & vmask;
                    }
                    pending = false;
                }
                if (cand) {
                    cand = false;
                    if (claim_bucket<128u>(L.claim, h, ci)) {
                        isnew = true;
                        __hip_atomic_store(&vis[h], nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else { pending = true; h = (h + 1u) & vmask; }
                }
                if (!__ballot(pending)) break;
            }
            if (isnew) L.claim[ci] = 0u;
        }
        const unsigned long long nbal = __ballot(isnew);
        const uint32_t nn = (uint32_t)__popcll(nbal);
        BK_T(2);
        if (nn == 0) continue;
        vis_count += nn;
        if (vis_count > (1u << vlog2) / 2u) { C.status = RADHIP_E_CAPACITY; break; }
        const uint32_t rank = (uint32_t)__popcll(nbal & lt_mask);
        if (isnew) L.u32a[rank] = nb;
        WSYNC();
        eval_rows<LPR>(G.fp, qv, qpop, L, nn, lane);
        C.evals += nn;
        BK_T(3);
        unsigned long long key = BK_INF;
        if (lane < nn) key = bk_key(rh_q24_dev(L.u32b[lane], L.u32c[lane]), L.u32a[lane]);
        bool keep = lane < nn;
        if (n_top == ef) keep = keep && ((key >> 1) < (bk_top_get<EPL>(t, ef - 1u) >> 1));
        const unsigned long long kb = __ballot(keep);
        const uint32_t m = (uint32_t)__popcll(kb);
#ifdef BK_PROFILE
        pacc[6] += 1; pacc[7] += m;
#endif
        BK_T(4);
        if (m == 0) continue;
        // rank of every kept key among the kept keys, without LDS: the kept lanes' keys pass by as scalars
        uint32_t r_new = 0;
        for (unsigned long long rest = kb; rest; rest &= rest - 1ull) {
            const unsigned long long ko = bk_readlane64(key, (uint32_t)__builtin_ctzll(rest));
            r_new += (keep && ko < key) ? 1u : 0u;
        }
        uint32_t shift[EPL];
#pragma unroll
        for (int k = 0; k < EPL; ++k) shift[k] = 0;
        for (uint32_t j = 0; j < m; ++j) {
            const unsigned long long kj = bk_readlane64(key, (uint32_t)__builtin_ctzll(__ballot(keep && r_new == j)));   // the j-th new key
            uint32_t lo = 0;
#pragma unroll
            for (int k = 0; k < EPL; ++k) {
                lo += (uint32_t)__popcll(__ballot((t[k] >> 1) < (kj >> 1)));      // old entries below the new key (BK_INF never is)
                shift[k] += ((kj >> 1) < (t[k] >> 1)) ? 1u : 0u;                  // ... and the old entries it pushes up
            }
            const uint32_t p = j + lo;
            if (lane == 0 && p < ef) L.topB[p] = kj;
        }
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
            const uint32_t i = 64u * k + lane, p = i + shift[k];
            if (i < n_top && p < ef) L.topB[p] = t[k];
        }
        n_top = n_top + m < ef ? n_top + m : ef;
        WSYNC();
#pragma unroll
        for (int k = 0; k < EPL; ++k) { const uint32_t i = 64u * k + lane; t[k] = i < n_top ? L.topB[i] : BK_INF; }
        BK_T(5);
    }
    WSYNC();
#pragma unroll
    for (int k = 0; k < EPL; ++k) { const uint32_t i = 64u * k + lane; if (i < n_top) L.topA[i] = t[k]; }
    WSYNC();
    return n_top;
}

// the buffer in registers when it fits eight per lane, in LDS beyond (expansion_add > 512)
template <int LPR>
__device__ __forceinline__ uint32_t search_layer_any(const GraphView &G, const uint4 qv, uint32_t qpop, uint32_t level, uint32_t ef,
                                                     uint32_t ef_cap, uint32_t n_top, WaveLds &L, uint32_t *vis, uint32_t vlog2,
                                                     uint32_t &vis_count, SearchCounters &C, uint32_t lane) {
    if (ef_cap <= 128u) return search_layer_reg<LPR, 2>(G, qv, qpop, level, ef, n_top, L, vis, vlog2, vis_count, C, lane);
    if (ef_cap <= BK_REG_EF_CAP) return search_layer_reg<LPR, 8>(G, qv, qpop, level, ef, n_top, L, vis, vlog2, vis_count, C, lane);
    return search_layer<LPR>(G, qv, qpop, level, ef, n_top, L, vis, vlog2, vis_count, C, lane);
}

// greedy 1-best move on one level until no neighbour is closer
template <int LPR>
__device__ unsigned long long greedy_level(const GraphView &G, const uint4 qv, uint32_t qpop, uint32_t level,
                                           unsigned long long curk, WaveLds &L, SearchCounters &C, uint32_t lane) {
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (;;) {
        C.pops++;
        uint32_t cap;
        const uint32_t *row = gv_row(G, bk_slot(curk), level, &cap);
        uint32_t nb = RADHIP_NO_SLOT;
        if (lane < cap) nb = row[lane];
        const bool valid = nb != RADHIP_NO_SLOT;
        const unsigned long long vb = __ballot(valid);
        const uint32_t nn = (uint32_t)__popcll(vb);
        if (nn == 0) break;
        const uint32_t rank = (uint32_t)__popcll(vb & lt_mask);
        if (valid) L.u32a[rank] = nb;
        WSYNC();
        eval_rows<LPR>(G.fp, qv, qpop, L, nn, lane);
        C.evals += nn;
        unsigned long long key = BK_INF;
        if (lane < nn) key = bk_key(rh_q24_dev(L.u32b[lane], L.u32c[lane]), L.u32a[lane]);
        const unsigned long long best = bk_wave_min_u64(key);
        WSYNC();
        if (best < curk) curk = best; else break;
    }
    return curk;
}

// neighbour-selection heuristic over sorted candidates cand[0..n) (keys hold the distance to
// the base node); writes the kept slots to sel[0..k), returns k <= cap
template <int LPR>
__device__ uint32_t select_heuristic(const uint4 *fp, const unsigned long long *cand, uint32_t n, uint32_t cap,
                                     uint32_t *sel, uint32_t lane) {
    constexpr uint32_t RPP = 64 / LPR;
    const uint32_t chunk = lane % LPR;
    uint32_t k = 0;
    for (uint32_t ci = 0; ci < n && k < cap; ++ci) {
        const unsigned long long ck = cand[ci];
        const uint32_t c = bk_slot(ck), qc = bk_q(ck);
        const uint4 cv = fp[(uint64_t)c * LPR + chunk];
        const uint32_t cpop = rh_group_sum<LPR>(rh_popc4(cv));
        bool bad = false;
        for (uint32_t base = 0; base < k; base += RPP) {
            const uint32_t ai = base + lane / LPR;
            uint4 av = make_uint4(0, 0, 0, 0);
            if (ai < k) av = fp[(uint64_t)sel[ai] * LPR + chunk];
            const uint32_t apop = rh_group_sum<LPR>(rh_popc4(av));
            const uint32_t aa = rh_group_sum<LPR>(rh_popc4_and(av, cv));
            if (ai < k && rh_q24_dev(aa, apop + cpop - aa) < qc) bad = true;
        }
        if (!__ballot(bad)) {
            if (lane == 0) sel[k] = c;
            k++;
            WSYNC();
        }
    }
    return k;
}

// The same selection for rows of at most 2 * (64 / LPR) slots (connectivity 8 at 1024 bits: 16 / 8), without a dependent HBM
// round trip per candidate (round 4: select_heuristic was 15 % of an insert at expansion_add 400 — 400 candidates, each one
// load of its fingerprint and a reload of every selected row).  Candidates come in chunks of 64 / LPR rows: one wave-load per
// chunk (the next chunk's is in flight meanwhile), staged in 1 KB of LDS (`tile`) from which every lane group reads the
// candidate's row; the selected rows stay in REGISTERS — selected row s in lane group s % RPP, register s / RPP — so a
// candidate meets all of them in two passes of popcounts and no load.  Same candidates examined in the same order, same
// comparisons: same selection.
// (not inlined: inside the kernel its registers cost the search loop spills — 8.92 s inlined, 8.33 s as a call, 10M rows)
template <int LPR>
__device__ __attribute__((noinline)) uint32_t select_heuristic_regs(const uint4 *fp, const unsigned long long *cand, uint32_t n, uint32_t cap,
                                          uint32_t *sel, uint4 *tile, uint32_t lane) {
    constexpr uint32_t RPP = 64 / LPR;
    const uint32_t chunk = lane % LPR, grp = lane / LPR;
    uint4 sv0 = make_uint4(0, 0, 0, 0), sv1 = make_uint4(0, 0, 0, 0);
    uint32_t sp0 = 0, sp1 = 0, k = 0;
    auto load = [&](uint32_t base) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (base + grp < n) v = fp[(uint64_t)bk_slot(cand[base + grp]) * LPR + chunk];
        return v;
    };
    uint4 cur = load(0);
    for (uint32_t base = 0; base < n && k < cap; base += RPP) {
        const uint4 nxt = base + RPP < n ? load(base + RPP) : make_uint4(0, 0, 0, 0);
        tile[lane] = cur;
        WSYNC();
        for (uint32_t j = 0; j < RPP && base + j < n && k < cap; ++j) {
            const unsigned long long ck = cand[base + j];
            const uint32_t c = bk_slot(ck), qc = bk_q(ck);
            const uint4 cv = tile[j * LPR + chunk];
            const uint32_t cpop = rh_group_sum<LPR>(rh_popc4(cv));
            bool bad = false;
            {
                const uint32_t aa = rh_group_sum<LPR>(rh_popc4_and(sv0, cv));
                if (grp < k && rh_q24_dev(aa, sp0 + cpop - aa) < qc) bad = true;
            }
            if (k > RPP) {
                const uint32_t aa = rh_group_sum<LPR>(rh_popc4_and(sv1, cv));
                if (RPP + grp < k && rh_q24_dev(aa, sp1 + cpop - aa) < qc) bad = true;
            }
            if (!__ballot(bad)) {
                if (lane == 0) sel[k] = c;
                if (k < RPP) { if (grp == k) { sv0 = cv; sp0 = cpop; } }
                else if (grp == k - RPP) { sv1 = cv; sp1 = cpop; }
                k++;
            }
        }
        WSYNC();
        cur = nxt;
    }
    WSYNC();
    return k;
}

// ------------------------------------------------------------- search kernel
struct SearchParams {
    GraphView G;
    const uint4 *queries;
    const uint32_t *qpop;
    uint32_t entry;
    int32_t max_level;
    uint32_t k, ef, ef_cap;
    uint32_t *vis;
    uint32_t vlog2;
    uint32_t *out_slots, *out_and, *out_or, *out_counts;
    unsigned long long *out_evals, *out_pops;
    int32_t *out_status;
};

template <int LPR>
__device__ __forceinline__ void clear_vis(uint32_t *vis, uint32_t vlog2, uint32_t lane) {
    uint4 *v4 = reinterpret_cast<uint4 *>(vis);
    const uint32_t n4 = (1u << vlog2) / 4u;
    const uint4 e = make_uint4(VIS_EMPTY, VIS_EMPTY, VIS_EMPTY, VIS_EMPTY);
    for (uint32_t i = lane; i < n4; i += 64) v4[i] = e;
    __threadfence_block();
}

template <int LPR>
__global__ __launch_bounds__(64) void search_kernel(SearchParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    WaveLds L = carve_lds(smem, P.ef_cap);
    const uint32_t lane = threadIdx.x, q = blockIdx.x;
    L.claim[lane] = 0u; L.claim[lane + 64u] = 0u;
    const uint4 qv = P.queries[(uint64_t)q * LPR + lane % LPR];
    const uint32_t qpop = P.qpop[q];
    uint32_t *vis = P.vis + ((uint64_t)q << P.vlog2);
    SearchCounters C = {0, 0, 0};
    // distance to the entry point
    if (lane == 0) L.u32a[0] = P.entry;
    WSYNC();
    eval_rows<LPR>(P.G.fp, qv, qpop, L, 1, lane);
    C.evals++;
    unsigned long long curk = bk_key(rh_q24_dev(L.u32b[0], L.u32c[0]), P.entry);
    WSYNC();
    for (int32_t l = P.max_level; l > 0; --l) curk = greedy_level<LPR>(P.G, qv, qpop, (uint32_t)l, curk, L, C, lane);
    clear_vis<LPR>(vis, P.vlog2, lane);
    uint32_t vis_count = 1;
    if (lane == 0) { vis_test_and_set(vis, (1u << P.vlog2) - 1u, 32u - P.vlog2, bk_slot(curk)); L.topA[0] = curk; }
    WSYNC();
    uint32_t n_top = search_layer_any<LPR>(P.G, qv, qpop, 0, P.ef, P.ef_cap, 1, L, vis, P.vlog2, vis_count, C, lane);
    const uint32_t nk = n_top < P.k ? n_top : P.k;
    // (and, or) are recomputed for the k results from their slots
    for (uint32_t base = 0; base < nk; base += 64) {
        const uint32_t cntb = nk - base < 64 ? nk - base : 64;
        if (lane < cntb) L.u32a[lane] = bk_slot(L.topA[base + lane]);
        WSYNC();
        eval_rows<LPR>(P.G.fp, qv, qpop, L, cntb, lane);
        if (lane < cntb) {
            P.out_slots[(uint64_t)q * P.k + base + lane] = L.u32a[lane];
            P.out_and[(uint64_t)q * P.k + base + lane] = L.u32b[lane];
            P.out_or[(uint64_t)q * P.k + base + lane] = L.u32c[lane];
        }
        WSYNC();
    }
    // fewer than k results (small or disconnected graph): the tail of the row is defined too
    for (uint32_t i = nk + lane; i < P.k; i += 64) {
        P.out_slots[(uint64_t)q * P.k + i] = RADHIP_NO_SLOT;
        P.out_and[(uint64_t)q * P.k + i] = 0u;
        P.out_or[(uint64_t)q * P.k + i] = 0u;
    }
    if (lane == 0) {
        P.out_counts[q] = nk;
        P.out_evals[q] = C.evals;
        P.out_pops[q] = C.pops;
        P.out_status[q] = C.status;
    }
}

// -------------------------------------------------------------- build kernels
struct BuildParams {
    GraphView G;
    uint32_t batch_start, batch_n;
    uint32_t snap_entry;
    int32_t snap_max_level;
    uint32_t ef, ef_cap;
    uint32_t *vis;
    uint32_t vlog2;
    uint4 *req;                    // {target, level, source, 0}
    unsigned long long *req_count;
    uint32_t req_cap;
    int32_t *status;               // [batch_n]
};

// Six wavefronts per SIMD (80 VGPRs instead of 81) beside 6.4 KB of LDS per wavefront at expansion_add 400: 24 searches per CU
// instead of 15 — the search is a chain of dependent memory accesses, so the wavefronts in flight are its throughput
// (10M rows: 9.44 -> 8.84 s; with two LDS buffers and 15 wavefronts per CU 11.2 s; -DBK_WAVES_PER_EU=5 for the A/B).
#ifndef BK_WAVES_PER_EU
#define BK_WAVES_PER_EU 6
#endif
#define BK_OCC_ATTR __attribute__((amdgpu_waves_per_eu(BK_WAVES_PER_EU, BK_WAVES_PER_EU)))
template <int LPR>
__global__ __launch_bounds__(64) BK_OCC_ATTR void build_insert_kernel(BuildParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    WaveLds L = carve_lds(smem, P.ef_cap);
    const uint32_t lane = threadIdx.x;
#ifdef BK_PROFILE
    struct Whole { unsigned long long t0; uint32_t lane; __device__ ~Whole() { if (lane == 0) atomicAdd(&bk_prof[8], __builtin_amdgcn_s_memtime() - t0); } } whole_{__builtin_amdgcn_s_memtime(), lane};
#endif
    L.claim[lane] = 0u; L.claim[lane + 64u] = 0u;
    const uint32_t i = P.batch_start + blockIdx.x;
    const uint4 qv = P.G.fp[(uint64_t)i * LPR + lane % LPR];
    const uint32_t qpop = rh_group_sum<LPR>(rh_popc4(qv));
    uint32_t *vis = P.vis + ((uint64_t)blockIdx.x << P.vlog2);
    SearchCounters C = {0, 0, 0};
    const int32_t lv = P.G.levels[i];
    if (lane == 0) L.u32a[0] = P.snap_entry;
    WSYNC();
    eval_rows<LPR>(P.G.fp, qv, qpop, L, 1, lane);
    unsigned long long curk = bk_key(rh_q24_dev(L.u32b[0], L.u32c[0]), P.snap_entry);
    WSYNC();
    for (int32_t l = P.snap_max_level; l > lv; --l) curk = greedy_level<LPR>(P.G, qv, qpop, (uint32_t)l, curk, L, C, lane);
    for (int32_t l = lv < P.snap_max_level ? lv : P.snap_max_level; l >= 0 && C.status == 0; --l) {
        clear_vis<LPR>(vis, P.vlog2, lane);
        uint32_t vis_count = 1;
        if (lane == 0) { vis_test_and_set(vis, (1u << P.vlog2) - 1u, 32u - P.vlog2, bk_slot(curk)); L.topA[0] = curk & ~1ull; }
        WSYNC();
        const uint32_t n_top = search_layer_any<LPR>(P.G, qv, qpop, (uint32_t)l, P.ef, P.ef_cap, 1, L, vis, P.vlog2, vis_count, C, lane);
        if (C.status) break;
        uint32_t cap;
        uint32_t *row = gv_row(P.G, i, (uint32_t)l, &cap);
#ifdef BK_PROFILE
        const unsigned long long ts0_ = __builtin_amdgcn_s_memtime();
#endif
        const uint32_t k = cap <= 2u * (64u / LPR)
                               ? select_heuristic_regs<LPR>(P.G.fp, L.topA, n_top, cap, L.u32a, reinterpret_cast<uint4 *>(L.newk), lane)
                               : select_heuristic<LPR>(P.G.fp, L.topA, n_top, cap, L.u32a, lane);
#ifdef BK_PROFILE
        if (lane == 0) { atomicAdd(&bk_prof[9], __builtin_amdgcn_s_memtime() - ts0_); atomicAdd(&bk_prof[10], (unsigned long long)n_top); atomicAdd(&bk_prof[11], 1ull); }
#endif
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(P.req_count, (unsigned long long)k);
        base = __shfl(base, 0, RH_WAVE);
        if (lane < k) {
            row[lane] = L.u32a[lane];
            if (base + lane < P.req_cap) P.req[base + lane] = make_uint4(L.u32a[lane], (uint32_t)l, i, 0u);
        }
        curk = L.topA[0] & ~1ull;
        WSYNC();
    }
    if (lane == 0) P.status[blockIdx.x] = C.status;
}

struct ReverseParams {
    GraphView G;
    const uint4 *req;          // sorted by (target, level, source)
    const uint32_t *group_off;  // [n_groups + 1]
    uint32_t n_groups;
};

template <int LPR>
__global__ __launch_bounds__(64) void build_reverse_kernel(ReverseParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    WaveLds L = carve_lds(smem, 64);
    const uint32_t lane = threadIdx.x, g = blockIdx.x;
    const uint32_t r0 = P.group_off[g], r1 = P.group_off[g + 1];
    const uint4 first = P.req[r0];
    const uint32_t t = first.x, l = first.y;
    uint32_t cap;
    uint32_t *row = gv_row(P.G, t, l, &cap);
    const uint4 tv = P.G.fp[(uint64_t)t * LPR + lane % LPR];
    const uint32_t tpop = rh_group_sum<LPR>(rh_popc4(tv));
    uint32_t mine = lane < cap ? row[lane] : RADHIP_NO_SLOT;   // lane j holds row[j]
    uint32_t cnt = (uint32_t)__popcll(__ballot(mine != RADHIP_NO_SLOT));
    for (uint32_t r = r0; r < r1; ++r) {
        const uint32_t s = P.req[r].z;
        if (cnt < cap) {
            if (lane == cnt) mine = s;
            cnt++;
            continue;
        }
        // re-select among the cap existing neighbours and s, by distance to t
        const uint32_t n = cap + 1u;
        unsigned long long *cand = L.newk;  // [128]
        for (uint32_t base = 0; base < n; base += 64) {
            const uint32_t cb = n - base < 64 ? n - base : 64;
            const uint32_t j = base + lane;
            const uint32_t from_row = __shfl(mine, (int)(j & 63u), RH_WAVE);  // all lanes take part
            if (lane < cb) L.u32a[lane] = j < cap ? from_row : s;
            WSYNC();
            eval_rows<LPR>(P.G.fp, tv, tpop, L, cb, lane);
            if (lane < cb) cand[base + lane] = bk_key(rh_q24_dev(L.u32b[lane], L.u32c[lane]), L.u32a[lane]);
            WSYNC();
        }
        for (uint32_t j = n + lane; j < 128; j += 64) cand[j] = BK_INF;
        WSYNC();
        // ascending bitonic sort of 128 keys in LDS
        for (uint32_t kk = 2; kk <= 128; kk <<= 1)
            for (uint32_t jj = kk >> 1; jj > 0; jj >>= 1) {
                const uint32_t tt = lane;
                const uint32_t a_i = ((tt & ~(jj - 1)) << 1) | (tt & (jj - 1));
                const uint32_t b_i = a_i | jj;
                const bool up = (a_i & kk) == 0;
                const unsigned long long a = cand[a_i], b = cand[b_i];
                if ((a > b) == up) { cand[a_i] = b; cand[b_i] = a; }
                WSYNC();
            }
        const uint32_t k = select_heuristic<LPR>(P.G.fp, cand, n, cap, L.u32a, lane);
        mine = lane < k ? L.u32a[lane] : RADHIP_NO_SLOT;
        cnt = k;
        WSYNC();
    }
    if (lane < cap) row[lane] = mine;
}

// ================================================================= host side
static uint64_t bh_h64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}
// integer-only geometric level draw, P(level >= l) = M^-l (same as orc_hnsw_level_of)
static int level_of(uint64_t seed, uint64_t slot, uint32_t M) {
    uint64_t h = bh_h64(seed ^ (slot * 0x9E3779B97F4A7C15ULL + 0x632BE59BD9B4E019ULL));
    int l = 0;
    while (l < 15 && (h % M) == 0) { h /= M; l++; }
    return l;
}
extern "C" int radhip_level_of(uint64_t seed, uint64_t slot, uint32_t connectivity) {
    return level_of(seed, slot, connectivity);
}

static uint32_t next_pow2(uint32_t x) { uint32_t p = 1; while (p < x) p <<= 1; return p; }
static uint32_t log2u(uint32_t x) { uint32_t l = 0; while ((1u << l) < x) l++; return l; }

static GraphView make_view(radhip_index *idx) {
    GraphView G;
    G.fp = idx->d_fp; G.levels = idx->d_levels; G.adj0 = idx->d_adj0; G.upper_row = idx->d_upper_row;
    G.adjU = idx->d_adjU; G.cap0 = idx->cap0; G.capU = idx->M;
    return G;
}

template <typename T>
static int grow_dev(T **ptr, uint64_t old_elems, uint64_t new_elems, hipStream_t s, uint64_t *acct) {
    T *np_ = nullptr;
    RH_HIP(hipMalloc((void **)&np_, std::max<uint64_t>(new_elems, 4) * sizeof(T)));
    if (*ptr && old_elems) RH_HIP(hipMemcpyAsync(np_, *ptr, old_elems * sizeof(T), hipMemcpyDeviceToDevice, s));
    RH_HIP(hipStreamSynchronize(s));
    if (*ptr) (void)hipFree(*ptr);
    *ptr = np_;
    *acct += (new_elems - old_elems) * sizeof(T);
    return RADHIP_OK;
}

// ---- reverse-edge requests: order by (target, level, source) and group by (target, level), on the
// device.  A source is a node of the current batch, so (source - batch_start) fits 16 bits and the whole
// order fits one 52-bit radix key; the request is rebuilt from the key, there is no payload to move.
__global__ void req_pack_kernel(const uint4 *req, uint64_t n, uint32_t batch_start, unsigned long long *keys) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 r = req[j];
        keys[j] = ((unsigned long long)r.x << 20) | ((unsigned long long)(r.y & 15u) << 16) | (unsigned long long)(r.z - batch_start);
    }
}
__global__ void req_unpack_kernel(const unsigned long long *keys, uint64_t n, uint32_t batch_start, uint4 *req,
                                  unsigned char *group_start) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long k = keys[j];
        req[j] = make_uint4((uint32_t)(k >> 20), (uint32_t)(k >> 16) & 15u, batch_start + (uint32_t)(k & 0xFFFFull), 0u);
        group_start[j] = (j == 0 || (keys[j - 1] >> 16) != (k >> 16)) ? 1 : 0;
    }
}

// rows == nullptr: the `count` rows behind the graph's last node are resident already (radhip_index_link_resident)
static int add_impl(radhip_index *idx, const uint8_t *rows, uint64_t count, uint64_t seed, uint32_t max_batch) {
    const bool resident = rows == nullptr;
    if (count == 0) return RADHIP_OK;
    if (max_batch < 1) max_batch = 1;
    if (max_batch > 65536) max_batch = 65536;
    RH_REQUIRE_FULL_CORPUS(idx);
    RH_REQUIRE_SOUND(idx);
    if (idx->has_graph && idx->g_n && (idx->h_levels.size() != idx->g_n || idx->h_upper_row.size() != idx->g_n)) {
        // graph generated on the device (synthetic): mirror it once; after that add() keeps levels / upper rows
        // itself (the adjacency rows are not needed on the host here)
        RH_TRY(rh_ensure_host_graph(idx));
    }
    if (!resident) {
        if (idx->has_vectors && idx->has_graph && idx->n != idx->g_n)
            RH_FAIL(RADHIP_E_STATE, "add() needs corpus and graph of equal size (%llu vs %llu)",
                    (unsigned long long)idx->n, (unsigned long long)idx->g_n);
        if (idx->has_vectors != idx->has_graph && (idx->n || idx->g_n))
            RH_FAIL(RADHIP_E_STATE, "add() cannot extend an index that has vectors without a graph (or vice versa)");
    }
    RH_TRY(rh_ensure_device(idx));
    rh_layout_invalidate(idx);
    const uint64_t first = idx->has_graph ? idx->g_n : 0, total = first + count;
    if (total >= 0xFFFFFFF0ull) RH_FAIL(RADHIP_E_INVALID, "too many nodes");
    const uint32_t lpr = idx->lpr;

    // ---- host bookkeeping: levels + upper-row allocation of every new node -------
    std::vector<int8_t> &hl = idx->h_levels;
    std::vector<uint32_t> &hu = idx->h_upper_row;
    hl.resize(total);
    hu.resize(total);
    uint64_t nu = idx->n_upper_rows;
    for (uint64_t i = first; i < total; ++i) {
        const int lv = level_of(seed, i, idx->M);
        hl[i] = (int8_t)lv;
        hu[i] = lv > 0 ? (uint32_t)nu : RADHIP_NO_SLOT;
        nu += (uint64_t)lv;
    }
    const uint64_t old_nu = idx->n_upper_rows;

    // The index object (sizes, entry point, host mirror) changes only when the last batch is linked.  A failure
    // before the first reverse-link pass leaves the index as it was and extendable (the arrays may keep their
    // larger allocations, their contents beyond the old size are never read).  A failure AFTER a reverse-link
    // pass does not: build_reverse_kernel has written links to the new slots into the rows of existing nodes,
    // and those slots were never committed — the index is marked poisoned and refuses every further use
    // (RADHIP_E_STATE) instead of serving edges into rows that do not exist.
    bool committed = false, reverse_ran = false;
    auto rollback = [&]() { hl.resize(first); hu.resize(first); if (reverse_ran && first) idx->poisoned = true; };
    // ---- grow device arrays: by half at least, so that many small add() calls stay linear ---------
    {
        const uint64_t need = total, have = idx->d_fp ? idx->fp_cap_rows : 0;
        if (!resident && need > have) {
            uint64_t ncap = std::max<uint64_t>(need, have + have / 2);
            uint4 *nfp = nullptr;
            hipError_t e = hipMalloc((void **)&nfp, ncap * idx->row_stride);
            if (e != hipSuccess && ncap > need) { ncap = need; e = hipMalloc((void **)&nfp, ncap * idx->row_stride); }   // tight on memory: exact size
            if (e != hipSuccess) { (void)hipGetLastError(); rollback(); RH_FAIL(e == hipErrorOutOfMemory ? RADHIP_E_NOMEM : RADHIP_E_HIP, "hipMalloc for %llu rows failed: %s", (unsigned long long)need, hipGetErrorString(e)); }
            if (idx->d_fp && first && (hipMemcpyAsync(nfp, idx->d_fp, first * idx->row_stride, hipMemcpyDeviceToDevice, idx->stream) != hipSuccess ||
                                       hipStreamSynchronize(idx->stream) != hipSuccess)) {
                (void)hipFree(nfp); rollback(); RH_FAIL(RADHIP_E_HIP, "device copy of the corpus failed");
            }
            if (idx->d_fp) { (void)hipFree(idx->d_fp); idx->device_bytes -= std::min<uint64_t>(idx->device_bytes, have * idx->row_stride); }
            idx->d_fp = nfp;
            idx->fp_cap_rows = ncap;
            idx->device_bytes += ncap * idx->row_stride;
        }
        // the new rows go up in pieces of 64 MB: the staging copy of a 5M-row add() would otherwise be 640 MB of host memory
        if (!resident) {
            const uint64_t piece = std::max<uint64_t>(1, (64ull << 20) / idx->row_stride);
            std::vector<uint8_t> stage;
            try { stage.assign((size_t)std::min<uint64_t>(piece, count) * idx->row_stride, 0); }
            catch (...) { rollback(); RH_FAIL(RADHIP_E_NOMEM, "out of host memory staging the new rows"); }
            for (uint64_t f = 0; f < count; f += piece) {
                const uint64_t c = std::min<uint64_t>(piece, count - f);
                for (uint64_t i = 0; i < c; ++i)
                    memcpy(stage.data() + i * idx->row_stride, rows + (f + i) * idx->row_bytes, idx->row_bytes);
                if (hipMemcpyAsync((uint8_t *)idx->d_fp + (first + f) * idx->row_stride, stage.data(), (size_t)c * idx->row_stride, hipMemcpyHostToDevice, idx->stream) != hipSuccess ||
                    hipStreamSynchronize(idx->stream) != hipSuccess) { rollback(); RH_FAIL(RADHIP_E_HIP, "upload of the new rows failed"); }
            }
        }
    }
    {
        int rc = RADHIP_OK;
        if (total > idx->cap_nodes || !idx->d_levels) {
            const uint64_t ncap = std::max<uint64_t>(total, idx->cap_nodes + idx->cap_nodes / 2);
            rc = grow_dev(&idx->d_levels, first, ncap, idx->stream, &idx->device_bytes);
            if (rc == RADHIP_OK) rc = grow_dev(&idx->d_adj0, first * idx->cap0, ncap * idx->cap0, idx->stream, &idx->device_bytes);
            if (rc == RADHIP_OK) rc = grow_dev(&idx->d_upper_row, first, ncap, idx->stream, &idx->device_bytes);
            if (rc == RADHIP_OK) idx->cap_nodes = ncap;
        }
        if (rc == RADHIP_OK && (nu > idx->cap_upper || !idx->d_adjU)) {
            const uint64_t ncap = std::max<uint64_t>(nu, idx->cap_upper + idx->cap_upper / 2);
            rc = grow_dev(&idx->d_adjU, old_nu * idx->M, ncap * idx->M, idx->stream, &idx->device_bytes);
            if (rc == RADHIP_OK) idx->cap_upper = ncap;
        }
        if (rc != RADHIP_OK) { rollback(); return rc; }
    }
    // On the stream the build kernels run on, and waited for: hipMemset of device memory returns before the fill has run,
    // and the library's stream is non-blocking — it does not wait for the null stream.  The first insert kernels of a
    // 100M-row call started while the 6.4 GB fill of the adjacency was still in flight and read whatever the memory held
    // before (zero pages in a fresh process, another index's rows in a used one: a memory fault; profiles/r03).
    if (hipMemcpyAsync(idx->d_levels + first, hl.data() + first, count, hipMemcpyHostToDevice, idx->stream) != hipSuccess ||
        hipMemcpyAsync(idx->d_upper_row + first, hu.data() + first, count * 4, hipMemcpyHostToDevice, idx->stream) != hipSuccess ||
        hipMemsetAsync(idx->d_adj0 + first * idx->cap0, 0xFF, count * idx->cap0 * 4, idx->stream) != hipSuccess ||
        (nu > old_nu && hipMemsetAsync(idx->d_adjU + old_nu * idx->M, 0xFF, (nu - old_nu) * idx->M * 4, idx->stream) != hipSuccess) ||
        hipStreamSynchronize(idx->stream) != hipSuccess) {
        rollback();
        RH_FAIL(RADHIP_E_HIP, "initialising the new graph rows failed");
    }

    // ---- per-batch scratch ----------------------------------------------------------
    const uint32_t ef = std::max<uint32_t>(idx->ef_add, 1);
    const uint32_t ef_cap = next_pow2(std::max<uint32_t>(ef, 64));
    const size_t lds = wave_lds_bytes(ef_cap);
    if (lds > 64 * 1024) { rollback(); RH_FAIL(RADHIP_E_INVALID, "expansion_add %u too large for the LDS top buffer", ef); }
    uint32_t vlog2 = std::min<uint32_t>(20, std::max<uint32_t>(10, log2u(4u * (ef + 64u) * idx->cap0)));
    const uint64_t bmax = std::min<uint64_t>(max_batch, count);
    uint32_t *d_vis = nullptr;
    int32_t *d_status = nullptr;
    unsigned long long *d_req_count = nullptr;
    uint4 *d_req = nullptr;
    uint32_t *d_goff = nullptr, *d_ng = nullptr;
    unsigned long long *d_keys_a = nullptr, *d_keys_b = nullptr;
    unsigned char *d_gstart = nullptr;
    void *d_tmp = nullptr;
    size_t tmp_bytes = 0;
    uint64_t req_cap = 0;
    auto free_req = [&]() {
        void *ps[] = {d_req, d_goff, d_keys_a, d_keys_b, d_gstart, d_tmp};
        for (void *p : ps) if (p) (void)hipFree(p);
        d_req = nullptr; d_goff = nullptr; d_keys_a = d_keys_b = nullptr; d_gstart = nullptr; d_tmp = nullptr; tmp_bytes = 0;
    };
    auto cleanup = [&]() {   // (failure paths only reach it before the commit at the end: the index rolls back)
        if (d_vis) (void)hipFree(d_vis); if (d_status) (void)hipFree(d_status);
        if (d_req_count) (void)hipFree(d_req_count); if (d_ng) (void)hipFree(d_ng);
        free_req();
        if (!committed) rollback();
    };
#define BH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { radhip_set_error("%s failed: %s", #x, hipGetErrorString(e_)); (void)hipGetLastError(); cleanup(); return e_ == hipErrorOutOfMemory ? RADHIP_E_NOMEM : RADHIP_E_HIP; } } while (0)
    BH(hipMalloc((void **)&d_vis, (bmax << vlog2) * 4));
    BH(hipMalloc((void **)&d_status, bmax * 4));
    BH(hipMalloc((void **)&d_req_count, 8));
    BH(hipMalloc((void **)&d_ng, 4));

    std::vector<int32_t> hstatus(bmax);
    uint32_t entry = idx->has_graph ? idx->entry : RADHIP_NO_SLOT;
    int32_t max_level = idx->has_graph ? idx->max_level : -1;

    uint64_t start = first;
    while (start < total) {
        uint64_t bs = start / 16;
        if (bs < 1) bs = 1;
        if (bs > max_batch) bs = max_batch;
        if (start + bs > total) bs = total - start;
        const uint64_t end = start + bs;
        if (entry != RADHIP_NO_SLOT) {
            uint64_t need = 0;
            for (uint64_t i = start; i < end; ++i) need += idx->cap0 + (uint64_t)hl[i] * idx->M;
            if (need > req_cap) {
                free_req();
                req_cap = need * 2;
                BH(hipMalloc((void **)&d_req, req_cap * sizeof(uint4)));
                BH(hipMalloc((void **)&d_keys_a, req_cap * 8));
                BH(hipMalloc((void **)&d_keys_b, req_cap * 8));
                BH(hipMalloc((void **)&d_gstart, req_cap));
                BH(hipMalloc((void **)&d_goff, (req_cap + 1) * 4));
                size_t t1 = 0, t2 = 0;
                BH(rocprim::radix_sort_keys(nullptr, t1, d_keys_a, d_keys_b, (size_t)req_cap, 0u, 52u, idx->stream));
                BH(rocprim::select(nullptr, t2, rocprim::counting_iterator<uint32_t>(0), d_gstart, d_goff, d_ng, (size_t)req_cap, idx->stream));
                tmp_bytes = std::max(t1, t2) + 256;
                BH(hipMalloc(&d_tmp, tmp_bytes));
            }
            unsigned long long nreq = 0;
            for (;;) {
                BH(hipMemsetAsync(d_req_count, 0, 8, idx->stream));
                BuildParams BP;
                BP.G = make_view(idx);
                BP.batch_start = (uint32_t)start; BP.batch_n = (uint32_t)bs;
                BP.snap_entry = entry; BP.snap_max_level = max_level;
                BP.ef = ef; BP.ef_cap = ef_cap; BP.vis = d_vis; BP.vlog2 = vlog2;
                BP.req = d_req; BP.req_count = d_req_count; BP.req_cap = (uint32_t)std::min<uint64_t>(req_cap, 0xFFFFFFFFull);
                BP.status = d_status;
                switch (lpr) {
                    case 1: hipLaunchKernelGGL(build_insert_kernel<1>, dim3((uint32_t)bs), dim3(64), lds, idx->stream, BP); break;
                    case 2: hipLaunchKernelGGL(build_insert_kernel<2>, dim3((uint32_t)bs), dim3(64), lds, idx->stream, BP); break;
                    case 4: hipLaunchKernelGGL(build_insert_kernel<4>, dim3((uint32_t)bs), dim3(64), lds, idx->stream, BP); break;
                    case 8: hipLaunchKernelGGL(build_insert_kernel<8>, dim3((uint32_t)bs), dim3(64), lds, idx->stream, BP); break;
                    default: hipLaunchKernelGGL(build_insert_kernel<16>, dim3((uint32_t)bs), dim3(64), lds, idx->stream, BP); break;
                }
                BH(hipGetLastError());
                BH(hipMemcpyAsync(&nreq, d_req_count, 8, hipMemcpyDeviceToHost, idx->stream));
                BH(hipMemcpyAsync(hstatus.data(), d_status, bs * 4, hipMemcpyDeviceToHost, idx->stream));
                BH(hipStreamSynchronize(idx->stream));
                bool overflow = false;
                for (uint64_t b = 0; b < bs; ++b)
                    if (hstatus[b] == RADHIP_E_CAPACITY) overflow = true;
                    else if (hstatus[b] != 0) { const int st_ = hstatus[b]; cleanup(); RH_FAIL(st_, "insert of node %llu failed on the device (status %d)", (unsigned long long)(start + b), st_); }
                if (!overflow) break;
                // A search of this batch filled its visited table (duplicate-heavy corpora expand far more nodes than
                // expansion_add suggests): the batch has only written rows of its own nodes, which are reset, and runs
                // again with a table twice the size — the result is the one a large table would have given at once.
                if (vlog2 >= 26) { cleanup(); RH_FAIL(RADHIP_E_CAPACITY, "a search of batch %llu overflowed a 2^26-entry visited table", (unsigned long long)start); }
                vlog2++;
                (void)hipFree(d_vis); d_vis = nullptr;
                BH(hipMalloc((void **)&d_vis, (bmax << vlog2) * 4));
                BH(hipMemsetAsync(idx->d_adj0 + start * idx->cap0, 0xFF, bs * idx->cap0 * 4, idx->stream));
                {
                    uint64_t u0 = RADHIP_NO_SLOT, u1 = 0;
                    for (uint64_t i = start; i < end; ++i) if (hl[i] > 0) { if (u0 == RADHIP_NO_SLOT) u0 = hu[i]; u1 = (uint64_t)hu[i] + (uint64_t)hl[i]; }
                    if (u0 != RADHIP_NO_SLOT) BH(hipMemsetAsync(idx->d_adjU + u0 * idx->M, 0xFF, (u1 - u0) * idx->M * 4, idx->stream));
                }
            }
            if (nreq > req_cap) { cleanup(); RH_FAIL(RADHIP_E_CAPACITY, "reverse-edge request buffer overflow"); }
            if (nreq) {
                const uint32_t sgrid = (uint32_t)std::min<uint64_t>((nreq + 255) / 256, 4096);
                hipLaunchKernelGGL(req_pack_kernel, dim3(sgrid), dim3(256), 0, idx->stream, d_req, (uint64_t)nreq, (uint32_t)start, d_keys_a);
                size_t tb = tmp_bytes;
                BH(rocprim::radix_sort_keys(d_tmp, tb, d_keys_a, d_keys_b, (size_t)nreq, 0u, 52u, idx->stream));
                hipLaunchKernelGGL(req_unpack_kernel, dim3(sgrid), dim3(256), 0, idx->stream, d_keys_b, (uint64_t)nreq, (uint32_t)start, d_req, d_gstart);
                tb = tmp_bytes;
                BH(rocprim::select(d_tmp, tb, rocprim::counting_iterator<uint32_t>(0), d_gstart, d_goff, d_ng, (size_t)nreq, idx->stream));
                uint32_t ng = 0;
                BH(hipMemcpyAsync(&ng, d_ng, 4, hipMemcpyDeviceToHost, idx->stream));
                BH(hipStreamSynchronize(idx->stream));
                const uint32_t nreq32 = (uint32_t)nreq;
                BH(hipMemcpyAsync(d_goff + ng, &nreq32, 4, hipMemcpyHostToDevice, idx->stream));
                ReverseParams RP;
                RP.G = make_view(idx); RP.req = d_req; RP.group_off = d_goff; RP.n_groups = ng;
                const size_t rlds = wave_lds_bytes(64);
                switch (lpr) {
                    case 1: hipLaunchKernelGGL(build_reverse_kernel<1>, dim3(ng), dim3(64), rlds, idx->stream, RP); break;
                    case 2: hipLaunchKernelGGL(build_reverse_kernel<2>, dim3(ng), dim3(64), rlds, idx->stream, RP); break;
                    case 4: hipLaunchKernelGGL(build_reverse_kernel<4>, dim3(ng), dim3(64), rlds, idx->stream, RP); break;
                    case 8: hipLaunchKernelGGL(build_reverse_kernel<8>, dim3(ng), dim3(64), rlds, idx->stream, RP); break;
                    default: hipLaunchKernelGGL(build_reverse_kernel<16>, dim3(ng), dim3(64), rlds, idx->stream, RP); break;
                }
                reverse_ran = true;
                BH(hipGetLastError());
                BH(hipStreamSynchronize(idx->stream));
            }
        }
        for (uint64_t i = start; i < end; ++i)
            if (hl[i] > max_level) { max_level = hl[i]; entry = (uint32_t)i; }
        start = end;
    }
    committed = true;
    cleanup();
#undef BH
    idx->n_upper_rows = nu;
    idx->n = total;
    idx->has_vectors = true;
    idx->g_n = total;
    idx->entry = entry;
    idx->max_level = max_level;
    idx->has_graph = true;
    idx->d_graph_valid = true;
    idx->h_graph_valid = false;   // adjacency rows changed on the device; levels/upper_row stay valid
    idx->h_top.clear();
    for (uint64_t i = 0; i < total; ++i)
        if (hl[i] == max_level) idx->h_top.push_back((uint32_t)i);
    if (idx->d_top) { (void)hipFree(idx->d_top); idx->d_top = nullptr; }
    idx->n_top = (uint32_t)idx->h_top.size();
    RH_HIP(hipMalloc((void **)&idx->d_top, std::max<size_t>(idx->n_top, 4) * 4));
    RH_HIP(hipMemcpy(idx->d_top, idx->h_top.data(), (size_t)idx->n_top * 4, hipMemcpyHostToDevice));
    return RADHIP_OK;
}

extern "C" int radhip_index_add(radhip_index_t *idx, const uint8_t *rows, uint64_t count, uint64_t seed,
                                uint32_t max_batch) {
    if (!idx || (!rows && count)) RH_FAIL(RADHIP_E_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    return add_impl(idx, rows, count, seed, max_batch);
}

#ifdef BK_PROFILE
static void bk_prof_print() {
    unsigned long long h[12];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(bk_prof), sizeof h) != hipSuccess) return;
    unsigned long long tot = 0; for (int i = 0; i < 6; ++i) tot += h[i];
    const char *nm[6] = {"loop-top", "scan+mark", "adjacency+probe", "eval", "keys+filter", "merge"};
    fprintf(stderr, "[bk_prof] %llu ticks:", tot);
    for (int i = 0; i < 6; ++i) fprintf(stderr, " %s=%.1f%%", nm[i], 100.0 * h[i] / (tot ? tot : 1));
    fprintf(stderr, "; %llu pops with new neighbours, %.2f kept keys each\n", h[6], h[6] ? (double)h[7] / h[6] : 0.0);
    fprintf(stderr, "[bk_prof] whole inserts %llu ticks: the searches' sections %.1f%%, select_heuristic %.1f%% (%llu calls over %.1f candidates each)\n",
            h[8], 100.0 * tot / (h[8] ? h[8] : 1), 100.0 * h[9] / (h[8] ? h[8] : 1), h[11], h[11] ? (double)h[10] / h[11] : 0.0);
}
#endif
extern "C" int radhip_index_link_resident(radhip_index_t *idx, uint64_t seed, uint32_t max_batch) {
    if (!idx) RH_FAIL(RADHIP_E_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    if (!idx->has_vectors) RH_FAIL(RADHIP_E_STATE, "no vectors loaded");
    const uint64_t linked = idx->has_graph ? idx->g_n : 0;
    if (idx->n < linked) RH_FAIL(RADHIP_E_STATE, "the graph has more nodes (%llu) than the corpus has rows (%llu)",
                                 (unsigned long long)linked, (unsigned long long)idx->n);
#ifdef BK_PROFILE
    const int rc_ = add_impl(idx, nullptr, idx->n - linked, seed, max_batch);
    bk_prof_print();
    return rc_;
#else
    return add_impl(idx, nullptr, idx->n - linked, seed, max_batch);
#endif
}

extern "C" int radhip_search(radhip_index_t *idx, const uint8_t *queries, uint32_t nq, uint32_t k, uint32_t ef,
                             uint32_t *out_slots, uint32_t *out_and, uint32_t *out_or, uint32_t *out_counts,
                             uint64_t *out_evals, uint64_t *out_pops) {
    if (!idx || !queries || !out_slots || !out_counts) RH_FAIL(RADHIP_E_INVALID, "null argument");
    if (!idx->has_vectors || !idx->has_graph) RH_FAIL(RADHIP_E_STATE, "index needs vectors and a graph");
    RH_REQUIRE_FULL_CORPUS(idx);
    if (nq == 0 || k == 0) return RADHIP_OK;
    if (ef < k) ef = k;
    std::lock_guard<std::mutex> lk(idx->mu);
    RH_TRY(rh_ensure_device(idx));
    const uint32_t ef_cap = next_pow2(std::max<uint32_t>(ef, 64));
    const size_t lds = wave_lds_bytes(ef_cap);
    if (lds > 64 * 1024) RH_FAIL(RADHIP_E_INVALID, "ef %u too large for the LDS top buffer", ef);
    const uint32_t vlog2 = std::min<uint32_t>(20, std::max<uint32_t>(10, log2u(4u * (ef + 64u) * idx->cap0)));
    std::vector<uint8_t> padded((size_t)nq * idx->row_stride, 0);
    std::vector<uint32_t> pop(nq, 0);
    for (uint32_t i = 0; i < nq; ++i) {
        memcpy(padded.data() + (size_t)i * idx->row_stride, queries + (size_t)i * idx->row_bytes, idx->row_bytes);
        for (uint32_t b = 0; b < idx->row_bytes; ++b) pop[i] += (uint32_t)__builtin_popcount(queries[(size_t)i * idx->row_bytes + b]);
    }
    uint4 *dq = nullptr;
    uint32_t *dpop = nullptr, *dvis = nullptr, *ds = nullptr, *da = nullptr, *dorr = nullptr, *dc = nullptr;
    unsigned long long *de = nullptr, *dp = nullptr;
    int32_t *dst = nullptr;
    auto cleanup = [&]() {
        void *ps[] = {dq, dpop, dvis, ds, da, dorr, dc, de, dp, dst};
        for (void *p : ps) if (p) (void)hipFree(p);
    };
#define SH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { radhip_set_error("%s failed: %s", #x, hipGetErrorString(e_)); (void)hipGetLastError(); cleanup(); return e_ == hipErrorOutOfMemory ? RADHIP_E_NOMEM : RADHIP_E_HIP; } } while (0)
    SH(hipMalloc((void **)&dq, padded.size()));
    SH(hipMalloc((void **)&dpop, (size_t)nq * 4));
    SH(hipMalloc((void **)&dvis, ((size_t)nq << vlog2) * 4));
    SH(hipMalloc((void **)&ds, (size_t)nq * k * 4));
    SH(hipMalloc((void **)&da, (size_t)nq * k * 4));
    SH(hipMalloc((void **)&dorr, (size_t)nq * k * 4));
    SH(hipMalloc((void **)&dc, (size_t)nq * 4));
    SH(hipMalloc((void **)&de, (size_t)nq * 8));
    SH(hipMalloc((void **)&dp, (size_t)nq * 8));
    SH(hipMalloc((void **)&dst, (size_t)nq * 4));
    SH(hipMemcpyAsync(dq, padded.data(), padded.size(), hipMemcpyHostToDevice, idx->stream));
    SH(hipMemcpyAsync(dpop, pop.data(), (size_t)nq * 4, hipMemcpyHostToDevice, idx->stream));
    SearchParams SP;
    SP.G = make_view(idx); SP.queries = dq; SP.qpop = dpop; SP.entry = idx->entry; SP.max_level = idx->max_level;
    SP.k = k; SP.ef = ef; SP.ef_cap = ef_cap; SP.vis = dvis; SP.vlog2 = vlog2;
    SP.out_slots = ds; SP.out_and = da; SP.out_or = dorr; SP.out_counts = dc; SP.out_evals = de; SP.out_pops = dp;
    SP.out_status = dst;
    switch (idx->lpr) {
        case 1: hipLaunchKernelGGL(search_kernel<1>, dim3(nq), dim3(64), lds, idx->stream, SP); break;
        case 2: hipLaunchKernelGGL(search_kernel<2>, dim3(nq), dim3(64), lds, idx->stream, SP); break;
        case 4: hipLaunchKernelGGL(search_kernel<4>, dim3(nq), dim3(64), lds, idx->stream, SP); break;
        case 8: hipLaunchKernelGGL(search_kernel<8>, dim3(nq), dim3(64), lds, idx->stream, SP); break;
        default: hipLaunchKernelGGL(search_kernel<16>, dim3(nq), dim3(64), lds, idx->stream, SP); break;
    }
    SH(hipGetLastError());
    std::vector<int32_t> st(nq);
    SH(hipMemcpyAsync(out_slots, ds, (size_t)nq * k * 4, hipMemcpyDeviceToHost, idx->stream));
    if (out_and) SH(hipMemcpyAsync(out_and, da, (size_t)nq * k * 4, hipMemcpyDeviceToHost, idx->stream));
    if (out_or) SH(hipMemcpyAsync(out_or, dorr, (size_t)nq * k * 4, hipMemcpyDeviceToHost, idx->stream));
    SH(hipMemcpyAsync(out_counts, dc, (size_t)nq * 4, hipMemcpyDeviceToHost, idx->stream));
    if (out_evals) SH(hipMemcpyAsync(out_evals, de, (size_t)nq * 8, hipMemcpyDeviceToHost, idx->stream));
    if (out_pops) SH(hipMemcpyAsync(out_pops, dp, (size_t)nq * 8, hipMemcpyDeviceToHost, idx->stream));
    SH(hipMemcpyAsync(st.data(), dst, (size_t)nq * 4, hipMemcpyDeviceToHost, idx->stream));
    SH(hipStreamSynchronize(idx->stream));
    cleanup();
#undef SH
    for (uint32_t i = 0; i < nq; ++i)
        if (st[i] != 0) RH_FAIL(st[i], "search %u overflowed its visited table", i);
    return RADHIP_OK;
}
