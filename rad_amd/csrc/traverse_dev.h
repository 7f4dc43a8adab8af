// traverse_dev.h — what the traversal kernels share: state layout in HBM (TravHeader, TravParams), queue-key tables,
// small device helpers.  Included by traverse.hip (trav_kernel + the host side) and by the trav4_*.hip translation units,
// each of which instantiates a subset of trav4_kernel (traverse4.inc) so that the library builds in parallel.
#pragma once
#include "common.h"

#include <algorithm>
#include <new>

#ifndef RH_S_CAP
#define RH_S_CAP 512
#endif
#ifndef RH_MAX_RUNS
#define RH_MAX_RUNS 8192
#endif
#define S_CAP ((uint32_t)RH_S_CAP)
#define MAX_RUNS ((uint32_t)RH_MAX_RUNS)
#define RK 4
#define HT_EMPTY64 0xFFFFFFFFFFFFFFFFull
#define VAL_V0 (1u << 24)
// Lazy clearing: bits 31..25 of an entry's value word carry the epoch of the batch that wrote it
// (0x7F, what the 0xFF memset leaves, is never a live epoch); entries of older epochs read as empty,
// so re-arming the state for a new batch of queries does not touch the tables (49 GB at bench size).
// Home bucket of a slot: multiplicative hash.  (A variant that keeps 16 consecutive slots in one
// 128-B line of the table was measured slower, with and without a cluster-contiguous
// renumbering of the corpus: profiles/r01/README.md.)
#define RH_HT_HASH(s) (((s) * 2654435769u) >> ht_shift)
#define VAL_EPOCH_SHIFT 25
#define EPOCH_LIMIT 127u
#define DQ_INIT (1u << 14)
#define DQ_MAX (1u << 23)

// One wave per workgroup: LDS traffic of a single wave is executed in issue order, so
// cross-lane hand-offs through LDS need only a compiler barrier — not the
// s_waitcnt vmcnt(0) that __syncthreads() adds (it would stall on every outstanding store).
#define WSYNC()                                                  \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)

struct TravHeader {
    uint64_t n_scored, n_pops, n_nbr, pq_used, n_upper;
    uint64_t target;        // stop once n_scored >= target (checked before every pop)
    uint64_t frontier_key;  // best queue key when the kernel last returned (RH_KEY_INF = empty)
    uint64_t pivot;
    uint64_t n_repivot, n_flush;
    uint32_t stg_cnt, n_runs, qpop, primed;
    uint32_t dq, mid_pos;
    int32_t status;
    uint32_t dm;
    // trav4_kernel's three-level queue: keys below mid_limit are in registers / staging / the mid run
    // [mid_pos, mid_end) of the key pool; the far runs hold keys >= mid_limit only
    uint64_t mid_limit, far_min;
    uint32_t mid_end, n_remid;
    // row-sharded form of trav4_kernel: candidates out (count | level << 8 | lane rotation << 16), entry points primed
    uint32_t sh_pend, sh_prime_at;
};

struct TravParams {
    const uint4 *fp;
    const uint32_t *adj0, *upper_row, *adjU, *top;
    uint32_t n_top, cap0, capU, nq;
    int32_t start_level;
    uint32_t epoch;          // current batch (see VAL_EPOCH_SHIFT)
    uint32_t spread_shift;   // new keys of an expansion go to lanes (i << spread_shift) + rot
    uint32_t spec_passes;    // 1 or 2: speculative row gathers cover every neighbour; 0: disabled
    uint64_t n_to_score, max_pops;
    TravHeader *hdr;
    const uint4 *queries;
    unsigned long long *ht;  // {slot | val<<32}
    uint32_t ht_log2;
    // bucket table (trav4_kernel<.., BT = true>, traverse4.inc): per traversal 2^bt_log2 buckets of four u32 entries
    // (slot + 1) | epoch << bt_sbits | v0 << 31; epochs 1 .. 2^(31 - bt_sbits) - 1, 0 = cleared
    uint32_t *bt;
    uint32_t bt_log2, bt_sbits;
    unsigned long long *ut;
    uint32_t ut_log2;
    // grouped visited/scored table (GT kernels; needs the index's graph-locality layout, layout.hip):
    // per traversal 2^gt_log2 lines of 8 chunks {tag, 48 seen bits, 48 pend bits}, see traverse4.inc
    const uint2 *adjx0, *adjxU, *topx;   // {slot, layout id} pair rows
    const uint32_t *lid;
    unsigned long long *gt;              // [nq << (gt_log2 + 4)] (two u64 per chunk)
    uint32_t gt_log2;
    uint2 *scored;
    uint64_t scored_cap;
    unsigned long long *pq;
    uint64_t pq_cap;
    unsigned long long *stg_save;  // [nq * S_CAP]
    unsigned long long *r_save;    // [nq * RK * 64] near keys
    uint32_t max_runs;             // run-table entries per traversal: 8192, more for n_to_score beyond ~400k
    uint2 *runs;                   // [nq * max_runs] {pos, end}
    unsigned long long *rhead;     // [nq * MAX_RUNS] head key of every run (INF = exhausted)
    unsigned long long *midpool;   // trav4_kernel: [nq * 256] the sorted mid run of every traversal
    uint32_t *q_next;              // trav4_kernel: the next traversal of the batch a free row takes (zeroed before every launch)
    uint32_t q_static;             // 1: a row keeps the traversal its block index names and takes no other (grid = nq / 4)
    // SLOT form (traverse4.inc): the tables / key pool / run table / mid pool above exist once per resident row (`slots` of them,
    // row = blockIdx.x * 4 + g), hdr / queries / scored / pop log once per traversal
    uint32_t slots;                // 0 = state per traversal
    uint32_t epoch_first, epoch_max;   // epochs a row's table entries can carry (bucket table: 1 .. 2^(31 - bt_sbits) - 1; others 0 .. 126)
    uint32_t *slot_epoch;          // [slots] epoch of the last traversal each row worked on (epoch_first - 1 = cleared tables, nothing yet)
    uint32_t list_mask;            // 0xFFFFFFFF: one scored list per traversal; else ring - 1: traversal i writes list i & list_mask (SLOT form)
    uint32_t *poplog_nodes;
    uint8_t *poplog_levels;
    uint64_t poplog_cap;
    unsigned long long *prof;   // RH_PROFILE builds only: per-section cycle sums
    // row-sharded form (trav4_kernel<LPR, false, true>, shard.hip): candidate slots out, packed counts in
    uint32_t *sh_req;              // [nq * sh_W + 16]: this step's candidates (NO_SLOT padded), then the live count
    const uint32_t *sh_in;         // [nq * sh_W]: and | or << 16 of the last step's candidates
    uint32_t *sh_pend_h;           // [nq * 16]: the table bucket every candidate out has claimed
    uint32_t sh_W;
};

__device__ __forceinline__ unsigned long long ld64(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool ht_is_empty(unsigned long long e, uint32_t epoch) { return (uint32_t)(e >> 57) != epoch; }
// test-and-set of (slot<<4|level) in the upper-level visited set; entries of older epochs are free
__device__ __forceinline__ bool ut_test_and_set(unsigned long long *ut, uint32_t mask, uint32_t shift,
                                                unsigned long long body, uint32_t epoch) {
    const unsigned long long kk = (((unsigned long long)epoch << 40) | body) + 1ull;
    uint32_t h = (uint32_t)(((body + 1ull) * 0x9E3779B97F4A7C15ull) >> shift);
    for (;;) {
        const unsigned long long old = atomicCAS(&ut[h], 0ull, kk);
        if (old == 0ull) return true;
        if (old == kk) return false;
        if (((old - 1ull) >> 40) != (unsigned long long)epoch) {   // stale: take it over
            if (atomicCAS(&ut[h], old, kk) == old) return true;
            continue;                                              // another lane got there first: look again
        }
        h = (h + 1u) & mask;
    }
}
__device__ __forceinline__ void st_relaxed(uint32_t *p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Decimal tables of the queue key (the member string "{node}:{level}" is kept as a zero-padded decimal).  A select
// chain or comparison tree over a per-lane value compiles to a tree of exec-mask branches, ~50 scalar instructions
// per use; a table read from LDS is one.
struct KeyTabs {
    uint32_t pow10[12];      // 10^e, e = 0..9 (12 entries: the LDS block of trav4_kernel is counted in bytes)
    uint2 div10[12];         // {m, s}: n / 10^e == umulhi(n, m) >> s for n < 2^30, e = 1..9
};
__device__ __forceinline__ void keytabs_init(KeyTabs &T, uint32_t lane) {   // lanes 0..15 of a wavefront; sync before use
    if (lane < 12u) {
        uint32_t p = 1u;
        for (uint32_t i = 0; i < lane && i < 9u; ++i) p *= 10u;
        T.pow10[lane] = p;
        uint32_t l = 0;
        while ((1u << l) < p) l++;
        // m = floor(2^(30+l) / p) + 1 < 2^32; exact for n < 2^30 because 2^l > p (p is not a power of two for e >= 1)
        T.div10[lane] = lane == 0u ? make_uint2(0u, 0u) : make_uint2((uint32_t)(((1ull << (30u + l)) / p) + 1ull), l - 2u);
    }
}
// slot of a key: (p + 1) / 10^dl - 1
__device__ __forceinline__ uint32_t key_slot_tab(const KeyTabs &T, unsigned long long key) {
    const uint32_t n = ((uint32_t)(key >> 8) & 0x3FFFFFFFu) + 1u, e = (uint32_t)(key >> 4) & 0xFu;
    const uint2 ms = T.div10[e];
    return (e ? (__umulhi(n, ms.x) >> ms.y) : n) - 1u;
}
// rh_make_key (common.h): digits = floor(log10) estimate from the bit length, one compare to fix it
__device__ __forceinline__ unsigned long long make_key_tab(const KeyTabs &T, uint32_t q24, uint32_t slot, uint32_t level) {
    const uint32_t bits = 32u - (uint32_t)__clz((int)(slot | 1u));
    const uint32_t t = (bits * 1233u) >> 12;                    // digits - 1 or digits
    const uint32_t d = t + ((slot | 1u) >= T.pow10[t] ? 1u : 0u);
    const uint32_t dl = 9u - d;
    const uint32_t p = (slot + 1u) * T.pow10[dl] - 1u;
    return ((unsigned long long)q24 << 38) | ((unsigned long long)p << 8) | ((unsigned long long)dl << 4) | (unsigned long long)rh_level_rank(level);
}

struct TravLds {
    KeyTabs kt;
    unsigned long long stg[S_CAP];
    uint32_t new_slot[64];
    uint32_t new_h[64];
    uint32_t new_and[64];
    uint32_t new_or[64];
    uint32_t claim[128];
    uint32_t claimtab[128];   // buckets claimed by the expansion in flight (0 = free)
};

// ascending in-place bitonic sort of s[0..P), P a power of two, by one wave
__device__ void lds_bitonic_sort(unsigned long long *s, uint32_t P, uint32_t lane) {
    for (uint32_t k = 2; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = lane; t < (P >> 1); t += 64) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const uint32_t ixj = i | j;
                const bool up = (i & k) == 0;
                const unsigned long long a = s[i], b = s[ixj];
                if ((a > b) == up) { s[i] = b; s[ixj] = a; }
            }
            WSYNC();
        }
    }
}

// insert x into the lane's sorted keys k0<=k1<=k2<=k3 (RH_KEY_INF = free); returns the key that
// fell off the end (the largest of the five), RH_KEY_INF if there was room or x was INF
__device__ __forceinline__ unsigned long long r_insert(unsigned long long x, unsigned long long &k0,
                                                       unsigned long long &k1, unsigned long long &k2,
                                                       unsigned long long &k3) {
    unsigned long long t = x, a;
    if (t < k0) { a = k0; k0 = t; t = a; }
    if (t < k1) { a = k1; k1 = t; t = a; }
    if (t < k2) { a = k2; k2 = t; t = a; }
    if (t < k3) { a = k3; k3 = t; t = a; }
    return t;
}

__device__ __forceinline__ uint32_t key_slot(unsigned long long key) {
    const uint32_t p1 = ((uint32_t)(key >> 8) & 0x3FFFFFFFu) + 1u;
    switch ((uint32_t)(key >> 4) & 0xFu) {   // wave-uniform at the only call site
        case 0: return p1 - 1u;
        case 1: return p1 / 10u - 1u;
        case 2: return p1 / 100u - 1u;
        case 3: return p1 / 1000u - 1u;
        case 4: return p1 / 10000u - 1u;
        case 5: return p1 / 100000u - 1u;
        case 6: return p1 / 1000000u - 1u;
        case 7: return p1 / 10000000u - 1u;
        default: return p1 / 100000000u - 1u;
    }
}

// ---- trav4_kernel (traverse4.inc) is instantiated in trav4_*.hip; traverse.hip launches it through these
enum { RH_T4_HASH = 0, RH_T4_BUCKET = 1, RH_T4_GROUPED = 2, RH_T4_LOCAL = 3 };
// table: RH_T4_*; wide: adjacency rows of 17..64 slots; slot: heavy state per resident row (P.slots) instead of per traversal
int rh_trav4_launch(int table, bool wide, bool slot, int lpr, uint32_t grid, hipStream_t st, const TravParams &P);
int rh_trav4_launch_sharded(uint32_t grid, hipStream_t st, const TravParams &P);
int rh_trav4_occupancy(int lpr, int *per_cu);
// per-table launchers (one translation unit each)
int rh_trav4_launch_hash(int lpr, uint32_t grid, hipStream_t st, const TravParams &P);
int rh_trav4_launch_bucket(bool wide, bool slot, int lpr, uint32_t grid, hipStream_t st, const TravParams &P);
int rh_trav4_launch_grouped(bool wide, bool slot, int lpr, uint32_t grid, hipStream_t st, const TravParams &P);
int rh_trav4_launch_local(bool wide, bool slot, int lpr, uint32_t grid, hipStream_t st, const TravParams &P);
uint32_t rh_trav4_staging_capacity();
