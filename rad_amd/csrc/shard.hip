// shard.hip — the row-sharded multi-GPU traversal (BASELINE.json north_star): the
// fingerprint corpus is partitioned over the GPUs of a node by contiguous slot range, ONE layered graph
// (adjacency replicated), and the traversals of a batch are partitioned over the ranks too.  Per frontier
// step every rank
//   1. advances each of ITS traversals to the point where a fingerprint would be read: apply the scores that
//      came back for the last step (scored insert + queue insert, rad/coordination_service.py:379-389), then
//      pop / expand (visited test-and-set, rad/visited.py:17-29) until some neighbour is not in the scored
//      set (rad/distributed_worker.py:296-305) — those slots are the step's frontier candidates;
//   2. all-gathers the candidates of all ranks (RCCL ncclAllGather over xGMI);
//   3. evaluates the candidates whose rows it owns against the query of the traversal that asked
//      (Tanimoto counts, the K2 gather shape);
//   4. returns the scores to the asking ranks (ncclReduceScatter of disjoint contributions: every candidate
//      is owned by exactly one rank, the others add 0).
// The control flow per traversal is exactly the single-GPU one (strict best-first, one pop at a time), so
// the scored lists and pop logs are bit-identical to a single-GPU traversal of the same corpus and graph
// whatever the number of ranks (tests/test_sharded.py on CPU with the oracle's stepper as the local
// engine, tests/test_gpu_sharded.py through these kernels).  What is exchanged per step and rank:
// nq x W candidate slots (4 B each) out, nq x W packed (and | or << 16) scores back.
//
// Three step engines, same results.  "row" (the default since round 3): shard_step_row_kernel, sixteen lanes per
// traversal, a 16-ary heap whose levels are 128-B lines, neighbours probed one per lane, state in HBM.  "thread"
// (RADHIP_SHARD_ENGINE=thread): shard_step_kernel, one thread per traversal, an 8-ary heap and open-addressed sets in HBM
// — the oracle's stepper restated (oracle/rad_oracle.c orc_stepper_step).  "wave" (RADHIP_SHARD_ENGINE=wave; adjacency
// rows of at most 16 slots): trav4_kernel's sharded form (traverse4.inc, SH = true) — the single-GPU kernel with its
// three-level queue and tables, four traversals per wavefront, cut at the fingerprint read and resumed by the next
// launch.  A step is a launch that ends when its slowest traversal has its candidates out, so what counts is the
// worst-case latency of ONE pop, not throughput: the heaps' O(log n) dependent reads are evenly short, the wave
// kernel's register/LDS queue pays a state restore + save per launch and now and then a 256-key sort or a pass over
// its far runs.  Measured on one MI355X, 20M rows, world 1 (profiles/r03/sharded_engines): without speculation the row
// engine takes 75 us per step at 16384 traversals and 119 at 32768, the thread engine 87 and 107 (wave, round 2: 91 /
// 171 at 8192 / 30720); with two speculative queue heads per step (the row engine's default) the row engine needs 0.54 x
// the steps at 1.6 x the time each: 186 / 228 / 256 M expansions/s at 16384 / 32768 / 49152 traversals per rank.
#include "common.h"
#include "comm.h"

struct radhip_traversal;
int rh_trav_create_sharded(radhip_index *idx, const uint8_t *queries, uint32_t nq, uint64_t n_to_score, uint32_t flags,
                           radhip_traversal **out);                                                       // traverse.hip
void rh_trav_bind_shard(radhip_traversal *t, uint32_t *d_req, const uint32_t *d_in, uint32_t W, uint32_t max_inner);
int rh_trav_enqueue_shard_step(radhip_traversal *t);

#include <algorithm>
#include <chrono>
#include <new>
#include <thread>

#define SH_EMPTY64 0ull

struct alignas(128) ShardHeader {   // one 128-B line: a slot's header is one request to load, one to store
    uint64_t n_scored, n_pops, n_nbr, heap_n;
    uint32_t prime_at, n_pend, pend_level;
    int32_t status;      // 0 running, 1 n_to_score reached, 2 queue empty, < 0 error
    uint32_t n_vis;      // entries of the visited set (checked against 3/4 of its size: the probes are unbounded loops)
    uint32_t n_spec;     // speculative candidates out with the last step (req[W .. W + n_spec))
    // the speculative expansions of the last step: queue head s was node spec_node[s] on level spec_level[s]; its
    // unscored neighbours are the spec_cnt[s] candidates from req[W + spec_off[s]] on, in row order
    uint32_t spec_node[2];
    uint8_t spec_level[2], spec_cnt[2], spec_off[2], no_more, pad_;   // no_more: the batch had no traversal left when this slot asked (row engine)
    uint64_t spec_req, spec_hit, spec_used;   // statistics: speculative scores asked for / expansions finished from them / scores used
    // row engine: the traversal this slot works on (+ 1; 0 = none yet) and the epoch its set entries carry — a slot that is
    // done takes the next traversal of the batch (P.next_t) and leaves the last one's entries behind as stale
    uint32_t tid1, epoch;
};
// what a traversal has to show, by traversal number (a slot writes it at the end of every step: the slot moves on)
struct ShardResult { uint64_t n_scored, n_pops, n_nbr; int32_t status; uint32_t n_vis; };

struct ShardParams {
    const uint32_t *adj0, *upper_row, *adjU, *top;
    uint32_t n_top, cap0, capU, nq;
    uint32_t W;                  // widest adjacency row = candidates one expansion can need
    uint32_t Wt;                 // request slots per traversal and step: W needed + spec * W speculative
    uint32_t spec;               // queue heads expanded speculatively per step (0, 1 or 2)
    int32_t start_level;
    uint64_t n_to_score;
    ShardHeader *hdr;
    unsigned long long *heap; uint64_t heap_cap, heap_stride;   // heap_stride keys per traversal (a multiple of 16, >= heap_cap + 16)
    unsigned long long *vis; uint32_t vlog2;      // visited on levels >= 1: ((slot << 4) | level) + 1
    unsigned long long *sc; uint32_t slog2;       // scored set: (slot + 1) | (and | or << 12 | v0 << 24) << 32; v0 = visited on level 0
    uint2 *scored; uint64_t scored_cap;
    uint32_t *req;               // [nq * Wt + 16]: candidate slots of this step, then the live count
    const uint32_t *scores_in;   // [nq * Wt]: and | or << 16 of the previous step's candidates
    uint32_t *poplog_nodes; uint8_t *poplog_levels; uint64_t poplog_cap;
    uint32_t max_inner;          // pops per step at most while nothing needs a score
    uint32_t ns;                 // slots (<= nq): heap / sets / request rows exist per SLOT, scored lists / pop logs / results per traversal
    uint32_t *next_t;            // row engine: traversals taken beyond the first of every slot (the next one is ns + *next_t)
    ShardResult *res;            // [nq]
};
// the live word behind a rank's candidates: live traversals in bits 0..30, bit 31 = a traversal of this rank
// failed on the device (every rank sees it in the all-gather and the loop ends everywhere at the same step)
#define SH_POISON 0x80000000u
// The scored set (rad/scored.py) also carries the level-0 half of the visited set (rad/visited.py:17-29), as the
// single-GPU kernel's table does: a node is visited on level 0 only after it was scored, so presence + one bit answers
// both questions with ONE probe per level-0 neighbour; the separate visited set only holds (node, level >= 1) pairs.
#define SH_V0 (1u << 24)
__device__ __forceinline__ uint32_t sh_pack(uint32_t wire) { return (wire & 0xFFFFu) | ((wire >> 16) << 12); }   // and | or << 16 -> and | or << 12

__device__ __forceinline__ uint64_t sh_h64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}
// true if the key was already present, else inserts it (n_ins counts the inserts: the caller bounds the fill)
__device__ __forceinline__ bool sh_vis_tas(unsigned long long *vis, uint32_t vlog2, uint32_t slot, uint32_t level, uint32_t &n_ins) {
    const unsigned long long k1 = (((unsigned long long)slot << 4) | level) + 1ull;
    const uint64_t mask = (1ull << vlog2) - 1ull;
    uint64_t i = sh_h64(k1) & mask;
    for (;;) {
        const unsigned long long e = vis[i];
        if (e == SH_EMPTY64) { vis[i] = k1; n_ins++; return false; }
        if (e == k1) return true;
        i = (i + 1) & mask;
    }
}
// 8-ary min-heap of u64 keys: a level is one 64-B line, so a pop costs ~log8(n) dependent line reads instead
// of 2 log2(n) (the step kernel is one thread per traversal: dependent reads are what a step costs)
#ifndef SH_D
#define SH_D 8ull
#endif
// The ancestors of the new leaf are known before any key is read: up to SH_ANC of them are loaded together (one round
// trip) and the sift-up runs in registers; deeper heaps (> 8^7 entries at SH_D = 8) finish with the classic loop.
#define SH_ANC 7
__device__ __forceinline__ void sh_heap_push(unsigned long long *h, uint64_t &n, unsigned long long key) {
    uint64_t i = n++;
    uint64_t ai[SH_ANC];
    unsigned long long av[SH_ANC];
    {
        uint64_t j = i;
#pragma unroll
        for (int l = 0; l < SH_ANC; ++l) {
            const bool has = j > 0;
            j = has ? (j - 1) / SH_D : 0;
            ai[l] = has ? j : ~0ull;
        }
#pragma unroll
        for (int l = 0; l < SH_ANC; ++l) av[l] = ai[l] != ~0ull ? h[ai[l]] : 0ull;   // (0 <= every key: stops the sift)
    }
    bool up = true;
#pragma unroll
    for (int l = 0; l < SH_ANC; ++l) {
        if (up) {
            if (ai[l] == ~0ull || av[l] <= key) up = false;
            else { h[i] = av[l]; i = ai[l]; }
        }
    }
    if (up) {   // more than SH_ANC levels
        while (i > 0) {
            const uint64_t p = (i - 1) / SH_D;
            const unsigned long long pk = h[p];
            if (pk <= key) break;
            h[i] = pk;
            i = p;
        }
    }
    h[i] = key;
}
__device__ __forceinline__ unsigned long long sh_heap_pop(unsigned long long *h, uint64_t &n) {
    const unsigned long long top = h[0];
    const unsigned long long last = h[--n];
    uint64_t i = 0;
    for (;;) {
        const uint64_t c0 = SH_D * i + 1;
        if (c0 >= n) break;
        const uint64_t c1 = c0 + SH_D < n ? c0 + SH_D : n;
        unsigned long long ck[SH_D];
#pragma unroll
        for (uint64_t j = 0; j < SH_D; ++j) ck[j] = c0 + j < c1 ? h[c0 + j] : RH_KEY_INF;   // independent loads of one line
        uint64_t m = c0;
        unsigned long long mk = ck[0];
#pragma unroll
        for (uint64_t j = 1; j < SH_D; ++j) if (ck[j] < mk) { mk = ck[j]; m = c0 + j; }
        if (mk >= last) break;
        h[i] = mk;
        i = m;
    }
    if (n) h[i] = last;
    return top;
}

// One frontier step of one traversal per thread.  Speculation (P.spec > 0): a score is a pure function of (query,
// row), so besides the candidates the traversal NEEDS (the unscored neighbours of the expansion it stopped at) a
// step also asks for the unscored neighbours of the next P.spec queue heads.  Nothing of that is committed: the
// scored list, the queue and the visited set only change in strict pop order, exactly as without speculation
// (rad/coordination_service.py:369-395).  When the next pop IS the head that was expanded speculatively (3 times
// out of 4: the scores that just arrived rarely beat it) and every unscored neighbour it has now is among the
// speculative candidates, the expansion finishes at once from their scores instead of costing another step.
__global__ __launch_bounds__(64) void shard_step_kernel(ShardParams P) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= P.nq) return;
    ShardHeader H = P.hdr[q];
    uint32_t *req = P.req + (uint64_t)q * P.Wt;
    uint32_t *live = P.req + (uint64_t)P.ns * P.Wt + P.ns;     // (thread engine: one slot per traversal, ns == nq)
    P.req[(uint64_t)P.ns * P.Wt + q] = q;
    if (H.status != 0) {
        if (H.n_pend || H.n_spec) { for (uint32_t i = 0; i < P.Wt; ++i) req[i] = RADHIP_NO_SLOT; H.n_pend = 0; H.n_spec = 0; P.hdr[q] = H; }
        if (H.status < 0) atomicOr(live, SH_POISON);
        return;
    }
    unsigned long long *heap = P.heap + (uint64_t)q * P.heap_stride;
    unsigned long long *vis = P.vis + ((uint64_t)q << P.vlog2);
    unsigned long long *sc = P.sc + ((uint64_t)q << P.slog2);
    uint2 *scored = P.scored + (uint64_t)q * P.scored_cap;
    const uint32_t *sin = P.scores_in + (uint64_t)q * P.Wt;
    const uint64_t vmask = (1ull << P.vlog2) - 1ull, smask = (1ull << P.slog2) - 1ull;
    const uint32_t vis_limit = (uint32_t)(((1ull << P.vlog2) / 4ull) * 3ull);

    // ---- commit up to 16 scored candidates in order: scored list, scored set, queue (rad/coordination_service.py:
    // 377-389).  The first scored-set bucket of every candidate is loaded together (one round trip); a preloaded
    // empty bucket is trusted only while no earlier candidate of the block was stored there.
    auto commit16 = [&](const uint32_t (&sl)[16], const uint32_t (&sv)[16], uint32_t n, uint32_t level) {
        uint64_t hb[16], ins[16];
        unsigned long long eb[16];
#pragma unroll
        for (uint32_t j = 0; j < 16; ++j) {
            hb[j] = sh_h64((uint64_t)sl[j] + 1ull) & smask;
            ins[j] = ~0ull;
            eb[j] = j < n ? sc[hb[j]] : SH_EMPTY64;
        }
#pragma unroll
        for (uint32_t j = 0; j < 16; ++j) {
            if (j >= n || H.status != 0) continue;
            if (H.n_scored >= P.scored_cap || H.heap_n >= P.heap_cap) { H.status = RADHIP_E_CAPACITY; continue; }
            const uint32_t slot = sl[j], v = sv[j];
            scored[H.n_scored++] = make_uint2(slot, v);
            uint64_t bi = hb[j];
            unsigned long long e = eb[j];
            if (e == SH_EMPTY64) {
                bool taken = false;
#pragma unroll
                for (uint32_t t = 0; t < 16; ++t) taken = taken || (t < j && ins[t] == bi);
                if (taken) e = sc[bi];
            }
            while (e != SH_EMPTY64) { bi = (bi + 1) & smask; e = sc[bi]; }
            sc[bi] = (unsigned long long)(slot + 1u) | ((unsigned long long)(sh_pack(v) | (level == 0u ? SH_V0 : 0u)) << 32);
            ins[j] = bi;
            sh_heap_push(heap, H.heap_n, rh_make_key_dev(rh_q24_dev(v & 0xFFFFu, v >> 16), slot, level));
        }
    };

    // ---- finish: the candidates of the last step are scored now
    if (H.n_pend) {
        for (uint32_t base = 0; base < H.n_pend && H.status == 0; base += 16) {
            uint32_t sl[16], sv[16];
#pragma unroll
            for (uint32_t j = 0; j < 16; ++j) {
                const bool on = base + j < H.n_pend;
                sl[j] = on ? req[base + j] : RADHIP_NO_SLOT;
                sv[j] = on ? sin[base + j] : 0u;
            }
            commit16(sl, sv, H.n_pend - base < 16u ? H.n_pend - base : 16u, H.pend_level);
        }
    }
    H.n_pend = 0;
    const uint32_t n_cache = H.n_spec;     // speculative scores of the last step: valid for this step only
    uint32_t k = 0;
    bool primed_now = false;
    if (H.status == 0 && H.prime_at < P.n_top) {
        // ---- prime (rad/traverser.py:141-170): Wt top-level nodes per step; distinct, nothing scored yet
        while (H.prime_at < P.n_top && k < P.Wt) {
            const uint32_t slot = P.top[H.prime_at++];
            if (P.start_level > 0) (void)sh_vis_tas(vis, P.vlog2, slot, (uint32_t)P.start_level, H.n_vis);   // (level 0: the commit sets v0)
            req[k++] = slot;
        }
        H.pend_level = (uint32_t)P.start_level;
        primed_now = true;
    } else if (H.status == 0) {
        // ---- pop / expand until a neighbour needs a score it does not have (or max_inner pops without one)
        uint32_t hits = 0;
        for (uint32_t it = 0; it < P.max_inner + hits; ++it) {
            if (H.n_scored >= P.n_to_score) { H.status = 1; break; }
            if (H.heap_n == 0) { H.status = 2; break; }
            if (H.n_vis > vis_limit) { H.status = RADHIP_E_CAPACITY; break; }
            const unsigned long long key = sh_heap_pop(heap, H.heap_n);
            uint32_t slot, level;
            rh_decode_key(key, &slot, &level);
            if (P.poplog_nodes && H.n_pops < P.poplog_cap) {
                P.poplog_nodes[(uint64_t)q * P.poplog_cap + H.n_pops] = slot;
                P.poplog_levels[(uint64_t)q * P.poplog_cap + H.n_pops] = (uint8_t)level;
            }
            H.n_pops++;
            const uint32_t cap = level == 0 ? P.cap0 : P.capU;
            const uint32_t *row = level == 0 ? P.adj0 + (uint64_t)slot * P.cap0
                                             : P.adjU + ((uint64_t)P.upper_row[slot] + (level - 1u)) * P.capU;
            // 16 neighbours at a time: their first visited-set and scored-set buckets are loaded together (independent
            // loads, one round trip) before the entries are resolved in row order — a thread's dependent reads are
            // what a step costs.  A preloaded EMPTY visited bucket is only trusted if no earlier neighbour of this row
            // was inserted there; the scored set does not change inside this loop.
            bool row_end = false;
            for (uint32_t base = 0; base < cap && !row_end && H.status == 0; base += 16) {
                uint32_t nbv[16];
                unsigned long long ev[16], es[16];
                uint64_t hv[16], hs[16], ins[16];
                uint32_t cnt16 = 0;
#pragma unroll
                for (uint32_t j = 0; j < 16; ++j) {
                    nbv[j] = (base + j < cap && !row_end) ? row[base + j] : RADHIP_NO_SLOT;
                    if (nbv[j] == RADHIP_NO_SLOT) row_end = true; else cnt16 = j + 1;
                }
#pragma unroll
                for (uint32_t j = 0; j < 16; ++j) {
                    ev[j] = 0ull; es[j] = 0ull; hv[j] = 0; hs[j] = 0; ins[j] = ~0ull;
                    if (j < cnt16) {
                        hs[j] = sh_h64((uint64_t)nbv[j] + 1ull) & smask;
                        es[j] = sc[hs[j]];
                        if (level > 0u) {
                            hv[j] = sh_h64((((unsigned long long)nbv[j] << 4) | level) + 1ull) & vmask;
                            ev[j] = vis[hv[j]];
                        }
                    }
                }
#pragma unroll
                for (uint32_t j = 0; j < 16; ++j) {
                    if (j >= cnt16 || H.status != 0) continue;
                    const uint32_t nb = nbv[j];
                    H.n_nbr++;
                    if (level > 0u) {   // visited test-and-set on an upper level: the small set
                        const unsigned long long k1 = (((unsigned long long)nb << 4) | level) + 1ull;
                        uint64_t i = hv[j];
                        unsigned long long e = ev[j];
                        if (e == SH_EMPTY64) {
                            bool taken = false;
#pragma unroll
                            for (uint32_t t = 0; t < 16; ++t) taken = taken || (t < j && ins[t] == i);
                            if (taken) e = vis[i];
                        }
                        bool seen = false;
                        for (;;) {
                            if (e == SH_EMPTY64) { vis[i] = k1; ins[j] = i; H.n_vis++; break; }
                            if (e == k1) { seen = true; break; }
                            i = (i + 1) & vmask;
                            e = vis[i];
                        }
                        if (seen) continue;
                    }
                    uint64_t si = hs[j];
                    unsigned long long se = es[j];
                    bool found = false;
                    uint32_t v = 0;
                    for (;;) {
                        if (se == SH_EMPTY64) break;
                        if ((uint32_t)se == nb + 1u) { found = true; v = (uint32_t)(se >> 32); break; }
                        si = (si + 1) & smask;
                        se = sc[si];
                    }
                    if (found) {
                        if (level == 0u) {   // scored before: visited on level 0 iff its v0 bit is set
                            if (v & SH_V0) continue;
                            sc[si] = se | ((unsigned long long)SH_V0 << 32);
                        }
                        if (H.heap_n >= P.heap_cap) { H.status = RADHIP_E_CAPACITY; continue; }
                        sh_heap_push(heap, H.heap_n, rh_make_key_dev(rh_q24_dev(v & 0xFFFu, (v >> 12) & 0xFFFu), nb, level));
                    } else req[k++] = nb;   // new: scored at the next step (the commit marks it visited on level 0)
                }
            }
            // ---- every unscored neighbour among the speculative candidates of the last step?  Then their scores are
            // here already: commit them now, in row order, and go on popping.  (The new nodes of this expansion are a
            // subsequence of what the speculative expansion of the same (node, level) found: nothing gets unscored.)
            if (k && H.status == 0 && n_cache && k <= 16u) {
                int seg = -1;
                if (H.spec_cnt[0] && H.spec_node[0] == slot && H.spec_level[0] == level) seg = 0;
                else if (H.spec_cnt[1] && H.spec_node[1] == slot && H.spec_level[1] == level) seg = 1;
                if (seg >= 0) {
                    const uint32_t off = P.W + H.spec_off[seg], cn = H.spec_cnt[seg];
                    uint32_t sl[16], sv[16];
                    uint32_t j = 0, got = 0;
                    for (uint32_t i = 0; i < 16u; ++i) { sl[i] = RADHIP_NO_SLOT; sv[i] = 0u; }
                    for (uint32_t i = 0; i < 16u; ++i) {
                        if (i >= k) break;
                        const uint32_t want = req[i];
                        while (j < cn && req[off + j] != want) ++j;
                        if (j >= cn) break;
                        sl[i] = want; sv[i] = sin[off + j];
                        ++j; ++got;
                    }
                    if (got == k) {
                        commit16(sl, sv, k, level);
                        H.spec_hit++; H.spec_used += k;
                        k = 0;
                        if (hits < P.spec) hits++;
                    }
                }
            }
            H.pend_level = level;
            if (H.status == 0 && level > 0) {
                const uint32_t nl = level - 1u;
                bool fresh;
                if (nl > 0u) fresh = !sh_vis_tas(vis, P.vlog2, slot, nl, H.n_vis);
                else {   // the node is scored, hence in the scored set: visited(node, 0) is its v0 bit
                    fresh = false;
                    uint64_t si = sh_h64((uint64_t)slot + 1ull) & smask;
                    for (uint64_t tries = 0; tries <= smask; ++tries) {
                        const unsigned long long se = sc[si];
                        if (se == SH_EMPTY64) break;   // (unreachable by construction)
                        if ((uint32_t)se == slot + 1u) {
                            if (!((uint32_t)(se >> 32) & SH_V0)) { sc[si] = se | ((unsigned long long)SH_V0 << 32); fresh = true; }
                            break;
                        }
                        si = (si + 1) & smask;
                    }
                }
                if (fresh) {
                    if (H.heap_n >= P.heap_cap) H.status = RADHIP_E_CAPACITY;
                    else sh_heap_push(heap, H.heap_n, rh_make_key_dev((uint32_t)(key >> 38), slot, nl));
                }
            }
            if (k || H.status != 0) break;
        }
    }
    for (uint32_t i = k; i < P.W && i < P.Wt; ++i) req[i] = RADHIP_NO_SLOT;
    H.n_pend = k;
    // ---- speculate: the unscored neighbours of the next queue heads (the head itself, then the smaller of its
    // children in the 8-ary heap: the runner-up is one of them).  Read-only: no visited mark, nothing inserted.
    uint32_t ns = 0;
    H.spec_cnt[0] = H.spec_cnt[1] = 0;
    if (P.spec && H.status == 0 && !primed_now && H.prime_at >= P.n_top && H.heap_n > 0) {
        for (uint32_t s = 0; s < P.spec && s < 2u; ++s) {
            unsigned long long hk = RH_KEY_INF;
            if (s == 0) hk = heap[0];
            else {
                const uint64_t c1 = H.heap_n < 1 + SH_D ? H.heap_n : 1 + SH_D;
                unsigned long long ck[SH_D];
#pragma unroll
                for (uint64_t j = 0; j < SH_D; ++j) ck[j] = 1 + j < c1 ? heap[1 + j] : RH_KEY_INF;
#pragma unroll
                for (uint64_t j = 0; j < SH_D; ++j) hk = ck[j] < hk ? ck[j] : hk;
            }
            if (hk == RH_KEY_INF) break;
            uint32_t slot, level;
            rh_decode_key(hk, &slot, &level);
            const uint32_t cap = level == 0 ? P.cap0 : P.capU;
            const uint32_t *row = level == 0 ? P.adj0 + (uint64_t)slot * P.cap0
                                             : P.adjU + ((uint64_t)P.upper_row[slot] + (level - 1u)) * P.capU;
            const uint32_t off0 = ns;
            bool row_end = false;
            for (uint32_t base = 0; base < cap && !row_end; base += 16) {
                uint32_t nbv[16];
                unsigned long long es[16];
                uint64_t hs[16];
                uint32_t cnt16 = 0;
#pragma unroll
                for (uint32_t j = 0; j < 16; ++j) {
                    nbv[j] = (base + j < cap && !row_end) ? row[base + j] : RADHIP_NO_SLOT;
                    if (nbv[j] == RADHIP_NO_SLOT) row_end = true; else cnt16 = j + 1;
                }
#pragma unroll
                for (uint32_t j = 0; j < 16; ++j) {
                    es[j] = 0ull; hs[j] = 0;
                    if (j < cnt16) { hs[j] = sh_h64((uint64_t)nbv[j] + 1ull) & smask; es[j] = sc[hs[j]]; }
                }
#pragma unroll
                for (uint32_t j = 0; j < 16; ++j) {
                    if (j >= cnt16) continue;
                    const uint32_t nb = nbv[j];
                    uint64_t si = hs[j];
                    unsigned long long se = es[j];
                    bool found = false;
                    for (;;) {
                        if (se == SH_EMPTY64) break;
                        if ((uint32_t)se == nb + 1u) { found = true; break; }
                        si = (si + 1) & smask;
                        se = sc[si];
                    }
                    if (!found && P.W + ns < P.Wt) req[P.W + ns++] = nb;
                }
            }
            H.spec_node[s] = slot; H.spec_level[s] = (uint8_t)level; H.spec_off[s] = (uint8_t)off0;
            H.spec_cnt[s] = (uint8_t)(ns - off0 > 255u ? 255u : ns - off0);
        }
        H.spec_req += ns;
    }
    if (!primed_now) for (uint32_t i = P.W + ns; i < P.Wt; ++i) req[i] = RADHIP_NO_SLOT;
    else for (uint32_t i = (k > P.W ? k : P.W); i < P.Wt; ++i) req[i] = RADHIP_NO_SLOT;
    H.n_spec = ns;
    P.hdr[q] = H;
    { ShardResult R; R.n_scored = H.n_scored; R.n_pops = H.n_pops; R.n_nbr = H.n_nbr; R.status = H.status; R.n_vis = H.n_vis; P.res[q] = R; }
    if (H.status == 0) atomicAdd(live, 1u);   // live traversals of this rank
    else if (H.status < 0) atomicOr(live, SH_POISON);
}

// ---- the "row" engine: the same step, sixteen lanes per traversal -------------------------------------------------
// One thread per traversal makes every load instruction of a wavefront touch 64 traversals' state — 64 lines in 64
// different pages — and walks a row's neighbours and a heap level's children one instruction at a time.  Here a
// traversal is a ROW of 16 lanes (4 rows per wavefront, as in trav4_kernel): a 16-ary heap whose level is one 128-B line
// read by the row in one request (a pop costs ~log16 n dependent reads, a push ONE: all ancestors are known in advance and
// are loaded together), the neighbours of an adjacency row probed one per lane, set inserts by compare-and-swap (keys of
// one expansion are distinct, so the sets end up with the same members whatever lane wins a bucket), candidates ranked by
// ballot so that they leave in row order.  All state stays in HBM between steps, as in the thread engine — nothing to
// save or restore — and the results are the same bit for bit: the pop order is a function of the queue's key SET only.
// Heap node j lives at h[j + 15]: the 16 children of a node start at a multiple of 16 keys.
#define SHR_AT(j) ((j) + 15ull)
__device__ __forceinline__ unsigned long long shr_ld(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void shr_st(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t shr_ld32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long shr_row_min(unsigned long long v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(v, o, 16); v = t < v ? t : v; }
    return v;
}
__device__ __forceinline__ uint32_t shr_ballot(bool p, uint32_t gshift) { return (uint32_t)(__ballot(p) >> gshift) & 0xFFFFu; }
// every lane of the row passes the same key
__device__ __forceinline__ void shr_push(unsigned long long *h, uint64_t &n, unsigned long long key, uint32_t gl, uint32_t gshift) {
    const uint64_t i = n++;
    // lane l < 6 owns ancestor l of the new leaf (16^6 > any queue this engine is created for)
    uint64_t a = i, my = ~0ull;
    bool valid = true;
#pragma unroll
    for (uint32_t t = 0; t < 6; ++t) {
        valid = valid && a > 0;
        a = valid ? (a - 1) >> 4 : 0;
        if (t == gl) my = valid ? a : ~0ull;
    }
    // parent and grandparent first: a new key rarely rises further (the loop is request-bound: 6 lines per push were a
    // third of a step's memory requests), the other four ancestors only when it does
    unsigned long long av = (gl < 2u && my != ~0ull) ? shr_ld(&h[SHR_AT(my)]) : 0ull;   // (0 <= every key: ends the sift)
    uint32_t b = shr_ballot(gl < 2u && my != ~0ull && av > key, gshift);
    if (b == 3u) {
        if (gl >= 2u && my != ~0ull) av = shr_ld(&h[SHR_AT(my)]);
        b = shr_ballot(my != ~0ull && av > key, gshift);
    }
    const uint32_t u = (uint32_t)__builtin_ctz(~b);                               // ancestors that move down one level
    const uint64_t below = __shfl_up(my, 1, 16);
    if (gl < u) shr_st(&h[SHR_AT(gl == 0 ? i : below)], av);
    if (gl == (u ? u - 1u : 0u)) shr_st(&h[SHR_AT(u ? my : i)], key);
}
// The keys of lanes 0 .. n-1 (distinct) enter the heap — one round trip per group of new leaves that share a parent (all of
// them three times out of four) instead of one per key.  The path from the root to that parent (<= 6 keys, loaded together)
// and the new keys are ranked against each other in registers: the d smallest go onto the path, top down, the others into
// the leaves.  A path node only ever receives a key that is not larger than the one it held (the (l+1)-th smallest of the
// union cannot exceed the (l+1)-th path key), so its other children stay below it; any valid heap pops the same order.
__device__ __forceinline__ void shr_push_bulk(unsigned long long *h, uint64_t &hn, unsigned long long key, uint32_t n, uint32_t gl, uint32_t gshift) {
    uint32_t done = 0;
    while (done < n) {
        const uint64_t a = hn;                                   // the first free leaf
        if (a == 0) {                                            // empty heap: the first key is the root
            const unsigned long long k0 = __shfl(key, (int)done, 16);
            if (gl == 0) shr_st(&h[SHR_AT(0)], k0);
            hn = 1; done++;
            continue;
        }
        const uint64_t p = (a - 1) >> 4;
        const uint64_t room = 16ull * p + 17ull - a;             // leaves a .. 16p + 16 are children of p
        const uint32_t m = (uint64_t)(n - done) < room ? n - done : (uint32_t)room;
        // lane l < 6 owns ancestor l of leaf a (l = 0: p)
        uint64_t an = a, my = ~0ull;
        bool valid = true;
#pragma unroll
        for (uint32_t t = 0; t < 6; ++t) {
            valid = valid && an > 0;
            an = valid ? (an - 1) >> 4 : 0;
            if (t == gl) my = valid ? an : ~0ull;
        }
        const unsigned long long cv = my != ~0ull ? shr_ld(&h[SHR_AT(my)]) : RH_KEY_INF;
        const uint32_t d = (uint32_t)__popc(shr_ballot(my != ~0ull, gshift));
        const unsigned long long nk_ = __shfl(key, (int)((done + gl) & 15u), 16);                // (every lane of the row shuffles)
        const unsigned long long nk = gl < m ? nk_ : RH_KEY_INF;
        uint32_t rn = 0, rc = 0;                                 // rank of this lane's new key / path key in the union
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const unsigned long long kt = __shfl(nk, t, 16);
            rn += kt < nk ? 1u : 0u; rc += kt < cv ? 1u : 0u;    // (absent keys are INF: they rank behind everything)
        }
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const unsigned long long ct = __shfl(cv, t, 16);
            rn += ct < nk ? 1u : 0u; rc += ct < cv ? 1u : 0u;
        }
        // rank r < d: path position r from the top = ancestor d - 1 - r; r >= d: leaf a + (r - d)
        const uint64_t pn = __shfl(my, (int)((d - 1u - rn) & 15u), 16), pc = __shfl(my, (int)((d - 1u - rc) & 15u), 16);
        if (gl < m) shr_st(&h[SHR_AT(rn < d ? pn : a + (rn - d))], nk);
        if (gl < d && rc != d - 1u - gl) shr_st(&h[SHR_AT(rc < d ? pc : a + (rc - d))], cv);   // (a path key that keeps its place is not rewritten)
        hn += m; done += m;
    }
}
__device__ __forceinline__ unsigned long long shr_pop(unsigned long long *h, uint64_t &n, uint32_t gl, uint32_t gshift) {
    const unsigned long long top = shr_ld(&h[SHR_AT(0)]);
    const unsigned long long last = shr_ld(&h[SHR_AT(n - 1)]);
    --n;
    if (n == 0) return top;
    uint64_t i = 0;
    for (;;) {
        const uint64_t c0 = 16ull * i + 1ull;
        if (c0 >= n) break;
        const unsigned long long ck = c0 + gl < n ? shr_ld(&h[SHR_AT(c0 + gl)]) : RH_KEY_INF;   // one line
        const unsigned long long mk = shr_row_min(ck);
        if (mk >= last) break;
        const uint32_t m = (uint32_t)__builtin_ctz(shr_ballot(ck == mk, gshift));
        if (gl == 0) shr_st(&h[SHR_AT(i)], mk);
        i = c0 + m;
    }
    if (gl == 0) shr_st(&h[SHR_AT(i)], last);
    return top;
}
// Set entries of the row engine carry the slot's epoch (scored set: bits 57..63, visited set: bits 40..46): what an earlier
// traversal of the slot left behind reads as empty.
#define SHR_SC_EPOCH_SHIFT 57
#define SHR_EPOCHS_PER_SLOT 127u   // traversals one slot can take between two clears of its sets (reset()): epochs 1..127
#define SHR_VIS_EPOCH_SHIFT 40
__device__ __forceinline__ bool shr_sc_free(unsigned long long e, uint32_t epoch) { return e == SH_EMPTY64 || (uint32_t)(e >> SHR_SC_EPOCH_SHIFT) != epoch; }
__device__ __forceinline__ bool shr_vis_free(unsigned long long e, uint32_t epoch) { return e == SH_EMPTY64 || (uint32_t)(e >> SHR_VIS_EPOCH_SHIFT) != epoch; }
// true if (slot, level) was already in the set, else inserts it: ONE lane per key, distinct keys per call site
__device__ __forceinline__ bool shr_vis_tas(unsigned long long *vis, uint64_t vmask, uint32_t slot, uint32_t level, uint32_t epoch) {
    const unsigned long long body = (((unsigned long long)slot << 4) | level) + 1ull;
    const unsigned long long k1 = body | ((unsigned long long)epoch << SHR_VIS_EPOCH_SHIFT);
    uint64_t i = sh_h64(body) & vmask;
    unsigned long long expect = SH_EMPTY64;             // (one round trip where the bucket was never used: the common case)
    for (;;) {
        const unsigned long long cur = atomicCAS(&vis[i], expect, k1);
        if (cur == expect) return false;                // inserted
        if (cur == k1) return true;
        if (shr_vis_free(cur, epoch)) { expect = cur; continue; }   // a stale entry (or another lane's fresh one: looked at again)
        expect = SH_EMPTY64;
        i = (i + 1) & vmask;
    }
}

struct ShardRowLds { uint32_t cand[4][16]; uint32_t sp_req[4][32]; uint32_t sp_sc[4][32]; };

__global__ __launch_bounds__(64) void shard_step_row_kernel(ShardParams P) {
    __shared__ ShardRowLds L;
    const uint32_t lane = threadIdx.x, g = lane >> 4, gl = lane & 15u, gshift = g * 16u;
    const uint32_t lt = (1u << gl) - 1u;
    const uint32_t q = blockIdx.x * 4u + g;   // the SLOT: heap, sets, request row
    if (q >= P.ns) return;                    // (whole rows leave: everything below is row-uniform control flow)
    ShardHeader H = P.hdr[q];
    uint32_t *req = P.req + (uint64_t)q * P.Wt;
    uint32_t *tidw = P.req + (uint64_t)P.ns * P.Wt + q;         // which traversal's query the owners score this slot's candidates against
    uint32_t *live = P.req + (uint64_t)P.ns * P.Wt + P.ns;
    // (the counter is asked, and its answer spread over the row, by every lane that is still here — outside the branch of the
    // rows that want a traversal: traverse4.inc `take` found out why)
    const bool wants = H.tid1 != 0u && H.status > 0 && H.epoch < SHR_EPOCHS_PER_SLOT && !H.no_more;   // (a failed traversal keeps its slot: the error stays visible)
    uint32_t t_next = 0xFFFFFFFFu;
    if (wants && gl == 0) t_next = P.ns + atomicAdd(P.next_t, 1u);
    t_next = (uint32_t)__shfl((int)t_next, 0, 16);
    if (H.tid1 == 0u || H.status != 0) {
        // nothing to work on: the slot's traversal is done (its results are in P.res), or this is the first step
        if (H.n_pend || H.n_spec) {
            for (uint32_t i = gl; i < P.Wt; i += 16u) req[i] = RADHIP_NO_SLOT;
            H.n_pend = 0; H.n_spec = 0;
        }
        uint32_t t = 0xFFFFFFFFu;
        if (H.tid1 == 0u) t = q;                 // the first traversal of slot s is traversal s (what the host-staged exchange assumes)
        else if (wants) t = t_next;
        if (t >= P.nq) {
            // (asked once: thousands of finished slots adding to one counter at every step cost more than the step)
            const bool first_no = !H.no_more;
            H.no_more = 1;
            if (gl == 0) { if (first_no || H.status < 0) P.hdr[q] = H; if (H.status < 0) atomicOr(live, SH_POISON); }
            return;
        }
        // take traversal t: an empty queue, sets whose old entries are stale, the entry points still to come
        H.n_scored = 0; H.n_pops = 0; H.n_nbr = 0; H.heap_n = 0; H.prime_at = 0; H.pend_level = 0; H.status = 0; H.n_vis = 0;
        H.spec_cnt[0] = H.spec_cnt[1] = 0;
        H.tid1 = t + 1u; H.epoch += 1u; H.no_more = 0;
        if (gl == 0) *tidw = t;
    }
    const uint32_t tid = H.tid1 - 1u, epoch = H.epoch;
    unsigned long long *heap = P.heap + (uint64_t)q * P.heap_stride;
    unsigned long long *vis = P.vis + ((uint64_t)q << P.vlog2);
    unsigned long long *sc = P.sc + ((uint64_t)q << P.slog2);
    uint2 *scored = P.scored + (uint64_t)tid * P.scored_cap;
    const uint32_t *sin = P.scores_in + (uint64_t)q * P.Wt;
    const uint64_t vmask = (1ull << P.vlog2) - 1ull, smask = (1ull << P.slog2) - 1ull;
    const uint32_t vis_limit = (uint32_t)(((1ull << P.vlog2) / 4ull) * 3ull);

    // ---- commit n <= 16 scored candidates (lane j holds candidate j), in order: scored list, scored set, queue
    auto commit_row = [&](uint32_t slot, uint32_t v, uint32_t n, uint32_t level) {
        if (H.n_scored + n > P.scored_cap || H.heap_n + n > P.heap_cap) { H.status = RADHIP_E_CAPACITY; return; }
        const bool on = gl < n;
        if (on) {
            scored[H.n_scored + gl] = make_uint2(slot, v);
            const unsigned long long e = (unsigned long long)(slot + 1u) | ((unsigned long long)(sh_pack(v) | (level == 0u ? SH_V0 : 0u)) << 32) |
                                         ((unsigned long long)epoch << SHR_SC_EPOCH_SHIFT);
            uint64_t bi = sh_h64((uint64_t)slot + 1ull) & smask;
            unsigned long long expect = SH_EMPTY64;     // (one round trip where the bucket was never used: the common case)
            for (;;) {
                const unsigned long long cur = atomicCAS(&sc[bi], expect, e);
                if (cur == expect) break;
                if (shr_sc_free(cur, epoch)) { expect = cur; continue; }
                expect = SH_EMPTY64;
                bi = (bi + 1) & smask;
            }
        }
        H.n_scored += n;
        const unsigned long long key = rh_make_key_dev(rh_q24_dev(v & 0xFFFFu, v >> 16), slot, level);
        shr_push_bulk(heap, H.heap_n, key, n, gl, gshift);
    };

    // ---- finish: the candidates of the last step are scored now
    for (uint32_t base = 0; base < H.n_pend && H.status == 0; base += 16u) {
        const bool on = base + gl < H.n_pend;
        const uint32_t sl = on ? req[base + gl] : RADHIP_NO_SLOT, sv = on ? sin[base + gl] : 0u;
        commit_row(sl, sv, H.n_pend - base < 16u ? H.n_pend - base : 16u, H.pend_level);
    }
    H.n_pend = 0;
    const uint32_t n_cache = H.n_spec;
    uint32_t k = 0;
    bool primed_now = false;
    if (H.status == 0 && H.prime_at < P.n_top) {
        // ---- prime (rad/traverser.py:141-170): Wt top-level nodes per step
        const uint32_t cnt = P.n_top - H.prime_at < P.Wt ? P.n_top - H.prime_at : P.Wt;
        for (uint32_t i = gl; i < ((cnt + 15u) & ~15u); i += 16u) {
            bool ins = false;
            if (i < cnt) {
                const uint32_t slot = P.top[H.prime_at + i];
                if (P.start_level > 0) ins = !shr_vis_tas(vis, vmask, slot, (uint32_t)P.start_level, epoch);
                req[i] = slot;
            }
            H.n_vis += (uint32_t)__popc(shr_ballot(ins, gshift));
        }
        H.prime_at += cnt; k = cnt;
        H.pend_level = (uint32_t)P.start_level;
        primed_now = true;
    } else if (H.status == 0) {
        uint32_t hits = 0;
        for (uint32_t it = 0; it < P.max_inner + hits; ++it) {
            if (H.n_scored >= P.n_to_score) { H.status = 1; break; }
            if (H.heap_n == 0) { H.status = 2; break; }
            if (H.n_vis > vis_limit) { H.status = RADHIP_E_CAPACITY; break; }
            const unsigned long long key = shr_pop(heap, H.heap_n, gl, gshift);
            uint32_t slot, level;
            rh_decode_key(key, &slot, &level);
            if (P.poplog_nodes && H.n_pops < P.poplog_cap && gl == 0) {
                P.poplog_nodes[(uint64_t)tid * P.poplog_cap + H.n_pops] = slot;
                P.poplog_levels[(uint64_t)tid * P.poplog_cap + H.n_pops] = (uint8_t)level;
            }
            H.n_pops++;
            const uint32_t cap = level == 0 ? P.cap0 : P.capU;
            const uint32_t *row = level == 0 ? P.adj0 + (uint64_t)slot * P.cap0
                                             : P.adjU + ((uint64_t)P.upper_row[slot] + (level - 1u)) * P.capU;
            bool row_end = false;
            for (uint32_t base = 0; base < cap && !row_end && H.status == 0; base += 16u) {
                const uint32_t nb = base + gl < cap ? row[base + gl] : RADHIP_NO_SLOT;
                const uint32_t endm = shr_ballot(nb == RADHIP_NO_SLOT, gshift);
                const uint32_t cnt16 = endm ? (uint32_t)__builtin_ctz(endm) : 16u;   // the row ends at its first empty slot
                if (endm) row_end = true;
                H.n_nbr += cnt16;
                bool go = gl < cnt16;
                if (level > 0u) {            // visited test-and-set on an upper level: the small set
                    bool ins = false;
                    if (go) { const bool seen = shr_vis_tas(vis, vmask, nb, level, epoch); ins = !seen; go = !seen; }
                    H.n_vis += (uint32_t)__popc(shr_ballot(ins, gshift));
                }
                bool found = false;
                uint32_t v = 0;
                uint64_t si = 0;
                unsigned long long se = 0ull;
                if (go) {
                    si = sh_h64((uint64_t)nb + 1ull) & smask;
                    for (;;) {
                        se = shr_ld(&sc[si]);
                        if (shr_sc_free(se, epoch)) break;
                        if ((uint32_t)se == nb + 1u) { found = true; v = (uint32_t)(se >> 32); break; }
                        si = (si + 1) & smask;
                    }
                }
                const bool isnew = go && !found;
                bool old_push = go && found;
                if (old_push && level == 0u) {   // scored before: visited on level 0 iff its v0 bit is set
                    if (v & SH_V0) old_push = false;
                    else shr_st(&sc[si], se | ((unsigned long long)SH_V0 << 32));
                }
                const uint32_t nm = shr_ballot(isnew, gshift);
                if (isnew) {
                    const uint32_t at = k + (uint32_t)__popc(nm & lt);
                    req[at] = nb;
                    if (at < 16u) L.cand[g][at] = nb;
                }
                k += (uint32_t)__popc(nm);
                uint32_t om = shr_ballot(old_push, gshift);
                if (om) {
                    if (H.heap_n + (uint32_t)__popc(om) > P.heap_cap) { H.status = RADHIP_E_CAPACITY; om = 0u; }
                    const unsigned long long okey = rh_make_key_dev(rh_q24_dev(v & 0xFFFu, (v >> 12) & 0xFFFu), nb, level);
                    while (om) {
                        const uint32_t j = (uint32_t)__builtin_ctz(om);
                        om &= om - 1u;
                        shr_push(heap, H.heap_n, __shfl(okey, (int)j, 16), gl, gshift);
                    }
                }
            }
            // ---- every unscored neighbour among the speculative candidates of the last step: commit from their scores
            if (k && H.status == 0 && n_cache && k <= 16u) {
                int seg = -1;
                if (H.spec_cnt[0] && H.spec_node[0] == slot && H.spec_level[0] == level) seg = 0;
                else if (H.spec_cnt[1] && H.spec_node[1] == slot && H.spec_level[1] == level) seg = 1;
                if (seg >= 0 && H.spec_cnt[seg] <= 32u) {
                    const uint32_t off = P.W + H.spec_off[seg], cn = H.spec_cnt[seg];
                    for (uint32_t j = gl; j < cn; j += 16u) { L.sp_req[g][j] = req[off + j]; L.sp_sc[g][j] = sin[off + j]; }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const uint32_t want = gl < k ? L.cand[g][gl] : RADHIP_NO_SLOT;
                    uint32_t sv = 0u;
                    bool have = false;
                    if (gl < k) for (uint32_t j = 0; j < cn; ++j) if (L.sp_req[g][j] == want) { sv = L.sp_sc[g][j]; have = true; }
                    const uint32_t got = (uint32_t)__popc(shr_ballot(have, gshift));
                    __builtin_amdgcn_wave_barrier();
                    if (got == k) {
                        commit_row(want, sv, k, level);
                        H.spec_hit++; H.spec_used += k;
                        k = 0;
                        if (hits < P.spec) hits++;
                    }
                }
            }
            H.pend_level = level;
            if (H.status == 0 && level > 0) {
                const uint32_t nl = level - 1u;
                uint32_t fresh = 0u;
                if (gl == 0) {
                    if (nl > 0u) fresh = shr_vis_tas(vis, vmask, slot, nl, epoch) ? 0u : 1u;
                    else {   // the node is scored, hence in the scored set: visited(node, 0) is its v0 bit
                        uint64_t si = sh_h64((uint64_t)slot + 1ull) & smask;
                        for (uint64_t tries = 0; tries <= smask; ++tries) {
                            const unsigned long long se = shr_ld(&sc[si]);
                            if (shr_sc_free(se, epoch)) break;   // (unreachable by construction)
                            if ((uint32_t)se == slot + 1u) {
                                if (!((uint32_t)(se >> 32) & SH_V0)) { shr_st(&sc[si], se | ((unsigned long long)SH_V0 << 32)); fresh = 1u; }
                                break;
                            }
                            si = (si + 1) & smask;
                        }
                    }
                }
                fresh = (uint32_t)__shfl((int)fresh, 0, 16);
                if (fresh) {
                    if (nl > 0u) H.n_vis++;
                    if (H.heap_n >= P.heap_cap) H.status = RADHIP_E_CAPACITY;
                    else shr_push(heap, H.heap_n, rh_make_key_dev((uint32_t)(key >> 38), slot, nl), gl, gshift);
                }
            }
            if (k || H.status != 0) break;
        }
    }
    for (uint32_t i = k + gl; i < P.W && i < P.Wt; i += 16u) req[i] = RADHIP_NO_SLOT;
    H.n_pend = k;
    // ---- speculate: the unscored neighbours of the next queue heads (the head, then the smallest of its children:
    // the runner-up is one of them).  Read-only.
    uint32_t ns = 0;
    H.spec_cnt[0] = H.spec_cnt[1] = 0;
    if (P.spec && H.status == 0 && !primed_now && H.prime_at >= P.n_top && H.heap_n > 0) {
        for (uint32_t s = 0; s < P.spec && s < 2u; ++s) {
            unsigned long long hk;
            if (s == 0) hk = shr_ld(&heap[SHR_AT(0)]);
            else hk = shr_row_min(1ull + gl < H.heap_n ? shr_ld(&heap[SHR_AT(1ull + gl)]) : RH_KEY_INF);
            if (hk == RH_KEY_INF) break;
            uint32_t slot, level;
            rh_decode_key(hk, &slot, &level);
            const uint32_t cap = level == 0 ? P.cap0 : P.capU;
            const uint32_t *row = level == 0 ? P.adj0 + (uint64_t)slot * P.cap0
                                             : P.adjU + ((uint64_t)P.upper_row[slot] + (level - 1u)) * P.capU;
            const uint32_t off0 = ns;
            bool row_end = false;
            for (uint32_t base = 0; base < cap && !row_end; base += 16u) {
                const uint32_t nb = base + gl < cap ? row[base + gl] : RADHIP_NO_SLOT;
                const uint32_t endm = shr_ballot(nb == RADHIP_NO_SLOT, gshift);
                const uint32_t cnt16 = endm ? (uint32_t)__builtin_ctz(endm) : 16u;
                if (endm) row_end = true;
                bool unscored = false;
                if (gl < cnt16) {
                    uint64_t si = sh_h64((uint64_t)nb + 1ull) & smask;
                    unscored = true;
                    for (;;) {
                        const unsigned long long se = shr_ld(&sc[si]);
                        if (shr_sc_free(se, epoch)) break;
                        if ((uint32_t)se == nb + 1u) { unscored = false; break; }
                        si = (si + 1) & smask;
                    }
                }
                const uint32_t um = shr_ballot(unscored, gshift);
                const uint32_t at = P.W + ns + (uint32_t)__popc(um & lt);
                if (unscored && at < P.Wt) req[at] = nb;
                const uint32_t room = P.Wt - (P.W + ns);
                ns += (uint32_t)__popc(um) < room ? (uint32_t)__popc(um) : room;
            }
            H.spec_node[s] = slot; H.spec_level[s] = (uint8_t)level; H.spec_off[s] = (uint8_t)off0;
            H.spec_cnt[s] = (uint8_t)(ns - off0 > 255u ? 255u : ns - off0);
        }
        H.spec_req += ns;
    }
    if (!primed_now) { for (uint32_t i = P.W + ns + gl; i < P.Wt; i += 16u) req[i] = RADHIP_NO_SLOT; }
    else { for (uint32_t i = (k > P.W ? k : P.W) + gl; i < P.Wt; i += 16u) req[i] = RADHIP_NO_SLOT; }
    H.n_spec = ns;
    if (gl == 0) {
        P.hdr[q] = H;
        ShardResult R; R.n_scored = H.n_scored; R.n_pops = H.n_pops; R.n_nbr = H.n_nbr; R.status = H.status; R.n_vis = H.n_vis;
        P.res[tid] = R;
        // live = traversals of this rank still to finish: this one if it runs on, or (if it is done) the slot will try to take
        // another one at the next step — it counts as live while the batch has any left
        if (H.status == 0 || (H.status > 0 && P.ns + shr_ld32(P.next_t) < P.nq)) atomicAdd(live, 1u);
        else if (H.status < 0) atomicOr(live, SH_POISON);
    }
}

// ---- candidates of every rank x the rows this rank owns: LPR lanes per candidate, U in flight per lane
struct EvalParams {
    const uint4 *fp;             // rows [first, first + count) of the corpus
    uint64_t first, count;
    const uint4 *queries;        // [world * nq] query rows (padded to the row stride)
    const uint32_t *qpop;        // [world * nq]
    const uint32_t *req_all;     // [world][ns * W candidates | ns traversal numbers | 16 (live word first)]
    uint32_t *out;               // [world][ns * W]
    uint32_t *live;              // this rank's live count (behind its own block): zeroed here for the next step
    uint32_t world, nq, W;       // nq = traversals per rank (rows of `queries` per rank)
    uint32_t ns;                 // slots per rank
    uint32_t identity;           // 1: slot s works on traversal s (host-staged exchange: the traversal numbers do not travel)
};

template <int LPR>
__global__ __launch_bounds__(256) void shard_eval_kernel(EvalParams P) {
    constexpr int GPW = 64 / LPR, U = 4;
    const uint32_t lane = threadIdx.x & 63u, chunk = lane % LPR, grp = lane / LPR;
    const uint64_t per_rank = (uint64_t)P.ns * P.W, total = per_rank * P.world, blk = per_rank + P.ns + 16u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    if (blockIdx.x == 0 && threadIdx.x == 0) *P.live = 0u;   // (the all-gather before this kernel has taken it)
    for (uint64_t base = wave * (GPW * U); base < total; base += n_waves * (GPW * U)) {
        uint32_t sl[U], tr[U];
        uint64_t at[U];
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            at[u] = base + (uint64_t)u * GPW + grp;
            sl[u] = RADHIP_NO_SLOT; tr[u] = 0u;
            if (at[u] < total) {
                const uint32_t r = (uint32_t)(at[u] / per_rank), off = (uint32_t)(at[u] - (uint64_t)r * per_rank), sslot = off / P.W;
                sl[u] = P.req_all[(uint64_t)r * blk + off];
                // rank * nq + the traversal the asking slot works on (it travels behind the candidates; loaded beside them)
                tr[u] = r * P.nq + (P.identity ? sslot : P.req_all[(uint64_t)r * blk + per_rank + sslot]);
            }
            const bool mine = sl[u] != RADHIP_NO_SLOT && sl[u] >= P.first && sl[u] < P.first + P.count;
            v[u] = make_uint4(0, 0, 0, 0);
            if (mine) v[u] = P.fp[((uint64_t)sl[u] - P.first) * LPR + chunk];
            else sl[u] = RADHIP_NO_SLOT;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t res = 0u;
            if (sl[u] != RADHIP_NO_SLOT) {
                const uint64_t tq = tr[u];
                const uint4 qv = P.queries[tq * LPR + chunk];
                const uint32_t rp = rh_group_sum<LPR>(rh_popc4(v[u]));
                const uint32_t a = rh_group_sum<LPR>(rh_popc4_and(v[u], qv));
                res = a | ((P.qpop[tq] + rp - a) << 16);
            } else {
                (void)rh_group_sum<LPR>(0u); (void)rh_group_sum<LPR>(0u);
            }
            if (chunk == 0 && at[u] < total) P.out[at[u]] = res;
        }
    }
}

// ================================================================== host side
struct radhip_shard {
    radhip_index *idx = nullptr;
    int rank = 0, world = 1;
    uint32_t nq = 0, W = 0;       // W = request slots per traversal and step (row width x (1 + spec))
    uint32_t ns = 0;              // slots (== nq unless the row engine was given fewer: RADHIP_SHARD_SLOTS)
    uint32_t Wrow = 0, spec = 0;
    hipStream_t stream = nullptr; // the stream this shard's kernels and collectives run on (the index's, or its own: pairs)
    bool own_stream = false;
    uint64_t n_to_score = 0, first = 0, count = 0;
    ShardParams P{};
    EvalParams E{};
    uint4 *d_queries = nullptr;
    uint32_t *d_qpop = nullptr, *d_req = nullptr, *d_req_all = nullptr, *d_out = nullptr, *d_in = nullptr;
    uint64_t graph_gen = 0;
    size_t state_bytes = 0;
    radhip_traversal *wave = nullptr;   // the wave engine's state (null: thread or row engine)
    bool row = false;                   // the row engine (shard_step_row_kernel) instead of the thread engine
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double step_ms = 0.0, eval_ms = 0.0;
    uint64_t steps = 0, exchanged_bytes = 0;
};

static uint32_t sh_log2_ceil(uint64_t x) { uint32_t l = 0; while (((uint64_t)1 << l) < x) l++; return l; }

extern "C" int radhip_shard_destroy(radhip_shard_t *s) {
    if (!s) return RADHIP_OK;
    if (s->wave) (void)radhip_traversal_destroy(s->wave);
    if (s->idx && s->idx->dev_ready) (void)hipSetDevice(s->idx->device);
    void *ps[] = {s->d_queries, s->d_qpop, s->d_req, s->d_req_all, s->d_out, s->d_in, s->P.hdr, s->P.heap, s->P.vis, s->P.sc,
                  s->P.scored, s->P.poplog_nodes, s->P.poplog_levels, s->P.next_t, s->P.res};
    for (void *p : ps) if (p) (void)hipFree(p);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->own_stream && s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
    return RADHIP_OK;
}

extern "C" int radhip_shard_create(radhip_index_t *idx, int rank, int world, uint64_t row_first, uint64_t row_count,
                                   const uint8_t *queries_all, uint32_t nq, uint64_t n_to_score, uint32_t flags,
                                   radhip_shard_t **out) {
    if (!idx || !queries_all || !out) RH_FAIL(RADHIP_E_INVALID, "null argument");
    if (world < 1 || rank < 0 || rank >= world || nq == 0 || n_to_score == 0) RH_FAIL(RADHIP_E_INVALID, "bad argument");
    if (!idx->has_vectors || !idx->has_graph) RH_FAIL(RADHIP_E_STATE, "index needs vectors and a graph");
    if (idx->g_n > 1000000000ull) RH_FAIL(RADHIP_E_INVALID, "RAD traversal needs slots < 1e9");
    if (row_first + row_count > idx->g_n) RH_FAIL(RADHIP_E_RANGE, "rows [%llu, %llu) exceed the graph's %llu nodes",
                                                  (unsigned long long)row_first, (unsigned long long)(row_first + row_count),
                                                  (unsigned long long)idx->g_n);
    // the rows this rank evaluates must be resident: either the whole corpus or exactly this shard
    if (!(idx->shard_first <= row_first && row_first + row_count <= idx->shard_first + idx->n))
        RH_FAIL(RADHIP_E_STATE, "rows [%llu, %llu) are not resident in this index (it holds [%llu, %llu))",
                (unsigned long long)row_first, (unsigned long long)(row_first + row_count),
                (unsigned long long)idx->shard_first, (unsigned long long)(idx->shard_first + idx->n));
    radhip_traversal *wave = nullptr;
    bool want_row = false;
    {
        const char *e = getenv("RADHIP_SHARD_ENGINE");
        const bool want_wave = e && e[0] == 'w';
        want_row = !e || e[0] == 'r';   // the default since round 3
        if (want_wave && std::max<uint32_t>(idx->cap0, idx->M) > 16)
            RH_FAIL(RADHIP_E_INVALID, "RADHIP_SHARD_ENGINE=wave needs adjacency rows of at most 16 slots");
        if (want_wave)   // (takes the index lock itself)
            RH_TRY(rh_trav_create_sharded(idx, queries_all + (size_t)rank * nq * idx->row_bytes, nq, n_to_score, flags, &wave));
    }
    std::unique_lock<std::mutex> lk(idx->mu);
    {
        const int rc0 = rh_ensure_device(idx);
        if (rc0 != RADHIP_OK) { lk.unlock(); if (wave) (void)radhip_traversal_destroy(wave); return rc0; }
    }
    radhip_shard *s = new (std::nothrow) radhip_shard();
    if (!s) { lk.unlock(); if (wave) (void)radhip_traversal_destroy(wave); RH_FAIL(RADHIP_E_NOMEM, "out of host memory"); }
    s->wave = wave;
    s->stream = idx->stream;
    if ((flags & RADHIP_SHARD_OWN_STREAM) && !wave) {
        if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) { delete s; lk.unlock(); RH_FAIL(RADHIP_E_HIP, "hipStreamCreate failed"); }
        s->own_stream = true;
    }
    s->idx = idx; s->rank = rank; s->world = world; s->nq = nq; s->n_to_score = std::min<uint64_t>(n_to_score, idx->g_n);
    s->first = row_first; s->count = row_count; s->graph_gen = idx->graph_gen;
    // request slots per traversal and step: the widest adjacency row (what one expansion can need) plus as much again
    // for every queue head that is expanded speculatively (RADHIP_SHARD_SPEC = 0 (default), 1 or 2; thread engine only)
    const uint32_t Wrow = std::max<uint32_t>(idx->cap0, idx->M);
    // thread engine: off by default (it halves the frontier steps but makes each step longer by as much); row engine: 2
    // (0.54 x the steps — and collectives — at 1.6 x the step: profiles/r03)
    uint32_t spec = (want_row && !wave) ? 2u : 0u;
    if (const char *e = getenv("RADHIP_SHARD_SPEC")) { const int v = atoi(e); if (v >= 0 && v <= 2 && !wave) spec = (uint32_t)v; }
    // one row width for what a traversal needs + ONE more for everything it asks speculatively (two heads have ~5 unscored
    // neighbours each: they share 16 slots; a head whose candidates do not fit is simply not finished from the cache) —
    // a third less to all-gather, scan and reduce-scatter per step than a row width per head
    uint32_t spec_rows = spec ? 1u : 0u;
    if (const char *e = getenv("RADHIP_SHARD_SPEC_ROWS")) { const int v = atoi(e); if (v >= 1 && v <= 2 && spec) spec_rows = (uint32_t)v < spec ? (uint32_t)v : spec; }
    const uint32_t W = Wrow * (1u + spec_rows);
    s->W = W; s->Wrow = Wrow; s->spec = spec;
    const uint64_t n_top = idx->n_top;
    const uint64_t scored_cap = s->n_to_score + W + n_top;
    const uint64_t up_pairs = idx->n_upper_rows + n_top * (uint64_t)(idx->max_level + 1);
    // queue entries = scored nodes (one level-0 entry each) + visits above level 0: the traversal kernels' estimate
    // (scored_cap * 8 / connectivity) with a factor of two on top, never more than the graph has
    // queue entries = scored nodes (one level-0 entry each) + visits above level 0 (measured at most 0.11 x n_to_score on
    // the bench graphs, connectivity 8: sized for 2 / connectivity of the scored nodes, never more than the graph has; a
    // traversal that outgrows it fails with RADHIP_E_CAPACITY on every rank at the same step)
    const uint64_t up_est = std::min<uint64_t>(up_pairs, scored_cap * 2 / idx->M + 4096);
    uint64_t heap_cap = scored_cap + up_est + 64;
    // (test hook: a queue that is too small, so that a traversal fails on the device in the middle of a run)
    if (const char *e = getenv("RADHIP_SHARD_TEST_HEAP_CAP")) { const long long v = atoll(e); if (v > 0) heap_cap = (uint64_t)v; }
    const uint32_t vlog2 = std::max<uint32_t>(8, sh_log2_ceil(up_est + up_est / 2 + 64));   // (node, level >= 1) pairs only
    const uint32_t slog2 = std::max<uint32_t>(8, sh_log2_ceil(2 * scored_cap));
    ShardParams &P = s->P;
    P.adj0 = idx->d_adj0; P.upper_row = idx->d_upper_row; P.adjU = idx->d_adjU; P.top = idx->d_top;
    P.n_top = idx->n_top; P.cap0 = idx->cap0; P.capU = idx->M; P.nq = nq; P.W = Wrow; P.Wt = W; P.spec = spec;
    P.start_level = idx->max_level > 0 ? idx->max_level - 1 : 0;
    s->row = want_row && !wave && heap_cap <= (1ull << 24);   // (shr_push loads six ancestors: heaps below 16^6 entries)
    P.n_to_score = s->n_to_score; P.heap_cap = heap_cap; P.heap_stride = (heap_cap + 16 + 15) & ~15ull; P.vlog2 = vlog2; P.slog2 = slog2; P.scored_cap = scored_cap;
    // pops per step while nothing needs a score: the step ends with its slowest traversal, so a long inner loop
    // makes every step as slow as the unluckiest of thousands of traversals (measured: 2 beats 1, 4 and 8)
    P.max_inner = 2;
    // (row engine with slots that stay live: 4 — 270 against 253 M expansions/s and 28.4k against 32.5k steps for 65536
    // traversals on 32768 slots; 6, 8, 16: 263, 255, 230 M — profiles/r03/sharded_slots/spec_inner_sweep.log)
    if (s->row) P.max_inner = 4;
    if (const char *e = getenv("RADHIP_SHARD_INNER")) { const int v = atoi(e); if (v >= 1 && v <= 64) P.max_inner = (uint32_t)v; }
    // Slots: the heavy per-traversal structures (queue, sets: 3.6 MB at n_to_score = 100k) exist once per SLOT; a batch may
    // hold more traversals than slots (RADHIP_SHARD_SLOTS, row engine, product loop): a slot whose traversal is done takes
    // the next one.  The traversals of a batch differ in length (the longest needs twice the mean) and the loop runs until
    // the last one ends: with one slot per traversal half of the slot-steps of a batch are idle.
    uint32_t ns = nq;
    if (const char *e = getenv("RADHIP_SHARD_SLOTS")) { const long long v = atoll(e); if (v > 0 && (uint64_t)v < nq && s->row) ns = (uint32_t)v; }
    // a slot's set entries carry a 7-bit epoch, one epoch per traversal it takes (1..127; the sets are cleared by reset()):
    // with fewer than nq / 127 slots every slot would run out of epochs before the batch is done and the rest of the
    // traversals would never run
    if ((uint64_t)ns * SHR_EPOCHS_PER_SLOT < nq) {
        lk.unlock();
        radhip_shard_destroy(s);
        RH_FAIL(RADHIP_E_INVALID, "RADHIP_SHARD_SLOTS=%u is too few for %u traversals per batch: a slot takes at most %u (need >= %u slots)",
                ns, nq, SHR_EPOCHS_PER_SLOT, (nq + SHR_EPOCHS_PER_SLOT - 1u) / SHR_EPOCHS_PER_SLOT);
    }
    s->ns = ns; P.ns = ns;
    int rc = RADHIP_OK;
    const size_t per_rank = (size_t)ns * W;
    auto al = [&](void **p, size_t bytes, bool zero) {
        if (rc != RADHIP_OK) return;
        hipError_t e = hipMalloc(p, bytes ? bytes : 16);
        if (e != hipSuccess) {
            radhip_set_error("hipMalloc(%zu) for the sharded traversal state failed: %s", bytes, hipGetErrorString(e));
            rc = e == hipErrorOutOfMemory ? RADHIP_E_NOMEM : RADHIP_E_HIP;
            (void)hipGetLastError();
            return;
        }
        s->state_bytes += bytes;
        if (zero && hipMemsetAsync(*p, 0, bytes ? bytes : 16, idx->stream) != hipSuccess) rc = RADHIP_E_HIP;
    };
    if (!wave) {
        al((void **)&P.hdr, (size_t)ns * sizeof(ShardHeader), true);
        al((void **)&P.heap, (size_t)ns * P.heap_stride * 8, false);
        al((void **)&P.vis, ((size_t)ns << vlog2) * 8, true);
        al((void **)&P.sc, ((size_t)ns << slog2) * 8, true);
        al((void **)&P.scored, (size_t)nq * scored_cap * sizeof(uint2), false);
        al((void **)&P.res, (size_t)nq * sizeof(ShardResult), true);
        al((void **)&P.next_t, 64, true);
    } else s->state_bytes += radhip_traversal_state_bytes(wave);
    al((void **)&s->d_req, (per_rank + ns + 16) * 4, true);
    al((void **)&s->d_req_all, (size_t)world * (per_rank + ns + 16) * 4, true);
    al((void **)&s->d_out, (size_t)world * per_rank * 4, true);
    al((void **)&s->d_in, per_rank * 4, true);
    al((void **)&s->d_queries, (size_t)world * nq * idx->row_stride, false);
    al((void **)&s->d_qpop, (size_t)world * nq * 4, false);
    if (!wave && (flags & RADHIP_TRAV_LOG_POPS)) {
        P.poplog_cap = heap_cap;
        al((void **)&P.poplog_nodes, (size_t)nq * heap_cap * 4, false);
        al((void **)&P.poplog_levels, (size_t)nq * heap_cap, false);
    }
    if (rc == RADHIP_OK && (hipEventCreate(&s->ev0) != hipSuccess || hipEventCreate(&s->ev1) != hipSuccess)) rc = RADHIP_E_HIP;
    if (rc == RADHIP_OK) {
        const size_t tq = (size_t)world * nq;
        std::vector<uint8_t> padded(tq * idx->row_stride, 0);
        std::vector<uint32_t> pop(tq, 0);
        for (size_t i = 0; i < tq; ++i) {
            memcpy(padded.data() + i * idx->row_stride, queries_all + i * idx->row_bytes, idx->row_bytes);
            uint32_t p = 0;
            for (uint32_t b = 0; b < idx->row_bytes; ++b) p += (uint32_t)__builtin_popcount(queries_all[i * idx->row_bytes + b]);
            pop[i] = p;
        }
        if (hipMemcpyAsync(s->d_queries, padded.data(), padded.size(), hipMemcpyHostToDevice, idx->stream) != hipSuccess ||
            hipMemcpyAsync(s->d_qpop, pop.data(), tq * 4, hipMemcpyHostToDevice, idx->stream) != hipSuccess ||
            hipStreamSynchronize(idx->stream) != hipSuccess) {
            radhip_set_error("query upload failed");
            rc = RADHIP_E_HIP;
        }
    }
    if (rc != RADHIP_OK) { lk.unlock(); radhip_shard_destroy(s); return rc; }
    P.req = s->d_req; P.scores_in = s->d_in;
    if (wave) rh_trav_bind_shard(wave, s->d_req, s->d_in, W, P.max_inner);
    EvalParams &E = s->E;
    E.fp = idx->d_fp + (row_first - idx->shard_first) * idx->lpr;
    E.first = row_first; E.count = row_count; E.queries = s->d_queries; E.qpop = s->d_qpop;
    E.req_all = s->d_req_all; E.out = s->d_out; E.live = s->d_req + per_rank + ns; E.world = (uint32_t)world; E.nq = nq; E.W = W;
    E.ns = ns; E.identity = 0;
    *out = s;
    return RADHIP_OK;
}

// re-arm the same state for a new batch of world * nq queries (bench steps): headers and sets are cleared
// on the device, nothing is reallocated
extern "C" int radhip_shard_reset(radhip_shard_t *s, const uint8_t *queries_all) {
    if (!s || !queries_all) RH_FAIL(RADHIP_E_INVALID, "null argument");
    radhip_index *idx = s->idx;
    if (s->wave) RH_TRY(radhip_traversal_reset(s->wave, queries_all + (size_t)s->rank * s->nq * idx->row_bytes));   // (own lock, own generation check)
    std::lock_guard<std::mutex> lk(idx->mu);
    if (s->graph_gen != idx->graph_gen)
        RH_FAIL(RADHIP_E_STATE, "the index changed since this sharded traversal was created: create a new one");
    RH_HIP(hipSetDevice(idx->device));
    const size_t tq = (size_t)s->world * s->nq, per_rank = (size_t)s->ns * s->W;
    std::vector<uint8_t> padded(tq * idx->row_stride, 0);
    std::vector<uint32_t> pop(tq, 0);
    for (size_t i = 0; i < tq; ++i) {
        memcpy(padded.data() + i * idx->row_stride, queries_all + i * idx->row_bytes, idx->row_bytes);
        uint32_t p = 0;
        for (uint32_t b = 0; b < idx->row_bytes; ++b) p += (uint32_t)__builtin_popcount(queries_all[i * idx->row_bytes + b]);
        pop[i] = p;
    }
    hipStream_t st = s->stream;
    RH_HIP(hipMemcpyAsync(s->d_queries, padded.data(), padded.size(), hipMemcpyHostToDevice, st));
    RH_HIP(hipMemcpyAsync(s->d_qpop, pop.data(), tq * 4, hipMemcpyHostToDevice, st));
    if (!s->wave) {
        RH_HIP(hipMemsetAsync(s->P.hdr, 0, (size_t)s->ns * sizeof(ShardHeader), st));
        RH_HIP(hipMemsetAsync(s->P.vis, 0, ((size_t)s->ns << s->P.vlog2) * 8, st));
        RH_HIP(hipMemsetAsync(s->P.sc, 0, ((size_t)s->ns << s->P.slog2) * 8, st));
        RH_HIP(hipMemsetAsync(s->P.res, 0, (size_t)s->nq * sizeof(ShardResult), st));
        RH_HIP(hipMemsetAsync(s->P.next_t, 0, 64, st));
    }
    RH_HIP(hipMemsetAsync(s->d_req, 0, (per_rank + s->ns + 16) * 4, st));
    RH_HIP(hipMemsetAsync(s->d_in, 0, per_rank * 4, st));
    RH_HIP(hipStreamSynchronize(st));
    s->step_ms = 0.0; s->eval_ms = 0.0; s->steps = 0; s->exchanged_bytes = 0;
    return RADHIP_OK;
}

extern "C" uint32_t radhip_shard_width(const radhip_shard_t *s) { return s ? s->W : 0; }
extern "C" uint32_t radhip_shard_slots(const radhip_shard_t *s) { return s ? s->ns : 0; }
extern "C" int radhip_shard_speculation(const radhip_shard_t *s, uint32_t *out_depth, uint64_t *out_requested, uint64_t *out_used,
                                        uint64_t *out_hits) {
    if (!s) RH_FAIL(RADHIP_E_INVALID, "null argument");
    if (out_depth) *out_depth = s->spec;
    uint64_t rq = 0, us = 0, hi = 0;
    if (!s->wave && s->spec) {
        std::lock_guard<std::mutex> lk(s->idx->mu);
        RH_HIP(hipSetDevice(s->idx->device));
        std::vector<ShardHeader> hdr(s->ns);
        RH_HIP(hipMemcpy(hdr.data(), s->P.hdr, (size_t)s->ns * sizeof(ShardHeader), hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < s->ns; ++i) { rq += hdr[i].spec_req; us += hdr[i].spec_used; hi += hdr[i].spec_hit; }
    }
    if (out_requested) *out_requested = rq;
    if (out_used) *out_used = us;
    if (out_hits) *out_hits = hi;
    return RADHIP_OK;
}
extern "C" int radhip_shard_engine(const radhip_shard_t *s) { return !s ? 0 : s->wave ? 1 : s->row ? 2 : 0; }
extern "C" uint64_t radhip_shard_state_bytes(const radhip_shard_t *s) { return s ? s->state_bytes : 0; }

static int shard_check(radhip_shard *s) {
    if (s->graph_gen != s->idx->graph_gen)
        RH_FAIL(RADHIP_E_STATE, "the index changed since this sharded traversal was created: create a new one");
    RH_HIP(hipSetDevice(s->idx->device));
    return RADHIP_OK;
}

// enqueue one step kernel on the shard's stream (no synchronisation)
static int shard_enqueue_step(radhip_shard *s, bool zero_live) {
    // (in the product loop the evaluation kernel of the step before has zeroed the live count: one launch less)
    if (zero_live) RH_HIP(hipMemsetAsync(s->d_req + (size_t)s->ns * s->W + s->ns, 0, 64, s->stream));
    if (s->wave) return rh_trav_enqueue_shard_step(s->wave);
    if (s->row) hipLaunchKernelGGL(shard_step_row_kernel, dim3((s->ns + 3u) / 4u), dim3(64), 0, s->stream, s->P);
    else hipLaunchKernelGGL(shard_step_kernel, dim3((s->nq + 63u) / 64u), dim3(64), 0, s->stream, s->P);
    RH_HIP(hipGetLastError());
    return RADHIP_OK;
}
static int shard_enqueue_eval(radhip_shard *s) {
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, s->idx->device);
    const uint64_t total = (uint64_t)s->world * s->ns * s->W;
    const uint64_t per_block = 4ull * (64 / s->idx->lpr) * 4ull;
    const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((total + per_block - 1) / per_block, (uint64_t)n_cu * 8));
    switch (s->idx->lpr) {
        case 1: hipLaunchKernelGGL(shard_eval_kernel<1>, dim3(grid), dim3(256), 0, s->stream, s->E); break;
        case 2: hipLaunchKernelGGL(shard_eval_kernel<2>, dim3(grid), dim3(256), 0, s->stream, s->E); break;
        case 4: hipLaunchKernelGGL(shard_eval_kernel<4>, dim3(grid), dim3(256), 0, s->stream, s->E); break;
        case 8: hipLaunchKernelGGL(shard_eval_kernel<8>, dim3(grid), dim3(256), 0, s->stream, s->E); break;
        default: hipLaunchKernelGGL(shard_eval_kernel<16>, dim3(grid), dim3(256), 0, s->stream, s->E); break;
    }
    RH_HIP(hipGetLastError());
    return RADHIP_OK;
}

// ---- host-staged pieces (tests, rehearsal on one GPU, any exchange the host program has) -------------
// (the host-staged exchange moves candidates and scores only, not the slots' traversal numbers: one slot per traversal)
#define SH_REQUIRE_IDENTITY(s) do { if ((s)->ns != (s)->nq) RH_FAIL(RADHIP_E_STATE, "the host-staged exchange needs one slot per traversal (RADHIP_SHARD_SLOTS is for radhip_shard_run)"); } while (0)
extern "C" int radhip_shard_step(radhip_shard_t *s, uint32_t *out_live) {
    if (!s) RH_FAIL(RADHIP_E_INVALID, "null argument");
    SH_REQUIRE_IDENTITY(s);
    std::lock_guard<std::mutex> lk(s->idx->mu);
    RH_TRY(shard_check(s));
    RH_HIP(hipEventRecord(s->ev0, s->stream));
    RH_TRY(shard_enqueue_step(s, true));
    RH_HIP(hipEventRecord(s->ev1, s->stream));
    uint32_t live = 0;
    RH_HIP(hipMemcpyAsync(&live, s->d_req + (size_t)s->ns * s->W + s->ns, 4, hipMemcpyDeviceToHost, s->stream));
    RH_HIP(hipStreamSynchronize(s->stream));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s->ev0, s->ev1) == hipSuccess) s->step_ms += ms;
    s->steps++;
    // (a traversal that failed on the device keeps the rank "live" for its peers: they all stop at the next look)
    if (out_live) *out_live = (live & SH_POISON) ? (live & ~SH_POISON) + 1u : live;
    return RADHIP_OK;
}
extern "C" int radhip_shard_get_requests(radhip_shard_t *s, uint32_t *host) {
    if (!s || !host) RH_FAIL(RADHIP_E_INVALID, "null argument");
    SH_REQUIRE_IDENTITY(s);
    std::lock_guard<std::mutex> lk(s->idx->mu);
    RH_TRY(shard_check(s));
    RH_HIP(hipMemcpy(host, s->d_req, (size_t)s->nq * s->W * 4, hipMemcpyDeviceToHost));
    return RADHIP_OK;
}
extern "C" int radhip_shard_set_requests_all(radhip_shard_t *s, const uint32_t *host_all) {
    if (!s || !host_all) RH_FAIL(RADHIP_E_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(s->idx->mu);
    RH_TRY(shard_check(s));
    SH_REQUIRE_IDENTITY(s);
    const size_t per_rank = (size_t)s->nq * s->W;
    for (int r = 0; r < s->world; ++r)
        RH_HIP(hipMemcpy(s->d_req_all + (size_t)r * (per_rank + s->ns + 16), host_all + (size_t)r * per_rank, per_rank * 4, hipMemcpyHostToDevice));
    return RADHIP_OK;
}
extern "C" int radhip_shard_evaluate(radhip_shard_t *s) {
    if (!s) RH_FAIL(RADHIP_E_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(s->idx->mu);
    RH_TRY(shard_check(s));
    SH_REQUIRE_IDENTITY(s);
    s->E.identity = 1u;
    RH_HIP(hipEventRecord(s->ev0, s->stream));
    RH_TRY(shard_enqueue_eval(s));
    RH_HIP(hipEventRecord(s->ev1, s->stream));
    RH_HIP(hipStreamSynchronize(s->stream));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s->ev0, s->ev1) == hipSuccess) s->eval_ms += ms;
    return RADHIP_OK;
}
extern "C" int radhip_shard_get_scores_out(radhip_shard_t *s, uint32_t *host) {
    if (!s || !host) RH_FAIL(RADHIP_E_INVALID, "null argument");
    SH_REQUIRE_IDENTITY(s);
    std::lock_guard<std::mutex> lk(s->idx->mu);
    RH_TRY(shard_check(s));
    RH_HIP(hipMemcpy(host, s->d_out, (size_t)s->world * s->nq * s->W * 4, hipMemcpyDeviceToHost));
    return RADHIP_OK;
}
extern "C" int radhip_shard_set_scores_in(radhip_shard_t *s, const uint32_t *host) {
    if (!s || !host) RH_FAIL(RADHIP_E_INVALID, "null argument");
    SH_REQUIRE_IDENTITY(s);
    std::lock_guard<std::mutex> lk(s->idx->mu);
    RH_TRY(shard_check(s));
    RH_HIP(hipMemcpy(s->d_in, host, (size_t)s->nq * s->W * 4, hipMemcpyHostToDevice));
    return RADHIP_OK;
}

static int shard_first_error(radhip_shard *s);
static int shard_all_ran(radhip_shard *s);

// wait for a stream with a deadline: a peer that died or left the loop must not hang this rank in a collective
// for ever (RADHIP_SHARD_TIMEOUT_S, default 300 s per look; a look normally takes well under a second)
static int shard_sync(hipStream_t st) {
    static const double limit = [] { const char *e = getenv("RADHIP_SHARD_TIMEOUT_S"); const double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 300.0; }();
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t spins = 0;
    for (;;) {
        const hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) return RADHIP_OK;
        if (e != hipErrorNotReady) RH_FAIL(RADHIP_E_HIP, "the sharded loop's stream failed: %s", hipGetErrorString(e));
        if (++spins < 4096u) continue;                       // a look every few steps: the first answers come in microseconds
        std::this_thread::sleep_for(std::chrono::microseconds(50));
        if ((spins & 1023u) == 0u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit)
            RH_FAIL(RADHIP_E_COMM, "no answer from the sharded loop's stream for %.0f s (a peer left a collective?)", limit);
    }
}

// ---- the product loop: kernels and RCCL collectives, device buffers end to end.  One group of traversals on one
// stream, or TWO groups on two streams with two communicators (radhip_shard_run_pair): a frontier step is latency-
// bound end to end (a step kernel that ends with its slowest pop, then two small collectives), so the second group's
// step kernel runs while the first group's collectives are on the wire and the other way round.
static int shard_run_groups(radhip_shard **S, radhip_comm **C, int ng, uint64_t max_steps, uint64_t *out_steps) {
    radhip_index *idx = S[0]->idx;
    for (int g = 0; g < ng; ++g) {
        if (!S[g] || !C[g]) RH_FAIL(RADHIP_E_INVALID, "null argument");
        if (S[g]->idx != idx) RH_FAIL(RADHIP_E_INVALID, "the groups of one loop share one index");
        if (C[g]->world != S[g]->world || C[g]->rank != S[g]->rank) RH_FAIL(RADHIP_E_INVALID, "communicator and shard disagree on rank / world");
        if (ng > 1 && S[g]->wave) RH_FAIL(RADHIP_E_INVALID, "paired groups need the thread engine");
        for (int h = 0; h < g; ++h)
            if (S[h]->stream == S[g]->stream || C[h] == C[g]) RH_FAIL(RADHIP_E_INVALID, "paired groups need a stream and a communicator each (RADHIP_SHARD_OWN_STREAM)");
    }
    std::unique_lock<std::mutex> lk(idx->mu);
    for (int g = 0; g < ng; ++g) RH_TRY(shard_check(S[g]));
    const int world = S[0]->world;
    std::vector<uint32_t> live((size_t)world * ng);
    uint64_t steps = 0;
    // RADHIP_SHARD_TIMING=1: per-phase device time of the loop (diagnostic; one group only; adds five event records per step)
    const bool phases = ng == 1 && getenv("RADHIP_SHARD_TIMING") != nullptr;
    hipEvent_t pe[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    double pms[4] = {0, 0, 0, 0};
    if (phases) for (auto &e : pe) (void)hipEventCreate(&e);
    long long fail_at = -1;   // test hook: a host-side failure in the middle of the loop
    if (const char *e = getenv("RADHIP_SHARD_TEST_FAIL_AT")) fail_at = atoll(e);
    uint32_t poisoned_by = 0xFFFFFFFFu;
    auto loop = [&]() -> int {
        for (int g = 0; g < ng; ++g) RH_HIP(hipEventRecord(S[g]->ev0, S[g]->stream));
        for (;;) {
            if (fail_at >= 0 && (long long)steps == fail_at) RH_FAIL(RADHIP_E_HIP, "injected failure at step %lld (RADHIP_SHARD_TEST_FAIL_AT)", fail_at);
            for (int g = 0; g < ng; ++g) {
                radhip_shard *s = S[g];
                hipStream_t st = s->stream;
                const size_t per_rank = (size_t)s->ns * s->W, blk = per_rank + s->ns + 16;
                s->E.identity = 0u;
                if (phases) (void)hipEventRecord(pe[0], st);
                RH_TRY(shard_enqueue_step(s, steps == 0));
                if (phases) (void)hipEventRecord(pe[1], st);
                RH_TRY(rh_comm_allgather_dev(C[g], s->d_req, s->d_req_all, blk, st));
                if (phases) (void)hipEventRecord(pe[2], st);
                RH_TRY(shard_enqueue_eval(s));
                if (phases) (void)hipEventRecord(pe[3], st);
                RH_TRY(rh_comm_reduce_scatter_u32_dev(C[g], s->d_out, s->d_in, per_rank, st));
                if (phases) (void)hipEventRecord(pe[4], st);
                s->exchanged_bytes += (uint64_t)s->world * blk * 4 + (uint64_t)s->world * per_rank * 4;
            }
            steps++;
            // The live words of all ranks travel behind the candidates, so every rank sees the same numbers and stops
            // at the same step.  The host looks at them (a stream synchronisation) only every fourth step — and every
            // 64th while no traversal can have reached n_to_score yet (a step scores W nodes at most): a step of
            // finished traversals is a no-op, so looking late costs a few empty steps, never a different result.
            const bool last = max_steps && steps >= max_steps;
            const bool look = phases || last || (steps % (steps * S[0]->W < S[0]->n_to_score ? 64u : 4u)) == 0u;
            if (!look) continue;
            for (int g = 0; g < ng; ++g) {
                const size_t per_rank = (size_t)S[g]->ns * S[g]->W, blk = per_rank + S[g]->ns + 16;
                for (int r = 0; r < world; ++r)
                    RH_HIP(hipMemcpyAsync(&live[(size_t)g * world + r], S[g]->d_req_all + (size_t)r * blk + per_rank + S[g]->ns, 4, hipMemcpyDeviceToHost, S[g]->stream));
            }
            for (int g = 0; g < ng; ++g) RH_TRY(shard_sync(S[g]->stream));
            if (phases) for (int i = 0; i < 4; ++i) { float m = 0.f; if (hipEventElapsedTime(&m, pe[i], pe[i + 1]) == hipSuccess) pms[i] += m; }
            uint64_t tot = 0;
            for (size_t i = 0; i < live.size(); ++i) {
                if ((live[i] & SH_POISON) && poisoned_by == 0xFFFFFFFFu) poisoned_by = (uint32_t)(i % (size_t)world);
                tot += live[i] & ~SH_POISON;
            }
            if (poisoned_by != 0xFFFFFFFFu || tot == 0 || last) break;
        }
        for (int g = 0; g < ng; ++g) RH_HIP(hipEventRecord(S[g]->ev1, S[g]->stream));
        for (int g = 0; g < ng; ++g) RH_TRY(shard_sync(S[g]->stream));
        return RADHIP_OK;
    };
    const int rc = loop();
    float ms = 0.f;
    if (rc == RADHIP_OK) for (int g = 0; g < ng; ++g) if (hipEventElapsedTime(&ms, S[g]->ev0, S[g]->ev1) == hipSuccess) S[g]->step_ms += ms;
    if (phases) {
        if (rc == RADHIP_OK && steps)
            fprintf(stderr, "[shard] %llu steps, device ms per step: step kernel %.4f, all-gather %.4f, evaluation %.4f, reduce-scatter %.4f; whole loop %.4f\n",
                    (unsigned long long)steps, pms[0] / steps, pms[1] / steps, pms[2] / steps, pms[3] / steps, ms / steps);
        for (auto &e : pe) if (e) (void)hipEventDestroy(e);
    }
    for (int g = 0; g < ng; ++g) S[g]->steps += steps;
    if (out_steps) *out_steps = steps;
    if (rc != RADHIP_OK) {
        // This rank leaves the loop alone: its peers are (or will be) inside a collective it never enters.  Abort the
        // communicators, so that their collectives fail (or their look times out) instead of waiting for ever; the
        // stream is drained first so that nothing of this rank is left half-enqueued.
        char msg[400];
        snprintf(msg, sizeof msg, "%s", radhip_last_error());
        // Abort FIRST: when the failure is the deadline, or a peer that died, the stream still holds a collective
        // that can never complete — an unbounded hipStreamSynchronize in front of ncclCommAbort would wait for it for ever.
        // The abort makes the in-flight collective kernel exit; the stream is then drained with the same deadline as a look.
        for (int g = 0; g < ng; ++g) if (C[g]->world > 1) (void)rh_comm_abort(C[g]);
        bool drained = true;
        for (int g = 0; g < ng; ++g) if (shard_sync(S[g]->stream) != RADHIP_OK) drained = false;
        (void)hipGetLastError();
        radhip_set_error("radhip_shard_run left the loop after %llu steps: %s%s%s", (unsigned long long)steps, msg,
                         world > 1 ? " (communicator aborted: the peers' collectives fail instead of hanging)" : "",
                         drained ? "" : "; the stream did not drain within the deadline either");
        return drained ? rc : RADHIP_E_COMM;
    }
    lk.unlock();
    if (poisoned_by != 0xFFFFFFFFu) {
        // every rank saw the same word at the same step and left together: no abort needed.  The rank that owns the
        // failed traversal reports which one; the others report the remote failure.
        for (int g = 0; g < ng; ++g) { const int e = shard_first_error(S[g]); if (e != RADHIP_OK) return e; }
        RH_FAIL(RADHIP_E_CAPACITY, "a traversal of rank %u overflowed a fixed-capacity device structure: the loop ended on every rank after %llu steps",
                poisoned_by, (unsigned long long)steps);
    }
    for (int g = 0; g < ng; ++g) RH_TRY(shard_first_error(S[g]));
    if (!(max_steps && steps >= max_steps))   // the loop ended because nothing was live: then every traversal of the batch has run
        for (int g = 0; g < ng; ++g) RH_TRY(shard_all_ran(S[g]));
    return RADHIP_OK;
}

extern "C" int radhip_shard_run(radhip_shard_t *s, radhip_comm_t *comm, uint64_t max_steps, uint64_t *out_steps) {
    if (!s || !comm) RH_FAIL(RADHIP_E_INVALID, "null argument");
    radhip_shard *S[1] = {s};
    radhip_comm *C[1] = {comm};
    return shard_run_groups(S, C, 1, max_steps, out_steps);
}
extern "C" int radhip_shard_run_pair(radhip_shard_t *a, radhip_comm_t *comm_a, radhip_shard_t *b, radhip_comm_t *comm_b,
                                     uint64_t max_steps, uint64_t *out_steps) {
    if (!a || !b || !comm_a || !comm_b) RH_FAIL(RADHIP_E_INVALID, "null argument");
    radhip_shard *S[2] = {a, b};
    radhip_comm *C[2] = {comm_a, comm_b};
    return shard_run_groups(S, C, 2, max_steps, out_steps);
}

// a device-side failure of any local traversal is an error of the call
static int shard_first_error(radhip_shard *s) {
    if (s->wave) {
        std::vector<radhip_trav_stats_t> st(s->nq);
        RH_TRY(radhip_traversal_stats(s->wave, st.data()));
        for (uint32_t i = 0; i < s->nq; ++i)
            if (st[i].status < 0) RH_FAIL(st[i].status, "sharded traversal %u overflowed a fixed-capacity device structure (status %d); "
                                          "RADHIP_SHARD_ENGINE=thread sizes its queue for the whole graph", i, st[i].status);
        return RADHIP_OK;
    }
    std::vector<ShardResult> res(s->nq);
    RH_HIP(hipMemcpy(res.data(), s->P.res, (size_t)s->nq * sizeof(ShardResult), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < s->nq; ++i)
        if (res[i].status < 0) RH_FAIL(res[i].status, "sharded traversal %u overflowed a fixed-capacity device structure (status %d)", i, res[i].status);
    return RADHIP_OK;
}

// the loop ended with no live traversal anywhere: a traversal whose result is still "not started" (status 0, nothing
// scored) was never taken by a slot — results must not be silently missing
static int shard_all_ran(radhip_shard *s) {
    if (s->wave) return RADHIP_OK;
    std::vector<ShardResult> res(s->nq);
    RH_HIP(hipMemcpy(res.data(), s->P.res, (size_t)s->nq * sizeof(ShardResult), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < s->nq; ++i)
        if (res[i].status == 0) RH_FAIL(RADHIP_E_STATE, "sharded traversal %u of %u never ran (%u slots ran out of epochs?): results are incomplete", i, s->nq, s->ns);
    return RADHIP_OK;
}

extern "C" int radhip_shard_stats(const radhip_shard_t *s, radhip_trav_stats_t *out) {
    if (!s || !out) RH_FAIL(RADHIP_E_INVALID, "null argument");
    if (s->wave) return radhip_traversal_stats(s->wave, out);
    std::lock_guard<std::mutex> lk(s->idx->mu);
    RH_HIP(hipSetDevice(s->idx->device));
    std::vector<ShardResult> res(s->nq);
    RH_HIP(hipMemcpy(res.data(), s->P.res, (size_t)s->nq * sizeof(ShardResult), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < s->nq; ++i) {
        out[i].n_scored = res[i].n_scored; out[i].n_pops = res[i].n_pops; out[i].n_nbr = res[i].n_nbr;
        out[i].n_repivot = 0; out[i].n_flush = 0; out[i].status = res[i].status; out[i].n_remid = 0;
        out[i].n_upper = res[i].n_vis;
    }
    return RADHIP_OK;
}

extern "C" int radhip_shard_results(const radhip_shard_t *s, uint32_t q, uint32_t *out_slots, uint32_t *out_and,
                                    uint32_t *out_or, uint64_t cap, uint64_t *out_n) {
    if (!s || !out_n) RH_FAIL(RADHIP_E_INVALID, "null argument");
    if (q >= s->nq) RH_FAIL(RADHIP_E_RANGE, "traversal %u out of range", q);
    if (s->wave) return radhip_traversal_results(s->wave, q, out_slots, out_and, out_or, cap, out_n);
    std::lock_guard<std::mutex> lk(s->idx->mu);
    RH_HIP(hipSetDevice(s->idx->device));
    ShardResult h;
    RH_HIP(hipMemcpy(&h, s->P.res + q, sizeof h, hipMemcpyDeviceToHost));
    *out_n = h.n_scored;
    const uint64_t n = std::min<uint64_t>(h.n_scored, cap);
    if (n == 0) return RADHIP_OK;
    std::vector<uint2> buf(n);
    RH_HIP(hipMemcpy(buf.data(), s->P.scored + (uint64_t)q * s->P.scored_cap, n * sizeof(uint2), hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < n; ++i) {
        if (out_slots) out_slots[i] = buf[i].x;
        if (out_and) out_and[i] = buf[i].y & 0xFFFFu;
        if (out_or) out_or[i] = buf[i].y >> 16;
    }
    return RADHIP_OK;
}

extern "C" int radhip_shard_pop_log(const radhip_shard_t *s, uint32_t q, uint32_t *out_nodes, uint8_t *out_levels,
                                    uint64_t cap, uint64_t *out_n) {
    if (!s || !out_n) RH_FAIL(RADHIP_E_INVALID, "null argument");
    if (q >= s->nq) RH_FAIL(RADHIP_E_RANGE, "traversal %u out of range", q);
    if (s->wave) return radhip_traversal_pop_log(s->wave, q, out_nodes, out_levels, cap, out_n);
    if (!s->P.poplog_nodes) RH_FAIL(RADHIP_E_STATE, "sharded traversal was created without RADHIP_TRAV_LOG_POPS");
    std::lock_guard<std::mutex> lk(s->idx->mu);
    RH_HIP(hipSetDevice(s->idx->device));
    ShardResult h;
    RH_HIP(hipMemcpy(&h, s->P.res + q, sizeof h, hipMemcpyDeviceToHost));
    const uint64_t m = std::min<uint64_t>(h.n_pops, s->P.poplog_cap);
    *out_n = m;
    const uint64_t n = std::min<uint64_t>(m, cap);
    if (n == 0) return RADHIP_OK;
    if (out_nodes) RH_HIP(hipMemcpy(out_nodes, s->P.poplog_nodes + (uint64_t)q * s->P.poplog_cap, n * 4, hipMemcpyDeviceToHost));
    if (out_levels) RH_HIP(hipMemcpy(out_levels, s->P.poplog_levels + (uint64_t)q * s->P.poplog_cap, n, hipMemcpyDeviceToHost));
    return RADHIP_OK;
}

extern "C" int radhip_shard_timing(const radhip_shard_t *s, double *out_step_ms, double *out_eval_ms, uint64_t *out_steps,
                                   uint64_t *out_exchanged_bytes) {
    if (!s) RH_FAIL(RADHIP_E_INVALID, "null argument");
    if (out_step_ms) *out_step_ms = s->step_ms;
    if (out_eval_ms) *out_eval_ms = s->eval_ms;
    if (out_steps) *out_steps = s->steps;
    if (out_exchanged_bytes) *out_exchanged_bytes = s->exchanged_bytes;
    return RADHIP_OK;
}
