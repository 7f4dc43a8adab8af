// traverse.hip — K3: RAD best-first traversal, Tanimoto-scored, one wavefront
// per traversal.  gfx950 only (wave64, LDS, DPP; no MFMA: this is bit counting).
//
// Reference control flow restated on the device (paths relative to the
// reference tree):
//   prime                 rad/traverser.py:141-170
//   pop-min               rad/priority_queue.py:22-39   (ZSET order, see common.h key)
//   neighbors             index.get_neighbors, rad/hnsw_service.py:222
//   score-if-unscored     rad/distributed_worker.py:296-305
//   visited test-and-set  rad/visited.py:17-29, key (node, level)
//   scored insert         rad/scored.py:37-47, key node, insertion order
//   queue insert/descend  rad/coordination_service.py:369-395
//
// Per-traversal state lives in HBM (sized for n_to_score):
//   ht      open-addressing table of u64 {slot | (and|or<<12|v0<<24)<<32}: presence ==
//           scored, v0 == visited on level 0
//   ut      open-addressing set of (slot<<4|level)+1 for levels >= 1
//   scored  {slot, and|or<<16} in insertion order (the output)
//   pq      "far" keys: sorted runs of u64 keys written once by staging flushes, with a
//           run table {pos,end} and the head key of every run
// The queue is a two-level "pivot queue":
//   near    every key below `pivot` lives in registers: RK sorted keys per lane (lane-private
//           insertion is SIMD-parallel, pop-min is one DPP wave-min)
//   far     keys >= pivot are appended unsorted to LDS staging (S_CAP keys) and flushed as
//           sorted runs; they are not looked at again until the near set runs dry, when the
//           pivot is raised and the run prefixes / staging keys below it move to registers.
//   Invariant: every far key >= pivot, so a near key below the pivot is the global minimum.
#include "traverse_dev.h"


template <int LPR>
__global__ __launch_bounds__(64, 6) void trav_kernel(TravParams P) {
    __shared__ TravLds L;
    const uint32_t lane = threadIdx.x;
    const uint32_t q = blockIdx.x;
    TravHeader *H = P.hdr + q;
    int32_t status = H->status;
    if (status == 3 && H->n_scored < H->target) status = 0;  // target was raised: resume
    if (status != 0) return;
    L.claimtab[lane] = 0u; L.claimtab[lane + 64u] = 0u;

    uint64_t n_scored = H->n_scored, n_pops = H->n_pops, n_nbr = H->n_nbr, pq_used = H->pq_used,
             n_upper = H->n_upper, n_repivot = H->n_repivot, n_flush = H->n_flush;
    uint32_t cnt = H->stg_cnt, n_runs = H->n_runs, dq = H->dq;
    const uint32_t qpop = H->qpop;
    uint32_t primed = H->primed;
    const uint64_t target = H->target;
    unsigned long long pivot = H->pivot;

    unsigned long long *ht = P.ht + ((uint64_t)q << P.ht_log2);
    const uint32_t ht_shift = 32u - P.ht_log2;
    const uint32_t ht_mask = (1u << P.ht_log2) - 1u;
    unsigned long long *ut = P.ut + ((uint64_t)q << P.ut_log2);
    const uint32_t ut_shift = 64u - P.ut_log2;
    const uint32_t ut_mask = (1u << P.ut_log2) - 1u;
    const uint64_t ut_limit = ((uint64_t)1 << P.ut_log2) - ((uint64_t)1 << P.ut_log2) / 4;
    uint2 *scored = P.scored + (uint64_t)q * P.scored_cap;
    unsigned long long *pq = P.pq + (uint64_t)q * P.pq_cap;
    uint2 *runs = P.runs + (uint64_t)q * P.max_runs;
    unsigned long long *rhead = P.rhead + (uint64_t)q * P.max_runs;
    const uint32_t chunk = lane % LPR;
    const uint4 qv = P.queries[(uint64_t)q * LPR + chunk];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    // ---- restore near keys + staging -----------------------------------------
    unsigned long long k0 = RH_KEY_INF, k1 = RH_KEY_INF, k2 = RH_KEY_INF, k3 = RH_KEY_INF;
    if (primed) {
        const unsigned long long *rs = P.r_save + (uint64_t)q * (RK * 64);
        k0 = rs[lane]; k1 = rs[64 + lane]; k2 = rs[128 + lane]; k3 = rs[192 + lane];
    }
    for (uint32_t i = lane; i < cnt; i += 64) L.stg[i] = P.stg_save[(uint64_t)q * S_CAP + i];
    keytabs_init(L.kt, lane);
    WSYNC();

    // ---- far: append keys (one per lane where `has`) to staging ----------------
    auto far_append = [&](bool has, unsigned long long key) {
        const unsigned long long b = __ballot(has);
        if (b) {
            const uint32_t r = (uint32_t)__popcll(b & lt_mask);
            if (has) L.stg[cnt + r] = key;
            cnt += (uint32_t)__popcll(b);
            WSYNC();
        }
    };

    // ---- far: flush staging into a new sorted run --------------------------------
    auto flush = [&]() {
        if (cnt == 0) return;
        if (n_runs >= P.max_runs || pq_used + cnt > P.pq_cap) { status = RADHIP_E_CAPACITY; return; }
        uint32_t Pw = 2;
        while (Pw < cnt) Pw <<= 1;
        for (uint32_t i = cnt + lane; i < Pw; i += 64) L.stg[i] = RH_KEY_INF;
        WSYNC();
        lds_bitonic_sort(L.stg, Pw, lane);
        for (uint32_t i = lane; i < cnt; i += 64) pq[pq_used + i] = L.stg[i];
        if (lane == 0) {
            runs[n_runs] = make_uint2((uint32_t)pq_used, (uint32_t)(pq_used + cnt));
            rhead[n_runs] = L.stg[0];
        }
        n_runs++;
        n_flush++;
        pq_used += cnt;
        cnt = 0;
        __threadfence_block();  // run keys, {pos,end} and head are re-read by this wave later
        WSYNC();
    };

    // ---- raise the pivot and move every far key below it into registers ---------------
    auto repivot = [&]() {
        n_repivot++;
        if (cnt + 130u > S_CAP) { flush(); if (status) return; }
        unsigned long long fm = RH_KEY_INF;
        for (uint32_t i = lane; i < cnt; i += 64) { const unsigned long long v = L.stg[i]; fm = v < fm ? v : fm; }
        for (uint32_t r = lane; r < n_runs; r += 64) { const unsigned long long v = ld64(&rhead[r]); fm = v < fm ? v : fm; }
        fm = rh_wave_min_u64(fm);
        if (fm == RH_KEY_INF) { pivot = RH_KEY_INF; return; }   // far is empty
        const uint64_t nqv = (fm >> 38) + (uint64_t)dq;
        pivot = nqv >= (1ull << 24) ? RH_KEY_INF : (nqv << 38);
        unsigned long long newpiv = RH_KEY_INF;  // lane-private: smallest key this lane left in far below the pivot
        unsigned long long rej1 = RH_KEY_INF;    // at most one displaced near key per lane (goes to staging)
        // staging: a taken key leaves its slot; a displaced near key takes the slot over
        bool any_taken = false;
        for (uint32_t base = 0; base < cnt; base += 64) {
            const uint32_t i = base + lane;
            const unsigned long long v = i < cnt ? L.stg[i] : RH_KEY_INF;
            if (v < pivot) {
                if (k3 == RH_KEY_INF) { (void)r_insert(v, k0, k1, k2, k3); L.stg[i] = RH_KEY_INF; any_taken = true; }
                else if (v < k3) {
                    const unsigned long long rj = r_insert(v, k0, k1, k2, k3);
                    L.stg[i] = rj;
                    newpiv = rj < newpiv ? rj : newpiv;
                } else newpiv = v < newpiv ? v : newpiv;
            }
        }
        WSYNC();
        if (__ballot(any_taken)) {   // squeeze the holes out of staging
            uint32_t w = 0;
            for (uint32_t base = 0; base < cnt; base += 64) {
                const uint32_t i = base + lane;
                const unsigned long long v = i < cnt ? L.stg[i] : RH_KEY_INF;
                const bool keep = v != RH_KEY_INF;
                const unsigned long long b = __ballot(keep);
                if (keep) L.stg[w + (uint32_t)__popcll(b & lt_mask)] = v;
                w += (uint32_t)__popcll(b);
                WSYNC();
            }
            cnt = w;
        }
        // runs: lane l owns runs l, l+64, ...; a run gives up its prefix below the pivot
        for (uint32_t r = lane; r < n_runs; r += 64) {
            unsigned long long h = ld64(&rhead[r]);
            if (h >= pivot) continue;
            const unsigned long long pe64 = ld64(reinterpret_cast<const unsigned long long *>(&runs[r]));
            uint2 pe = make_uint2((uint32_t)pe64, (uint32_t)(pe64 >> 32));
            while (h < pivot) {
                if (k3 != RH_KEY_INF) {
                    if (h < k3 && rej1 == RH_KEY_INF) rej1 = r_insert(h, k0, k1, k2, k3);
                    else { newpiv = h < newpiv ? h : newpiv; break; }
                } else (void)r_insert(h, k0, k1, k2, k3);
                pe.x++;
                h = pe.x < pe.y ? ld64(&pq[pe.x]) : RH_KEY_INF;
            }
            runs[r] = pe;
            rhead[r] = h;
        }
        newpiv = rej1 < newpiv ? rej1 : newpiv;
        const unsigned long long np = rh_wave_min_u64(newpiv);
        const bool crowded = np != RH_KEY_INF;
        pivot = np < pivot ? np : pivot;
        far_append(rej1 != RH_KEY_INF, rej1);
        // adapt the pivot step to keep the registers about half full
        const uint32_t occ = (uint32_t)__popcll(__ballot(k0 != RH_KEY_INF)) + (uint32_t)__popcll(__ballot(k1 != RH_KEY_INF)) +
                             (uint32_t)__popcll(__ballot(k2 != RH_KEY_INF)) + (uint32_t)__popcll(__ballot(k3 != RH_KEY_INF));
        if (crowded) dq = dq > 1u ? dq >> 1 : 1u;
        else if (occ < 96u) dq = dq < DQ_MAX ? dq << 1 : DQ_MAX;
    };

    // ---- enqueue one key per lane where `push` ---------------------------------------------
    auto enqueue = [&](bool push, unsigned long long key) {
        const bool near = push && key < pivot;
        const unsigned long long rej = r_insert(near ? key : RH_KEY_INF, k0, k1, k2, k3);
        const bool rj = rej != RH_KEY_INF;
        if (__ballot(rj)) {   // a full lane pushed its largest key out: it becomes a far key
            const unsigned long long m = rh_wave_min_u64(rej);
            pivot = m < pivot ? m : pivot;
            far_append(rj, rej);
        }
        far_append(push && !near, key);
    };

    // ---- visited / scored / evaluate / enqueue for up to 64 candidate slots (one per lane,
    // distinct).  `prime` keeps the reference's unconditional queue insert (rad/traverser.py:158-168).
    auto process = [&](uint32_t slot, bool valid, uint32_t level, bool prime, uint32_t rot) {
        bool go = valid;
        if (level > 0) {
            if (go) {
                const bool fresh_u = ut_test_and_set(ut, ut_mask, ut_shift, ((unsigned long long)slot << 4) | level, P.epoch);
                go = fresh_u || prime;
            }
            n_upper += (uint64_t)__popcll(__ballot(go));
            if (n_upper > ut_limit) { status = RADHIP_E_CAPACITY; return; }
        }
        // Probe with plain (L2-served) loads: the table belongs to this wave alone, so the only
        // race is two lanes of THIS expansion wanting the same empty bucket — settled through an
        // LDS claim word.  No atomics reach HBM (they execute memory-side and capped v1).
        bool isnew = false;
        uint32_t h = 0, val = 0, ci = 0;
        {
            bool pending = go;          // still walking the probe sequence
            bool cand = false;          // stopped at an empty bucket, not yet confirmed
            if (go) h = RH_HT_HASH(slot);
            for (;;) {
                if (pending) {
                    for (;;) {
                        const unsigned long long e = ld64(&ht[h]);
                        if (ht_is_empty(e, P.epoch)) { cand = true; break; }
                        if ((uint32_t)e == slot) { val = (uint32_t)(e >> 32); break; }
                        h = (h + 1u) & ht_mask;
                    }
                    pending = false;
                }
                if (cand) {
                    cand = false;
                    if (claim_bucket<128u>(L.claimtab, h, ci)) isnew = true;                  // bucket is mine
                    else { pending = true; h = (h + 1u) & ht_mask; }                           // lost it: walk on
                }
                if (!__ballot(pending)) break;
            }
        }
        // scored before: re-enqueue on this level unless already visited here
        bool push_old = false;
        if (go && !isnew) {
            if (level == 0) {
                if (!(val & VAL_V0) || prime) {
                    st_relaxed(reinterpret_cast<uint32_t *>(&ht[h]) + 1, val | VAL_V0);
                    push_old = true;
                }
            } else push_old = true;
        }
        const unsigned long long nb = __ballot(isnew);
        const uint32_t nn = (uint32_t)__popcll(nb);
        if (nn) {
            const uint32_t rank = (uint32_t)__popcll(nb & lt_mask);
            if (isnew) { L.new_slot[rank] = slot; L.new_h[rank] = h; }
            WSYNC();
            constexpr uint32_t RPP = 64 / LPR;  // rows per pass
            // four row gathers in flight per lane before the first popcount
            for (uint32_t base = 0; base < nn; base += 4u * RPP) {
                uint4 v[4];
                uint32_t ri[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    ri[u] = base + (uint32_t)u * RPP + lane / LPR;
                    v[u] = make_uint4(0, 0, 0, 0);
                    if (ri[u] < nn) v[u] = P.fp[(uint64_t)L.new_slot[ri[u]] * LPR + chunk];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t rp = rh_group_sum<LPR>(rh_popc4(v[u]));
                    const uint32_t aa = rh_group_sum<LPR>(rh_popc4_and(v[u], qv));
                    if (ri[u] < nn && chunk == 0) { L.new_and[ri[u]] = aa; L.new_or[ri[u]] = qpop + rp - aa; }
                }
            }
            WSYNC();
            // finish the new nodes on lanes spread over the wave (so their keys land in
            // different lanes' registers): new index ni -> lane (ni << sh) + rot
            const uint32_t sh = (nn << P.spread_shift) <= 64u ? P.spread_shift : 0u;
            const uint32_t u = (lane - rot) & 63u;
            const uint32_t ni = u >> sh;
            const bool mine = ((u & ((1u << sh) - 1u)) == 0u) && ni < nn;
            unsigned long long key = RH_KEY_INF;
            if (mine) {
                const uint32_t s2 = L.new_slot[ni], a = L.new_and[ni], o = L.new_or[ni];
                __hip_atomic_store(&ht[L.new_h[ni]], (unsigned long long)s2 | ((unsigned long long)(a | (o << 12) | (level == 0 ? VAL_V0 : 0u) | (P.epoch << VAL_EPOCH_SHIFT)) << 32),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                scored[n_scored + ni] = make_uint2(s2, a | (o << 16));
                key = make_key_tab(L.kt, rh_q24_dev(a, o), s2, level);
            }
            n_scored += nn;
            enqueue(mine, key);
        }
        if (isnew) L.claimtab[ci] = 0u;   // entries are stored: the claims are spent
        if (__ballot(push_old)) {
            unsigned long long key = RH_KEY_INF;
            if (push_old) key = make_key_tab(L.kt, rh_q24_dev(val & 0xFFFu, (val >> 12) & 0xFFFu), slot, level);
            enqueue(push_old, key);
        }
    };

    // ---- same as process() for one adjacency row, with the fingerprint rows of ALL neighbours
    // gathered speculatively while the hash probes are in flight: one dependent HBM round trip
    // less per expansion, paid with the rows of already-scored neighbours.
    auto process_spec = [&](uint32_t slot, bool valid, uint32_t level, uint32_t rot) {
        constexpr uint32_t RPP = 64 / LPR;
        L.new_slot[lane] = valid ? slot : RADHIP_NO_SLOT;
        WSYNC();
        const uint32_t r0 = lane / LPR, r1 = (RPP + lane / LPR) & 63u;
        const uint32_t s0 = L.new_slot[r0];
        const uint32_t s1 = P.spec_passes > 1u ? L.new_slot[r1] : RADHIP_NO_SLOT;
        uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0;
        if (s0 != RADHIP_NO_SLOT) v0 = P.fp[(uint64_t)s0 * LPR + chunk];
        if (s1 != RADHIP_NO_SLOT) v1 = P.fp[(uint64_t)s1 * LPR + chunk];
        bool go = valid;
        if (level > 0) {
            if (go) {
                const bool fresh_u = ut_test_and_set(ut, ut_mask, ut_shift, ((unsigned long long)slot << 4) | level, P.epoch);
                go = fresh_u;
            }
            n_upper += (uint64_t)__popcll(__ballot(go));
            if (n_upper > ut_limit) { status = RADHIP_E_CAPACITY; return; }
        }
        bool isnew = false;
        uint32_t h = 0, val = 0, ci = 0;
        {
            bool pending = go, cand = false;
            if (go) h = RH_HT_HASH(slot);
            for (;;) {
                if (pending) {
                    for (;;) {
                        const unsigned long long e = ld64(&ht[h]);
                        if (ht_is_empty(e, P.epoch)) { cand = true; break; }
                        if ((uint32_t)e == slot) { val = (uint32_t)(e >> 32); break; }
                        h = (h + 1u) & ht_mask;
                    }
                    pending = false;
                }
                if (cand) {
                    cand = false;
                    if (claim_bucket<128u>(L.claimtab, h, ci)) isnew = true;
                    else { pending = true; h = (h + 1u) & ht_mask; }
                }
                if (!__ballot(pending)) break;
            }
        }
        bool push_old = false;
        if (go && !isnew) {
            if (level == 0) {
                if (!(val & VAL_V0)) {
                    st_relaxed(reinterpret_cast<uint32_t *>(&ht[h]) + 1, val | VAL_V0);
                    push_old = true;
                }
            } else push_old = true;
        }
        // counts of every gathered row
        {
            const uint32_t rp0 = rh_group_sum<LPR>(rh_popc4(v0)), aa0 = rh_group_sum<LPR>(rh_popc4_and(v0, qv));
            if (chunk == 0) { L.new_and[r0] = aa0; L.new_or[r0] = qpop + rp0 - aa0; }
            if (P.spec_passes > 1u) {
                const uint32_t rp1 = rh_group_sum<LPR>(rh_popc4(v1)), aa1 = rh_group_sum<LPR>(rh_popc4_and(v1, qv));
                if (chunk == 0) { L.new_and[r1] = aa1; L.new_or[r1] = qpop + rp1 - aa1; }
            }
        }
        const unsigned long long nb = __ballot(isnew);
        const uint32_t nn = (uint32_t)__popcll(nb);
        L.new_h[lane] = h;
        L.claim[lane] = isnew ? (uint32_t)__popcll(nb & lt_mask) : 0xFFFFFFFFu;
        WSYNC();
        if (nn) {
            const uint32_t sh = P.spread_shift;
            const uint32_t u = (lane - rot) & 63u;
            const uint32_t j = u >> sh;
            const uint32_t rk = ((u & ((1u << sh) - 1u)) == 0u) ? L.claim[j] : 0xFFFFFFFFu;
            const bool mine = rk != 0xFFFFFFFFu;
            unsigned long long key = RH_KEY_INF;
            if (mine) {
                const uint32_t s2 = L.new_slot[j], a = L.new_and[j], o = L.new_or[j];
                __hip_atomic_store(&ht[L.new_h[j]], (unsigned long long)s2 | ((unsigned long long)(a | (o << 12) | (level == 0 ? VAL_V0 : 0u) | (P.epoch << VAL_EPOCH_SHIFT)) << 32),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                scored[n_scored + rk] = make_uint2(s2, a | (o << 16));
                key = make_key_tab(L.kt, rh_q24_dev(a, o), s2, level);
            }
            n_scored += nn;
            enqueue(mine, key);
        }
        if (isnew) L.claimtab[ci] = 0u;
        if (__ballot(push_old)) {
            unsigned long long key = RH_KEY_INF;
            if (push_old) key = make_key_tab(L.kt, rh_q24_dev(val & 0xFFFu, (val >> 12) & 0xFFFu), slot, level);
            enqueue(push_old, key);
        }
        WSYNC();
    };

    // ---- prime ---------------------------------------------------------------
    if (!primed) {
        pivot = RH_KEY_INF;   // far is empty: everything is near until a lane overflows
        for (uint32_t base = 0; base < P.n_top && status == 0; base += 64) {
            if (cnt + 130u > S_CAP) flush();
            if (status) break;
            const uint32_t i = base + lane;
            const bool valid = i < P.n_top;
            const uint32_t slot = valid ? P.top[i] : RADHIP_NO_SLOT;
            process(slot, valid, (uint32_t)P.start_level, true, 0u);
        }
        primed = 1;
    }

    // ---- best-first loop -------------------------------------------------------
    uint64_t pops_here = 0;
    while (status == 0) {
        if (n_scored >= target) { status = target >= P.n_to_score ? 1 : 3; break; }
        if (P.max_pops && pops_here >= P.max_pops) break;
        if (cnt + 130u > S_CAP) { flush(); if (status) break; }
        // pop-min: near keys below the pivot are globally minimal
        unsigned long long mk = rh_wave_min_u64(k0);
        if (mk >= pivot) {
            if (mk == RH_KEY_INF && pivot == RH_KEY_INF && cnt == 0 && n_runs == 0) { status = 2; break; }
            repivot();
            if (status) break;
            mk = rh_wave_min_u64(k0);
            if (mk == RH_KEY_INF) { status = 2; break; }   // near and far both empty
            if (mk >= pivot) continue;                      // pivot was lowered below the near minimum
        }
        const int win = __ffsll((unsigned long long)__ballot(k0 == mk)) - 1;
        if ((int)lane == win) { k0 = k1; k1 = k2; k2 = k3; k3 = RH_KEY_INF; }
        const uint32_t node = key_slot_tab(L.kt, mk);
        const uint32_t level = rh_rank_level((uint32_t)mk & 0xFu);
        if (P.poplog_nodes && n_pops < P.poplog_cap && lane == 0) {
            P.poplog_nodes[(uint64_t)q * P.poplog_cap + n_pops] = node;
            P.poplog_levels[(uint64_t)q * P.poplog_cap + n_pops] = (uint8_t)level;
        }
        n_pops++;
        pops_here++;
        // adjacency row
        uint32_t nbr = RADHIP_NO_SLOT;
        if (level == 0) {
            if (lane < P.cap0) nbr = P.adj0[(uint64_t)node * P.cap0 + lane];
        } else {
            const uint32_t ur = P.upper_row[node];
            if (lane < P.capU) nbr = P.adjU[((uint64_t)ur + (level - 1u)) * P.capU + lane];
        }
        const bool valid = nbr != RADHIP_NO_SLOT;
        n_nbr += (uint64_t)__popcll(__ballot(valid));
        if (P.spec_passes) process_spec(nbr, valid, level, (uint32_t)(n_pops * 7u) & 63u);
        else process(nbr, valid, level, false, (uint32_t)(n_pops * 7u) & 63u);
        if (status) break;
        // descend: same node, one level down, same score
        if (level > 0) {
            const uint32_t nl = level - 1u;
            bool push0 = false;
            if (lane == 0) {
                if (nl > 0) {
                    push0 = ut_test_and_set(ut, ut_mask, ut_shift, ((unsigned long long)node << 4) | nl, P.epoch);
                } else {
                    uint32_t h = RH_HT_HASH(node);
                    unsigned long long e;
                    for (;;) {  // node is scored, hence present
                        e = __hip_atomic_load(&ht[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (ht_is_empty(e, P.epoch) || (uint32_t)e == node) break;  // empty unreachable by construction
                        h = (h + 1u) & ht_mask;
                    }
                    const uint32_t val = (uint32_t)(e >> 32);
                    if (!ht_is_empty(e, P.epoch) && (uint32_t)e == node && !(val & VAL_V0)) {
                        st_relaxed(reinterpret_cast<uint32_t *>(&ht[h]) + 1, val | VAL_V0);
                        push0 = true;
                    }
                }
            }
            const bool any0 = __ballot(push0) != 0;
            if (any0) {
                if (nl > 0) n_upper++;
                enqueue(push0, make_key_tab(L.kt, (uint32_t)(mk >> 38), node, nl));
            }
        }
    }

    // ---- frontier score: best key left in the queue -------------------------------
    unsigned long long fbest = k0;
    for (uint32_t i = lane; i < cnt; i += 64) fbest = L.stg[i] < fbest ? L.stg[i] : fbest;
    for (uint32_t r = lane; r < n_runs; r += 64) { const unsigned long long v = ld64(&rhead[r]); fbest = v < fbest ? v : fbest; }
    fbest = rh_wave_min_u64(fbest);
    // ---- persist ---------------------------------------------------------------
    {
        unsigned long long *rs = P.r_save + (uint64_t)q * (RK * 64);
        rs[lane] = k0; rs[64 + lane] = k1; rs[128 + lane] = k2; rs[192 + lane] = k3;
    }
    for (uint32_t i = lane; i < cnt; i += 64) P.stg_save[(uint64_t)q * S_CAP + i] = L.stg[i];
    if (lane == 0) {
        H->n_scored = n_scored; H->n_pops = n_pops; H->n_nbr = n_nbr; H->pq_used = pq_used;
        H->n_upper = n_upper; H->stg_cnt = cnt; H->n_runs = n_runs; H->primed = primed;
        H->pivot = pivot; H->dq = dq; H->n_repivot = n_repivot; H->n_flush = n_flush;
        H->status = status;
        H->frontier_key = fbest;
    }
}


// ================================================================== host side
struct radhip_traversal {
    radhip_index *idx = nullptr;
    uint32_t nq = 0;
    uint64_t n_to_score = 0;
    uint32_t flags = 0;
    TravParams P{};
    uint4 *d_queries = nullptr;
    size_t ht_bytes = 0, ut_bytes = 0, scored_bytes = 0, pq_bytes = 0, stg_bytes = 0, runs_bytes = 0,
           rhead_bytes = 0, rsave_bytes = 0, hdr_bytes = 0, log_bytes = 0;
    bool fresh_tables = true;   // tables not cleared yet (first upload)
    bool use4 = false;   // trav4_kernel (four traversals per wave)
    uint32_t resident4 = 0;   // traversals trav4_kernel holds resident on this device (grid = resident4 / 4 wavefronts)
    bool wide = false;   // ... its WIDE form: adjacency rows of 17..64 slots, walked in chunks of 16
    bool use_gt = false; // grouped visited/scored table (needs the index's graph-locality layout)
    bool use_bt = false; // bucket table: 16-B buckets of four entries, one request per probe (trav4_kernel's default)
    bool use_local = false;   // ... hashed by the graph-locality layout id instead of the slot (RADHIP_TABLE=local; needs the layout)
    size_t bt_bytes = 0;
    uint32_t epoch_max = EPOCH_LIMIT - 1u;   // last usable epoch of the table in use
    bool sharded = false; // the row-sharded form of trav4_kernel (shard.hip): stepped, never run()
    size_t gt_bytes = 0;
    uint64_t graph_gen = 0;   // generation of the index this state was sized for
    uint32_t ht_log2 = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double kernel_ms = 0.0;
    uint64_t launches = 0;
    uint64_t state_bytes = 0;
    // round 4: heavy state per resident row instead of per traversal (RADHIP_TRAV_SLOTS), and launches that do not wait
    // (radhip_traversal_start / _finish on a stream of the object's own, RADHIP_TRAV_OWN_STREAM)
    uint32_t sn = 0;               // state sets allocated: nq, or the slots
    uint32_t n_active = 0;         // traversals of the batch that is armed (<= nq: radhip_traversal_reset_count)
    uint32_t list_ring = 0;        // scored lists allocated when fewer than nq (a power of two): traversal i writes list i mod ring
    size_t slot_epoch_bytes = 0;
    hipStream_t stream = nullptr;  // the index's stream, or the object's own
    bool own_stream = false;
    bool in_flight = false;        // started, not finished yet
    bool first_pending = false;    // the launch in flight is the first of its batch (capacity fallbacks may re-run it)
};

static uint32_t log2_ceil(uint64_t x) {
    uint32_t l = 0;
    while (((uint64_t)1 << l) < x) l++;
    return l;
}

// Arm the first `na` traversals on the device: a fresh header each (its popcount formed from the query row that is already
// there), so that a chain of ten bench steps (655360 traversals) moves its 84 MB of queries to the device once and nothing else.
__global__ __launch_bounds__(256) void trav_arm_kernel(TravHeader *hdr, const uint4 *queries, uint32_t row_vec, uint32_t na, uint64_t target) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= na) return;
    const uint4 *q = queries + (size_t)i * row_vec;
    uint32_t p = 0;
    for (uint32_t w = 0; w < row_vec; ++w) { const uint4 v = q[w]; p += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w); }
    TravHeader h = {};
    h.qpop = p;
    h.target = target;
    h.frontier_key = RH_KEY_INF;
    h.pivot = RH_KEY_INF;
    h.dq = DQ_INIT;
    h.mid_limit = RH_KEY_INF;
    h.far_min = RH_KEY_INF;
    h.dm = 1u << 12;
    hdr[i] = h;
}
// the statistics of the first `na` traversals, packed for the host (56 bytes each instead of the whole header)
__global__ __launch_bounds__(256) void trav_stats_kernel(const TravHeader *hdr, radhip_trav_stats_t *out, uint32_t na) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= na) return;
    const TravHeader &h = hdr[i];
    radhip_trav_stats_t o;
    o.n_scored = h.n_scored; o.n_pops = h.n_pops; o.n_nbr = h.n_nbr; o.n_repivot = h.n_repivot; o.n_flush = h.n_flush;
    o.status = h.status; o.n_remid = (int32_t)h.n_remid; o.n_upper = h.n_upper;
    out[i] = o;
}

// queries == nullptr: the batch's queries are on the device already (the capacity fallbacks of run() re-arm the batch themselves)
static int trav_upload_queries(radhip_traversal *t, const uint8_t *queries) {
    radhip_index *idx = t->idx;
    const uint32_t na = t->n_active;       // (the first n_active traversals of the object are armed)
    t->P.nq = na;
    if (t->in_flight) RH_FAIL(RADHIP_E_STATE, "the traversal object has a launch in flight: radhip_traversal_finish first");
    hipStream_t st = t->stream;
    if (queries) {
        if (idx->row_stride == idx->row_bytes) {
            RH_HIP(hipMemcpyAsync(t->d_queries, queries, (size_t)na * idx->row_bytes, hipMemcpyHostToDevice, st));
        } else {   // rows padded to the device's stride: the padding is zero
            RH_HIP(hipMemsetAsync(t->d_queries, 0, (size_t)na * idx->row_stride, st));
            RH_HIP(hipMemcpy2DAsync(t->d_queries, idx->row_stride, queries, idx->row_bytes, idx->row_bytes, na, hipMemcpyHostToDevice, st));
        }
    }
    hipLaunchKernelGGL(trav_arm_kernel, dim3((na + 255u) / 256u), dim3(256), 0, st, t->P.hdr, (const uint4 *)t->d_queries,
                       (uint32_t)(idx->row_stride / 16u), na, (uint64_t)t->n_to_score);
    RH_HIP(hipGetLastError());
    if (t->P.slots) {
        // per-row epochs live on the device (a row bumps its own when it takes a traversal and clears its own tables when
        // they run out): re-arming a batch touches no table at all
        if (t->fresh_tables) {
            t->fresh_tables = false;
            if (t->P.ht) RH_HIP(hipMemsetAsync(t->P.ht, 0xFF, t->ht_bytes, st));
            if (t->P.gt) RH_HIP(hipMemsetAsync(t->P.gt, 0xFF, t->gt_bytes, st));
            if (t->P.bt) RH_HIP(hipMemsetAsync(t->P.bt, 0x00, t->bt_bytes, st));
            RH_HIP(hipMemsetAsync(t->P.ut, 0x00, t->ut_bytes, st));
            // "nothing yet": epoch_first - 1 (the bucket table's live epochs start at 1, the others' at 0 = 0xFFFFFFFF + 1)
            RH_HIP(hipMemsetAsync(t->P.slot_epoch, t->use_bt ? 0x00 : 0xFF, t->slot_epoch_bytes, st));
        }
    } else
    if (t->fresh_tables || ++t->P.epoch > t->epoch_max) {   // first use, or epoch space exhausted: really clear
        // (the bucket table's cleared state is 0 = epoch 0, so its live epochs start at 1; the other tables clear to
        // all-ones = epoch 0x7F and count from 0.  At 1B rows a bucket entry has one epoch bit: cleared every batch.)
        t->P.epoch = t->use_bt ? 1u : 0u;
        t->fresh_tables = false;
        if (t->P.ht) RH_HIP(hipMemsetAsync(t->P.ht, 0xFF, t->ht_bytes, st));
        if (t->P.gt) RH_HIP(hipMemsetAsync(t->P.gt, 0xFF, t->gt_bytes, st));
        if (t->P.bt) RH_HIP(hipMemsetAsync(t->P.bt, 0x00, t->bt_bytes, st));
        RH_HIP(hipMemsetAsync(t->P.ut, 0x00, t->ut_bytes, st));
    }
    RH_HIP(hipStreamSynchronize(st));
    t->kernel_ms = 0.0;
    t->launches = 0;
    return RADHIP_OK;
}

extern "C" int radhip_traversal_destroy(radhip_traversal_t *t) {
    if (!t) return RADHIP_OK;
    if (t->idx && t->idx->dev_ready) (void)hipSetDevice(t->idx->device);
    if (t->d_queries) (void)hipFree(t->d_queries);
    if (t->P.hdr) (void)hipFree(t->P.hdr);
    if (t->P.ht) (void)hipFree(t->P.ht);
    if (t->P.gt) (void)hipFree(t->P.gt);
    if (t->P.bt) (void)hipFree(t->P.bt);
    if (t->P.ut) (void)hipFree(t->P.ut);
    if (t->P.scored) (void)hipFree(t->P.scored);
    if (t->P.pq) (void)hipFree(t->P.pq);
    if (t->P.stg_save) (void)hipFree(t->P.stg_save);
    if (t->P.runs) (void)hipFree(t->P.runs);
    if (t->P.rhead) (void)hipFree(t->P.rhead);
    if (t->P.midpool) (void)hipFree(t->P.midpool);
    if (t->P.q_next) (void)hipFree(t->P.q_next);
    if (t->P.r_save) (void)hipFree(t->P.r_save);
    if (t->P.sh_pend_h) (void)hipFree(t->P.sh_pend_h);
    if (t->P.slot_epoch) (void)hipFree(t->P.slot_epoch);
    if (t->own_stream && t->stream) { (void)hipStreamSynchronize(t->stream); (void)hipStreamDestroy(t->stream); }
    if (t->P.poplog_nodes) (void)hipFree(t->P.poplog_nodes);
    if (t->P.poplog_levels) (void)hipFree(t->P.poplog_levels);
    if (t->ev0) (void)hipEventDestroy(t->ev0);
    if (t->ev1) (void)hipEventDestroy(t->ev1);
    delete t;
    return RADHIP_OK;
}

static int trav_base_event(radhip_index *idx, hipEvent_t *out);
static bool trav4_shape_ok(const radhip_index *idx);
static int trav_forced_kernel();
static int trav_capacity_of(radhip_index *idx, bool use4, uint32_t *out);

static int trav_create_impl(radhip_index_t *idx, const uint8_t *queries, uint32_t nq, uint64_t n_to_score, uint32_t flags,
                            bool sharded, radhip_traversal_t **out, uint32_t list_ring = 0) {
    if (!idx || !queries || !out) RH_FAIL(RADHIP_E_INVALID, "null argument");
    if (nq == 0) RH_FAIL(RADHIP_E_INVALID, "nq must be > 0");
    if (n_to_score == 0) RH_FAIL(RADHIP_E_INVALID, "n_to_score must be > 0");
    if (!idx->has_vectors || !idx->has_graph) RH_FAIL(RADHIP_E_STATE, "index needs vectors and a graph");
    if (!sharded) {   // (the sharded form never reads a fingerprint: the rows may be another rank's)
        RH_REQUIRE_FULL_CORPUS(idx);
        if (idx->g_n > idx->n) RH_FAIL(RADHIP_E_STATE, "graph has more nodes (%llu) than the corpus has rows (%llu)",
                                       (unsigned long long)idx->g_n, (unsigned long long)idx->n);
    }
    if (idx->g_n > 1000000000ull) RH_FAIL(RADHIP_E_INVALID, "RAD traversal needs slots < 1e9");
    std::lock_guard<std::mutex> lk(idx->mu);
    RH_TRY(rh_ensure_device(idx));
    {   // RADHIP_TABLE=group on an index without a layout: compute one first (how the parity suites cover
        // the grouped table on every graph they build)
        const char *e = getenv("RADHIP_TABLE");
        if (e && (e[0] == 'g' || e[0] == 'l') && !idx->layout_valid) RH_TRY(rh_optimize_layout_locked(idx, 0));
    }
    radhip_traversal *t = new (std::nothrow) radhip_traversal();
    if (!t) RH_FAIL(RADHIP_E_NOMEM, "out of host memory");
    t->idx = idx; t->nq = nq; t->n_active = nq; t->n_to_score = n_to_score; t->flags = flags; t->sharded = sharded;
    t->stream = idx->stream;
    if ((flags & RADHIP_TRAV_OWN_STREAM) && !sharded) {
        if (hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking) != hipSuccess) { delete t; (void)hipGetLastError(); RH_FAIL(RADHIP_E_HIP, "hipStreamCreate failed"); }
        t->own_stream = true;
    }
    if (sharded) {
        if (!trav4_shape_ok(idx)) { delete t; RH_FAIL(RADHIP_E_INVALID, "the wave engine needs adjacency rows of at most 16 slots"); }
        t->use4 = true;
    } else {
        const int forced = trav_forced_kernel();
        // rows wider than 16 slots (connectivity > 8): trav4_kernel's WIDE form walks a row in chunks of 16 (bucket or grouped
        // table: RADHIP_TABLE=hash keeps such an index on trav_kernel)
        const char *tb = getenv("RADHIP_TABLE");
        const bool narrow = trav4_shape_ok(idx), wide = !narrow && idx->cap0 <= 64 && idx->M <= 64 && !(tb && tb[0] == 'h');
        // measured (profiles/r03): rows of 32 slots (the reference notebook's connectivity 16) are 14-40 % faster on the WIDE
        // form than on trav_kernel; rows of 64 (BASELINE config[4]: four chunks, four gather passes per pop) are not —
        // those stay on trav_kernel unless RADHIP_TRAV=4 asks
        const bool wide_auto = wide && idx->cap0 <= 32 && idx->M <= 32;
        t->use4 = (narrow || (forced == 4 ? wide : wide_auto)) && forced != 1;
        t->wide = t->use4 && wide;
        if (t->use4 && forced != 4) {   // auto: four per wave beyond one resident round (and a quarter) of trav_kernel
            uint32_t cap1 = 0;
            int rc1 = trav_capacity_of(idx, false, &cap1);
            if (rc1 != RADHIP_OK) { delete t; return rc1; }
            t->use4 = (uint64_t)nq * 4ull > 5ull * cap1;
            t->wide = t->use4 && wide;
        }
    }
    const uint64_t n_top = idx->n_top;
    if (n_to_score > idx->g_n) n_to_score = idx->g_n;  // cannot score more than exist
    t->n_to_score = n_to_score;
    // a multiple of 16 entries: every traversal's scored list starts on a 128-B line
    const uint64_t scored_cap = (n_to_score + 64 + n_top + 15) & ~(uint64_t)15;
    const uint32_t ht_log2 = std::max<uint32_t>(10, log2_ceil(2 * scored_cap));
    const uint64_t up_pairs = idx->n_upper_rows + n_top * (uint64_t)(idx->max_level + 1);
    // (node, level >= 1) visits of a traversal: measured at most 0.11 x n_to_score on the bench graphs (mean 0.035,
    // connectivity 8; profiles/r03) — the set is sized for 2 / connectivity of the scored nodes at 3/4 load, and the key
    // pool (every live queue key) with it; a traversal that outgrows them re-runs with four times the room (trav_grow_upper)
    uint64_t ut_need = std::min<uint64_t>(2 * up_pairs + 64, scored_cap * 2 / idx->M + 4096);
    const uint32_t ut_log2 = std::max<uint32_t>(10, log2_ceil(ut_need));
    if (ht_log2 > 31 || ut_log2 > 31) { delete t; RH_FAIL(RADHIP_E_INVALID, "n_to_score too large"); }
    // every live queue key fits: a scored node is in the queue once per level at most; trav4_kernel collects
    // the garbage (consumed run prefixes) when the pool fills up
    const uint64_t pq_cap = scored_cap + ((uint64_t)1 << ut_log2);
    if (pq_cap >= 0xFFFFFFFFull) { delete t; RH_FAIL(RADHIP_E_INVALID, "n_to_score too large"); }
    TravParams &P = t->P;
    P.fp = idx->d_fp; P.adj0 = idx->d_adj0; P.upper_row = idx->d_upper_row; P.adjU = idx->d_adjU;
    P.top = idx->d_top; P.n_top = idx->n_top; P.cap0 = idx->cap0; P.capU = idx->M; P.nq = nq;
    P.start_level = idx->max_level > 0 ? idx->max_level - 1 : 0;
    P.n_to_score = n_to_score; P.max_pops = 0;
    {   // spread stride of new keys over the lanes: 64 / pow2ceil(widest adjacency row)
        uint32_t w = std::max<uint32_t>(idx->cap0, idx->M), p2 = 1, sh = 6;
        while (p2 < w) { p2 <<= 1; }
        while ((p2 << sh) > 64u && sh > 0) sh--;
        P.spread_shift = sh;
        const uint32_t rpp = 64u / idx->lpr;
        const uint32_t passes = (p2 + rpp - 1u) / rpp;   // p2 = pow2ceil(widest row) = lanes the spread covers
        P.spec_passes = passes <= 2u ? passes : 0u;
        // RADHIP_SPEC=0: no speculative gathers (profiling knob; measured 2-12 % slower at every batch size)
        if (getenv("RADHIP_SPEC") && getenv("RADHIP_SPEC")[0] == '0') P.spec_passes = 0u;
    }
    P.ht_log2 = ht_log2; P.ut_log2 = ut_log2; P.scored_cap = scored_cap; P.pq_cap = pq_cap;
    t->ht_log2 = ht_log2;
    t->graph_gen = idx->graph_gen;
    {   // Grouped table (RADHIP_TABLE=group; needs the index's graph-locality layout).  It cuts the table lines of an
        // expansion from one per neighbour to one per group the neighbours span (3.4 instead of 9.9 on the bench
        // graph), but the bench kernel is not bound by lines on graphs that have such locality (few new nodes per
        // expansion: latency-bound), and graphs whose expansions ARE line-bound (round 1's corpus) have no locality to
        // group by: measured 2-5 % slower than the per-slot table on every workload of profiles/r02, so the library
        // never picks it by itself.  8 chunks per line, one chunk per scored node in the worst case, <= 80 % load.
        const char *e = getenv("RADHIP_TABLE");
        const bool force_group = e && e[0] == 'g';
        const uint32_t gt_log2 = std::max<uint32_t>(7, log2_ceil((scored_cap * 5 + 31) / 32));
        const bool can = idx->layout_valid && idx->d_adjx0 && t->use4 && gt_log2 <= 17 && !sharded;
        t->use_gt = can && force_group;
        P.gt_log2 = gt_log2;
        // Bucket table: trav4_kernel's default (RADHIP_TABLE=hash keeps the one-entry-per-probe table, for A/B runs).
        // 2.5 entries per scored node at least (load <= 0.4: a probe needs a second bucket once in ~100).
        const bool force_hash = e && e[0] == 'h';
        t->use_bt = t->use4 && !t->use_gt && !sharded && !force_hash;
        P.bt_log2 = std::max<uint32_t>(6, log2_ceil((scored_cap * 5 + 7) / 8));
        // (the locality-hashed form gives every layout id a bucket of its own: twice the buckets — 2 MB per row of state at
        // n_to_score = 100k — keep two blocks of 8 ids from meeting on one line too often; measured +5 %, profiles/r04)
        if (e && e[0] == 'l' && idx->layout_valid && idx->d_adjx0 && t->use4 && !sharded && P.bt_log2 < 28) P.bt_log2 += 1;
        if (const char *x = getenv("RADHIP_BT_LOG2_ADD")) P.bt_log2 += (uint32_t)std::min(3, std::max(0, atoi(x)));   // (experiments: a larger table)
        P.bt_sbits = std::max<uint32_t>(8, log2_ceil(idx->g_n + 2));
        if (P.bt_sbits > 30 || P.bt_log2 > 28) t->use_bt = false;
        if (t->wide && !t->use_bt && !t->use_gt) { t->wide = false; t->use4 = false; }   // (the WIDE form has no per-slot hash table variant)
        if (t->use_bt) t->epoch_max = (1u << (31u - P.bt_sbits)) - 1u;
        // the locality-hashed form of the bucket table: same entries, the home bucket comes from the layout id (pair rows)
        t->use_local = t->use_bt && e && e[0] == 'l' && idx->layout_valid && idx->d_adjx0;
        P.adjx0 = idx->d_adjx0; P.adjxU = idx->d_adjxU; P.topx = idx->d_topx; P.lid = idx->d_lid;
    }
    // RADHIP_TRAV_SLOTS: tables, key pool, run table and mid pool once per RESIDENT ROW of trav4_kernel instead of once per
    // traversal (bucket or grouped table; the flag is ignored where that kernel is not the one that runs).  The rows are
    // the device's: a batch of any size needs the same 40 GB (n_to_score = 100k) plus 0.8 MB of scored list per traversal.
    t->sn = nq;
    if ((flags & RADHIP_TRAV_SLOTS) && t->use4 && (t->use_bt || t->use_gt) && !sharded) {
        uint32_t c = 0;
        int rc4 = trav_capacity_of(idx, true, &c);
        if (rc4 != RADHIP_OK) { radhip_traversal_destroy(t); return rc4; }
        t->resident4 = c ? c : 4u;
        if (nq > t->resident4) { t->sn = t->resident4; P.slots = t->sn; }   // (a batch that fits one resident round gains nothing)
        // (test hooks: a handful of rows for a small batch, so that every row works through many traversals; few epochs,
        // so that rows run out of them and clear their own tables)
        if (const char *e = getenv("RADHIP_TEST_SLOTS")) { const int v = atoi(e); if (v >= 4 && (uint32_t)v < nq) { t->sn = ((uint32_t)v + 3u) & ~3u; P.slots = t->sn; } }
        if (const char *e = getenv("RADHIP_TEST_EPOCH_MAX")) { const int v = atoi(e); if (v >= 1 && (uint32_t)v < t->epoch_max) t->epoch_max = (uint32_t)v; }
    }
    const size_t sn = t->sn;
    P.epoch_first = t->use_bt ? 1u : 0u;
    P.epoch_max = t->epoch_max;
    t->hdr_bytes = (size_t)nq * sizeof(TravHeader);
    t->gt_bytes = t->use_gt ? (sn << (P.gt_log2 + 4)) * 8 : 0;
    t->bt_bytes = t->use_bt ? (sn << (P.bt_log2 + 2)) * 4 : 0;
    t->ht_bytes = (t->use_gt || t->use_bt) ? 0 : (sn << ht_log2) * 8;
    t->ut_bytes = (sn << ut_log2) * 8;
    // a ring of scored lists for chained batches (per-row state only): traversal i writes list i mod ring, gated on traversal
    // i - ring being done; at least 4 resident rounds of lists, so that the gate never waits in practice
    P.list_mask = 0xFFFFFFFFu;
    if (list_ring && P.slots) {
        uint32_t r = 1; while (r < list_ring) r <<= 1;
        if (r < 2u * P.slots) r = 2u * P.slots;
        { uint32_t r2 = 1; while (r2 < r) r2 <<= 1; r = r2; }
        if (r < nq) { t->list_ring = r; P.list_mask = r - 1u; }
    }
    t->scored_bytes = (size_t)(t->list_ring ? t->list_ring : nq) * scored_cap * sizeof(uint2);
    t->pq_bytes = sn * pq_cap * 8;
    t->stg_bytes = P.slots ? 0 : (size_t)nq * S_CAP * 8;
    // a far run is written by a staging flush (64-256 keys) and stays in the table while it holds a key:
    // scored_cap / 48 entries cover that with a margin, 8192 at least (exhausted runs are collected on the device)
    P.max_runs = std::max<uint32_t>(MAX_RUNS, 1u << log2_ceil(scored_cap / 48 + 1));
    t->runs_bytes = sn * P.max_runs * sizeof(uint2);
    t->rhead_bytes = sn * P.max_runs * 8;
    t->rsave_bytes = P.slots ? 0 : (size_t)nq * RK * 64 * 8;
    t->slot_epoch_bytes = P.slots ? sn * 4 : 0;
    int rc = RADHIP_OK;
#define RH_A(ptr, bytes)                                                                   \
    do {                                                                                   \
        hipError_t e_ = hipMalloc((void **)&(ptr), (bytes) ? (bytes) : 16);                \
        if (e_ != hipSuccess) {                                                            \
            radhip_set_error("hipMalloc(%zu) for traversal state failed: %s", (size_t)(bytes), hipGetErrorString(e_)); \
            rc = e_ == hipErrorOutOfMemory ? RADHIP_E_NOMEM : RADHIP_E_HIP;                \
            (void)hipGetLastError();                                                       \
        } else t->state_bytes += (bytes);                                                  \
    } while (0)
    RH_A(t->d_queries, (size_t)nq * idx->row_stride);
    if (rc == 0) RH_A(P.hdr, t->hdr_bytes);
    if (rc == 0 && t->ht_bytes) RH_A(P.ht, t->ht_bytes);
    if (rc == 0 && t->gt_bytes) RH_A(P.gt, t->gt_bytes);
    if (rc == 0 && t->bt_bytes) RH_A(P.bt, t->bt_bytes);
    if (rc == 0) RH_A(P.ut, t->ut_bytes);
    if (rc == 0) RH_A(P.scored, t->scored_bytes);
    if (rc == 0) RH_A(P.pq, t->pq_bytes);
    if (rc == 0) RH_A(P.stg_save, t->stg_bytes);
    if (rc == 0) RH_A(P.runs, t->runs_bytes);
    if (rc == 0) RH_A(P.rhead, t->rhead_bytes);
    if (rc == 0) RH_A(P.midpool, sn * 256 * 8);
    if (rc == 0) RH_A(P.q_next, 64);
    if (rc == 0) RH_A(P.r_save, t->rsave_bytes);
    if (rc == 0 && P.slots) RH_A(P.slot_epoch, t->slot_epoch_bytes);
    if (rc == 0 && sharded) RH_A(P.sh_pend_h, (size_t)nq * 16 * 4);
    if (rc == 0 && (flags & RADHIP_TRAV_LOG_POPS)) {
        P.poplog_cap = pq_cap;
        RH_A(P.poplog_nodes, (size_t)nq * P.poplog_cap * 4);
        if (rc == 0) RH_A(P.poplog_levels, (size_t)nq * P.poplog_cap);
    }
#undef RH_A
    if (rc == 0 && hipEventCreate(&t->ev0) != hipSuccess) rc = RADHIP_E_HIP;
    if (rc == 0 && hipEventCreate(&t->ev1) != hipSuccess) rc = RADHIP_E_HIP;
    if (rc != 0) { radhip_traversal_destroy(t); return rc; }
    P.queries = t->d_queries;
    { hipEvent_t base_; (void)trav_base_event(idx, &base_); }
    rc = trav_upload_queries(t, queries);
    if (rc != 0) { radhip_traversal_destroy(t); return rc; }
    *out = t;
    return RADHIP_OK;
}

extern "C" int radhip_traversal_create(radhip_index_t *idx, const uint8_t *queries, uint32_t nq,
                                       uint64_t n_to_score, uint32_t flags, radhip_traversal_t **out) {
    return trav_create_impl(idx, queries, nq, n_to_score, flags, false, out);
}
// ... with a RING of scored lists (RADHIP_TRAV_SLOTS only; ignored otherwise): a batch of many resident rounds — several bench
// steps chained into ONE launch, so that the tail of a launch (its longest traversals running alone) is paid once — keeps the
// lists of its last `list_ring` traversals (rounded up to a power of two, at least two resident rounds); every traversal's
// counts stay in the statistics.  What a consumer that drains results as they complete would see (rad/scored.py:63-85).
extern "C" int radhip_traversal_create_ring(radhip_index_t *idx, const uint8_t *queries, uint32_t nq, uint64_t n_to_score,
                                            uint32_t flags, uint32_t list_ring, radhip_traversal_t **out) {
    return trav_create_impl(idx, queries, nq, n_to_score, flags | RADHIP_TRAV_SLOTS, false, out, list_ring);
}
extern "C" uint32_t radhip_traversal_list_ring(const radhip_traversal_t *t) { return t ? t->list_ring : 0; }

// ---- the row-sharded form (shard.hip owns the exchange buffers and the loop) ---------------------------------
int rh_trav_create_sharded(radhip_index *idx, const uint8_t *queries, uint32_t nq, uint64_t n_to_score, uint32_t flags,
                           radhip_traversal **out) {
    return trav_create_impl(idx, queries, nq, n_to_score, flags, true, out);
}
void rh_trav_bind_shard(radhip_traversal *t, uint32_t *d_req, const uint32_t *d_in, uint32_t W, uint32_t max_inner) {
    t->P.sh_req = d_req; t->P.sh_in = d_in; t->P.sh_W = W; t->P.max_pops = max_inner;
}
// one frontier step of every local traversal, enqueued on the index's stream (no synchronisation; caller holds idx->mu)
int rh_trav_enqueue_shard_step(radhip_traversal *t) {
    radhip_index *idx = t->idx;
    if (!t->sharded || !t->P.sh_req) RH_FAIL(RADHIP_E_STATE, "not a sharded traversal");
    const uint32_t grid = (t->nq + 3u) / 4u;
    // (the fingerprint width only matters to the gather, which this form does not have: one instantiation)
    RH_TRY(rh_trav4_launch_sharded(grid, idx->stream, t->P));
    t->launches++;
    return RADHIP_OK;
}
uint64_t rh_trav_graph_gen(const radhip_traversal *t) { return t->graph_gen; }

static int trav_reset_impl(radhip_traversal_t *t, const uint8_t *queries, uint32_t count) {
    if (!t || !queries) RH_FAIL(RADHIP_E_INVALID, "null argument");
    if (count == 0 || count > t->nq) RH_FAIL(RADHIP_E_RANGE, "a batch of this object holds 1..%u traversals (got %u)", t->nq, count);
    std::lock_guard<std::mutex> lk(t->idx->mu);
    if (t->graph_gen != t->idx->graph_gen)
        RH_FAIL(RADHIP_E_STATE, "the index changed since this traversal object was created: create a new one");
    RH_HIP(hipSetDevice(t->idx->device));
    if (t->in_flight) RH_FAIL(RADHIP_E_STATE, "the traversal object has a launch in flight: radhip_traversal_finish first");
    t->n_active = count;
    return trav_upload_queries(t, queries);
}
extern "C" int radhip_traversal_reset(radhip_traversal_t *t, const uint8_t *queries) {
    return trav_reset_impl(t, queries, t ? t->nq : 0);
}
// re-arm the first `count` (<= nq) traversals of the object with new queries; the others rest
extern "C" int radhip_traversal_reset_count(radhip_traversal_t *t, const uint8_t *queries, uint32_t count) {
    return trav_reset_impl(t, queries, count);
}

// one launch of the kernel the object is bound to, enqueued on the object's stream between its two events (no wait)
static int trav_enqueue(radhip_traversal *t) {
    radhip_index *idx = t->idx;
    hipStream_t st = t->stream;
#ifdef RH_PROFILE
    static unsigned long long *d_prof = nullptr;
    if (!d_prof) { (void)hipMalloc((void **)&d_prof, 256); }
    (void)hipMemset(d_prof, 0, 256);
    t->P.prof = d_prof;
#else
    t->P.prof = nullptr;
#endif
    // trav4_kernel: as many wavefronts as the device holds resident; their rows take the traversals of the batch from
    // a counter, one after the other (traverse4.inc)
    uint32_t grid4 = (t->n_active + 3u) / 4u;
    if (t->use4) {
        if (t->resident4 == 0) { uint32_t c = 0; RH_TRY(trav_capacity_of(idx, true, &c)); t->resident4 = c ? c : 4u; }
        // Rows that take their traversals from the counter: +2-4 % on rows of <= 16 slots (no row waits for the longest of
        // its wavefront's four), a draw on the WIDE form (-2.5 % / -0.4 % at two / four resident rounds, one binary:
        // profiles/r03/wide_rows_static_vs_counter.log), which keeps the static assignment.  RADHIP_TRAV_STATIC=1 / 0
        // forces either.
        t->P.q_static = t->wide ? 1u : 0u;
        if (const char *e = getenv("RADHIP_TRAV_STATIC")) t->P.q_static = e[0] == '1' ? 1u : 0u;
        if (t->P.slots) t->P.q_static = 0u;   // (a row's state serves whatever traversals the row takes)
        if (!t->P.q_static) grid4 = std::min<uint32_t>(grid4, t->resident4 / 4u);
        // (test hook: a grid of a few wavefronts, so that small batches exercise rows that take many traversals in a row)
        if (const char *e = getenv("RADHIP_TEST_GRID")) { const int v = atoi(e); if (v > 0 && !t->P.q_static) grid4 = std::min<uint32_t>(grid4, (uint32_t)v); }
        if (t->P.slots && grid4 * 4u > t->P.slots) grid4 = t->P.slots / 4u;
        RH_HIP(hipMemsetAsync(t->P.q_next, 0, 4, st));
    }
    RH_HIP(hipEventRecord(t->ev0, st));
    if (t->use4) {
        RH_TRY(rh_trav4_launch(t->use_gt ? RH_T4_GROUPED : t->use_local ? RH_T4_LOCAL : t->use_bt ? RH_T4_BUCKET : RH_T4_HASH, t->wide, t->P.slots != 0u, (int)idx->lpr, grid4, st, t->P));
    } else {
        switch (idx->lpr) {
            case 1: hipLaunchKernelGGL((trav_kernel<1>), dim3(t->n_active), dim3(64), 0, st, t->P); break;
            case 2: hipLaunchKernelGGL((trav_kernel<2>), dim3(t->n_active), dim3(64), 0, st, t->P); break;
            case 4: hipLaunchKernelGGL((trav_kernel<4>), dim3(t->n_active), dim3(64), 0, st, t->P); break;
            case 8: hipLaunchKernelGGL((trav_kernel<8>), dim3(t->n_active), dim3(64), 0, st, t->P); break;
            default: hipLaunchKernelGGL((trav_kernel<16>), dim3(t->n_active), dim3(64), 0, st, t->P); break;
        }
    }
    RH_HIP(hipGetLastError());
    RH_HIP(hipEventRecord(t->ev1, st));
    return RADHIP_OK;
}
// ... and its end: wait for the stream, kernel time accumulated
static int trav_collect(radhip_traversal *t) {
    RH_HIP(hipStreamSynchronize(t->stream));
    float ms = 0.f;
    RH_HIP(hipEventElapsedTime(&ms, t->ev0, t->ev1));
#ifdef RH_PROFILE
    {
        unsigned long long hp[32];
        (void)hipMemcpy(hp, t->P.prof, 256, hipMemcpyDeviceToHost);
        unsigned long long tot = 0; for (int i = 0; i < 9; ++i) tot += hp[i];
        const char *nm[10] = {"loophead", "flush", "pop", "decode+adj", "probe", "eval", "finish+enqueue", "old+descent-pre", "repivot", ""};
        fprintf(stderr, "[prof] total %llu cycles:", tot);
        for (int i = 0; i < 9; ++i) fprintf(stderr, " %s=%.1f%%", nm[i], 100.0 * hp[i] / (tot ? tot : 1));
        fprintf(stderr, "; %llu re-pivot events of a wavefront, %.0f cycles each\n", hp[9], hp[9] ? (double)hp[8] / hp[9] : 0.0);
        fprintf(stderr, "[prof] %llu wavefront rounds, %.0f cycles each; probe stages %llu with %.2f dependent round trips and %.1f lanes each\n",
                hp[30], hp[30] ? (double)tot / hp[30] : 0.0, hp[26], hp[26] ? (double)hp[27] / hp[26] : 0.0, hp[26] ? (double)hp[31] / hp[26] : 0.0);
        {   // inside the re-pivot (these sections are part of "repivot" above, whose own slot holds what is left)
            const char *n2[8] = {"lo-scan", "remid", "pivot-choice", "stg-scan", "squeeze", "mid-loop", "extract-tail", "dry+tail"};
            unsigned long long t2 = hp[8]; for (int i = 10; i < 18; ++i) t2 += hp[i];
            for (int i = 20; i < 25; ++i) t2 += hp[i];
            fprintf(stderr, "[prof] inside the re-mids (cycles per call): pre-flush %.0f, demote %.0f, scan+collect %.0f, settle %.0f, sort+mid run %.0f, tail %.0f; far runs at a re-mid %.0f\n",
                    (double)hp[20] / (hp[18] ? hp[18] : 1), (double)hp[21] / (hp[18] ? hp[18] : 1), (double)hp[22] / (hp[18] ? hp[18] : 1),
                    (double)hp[23] / (hp[18] ? hp[18] : 1), (double)hp[24] / (hp[18] ? hp[18] : 1), (double)hp[11] / (hp[18] ? hp[18] : 1),
                    (double)hp[25] / (hp[18] ? hp[18] : 1));
            fprintf(stderr, "[prof] re-pivot %.1f%% of all:", 100.0 * t2 / ((tot - hp[8] + t2) ? (tot - hp[8] + t2) : 1));
            for (int i = 10; i < 18; ++i) fprintf(stderr, " %s=%.1f%%", n2[i - 10], 100.0 * hp[i] / (t2 ? t2 : 1));
            fprintf(stderr, " rest=%.1f%%; %llu re-mid calls of a wavefront for %llu rows\n", 100.0 * hp[8] / (t2 ? t2 : 1), hp[18], hp[19]);
        }
    }
#endif
    t->kernel_ms += ms;
    t->launches++;
    return RADHIP_OK;
}
static int trav_launch(radhip_traversal *t) {
    RH_TRY(trav_enqueue(t));
    return trav_collect(t);
}

// The grouped table filled one of its chunk positions (layout ids that pile onto one position: possible
// only for an adversarial layout).  The batch has not returned anything yet: re-arm it with the per-slot
// hash table, which has no such limit below its sized capacity, and run it again from the start.
static int trav_fall_back_to_hash(radhip_traversal *t) {
    if (t->wide || t->P.slots) {   // the WIDE and the SLOT forms have no hash-table variant: their fallback is the bucket table
        const size_t bt_bytes = ((size_t)t->sn << (t->P.bt_log2 + 2)) * 4;
        uint32_t *n_bt = nullptr;
        hipError_t e = hipMalloc((void **)&n_bt, bt_bytes);
        if (e != hipSuccess) { (void)hipGetLastError(); RH_FAIL(e == hipErrorOutOfMemory ? RADHIP_E_NOMEM : RADHIP_E_HIP,
                                     "hipMalloc(%zu) for the bucket-table fallback failed: %s", bt_bytes, hipGetErrorString(e)); }
        if (t->P.gt) { (void)hipFree(t->P.gt); t->P.gt = nullptr; t->state_bytes -= t->gt_bytes; t->gt_bytes = 0; }
        t->use_gt = false; t->use_bt = true;
        t->P.bt = n_bt; t->bt_bytes = bt_bytes; t->state_bytes += bt_bytes;
        t->epoch_max = (1u << (31u - t->P.bt_sbits)) - 1u;
        t->P.epoch_first = 1u; t->P.epoch_max = t->epoch_max;
        t->fresh_tables = true;
        const double ms = t->kernel_ms;
        const uint64_t launches = t->launches;
        RH_TRY(trav_upload_queries(t, nullptr));
        t->kernel_ms = ms; t->launches = launches;
        return RADHIP_OK;
    }
    // allocate first, release the grouped table only when the per-slot table exists
    const size_t ht_bytes = ((size_t)t->nq << t->ht_log2) * 8;
    unsigned long long *n_ht = nullptr;
    hipError_t e = hipMalloc((void **)&n_ht, ht_bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); RH_FAIL(e == hipErrorOutOfMemory ? RADHIP_E_NOMEM : RADHIP_E_HIP,
                                 "hipMalloc(%zu) for the hash-table fallback failed: %s", ht_bytes, hipGetErrorString(e)); }
    if (t->P.gt) { (void)hipFree(t->P.gt); t->P.gt = nullptr; t->state_bytes -= t->gt_bytes; t->gt_bytes = 0; }
    t->use_gt = false;
    t->P.ht = n_ht;
    t->P.ht_log2 = t->ht_log2;
    t->ht_bytes = ht_bytes;
    t->state_bytes += t->ht_bytes;
    t->fresh_tables = true;
    const double ms = t->kernel_ms;
    const uint64_t launches = t->launches;
    RH_TRY(trav_upload_queries(t, nullptr));
    t->kernel_ms = ms; t->launches = launches;   // the aborted launch stays on the clock
    return RADHIP_OK;
}

// The set of (node, level >= 1) visits and the key pool are sized by an estimate of how much of a traversal
// happens above level 0 (scored_cap * 8 / connectivity): a traversal that stays on the upper levels (wide
// upper rows, tiny n_to_score against a tall graph) can exceed it.  The reference has no such limit
// (rad/visited.py is a Redis set): the batch is re-armed with four times the room and run again.
static int trav_grow_upper(radhip_traversal *t) {
    if (t->P.ut_log2 + 2 > 31) RH_FAIL(RADHIP_E_CAPACITY, "the upper-level visited set cannot grow any further");
    TravParams &P = t->P;
    // allocate first, swap on success: a failed regrow (four times the room is where memory runs out) leaves the
    // object exactly as it was — still bound to valid buffers, its batch still failed with RADHIP_E_CAPACITY
    const uint32_t ut_log2 = P.ut_log2 + 2;
    const uint64_t pq_cap = P.scored_cap + ((uint64_t)1 << ut_log2);
    if (pq_cap >= 0xFFFFFFFFull) RH_FAIL(RADHIP_E_CAPACITY, "the key pool cannot grow any further");
    const size_t ut_bytes = ((size_t)t->sn << ut_log2) * 8, pq_bytes = (size_t)t->sn * pq_cap * 8;
    unsigned long long *n_ut = nullptr, *n_pq = nullptr;
    uint32_t *n_pl = nullptr;
    uint8_t *n_plv = nullptr;
    hipError_t e = hipMalloc((void **)&n_ut, ut_bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&n_pq, pq_bytes);
    if (e == hipSuccess && P.poplog_nodes) {
        e = hipMalloc((void **)&n_pl, (size_t)t->nq * pq_cap * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&n_plv, (size_t)t->nq * pq_cap);
    }
    if (e != hipSuccess) {
        if (n_ut) (void)hipFree(n_ut);
        if (n_pq) (void)hipFree(n_pq);
        if (n_pl) (void)hipFree(n_pl);
        if (n_plv) (void)hipFree(n_plv);
        (void)hipGetLastError();
        (void)hipGetLastError();
        RH_FAIL(e == hipErrorOutOfMemory ? RADHIP_E_NOMEM : RADHIP_E_HIP, "growing the traversal state failed: %s", hipGetErrorString(e));
    }
    (void)hipFree(P.ut); (void)hipFree(P.pq);
    t->state_bytes -= t->ut_bytes + t->pq_bytes;
    P.ut = n_ut; P.pq = n_pq; P.ut_log2 = ut_log2; P.pq_cap = pq_cap;
    t->ut_bytes = ut_bytes; t->pq_bytes = pq_bytes;
    t->state_bytes += ut_bytes + pq_bytes;
    if (n_pl) {
        (void)hipFree(P.poplog_nodes); (void)hipFree(P.poplog_levels);
        P.poplog_nodes = n_pl; P.poplog_levels = n_plv; P.poplog_cap = pq_cap;
    }
    t->fresh_tables = true;
    const double ms = t->kernel_ms;
    const uint64_t launches = t->launches;
    // the re-arm starts every traversal again: the stop targets the caller had set (radhip_traversal_set_targets)
    // are put back afterwards
    std::vector<TravHeader> old(t->nq);
    RH_HIP(hipMemcpy(old.data(), P.hdr, t->hdr_bytes, hipMemcpyDeviceToHost));
    RH_TRY(trav_upload_queries(t, nullptr));
    {
        std::vector<TravHeader> hdr(t->nq);
        RH_HIP(hipMemcpy(hdr.data(), P.hdr, t->hdr_bytes, hipMemcpyDeviceToHost));
        bool any = false;
        for (uint32_t i = 0; i < t->nq; ++i) if (old[i].target != hdr[i].target) { hdr[i].target = old[i].target; any = true; }
        if (any) RH_HIP(hipMemcpy(P.hdr, hdr.data(), t->hdr_bytes, hipMemcpyHostToDevice));
    }
    t->kernel_ms = ms; t->launches = launches;
    return RADHIP_OK;
}

// what a finished launch left: running / failed traversals; the capacity fallbacks re-arm and re-run a batch whose FIRST
// launch hit a fixed-capacity structure (nothing of it has been returned to the caller yet)
static int trav_after_launch(radhip_traversal *t, bool first_launch, uint32_t *out_running) {
    std::vector<TravHeader> hdr(t->n_active);
    for (int attempt = 0;; ++attempt) {
        RH_HIP(hipMemcpy(hdr.data(), t->P.hdr, (size_t)t->n_active * sizeof(TravHeader), hipMemcpyDeviceToHost));
        uint32_t running = 0;
        int bad = 0;
        for (uint32_t i = 0; i < t->n_active; ++i) {
            if (hdr[i].status == 0) running++;  // status 3 (intermediate target reached) is parked, not running
            if (hdr[i].status < 0 && !bad) bad = hdr[i].status;
        }
        if (bad == RADHIP_E_CAPACITY && first_launch && attempt < 4) {
            if (t->use_gt) RH_TRY(trav_fall_back_to_hash(t));
            else RH_TRY(trav_grow_upper(t));
            RH_TRY(trav_launch(t));
            continue;
        }
        if (out_running) *out_running = running;
        if (bad) RH_FAIL(bad, "traversal state overflowed a fixed-capacity device structure (status %d)", bad);
        if (t->P.slots && running) RH_FAIL(RADHIP_E_STATE, "%u traversals of a per-row-state batch did not run to completion", running);
        return RADHIP_OK;
    }
}

static int trav_check_runnable(radhip_traversal *t, uint64_t max_pops) {
    radhip_index *idx = t->idx;
    // n_top, start_level, the table sizes and the layout were taken from the graph at create time: a
    // traversal object does not survive an add() / load_graph() (the kernel would read freed arrays)
    if (t->graph_gen != idx->graph_gen)
        RH_FAIL(RADHIP_E_STATE, "the index changed since this traversal object was created: create a new one");
    if (t->sharded) RH_FAIL(RADHIP_E_STATE, "a sharded traversal is stepped by radhip_shard_run / radhip_shard_step");
    if (t->in_flight) RH_FAIL(RADHIP_E_STATE, "the traversal object has a launch in flight: radhip_traversal_finish first");
    if (t->P.slots && max_pops) RH_FAIL(RADHIP_E_STATE, "a batch with per-row state (RADHIP_TRAV_SLOTS) runs to completion: max_pops must be 0");
    return RADHIP_OK;
}

extern "C" int radhip_traversal_run(radhip_traversal_t *t, uint64_t max_pops, uint32_t *out_running) {
    if (!t) RH_FAIL(RADHIP_E_INVALID, "null traversal");
    radhip_index *idx = t->idx;
    std::lock_guard<std::mutex> lk(idx->mu);
    RH_HIP(hipSetDevice(idx->device));
    RH_TRY(trav_check_runnable(t, max_pops));
    t->P.max_pops = max_pops;
    const bool first_launch = t->launches == 0;
    RH_TRY(trav_launch(t));
    return trav_after_launch(t, first_launch, out_running);
}

// ---- launches that do not wait (round 4).  A launch of a batch ends with its longest traversals running alone (~130 ms
// whatever its size: profiles/r03/README.md §8).  Two objects on two streams (RADHIP_TRAV_OWN_STREAM), each with its own
// rows' state and its own outputs: the next batch's wavefronts start as the last ones of this batch leave the device, and
// the tail is paid once per run instead of once per batch.  start = re-armed batch -> kernel enqueued; finish = wait +
// everything radhip_traversal_run does after its launch.
extern "C" int radhip_traversal_start(radhip_traversal_t *t) {
    if (!t) RH_FAIL(RADHIP_E_INVALID, "null traversal");
    radhip_index *idx = t->idx;
    std::lock_guard<std::mutex> lk(idx->mu);
    RH_HIP(hipSetDevice(idx->device));
    RH_TRY(trav_check_runnable(t, 0));
    t->P.max_pops = 0;
    t->first_pending = t->launches == 0;
    RH_TRY(trav_enqueue(t));
    t->in_flight = true;
    return RADHIP_OK;
}
extern "C" int radhip_traversal_finish(radhip_traversal_t *t, uint32_t *out_running) {
    if (!t) RH_FAIL(RADHIP_E_INVALID, "null traversal");
    radhip_index *idx = t->idx;
    if (!t->in_flight) RH_FAIL(RADHIP_E_STATE, "no launch in flight");
    RH_HIP(hipSetDevice(idx->device));
    // the wait happens outside the index lock: another object's start must be able to get in meanwhile
    {
        const hipError_t e = hipStreamSynchronize(t->stream);
        if (e != hipSuccess) { std::lock_guard<std::mutex> lk(idx->mu); t->in_flight = false; RH_FAIL(RADHIP_E_HIP, "the traversal's stream failed: %s", hipGetErrorString(e)); }
    }
    std::lock_guard<std::mutex> lk(idx->mu);
    t->in_flight = false;
    RH_TRY(trav_collect(t));
    return trav_after_launch(t, t->first_pending, out_running);
}
// milliseconds from the start of `from`'s last launch to the end of `to`'s last launch (both finished; events on the
// device's clock, whatever streams they were recorded on): the busy interval of a run of overlapping launches
extern "C" int radhip_traversal_elapsed_between(const radhip_traversal_t *from, const radhip_traversal_t *to, double *out_ms) {
    if (!from || !to || !out_ms) RH_FAIL(RADHIP_E_INVALID, "null argument");
    if (from->in_flight || to->in_flight) RH_FAIL(RADHIP_E_STATE, "a launch is still in flight");
    float ms = 0.f;
    RH_HIP(hipEventElapsedTime(&ms, from->ev0, to->ev1));
    *out_ms = ms;
    return RADHIP_OK;
}
// Start and end of the object's last (finished) launch in milliseconds on the device's clock since a fixed point of the
// process (an event recorded once per device): launches of different objects overlap (start / finish on two streams), so
// the time the device was busy is the union of their intervals, which the caller forms from these.
static std::mutex g_base_mu;
static hipEvent_t g_base_ev[64] = {nullptr};
static int trav_base_event(radhip_index *idx, hipEvent_t *out) {
    std::lock_guard<std::mutex> lk(g_base_mu);
    const int d = idx->device;
    if (d < 0 || d >= 64) RH_FAIL(RADHIP_E_INVALID, "device %d out of range", d);
    if (!g_base_ev[d]) {
        hipEvent_t e;
        RH_HIP(hipEventCreate(&e));
        if (hipEventRecord(e, idx->stream) != hipSuccess || hipEventSynchronize(e) != hipSuccess) { (void)hipEventDestroy(e); (void)hipGetLastError(); RH_FAIL(RADHIP_E_HIP, "recording the base event failed"); }
        g_base_ev[d] = e;
    }
    *out = g_base_ev[d];
    return RADHIP_OK;
}
extern "C" int radhip_traversal_launch_interval(const radhip_traversal_t *t, double *out_start_ms, double *out_end_ms) {
    if (!t || !out_start_ms || !out_end_ms) RH_FAIL(RADHIP_E_INVALID, "null argument");
    if (t->in_flight) RH_FAIL(RADHIP_E_STATE, "a launch is still in flight");
    if (t->launches == 0) RH_FAIL(RADHIP_E_STATE, "no launch yet");
    RH_HIP(hipSetDevice(t->idx->device));
    hipEvent_t base = nullptr;
    RH_TRY(trav_base_event(t->idx, &base));
    float a = 0.f, b = 0.f;
    RH_HIP(hipEventElapsedTime(&a, base, t->ev0));
    RH_HIP(hipEventElapsedTime(&b, base, t->ev1));
    *out_start_ms = a; *out_end_ms = b;
    return RADHIP_OK;
}
// rows whose state the batch shares (0 = state per traversal)
extern "C" uint32_t radhip_traversal_slots(const radhip_traversal_t *t) { return t ? t->P.slots : 0; }

extern "C" int radhip_traversal_stats(const radhip_traversal_t *t, radhip_trav_stats_t *out) {
    if (!t || !out) RH_FAIL(RADHIP_E_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(t->idx->mu);
    RH_HIP(hipSetDevice(t->idx->device));
    if (t->n_active < t->nq) memset(out + t->n_active, 0, (size_t)(t->nq - t->n_active) * sizeof *out);   // (not armed in this batch)
    radhip_trav_stats_t *d_out = nullptr;
    const size_t bytes = (size_t)t->n_active * sizeof *out;
    hipError_t e = hipMalloc((void **)&d_out, bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); RH_FAIL(e == hipErrorOutOfMemory ? RADHIP_E_NOMEM : RADHIP_E_HIP, "hipMalloc(%zu) for the statistics failed: %s", bytes, hipGetErrorString(e)); }
    hipLaunchKernelGGL(trav_stats_kernel, dim3((t->n_active + 255u) / 256u), dim3(256), 0, t->stream, t->P.hdr, d_out, t->n_active);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, t->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
    (void)hipFree(d_out);
    if (e != hipSuccess) RH_FAIL(RADHIP_E_HIP, "reading the statistics failed: %s", hipGetErrorString(e));
    return RADHIP_OK;
}

// with a ring of scored lists only the lists of the last `ring` traversals of the batch are still there
static int trav_list_kept(const radhip_traversal *t, uint32_t first, uint32_t count) {
    if (t->P.list_mask == 0xFFFFFFFFu || t->n_active <= t->list_ring) return RADHIP_OK;
    if (first < t->n_active - t->list_ring)
        RH_FAIL(RADHIP_E_STATE, "the scored lists of traversals below %u were overwritten (a ring of %u lists serves this batch of %u): "
                "their counts are in the statistics", t->n_active - t->list_ring, t->list_ring, t->n_active);
    (void)count;
    return RADHIP_OK;
}

extern "C" int radhip_traversal_results(const radhip_traversal_t *t, uint32_t q, uint32_t *out_slots,
                                        uint32_t *out_and, uint32_t *out_or, uint64_t cap, uint64_t *out_n) {
    if (!t || !out_n) RH_FAIL(RADHIP_E_INVALID, "null argument");
    if (q >= t->n_active) RH_FAIL(RADHIP_E_RANGE, "traversal %u out of range", q);
    RH_TRY(trav_list_kept(t, q, 1));
    std::lock_guard<std::mutex> lk(t->idx->mu);
    RH_HIP(hipSetDevice(t->idx->device));
    TravHeader h;
    RH_HIP(hipMemcpy(&h, t->P.hdr + q, sizeof h, hipMemcpyDeviceToHost));
    const uint64_t n = std::min<uint64_t>(h.n_scored, cap);
    *out_n = h.n_scored;
    if (n == 0) return RADHIP_OK;
    std::vector<uint2> buf(n);
    RH_HIP(hipMemcpy(buf.data(), t->P.scored + (uint64_t)(q & t->P.list_mask) * t->P.scored_cap, n * sizeof(uint2), hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < n; ++i) {
        if (out_slots) out_slots[i] = buf[i].x;
        if (out_and) out_and[i] = buf[i].y & 0xFFFFu;
        if (out_or) out_or[i] = buf[i].y >> 16;
    }
    return RADHIP_OK;
}

// ---- order-sensitive 64-bit hash of every traversal's scored list, formed on the device (a wrap-around sum of mixed
// (position, slot, and | or << 16) terms: any summation order gives the same value).  bench.py compares it with the
// oracle's orc_result_hash for its parity sample: the whole scored list, not three counters.
__device__ __forceinline__ unsigned long long th_mix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
__global__ __launch_bounds__(256) void result_hash_kernel(const uint2 *scored, uint64_t scored_cap, uint32_t list_mask, const TravHeader *hdr, uint32_t first_q,
                                                          unsigned long long *out) {
    const uint32_t q = first_q + blockIdx.x;
    const uint64_t n = hdr[q].n_scored < scored_cap ? hdr[q].n_scored : scored_cap;
    const uint2 *sc = scored + (uint64_t)(q & list_mask) * scored_cap;
    unsigned long long h = 0;
    for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint2 e = sc[i];
        h += th_mix64(((unsigned long long)i << 32 | (unsigned long long)e.x) + th_mix64((unsigned long long)e.y));
    }
    for (int o = 32; o > 0; o >>= 1) h += ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(h >> 32), o) << 32 | (uint32_t)__shfl_xor((int)(uint32_t)h, o));
    if ((threadIdx.x & 63u) == 0u) atomicAdd(&out[blockIdx.x], h);
}

extern "C" int radhip_traversal_result_hashes(const radhip_traversal_t *t, uint32_t first, uint32_t count, uint64_t *out) {
    if (!t || !out) RH_FAIL(RADHIP_E_INVALID, "null argument");
    if ((uint64_t)first + count > t->n_active) RH_FAIL(RADHIP_E_RANGE, "traversals [%u, %u) out of range", first, first + count);
    RH_TRY(trav_list_kept(t, first, count));
    if (count == 0) return RADHIP_OK;
    radhip_index *idx = t->idx;
    std::lock_guard<std::mutex> lk(idx->mu);
    RH_HIP(hipSetDevice(idx->device));
    unsigned long long *d = nullptr;
    RH_HIP(hipMalloc((void **)&d, (size_t)count * 8));
    int rc = RADHIP_OK;
    if (hipMemsetAsync(d, 0, (size_t)count * 8, t->stream) != hipSuccess) rc = RADHIP_E_HIP;
    if (rc == RADHIP_OK) {
        hipLaunchKernelGGL(result_hash_kernel, dim3(count), dim3(256), 0, t->stream, t->P.scored, t->P.scored_cap, t->P.list_mask, t->P.hdr, first, d);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(out, d, (size_t)count * 8, hipMemcpyDeviceToHost, t->stream) != hipSuccess ||
            hipStreamSynchronize(t->stream) != hipSuccess) rc = RADHIP_E_HIP;
    }
    (void)hipFree(d);
    if (rc != RADHIP_OK) radhip_set_error("radhip_traversal_result_hashes failed");
    return rc;
}

extern "C" int radhip_traversal_pop_log(const radhip_traversal_t *t, uint32_t q, uint32_t *out_nodes,
                                        uint8_t *out_levels, uint64_t cap, uint64_t *out_n) {
    if (!t || !out_n) RH_FAIL(RADHIP_E_INVALID, "null argument");
    if (q >= t->nq) RH_FAIL(RADHIP_E_RANGE, "traversal %u out of range", q);
    if (!t->P.poplog_nodes) RH_FAIL(RADHIP_E_STATE, "traversal was created without RADHIP_TRAV_LOG_POPS");
    std::lock_guard<std::mutex> lk(t->idx->mu);
    RH_HIP(hipSetDevice(t->idx->device));
    TravHeader h;
    RH_HIP(hipMemcpy(&h, t->P.hdr + q, sizeof h, hipMemcpyDeviceToHost));
    const uint64_t n = std::min<uint64_t>(std::min<uint64_t>(h.n_pops, t->P.poplog_cap), cap);
    *out_n = std::min<uint64_t>(h.n_pops, t->P.poplog_cap);
    if (n == 0) return RADHIP_OK;
    if (out_nodes) RH_HIP(hipMemcpy(out_nodes, t->P.poplog_nodes + (uint64_t)q * t->P.poplog_cap, n * 4, hipMemcpyDeviceToHost));
    if (out_levels) RH_HIP(hipMemcpy(out_levels, t->P.poplog_levels + (uint64_t)q * t->P.poplog_cap, n, hipMemcpyDeviceToHost));
    return RADHIP_OK;
}

extern "C" int radhip_traversal_kernel_time(const radhip_traversal_t *t, double *out_ms, uint64_t *out_launches) {
    if (!t) RH_FAIL(RADHIP_E_INVALID, "null traversal");
    if (out_ms) *out_ms = t->kernel_ms;
    if (out_launches) *out_launches = t->launches;
    return RADHIP_OK;
}

extern "C" uint64_t radhip_traversal_state_bytes(const radhip_traversal_t *t) { return t ? t->state_bytes : 0; }

extern "C" int radhip_traversal_set_targets(radhip_traversal_t *t, const uint64_t *targets) {
    if (t && t->P.slots) RH_FAIL(RADHIP_E_STATE, "a batch with per-row state (RADHIP_TRAV_SLOTS) runs to completion: it has no parked targets");
    if (!t || !targets) RH_FAIL(RADHIP_E_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(t->idx->mu);
    RH_HIP(hipSetDevice(t->idx->device));
    std::vector<TravHeader> hdr(t->nq);
    RH_HIP(hipMemcpy(hdr.data(), t->P.hdr, t->hdr_bytes, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < t->nq; ++i) {
        uint64_t tg = targets[i] < t->n_to_score ? targets[i] : t->n_to_score;
        hdr[i].target = tg;
        if (hdr[i].status == 3 && hdr[i].n_scored < tg) hdr[i].status = 0;
        if (hdr[i].status == 0 && hdr[i].primed && hdr[i].n_scored >= tg) hdr[i].status = tg >= t->n_to_score ? 1 : 3;
    }
    RH_HIP(hipMemcpy(t->P.hdr, hdr.data(), t->hdr_bytes, hipMemcpyHostToDevice));
    return RADHIP_OK;
}

extern "C" int radhip_traversal_frontier(const radhip_traversal_t *t, uint64_t *out_keys, uint64_t *out_scored) {
    if (!t) RH_FAIL(RADHIP_E_INVALID, "null traversal");
    std::lock_guard<std::mutex> lk(t->idx->mu);
    RH_HIP(hipSetDevice(t->idx->device));
    std::vector<TravHeader> hdr(t->nq);
    RH_HIP(hipMemcpy(hdr.data(), t->P.hdr, t->hdr_bytes, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < t->nq; ++i) {
        if (out_keys) out_keys[i] = hdr[i].frontier_key;
        if (out_scored) out_scored[i] = hdr[i].n_scored;
    }
    return RADHIP_OK;
}

// ---- test hook: the device restatements of the queue key, evaluated on the GPU -------------
// Both restatements must agree — the comparison-tree one of common.h (trav_kernel's prime path, other kernels) and
// the table one the traversal kernels use — and the table decode must give the slot back; a disagreement shows up
// as an all-ones key.
__global__ void debug_keys_kernel(const uint32_t *a, const uint32_t *o, const uint32_t *slot, const uint32_t *level,
                                  uint64_t n, unsigned long long *out) {
    __shared__ KeyTabs T;
    keytabs_init(T, threadIdx.x);
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long k1 = rh_make_key_dev(rh_q24_dev(a[i], o[i]), slot[i], level[i]);
        const unsigned long long k2 = make_key_tab(T, rh_q24_dev(a[i], o[i]), slot[i], level[i]);
        out[i] = (k1 == k2 && key_slot_tab(T, k2) == slot[i]) ? k2 : ~0ull;
    }
}

extern "C" uint32_t radhip_debug_staging_capacity(void) { return rh_trav4_staging_capacity(); }

extern "C" int radhip_debug_device_keys(radhip_index_t *idx, const uint32_t *a, const uint32_t *o, const uint32_t *slot,
                                        const uint32_t *level, uint64_t n, uint64_t *out_keys) {
    if (!idx || !a || !o || !slot || !level || !out_keys) RH_FAIL(RADHIP_E_INVALID, "null argument");
    if (n == 0) return RADHIP_OK;
    std::lock_guard<std::mutex> lk(idx->mu);
    RH_TRY(rh_ensure_device(idx));
    uint32_t *d[4] = {nullptr, nullptr, nullptr, nullptr};
    unsigned long long *dout = nullptr;
    const uint32_t *src[4] = {a, o, slot, level};
    int rc = RADHIP_OK;
    for (int i = 0; i < 4 && rc == RADHIP_OK; ++i) {
        if (hipMalloc((void **)&d[i], n * 4) != hipSuccess) rc = RADHIP_E_NOMEM;
        else if (hipMemcpy(d[i], src[i], n * 4, hipMemcpyHostToDevice) != hipSuccess) rc = RADHIP_E_HIP;
    }
    if (rc == RADHIP_OK && hipMalloc((void **)&dout, n * 8) != hipSuccess) rc = RADHIP_E_NOMEM;
    if (rc == RADHIP_OK) {
        hipLaunchKernelGGL(debug_keys_kernel, dim3(1024), dim3(256), 0, idx->stream, d[0], d[1], d[2], d[3], n, dout);
        if (hipStreamSynchronize(idx->stream) != hipSuccess) rc = RADHIP_E_HIP;
        else if (hipMemcpy(out_keys, dout, n * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = RADHIP_E_HIP;
    }
    for (int i = 0; i < 4; ++i) if (d[i]) (void)hipFree(d[i]);
    if (dout) (void)hipFree(dout);
    if (rc != RADHIP_OK) radhip_set_error("radhip_debug_device_keys failed (%d)", rc);
    return rc;
}

// ---- kernel choice ---------------------------------------------------------------------
// trav4_kernel packs four traversals into a wavefront (rows <= 16 wide only): the most
// expansions per HBM request slot when the chip is full.  trav_kernel gives a traversal a whole
// wavefront and gathers every neighbour's fingerprint speculatively while the probes are in
// flight: one dependent HBM round trip less per expansion, faster per traversal as long as
// all of them are resident at once (trav_kernel holds 6144).  Measured at the end of round 2 (20M rows,
// n_to_score 100k, trav_kernel vs trav4_kernel; scripts/crossover.sh): hierarchical corpus nq=1024 146 vs
// 225 ms, 4096 177 vs 258, 6144 197 vs 272, 8192 280 vs 278, 12288 349 vs 304; round 1's corpus 6144 81 vs
// 106, 8192 126 vs 113, 12288 156 vs 130.  So: four per wave when the batch exceeds 5/4 of trav_kernel's
// resident round (round 1's rule was two rounds; trav4_kernel has gained a third since).
// RADHIP_TRAV=1|4 forces a kernel (tests, profiling); RADHIP_NO_TRAV4 is the older spelling of 1.
static bool trav4_shape_ok(const radhip_index *idx) { return idx->cap0 <= 16 && idx->M <= 16; }

static int trav_forced_kernel() {
    if (getenv("RADHIP_NO_TRAV4")) return 1;
    const char *e = getenv("RADHIP_TRAV");
    if (e && e[0] == '1') return 1;
    if (e && e[0] == '4') return 4;
    return 0;
}

// traversals resident at once for one of the two kernels (device must be ready)
static int trav_capacity_of(radhip_index *idx, bool use4, uint32_t *out) {
    int per_cu = 0;
    hipError_t e;
    if (use4) { RH_TRY(rh_trav4_occupancy((int)idx->lpr, &per_cu)); e = hipSuccess; }
    else
    switch (idx->lpr) {
        case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, trav_kernel<1>, 64, 0); break;
        case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, trav_kernel<2>, 64, 0); break;
        case 4: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, trav_kernel<4>, 64, 0); break;
        case 8: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, trav_kernel<8>, 64, 0); break;
        default: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, trav_kernel<16>, 64, 0); break;
    }
    RH_HIP(e);
    hipDeviceProp_t prop;
    RH_HIP(hipGetDeviceProperties(&prop, idx->device));
    // The occupancy API over-reports by one wave per SIMD for SGPR-heavy kernels on gfx950 /
    // ROCm 7.2 (MI355X_MICROARCH.md, "Residency"): trav_kernel uses > 96 SGPRs, which admits
    // floor(800 / (112 + 16)) = 6 waves per SIMD = 24 single-wave workgroups per CU.
    if (per_cu > 24) per_cu = 24;
    *out = (uint32_t)per_cu * (uint32_t)prop.multiProcessorCount * (use4 ? 4u : 1u);
    return RADHIP_OK;
}

// The largest batch the chip holds resident at once (with the four-per-wave kernel where the
// index shape allows it): batches that are a multiple of it avoid a partially filled last round
extern "C" int radhip_traversal_resident_capacity(radhip_index_t *idx, uint32_t *out) {
    if (!idx || !out) RH_FAIL(RADHIP_E_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    RH_TRY(rh_ensure_device(idx));
    const char *tb = getenv("RADHIP_TABLE");
    const int forced = trav_forced_kernel();
    const bool wide_ok = idx->cap0 <= (forced == 4 ? 64u : 32u) && idx->M <= (forced == 4 ? 64u : 32u) && !(tb && tb[0] == 'h');
    return trav_capacity_of(idx, (trav4_shape_ok(idx) || wide_ok) && forced != 1, out);
}

// 4 = trav4_kernel (four traversals per wavefront), 1 = trav_kernel
extern "C" int radhip_traversal_kernel(const radhip_traversal_t *t) { return t ? (t->use4 ? 4 : 1) : 0; }
// 1 = grouped visited/scored table (2 bits per node, keyed by the graph-locality layout), 0 = per-slot hash table
extern "C" int radhip_traversal_table(const radhip_traversal_t *t) { return t ? (t->use_gt ? 1 : t->use_local ? 3 : t->use_bt ? 2 : 0) : -1; }

