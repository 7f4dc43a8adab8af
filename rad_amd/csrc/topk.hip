// topk.hip — exact k nearest rows of a corpus range for a batch of queries, reduced on the chip
// (gfx950).  The K1 scan (index.hip) writes 8 B per (query, row) pair, which at 8 queries per pass
// is a third of its HBM traffic and an O(N) host array per query; here the same tile loop feeds a
// per-wavefront candidate buffer in LDS through a running threshold instead, so the pass reads
// the corpus once and writes k keys per wavefront.  Order: (distance, slot) ascending, distance
// as the 24-bit quotient of common.h (strictly monotone in the exact Tanimoto distance) — the
// order of a brute-force scan, which is what tests/test_gpu_kernels.py compares it with.
// replaces: usearch exact search (`Index.search(..., exact=True)`), used for recall figures only.
#include "common.h"
#include "rows_tile.h"

#include <algorithm>
#include <vector>

void rh_stage_queries(const radhip_index *idx, const uint8_t *queries, uint32_t nq,
                      std::vector<uint8_t> &padded, std::vector<uint32_t> &pop);   // index.hip

#define TK_INF 0xFFFFFFFFFFFFFFFFull

// ascending in-place bitonic sort of s[0..P), P a power of two, by one wavefront
__device__ __forceinline__ void tk_sort(unsigned long long *s, uint32_t P, uint32_t lane) {
    for (uint32_t k = 2; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = lane; t < (P >> 1); t += 64) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const uint32_t ixj = i | j;
                const bool up = (i & k) == 0;
                const unsigned long long a = s[i], b = s[ixj];
                if ((a > b) == up) { s[i] = b; s[ixj] = a; }
            }
            RH_WAVE_SYNC();
        }
    }
}

// keep the k smallest of buf[0..cnt) (capacity C, a power of two): returns the new count and
// sets thr to the k-th smallest key (TK_INF while fewer than k are known)
__device__ __forceinline__ uint32_t tk_compact(unsigned long long *buf, uint32_t cnt, uint32_t C, uint32_t k,
                                               uint32_t lane, unsigned long long &thr) {
    RH_WAVE_SYNC();
    for (uint32_t i = cnt + lane; i < C; i += 64) buf[i] = TK_INF;
    RH_WAVE_SYNC();
    tk_sort(buf, C, lane);
    const uint32_t kept = cnt < k ? cnt : k;
    thr = kept == k ? buf[k - 1] : TK_INF;
    return kept;
}

// append key (one per lane where `pass`) to buf; cnt is wave-uniform
__device__ __forceinline__ void tk_append(unsigned long long *buf, uint32_t &cnt, bool pass, unsigned long long key,
                                          unsigned long long lt_mask) {
    const unsigned long long b = __ballot(pass);
    if (b) {
        if (pass) buf[cnt + (uint32_t)__popcll(b & lt_mask)] = key;
        cnt += (uint32_t)__popcll(b);
    }
}

// one row per lane (and-counts against the NQ queries, its own popcount): offer it to the NQ candidate buffers of the wavefront
template <int NQ>
__device__ __forceinline__ void tk_offer(unsigned long long *buf, uint32_t C, uint32_t k, uint32_t lane, unsigned long long lt_mask,
                                         bool in_range, uint32_t slot, uint32_t rp, const uint32_t (&a)[NQ], const uint32_t (&qp)[NQ],
                                         uint32_t (&cnt)[NQ], unsigned long long (&thr)[NQ]) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        if (cnt[i] + 64u > C) cnt[i] = tk_compact(buf + (size_t)i * C, cnt[i], C, k, lane, thr[i]);
        const uint32_t o = qp[i] + rp - a[i];
        // cheap exact pre-filter: floor(x * 2^23 / o) <= T  <=>  x * 2^23 < (T + 1) * o; the quotient
        // itself (a division) is computed only for the few rows that can enter the buffer
        const unsigned long long tq1 = (thr[i] >> 32) + 1ull;
        bool pass = in_range && (((unsigned long long)(o - a[i]) << 23) < tq1 * o || o == 0u);
        unsigned long long key = TK_INF;
        if (__ballot(pass)) {
            if (pass) {
                key = ((unsigned long long)rh_q24_dev(a[i], o) << 32) | slot;
                pass = key < thr[i];
            }
            tk_append(buf + (size_t)i * C, cnt[i], pass, key, lt_mask);
        }
    }
}

// (experiment knob: -DTK_WAVES_PER_EU=4 caps the scan at 128 VGPRs — four wavefronts per SIMD at the price of spills)
#ifdef TK_WAVES_PER_EU
#define TK_OCC_ATTR __attribute__((amdgpu_waves_per_eu(TK_WAVES_PER_EU, TK_WAVES_PER_EU)))
#else
#define TK_OCC_ATTR
#endif
template <int LPR, int NQ>
__global__ __launch_bounds__(256) TK_OCC_ATTR void topk_scan_kernel(const uint4 *__restrict__ fp, uint64_t first, uint64_t count,
                                                        const uint4 *__restrict__ queries, const uint32_t *__restrict__ qpop,
                                                        uint32_t k, uint32_t C, unsigned long long *__restrict__ cand) {
    extern __shared__ unsigned long long tk_smem[];   // [4 wavefronts][NQ][C]
    constexpr int RPL = 64 / LPR;
    constexpr int BATCH = LPR < 8 ? LPR : 8;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t chunk = lane % LPR, grp = lane / LPR;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    unsigned long long *buf = tk_smem + (size_t)wv * NQ * C;
    uint4 q[NQ];
    uint32_t qp[NQ], cnt[NQ];
    unsigned long long thr[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        q[i] = queries[i * LPR + chunk];
        qp[i] = qpop[i];
        cnt[i] = 0;
        thr[i] = TK_INF;
    }
    const uint64_t n_tiles = (count + 63) / 64;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t tile = wave; tile < n_tiles; tile += n_waves) {
        const uint64_t r0 = tile * 64;
        uint32_t keep_a[NQ], keep_rp = 0;
#pragma unroll
        for (int i = 0; i < NQ; ++i) keep_a[i] = 0;
#pragma unroll
        for (int b0 = 0; b0 < LPR; b0 += BATCH) {
            uint4 v[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const uint64_t r = r0 + (uint64_t)(b0 + u) * RPL + grp;
                v[u] = make_uint4(0, 0, 0, 0);
                if (r < count) v[u] = fp[(first + r) * LPR + chunk];
            }
            if constexpr (LPR == 8) {   // (one batch per tile: b0 == 0, lane c keeps row c of its group)
                const bool b2 = (chunk & 4u) != 0u, b1 = (chunk & 2u) != 0u, b0c = (chunk & 1u) != 0u;
                uint32_t p[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) p[u] = rh_popc4(v[u]);
                keep_rp = rh_transpose_sum8(p, b2, b1, b0c);
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) p[u] = rh_popc4_and(v[u], q[i]);
                    keep_a[i] = rh_transpose_sum8(p, b2, b1, b0c);
                }
            } else {
#pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    const uint32_t rp = rh_group_sum<LPR>(rh_popc4(v[u]));
                    if ((int)chunk == b0 + u) keep_rp = rp;
#pragma unroll
                    for (int i = 0; i < NQ; ++i) {
                        const uint32_t a = rh_group_sum<LPR>(rh_popc4_and(v[u], q[i]));
                        if ((int)chunk == b0 + u) keep_a[i] = a;
                    }
                }
            }
        }
        const uint64_t r = r0 + (uint64_t)chunk * RPL + grp;   // the row this lane kept
        tk_offer<NQ>(buf, C, k, lane, lt_mask, r < count, (uint32_t)(first + r), keep_rp, keep_a, qp, cnt, thr);
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const uint32_t kept = tk_compact(buf + (size_t)i * C, cnt[i], C, k, lane, thr[i]);
        unsigned long long *dst = cand + ((uint64_t)i * n_waves + wave) * k;
        for (uint32_t j = lane; j < k; j += 64) dst[j] = j < kept ? buf[(size_t)i * C + j] : TK_INF;
    }
}

// 1024-bit rows, ONE ROW PER LANE (round 4; the tile machinery is rows_tile.h's rh_rows_*).  The kernel above is bound by its
// instruction stream, not by the memory: 1055 VALU instructions per tile of 64 rows x 8 queries = 3.0 ms of issue for a 100M-row
// pass that HBM serves in 2.3 ms.  Here: 731 per tile, three wavefronts per SIMD, the next tile's loads in flight while this
// one is counted (profiles/r04/README.md section 10).
#define TK_ROWS_WAVES 2          // wavefronts per block: 2 x (candidate buffers + 9 KB tile) stays under 64 KB of dynamic LDS for every k
template <int LPR, int NQ>
__global__ __launch_bounds__(64 * TK_ROWS_WAVES) void topk_rows_kernel(const uint4 *__restrict__ fp, uint64_t first, uint64_t count,
                                                        const uint32_t *__restrict__ qd /* [NQ][4 * LPR] */, const uint32_t *__restrict__ qpop,
                                                        uint32_t k, uint32_t C, unsigned long long *__restrict__ cand) {
    extern __shared__ unsigned long long tk_smem[];   // [TK_ROWS_WAVES wavefronts][NQ][C] keys, then [TK_ROWS_WAVES][RH_ROWS_TR_VEC(LPR)] uint4
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    unsigned long long *buf = tk_smem + (size_t)wv * NQ * C;
    uint4 *tr = reinterpret_cast<uint4 *>(tk_smem + (size_t)TK_ROWS_WAVES * NQ * C) + (size_t)wv * RH_ROWS_TR_VEC(LPR);
    uint32_t qp[NQ], cnt[NQ];
    unsigned long long thr[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) { qp[i] = qpop[i]; cnt[i] = 0; thr[i] = TK_INF; }
    const uint64_t n_tiles = (count + 63) / 64;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    uint4 nv[LPR];
    uint64_t tile = wave;
    if (tile < n_tiles) rh_rows_load<LPR>(fp, first, count, tile, lane, nv);
    for (; tile < n_tiles; tile += n_waves) {
        uint4 v[LPR];
        rh_rows_turn<LPR>(nv, tr, lane, v);
        if (tile + n_waves < n_tiles) rh_rows_load<LPR>(fp, first, count, tile + n_waves, lane, nv);
        uint32_t rp, a[NQ];
        rh_rows_count<LPR, NQ>(v, qd, rp, a);
        const uint64_t r = tile * 64 + lane;
        tk_offer<NQ>(buf, C, k, lane, lt_mask, r < count, (uint32_t)(first + r), rp, a, qp, cnt, thr);
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const uint32_t kept = tk_compact(buf + (size_t)i * C, cnt[i], C, k, lane, thr[i]);
        unsigned long long *dst = cand + ((uint64_t)i * n_waves + wave) * k;
        for (uint32_t j = lane; j < k; j += 64) dst[j] = j < kept ? buf[(size_t)i * C + j] : TK_INF;
    }
}

// one wavefront per query: the k smallest of its n_cand candidate keys, then their exact counts
template <int LPR>
__global__ __launch_bounds__(64) void topk_merge_kernel(const uint4 *__restrict__ fp, const uint4 *__restrict__ queries,
                                                        const uint32_t *__restrict__ qpop, uint32_t k, uint32_t C,
                                                        const unsigned long long *__restrict__ cand, uint64_t n_cand,
                                                        uint32_t *__restrict__ out_slots, uint32_t *__restrict__ out_and,
                                                        uint32_t *__restrict__ out_or, uint32_t *__restrict__ out_counts) {
    extern __shared__ unsigned long long tk_smem[];   // [C]
    const uint32_t lane = threadIdx.x, qi = blockIdx.x;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const unsigned long long *src = cand + (uint64_t)qi * n_cand;
    uint32_t cnt = 0;
    unsigned long long thr = TK_INF;
    for (uint64_t base = 0; base < n_cand; base += 512) {       // eight loads in flight per lane
        unsigned long long key[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint64_t i = base + (uint64_t)u * 64 + lane;
            key[u] = i < n_cand ? src[i] : TK_INF;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (cnt + 64u > C) cnt = tk_compact(tk_smem, cnt, C, k, lane, thr);
            tk_append(tk_smem, cnt, key[u] < thr, key[u], lt_mask);      // TK_INF padding never passes
        }
    }
    const uint32_t kept = tk_compact(tk_smem, cnt, C, k, lane, thr);
    if (lane == 0) out_counts[qi] = kept;
    const uint32_t chunk = lane % LPR, grp = lane / LPR;
    const uint4 qv = queries[(uint64_t)qi * LPR + chunk];
    const uint32_t qp = qpop[qi];
    constexpr uint32_t RPP = 64 / LPR;
    for (uint32_t base = 0; base < k; base += RPP) {
        const uint32_t j = base + grp;
        const bool have = j < kept;
        const uint32_t slot = have ? (uint32_t)tk_smem[j] : RADHIP_NO_SLOT;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (have) v = fp[(uint64_t)slot * LPR + chunk];
        const uint32_t rp = rh_group_sum<LPR>(rh_popc4(v));
        const uint32_t a = rh_group_sum<LPR>(rh_popc4_and(v, qv));
        if (j < k && chunk == 0) {
            out_slots[(uint64_t)qi * k + j] = slot;
            out_and[(uint64_t)qi * k + j] = have ? a : 0u;
            out_or[(uint64_t)qi * k + j] = have ? qp + rp - a : 0u;
        }
    }
}

static uint32_t tk_pow2ceil(uint32_t x) { uint32_t p = 1; while (p < x) p <<= 1; return p; }

// The scan's wavefronts stride over the tiles, so its grid is exactly what the device holds resident at once: at 8 queries per
// pass the kernel takes 160 VGPRs (three blocks per CU), and a fourth block per CU would run alone on a mostly idle chip
// after the others (round 4: 3.36 ms -> see profiles/r04/README.md for the 100M-row pass).
template <int LPR, int NQV>
static uint32_t tk_resident_grid(uint32_t grid_cap, int n_cu, size_t lds) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, topk_scan_kernel<LPR, NQV>, 256, lds) != hipSuccess || nb < 1) { (void)hipGetLastError(); nb = 1; }
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(grid_cap, (uint64_t)n_cu * (uint64_t)nb));
}

template <int LPR>
static int tk_run_pass(radhip_index *idx, int nqp, uint64_t first, uint64_t count, const uint4 *dq, const uint32_t *dpop,
                       uint32_t k, uint32_t C, uint32_t grid_cap, int n_cu, unsigned long long *dcand, uint32_t *ds, uint32_t *da,
                       uint32_t *dorr, uint32_t *dc) {
    size_t lds_scan = (size_t)4 * nqp * C * 8;
    uint32_t grid = grid_cap, waves_per_block = 4;
    // 1024- and 2048-bit rows: one row per lane (topk_rows_kernel; RADHIP_TOPK_ROWS=0 keeps the row-across-eight-lanes kernel: the A/B of
    // profiles/r04)
    static const bool rows_ok = []() { const char *e = getenv("RADHIP_TOPK_ROWS"); return !(e && e[0] == '0'); }();
    constexpr int RL = (LPR == 8 || LPR == 16) ? LPR : 8;   // (the instantiation the other widths never launch)
    const bool rows = (LPR == 8 || LPR == 16) && rows_ok;
    if (rows) { waves_per_block = TK_ROWS_WAVES; lds_scan = (size_t)TK_ROWS_WAVES * ((size_t)nqp * C * 8 + (size_t)RH_ROWS_TR_VEC(RL) * 16); }
#define TK_CASE(NQV)                                                                                   \
    case NQV:                                                                                          \
        if (rows) {                                                                                    \
            int nb = 0;                                                                                \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, topk_rows_kernel<RL, NQV>, 64 * TK_ROWS_WAVES, lds_scan) != hipSuccess || nb < 1) { (void)hipGetLastError(); nb = 1; } \
            grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(grid_cap, (uint64_t)n_cu * (uint64_t)nb)); \
            hipLaunchKernelGGL((topk_rows_kernel<RL, NQV>), dim3(grid), dim3(64 * TK_ROWS_WAVES), lds_scan, idx->stream,  \
                               idx->d_fp, first, count, reinterpret_cast<const uint32_t *>(dq), dpop, k, C, dcand); \
            break;                                                                                     \
        }                                                                                              \
        grid = tk_resident_grid<LPR, NQV>(grid_cap, n_cu, lds_scan);                                   \
        hipLaunchKernelGGL((topk_scan_kernel<LPR, NQV>), dim3(grid), dim3(256), lds_scan, idx->stream, \
                           idx->d_fp, first, count, dq, dpop, k, C, dcand);                            \
        break;
    switch (nqp) {
        TK_CASE(1) TK_CASE(2) TK_CASE(3) TK_CASE(4) TK_CASE(5) TK_CASE(6) TK_CASE(7) TK_CASE(8)
        default: RH_FAIL(RADHIP_E_INVALID, "internal: queries per pass must be 1..8");
    }
#undef TK_CASE
    RH_HIP(hipGetLastError());
    const uint64_t n_waves = (uint64_t)grid * waves_per_block;
    hipLaunchKernelGGL((topk_merge_kernel<LPR>), dim3(nqp), dim3(64), (size_t)C * 8, idx->stream, idx->d_fp, dq, dpop, k, C,
                       dcand, n_waves * k, ds, da, dorr, dc);
    RH_HIP(hipGetLastError());
    return RADHIP_OK;
}

extern "C" int radhip_tanimoto_topk(radhip_index_t *idx, const uint8_t *queries, uint32_t nq, uint32_t k,
                                    uint64_t first, uint64_t count, uint32_t *out_slots, uint32_t *out_and,
                                    uint32_t *out_or, uint32_t *out_counts) {
    if (!idx || !queries || !out_slots || !out_counts) RH_FAIL(RADHIP_E_INVALID, "null argument");
    if (!idx->has_vectors) RH_FAIL(RADHIP_E_STATE, "no vectors loaded");
    RH_REQUIRE_FULL_CORPUS(idx);
    if (first + count > idx->n) RH_FAIL(RADHIP_E_RANGE, "rows [first, first+count) out of range");
    if (k == 0 || k > 1984) RH_FAIL(RADHIP_E_INVALID, "k must be in 1..1984 (got %u)", k);
    if (nq == 0) return RADHIP_OK;
    if (count == 0) {
        for (uint32_t i = 0; i < nq; ++i) out_counts[i] = 0;
        for (size_t i = 0; i < (size_t)nq * k; ++i) { out_slots[i] = RADHIP_NO_SLOT; if (out_and) out_and[i] = 0; if (out_or) out_or[i] = 0; }
        return RADHIP_OK;
    }
    std::lock_guard<std::mutex> lk(idx->mu);
    RH_TRY(rh_ensure_device(idx));
    const uint32_t C = tk_pow2ceil(k + 64);
    // queries per pass: the four wavefronts of a block keep [queries][C] keys each in <= 64 KB of LDS
    const uint32_t pass = std::max<uint32_t>(1, std::min<uint32_t>(8, (64u * 1024u) / (4u * C * 8u)));
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, idx->device);
    const uint64_t tiles = (count + 63) / 64;
    // (the most blocks any instantiation holds resident: tk_resident_grid takes what its own occupancy allows)
    const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((tiles + 3) / 4, (uint64_t)n_cu * 8));
    std::vector<uint8_t> padded;
    std::vector<uint32_t> pop;
    rh_stage_queries(idx, queries, nq, padded, pop);
    uint4 *dq = nullptr;
    uint32_t *dpop = nullptr, *ds = nullptr, *da = nullptr, *dorr = nullptr, *dc = nullptr;
    unsigned long long *dcand = nullptr;
    auto cleanup = [&]() {
        void *ps[] = {dq, dpop, ds, da, dorr, dc, dcand};
        for (void *p : ps) if (p) (void)hipFree(p);
    };
#define TK_G(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { radhip_set_error("%s failed: %s", #x, hipGetErrorString(e_)); (void)hipGetLastError(); cleanup(); return e_ == hipErrorOutOfMemory ? RADHIP_E_NOMEM : RADHIP_E_HIP; } } while (0)
    TK_G(hipMalloc((void **)&dq, padded.size()));
    TK_G(hipMalloc((void **)&dpop, (size_t)nq * 4));
    TK_G(hipMalloc((void **)&ds, (size_t)pass * k * 4));
    TK_G(hipMalloc((void **)&da, (size_t)pass * k * 4));
    TK_G(hipMalloc((void **)&dorr, (size_t)pass * k * 4));
    TK_G(hipMalloc((void **)&dc, (size_t)pass * 4));
    TK_G(hipMalloc((void **)&dcand, (size_t)pass * grid * 4 * k * 8));
    TK_G(hipMemcpyAsync(dq, padded.data(), padded.size(), hipMemcpyHostToDevice, idx->stream));
    TK_G(hipMemcpyAsync(dpop, pop.data(), (size_t)nq * 4, hipMemcpyHostToDevice, idx->stream));
    int rc = RADHIP_OK;
    for (uint32_t q0 = 0; q0 < nq && rc == RADHIP_OK; q0 += pass) {
        const int nqp = (int)std::min<uint32_t>(pass, nq - q0);
        const uint4 *dqk = dq + (size_t)q0 * idx->lpr;
        switch (idx->lpr) {
            case 1: rc = tk_run_pass<1>(idx, nqp, first, count, dqk, dpop + q0, k, C, grid, n_cu, dcand, ds, da, dorr, dc); break;
            case 2: rc = tk_run_pass<2>(idx, nqp, first, count, dqk, dpop + q0, k, C, grid, n_cu, dcand, ds, da, dorr, dc); break;
            case 4: rc = tk_run_pass<4>(idx, nqp, first, count, dqk, dpop + q0, k, C, grid, n_cu, dcand, ds, da, dorr, dc); break;
            case 8: rc = tk_run_pass<8>(idx, nqp, first, count, dqk, dpop + q0, k, C, grid, n_cu, dcand, ds, da, dorr, dc); break;
            default: rc = tk_run_pass<16>(idx, nqp, first, count, dqk, dpop + q0, k, C, grid, n_cu, dcand, ds, da, dorr, dc); break;
        }
        if (rc != RADHIP_OK) break;
        TK_G(hipMemcpyAsync(out_slots + (size_t)q0 * k, ds, (size_t)nqp * k * 4, hipMemcpyDeviceToHost, idx->stream));
        if (out_and) TK_G(hipMemcpyAsync(out_and + (size_t)q0 * k, da, (size_t)nqp * k * 4, hipMemcpyDeviceToHost, idx->stream));
        if (out_or) TK_G(hipMemcpyAsync(out_or + (size_t)q0 * k, dorr, (size_t)nqp * k * 4, hipMemcpyDeviceToHost, idx->stream));
        TK_G(hipMemcpyAsync(out_counts + q0, dc, (size_t)nqp * 4, hipMemcpyDeviceToHost, idx->stream));
        TK_G(hipStreamSynchronize(idx->stream));
    }
#undef TK_G
    cleanup();
    return rc;
}
