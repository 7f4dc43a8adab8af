// trav4_hash.hip — trav4_kernel (traverse4.inc) with the one-entry-per-probe hash table (RADHIP_TABLE=hash), its row-sharded
// form (shard.hip's wave engine), the occupancy query and the staging-sort test hook.  One of the translation units the
// traversal kernels are split over so that the library builds in parallel.
#include "traverse_dev.h"
#include "traverse4.inc"

#define RH_T4_CASES(K, GRID, ST, P)                                                          \
    switch (lpr) {                                                                           \
        case 1: hipLaunchKernelGGL((K(1)), dim3(GRID), dim3(64), 0, ST, P); break;           \
        case 2: hipLaunchKernelGGL((K(2)), dim3(GRID), dim3(64), 0, ST, P); break;           \
        case 4: hipLaunchKernelGGL((K(4)), dim3(GRID), dim3(64), 0, ST, P); break;           \
        case 8: hipLaunchKernelGGL((K(8)), dim3(GRID), dim3(64), 0, ST, P); break;           \
        default: hipLaunchKernelGGL((K(16)), dim3(GRID), dim3(64), 0, ST, P); break;         \
    }

int rh_trav4_launch_hash(int lpr, uint32_t grid, hipStream_t st, const TravParams &P) {
#define RH_K(LPR) trav4_kernel<LPR, false>
    RH_T4_CASES(RH_K, grid, st, P)
#undef RH_K
    RH_HIP(hipGetLastError());
    return RADHIP_OK;
}

// one frontier step of the wave engine of the row-sharded mode (the fingerprint width only matters to the gather, which
// this form does not have: one instantiation)
int rh_trav4_launch_sharded(uint32_t grid, hipStream_t st, const TravParams &P) {
    hipLaunchKernelGGL((trav4_kernel<8, false, true>), dim3(grid), dim3(64), 0, st, P);
    RH_HIP(hipGetLastError());
    return RADHIP_OK;
}

int rh_trav4_launch(int table, bool wide, bool slot, int lpr, uint32_t grid, hipStream_t st, const TravParams &P) {
    if (table == RH_T4_GROUPED) return rh_trav4_launch_grouped(wide, slot, lpr, grid, st, P);
    if (table == RH_T4_BUCKET) return rh_trav4_launch_bucket(wide, slot, lpr, grid, st, P);
    if (table == RH_T4_LOCAL) return rh_trav4_launch_local(wide, slot, lpr, grid, st, P);
    if (wide || slot) RH_FAIL(RADHIP_E_STATE, "the hash-table form of trav4_kernel has no wide-row / per-slot variant");
    return rh_trav4_launch_hash(lpr, grid, st, P);
}

// wavefronts of trav4_kernel one CU holds (every table form is built for the same register / LDS budget)
int rh_trav4_occupancy(int lpr, int *per_cu) {
    hipError_t e;
    switch (lpr) {
        case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, trav4_kernel<1, false>, 64, 0); break;
        case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, trav4_kernel<2, false>, 64, 0); break;
        case 4: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, trav4_kernel<4, false>, 64, 0); break;
        case 8: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, trav4_kernel<8, false>, 64, 0); break;
        default: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, trav4_kernel<16, false>, 64, 0); break;
    }
    RH_HIP(e);
    return RADHIP_OK;
}

uint32_t rh_trav4_staging_capacity() { return T4_S; }

// test hook: trav4_kernel's staging sort (t4_sort256: register-resident flip-form bitonic network, one wavefront)
__global__ __launch_bounds__(64) void debug_sort_kernel(const unsigned long long *in, const uint32_t *counts, unsigned long long *out) {
    __shared__ unsigned long long S[256];
    const uint32_t lane = threadIdx.x, n = counts[blockIdx.x];
    for (uint32_t i = lane; i < 256u; i += 64u) S[i] = i < n ? in[(uint64_t)blockIdx.x * 256u + i] : 0x1234ull;   // junk beyond n: the sort must not read it
    WSYNC();
    t4_sort256(S, n, lane);
    WSYNC();
    for (uint32_t i = lane; i < T4_S; i += 64u) out[(uint64_t)blockIdx.x * 256u + i] = S[i];
}

extern "C" int radhip_debug_sort_staging(radhip_index_t *idx, const uint64_t *keys, const uint32_t *counts, uint32_t batches,
                                         uint64_t *out) {
    if (!idx || !keys || !counts || !out) RH_FAIL(RADHIP_E_INVALID, "null argument");
    if (batches == 0) return RADHIP_OK;
    for (uint32_t b = 0; b < batches; ++b) if (counts[b] > T4_S) RH_FAIL(RADHIP_E_INVALID, "a staging buffer holds at most %u keys", T4_S);
    std::lock_guard<std::mutex> lk(idx->mu);
    RH_TRY(rh_ensure_device(idx));
    unsigned long long *din = nullptr, *dout = nullptr;
    uint32_t *dc = nullptr;
    int rc = RADHIP_OK;
    const size_t bytes = (size_t)batches * 256 * 8;
    if (hipMalloc((void **)&din, bytes) != hipSuccess || hipMalloc((void **)&dout, bytes) != hipSuccess ||
        hipMalloc((void **)&dc, (size_t)batches * 4) != hipSuccess) rc = RADHIP_E_NOMEM;
    if (rc == RADHIP_OK && (hipMemcpy(din, keys, bytes, hipMemcpyHostToDevice) != hipSuccess ||
                            hipMemcpy(dc, counts, (size_t)batches * 4, hipMemcpyHostToDevice) != hipSuccess ||
                            hipMemsetAsync(dout, 0, bytes, idx->stream) != hipSuccess)) rc = RADHIP_E_HIP;
    if (rc == RADHIP_OK) {
        hipLaunchKernelGGL(debug_sort_kernel, dim3(batches), dim3(64), 0, idx->stream, din, dc, dout);
        if (hipStreamSynchronize(idx->stream) != hipSuccess) rc = RADHIP_E_HIP;
        else if (hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = RADHIP_E_HIP;
    }
    if (din) (void)hipFree(din);
    if (dout) (void)hipFree(dout);
    if (dc) (void)hipFree(dc);
    if (rc != RADHIP_OK) radhip_set_error("radhip_debug_sort_staging failed (%d)", rc);
    return rc;
}
