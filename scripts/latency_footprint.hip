// Random-access latency and request rate of one MI355X as a function of the FOOTPRINT the accesses spread over
// (does address translation cost a traversal kernel whose state spans 100+ GB?).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/bin/latency_footprint scripts/latency_footprint.hip
//   scripts/bin/latency_footprint [max_GiB = 128]
// (a) dependent chain: one wavefront, lane 0 walks a random cycle of 128-B lines -> ns per hop (unloaded latency)
// (b) loaded: 256 x 16 wavefronts, every lane loads 16 B from a random line, 4 independent loads per iteration ->
//     line requests per second and the time one wavefront's batch of loads takes
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x;
}
// line i of the buffer holds the index of the next line: next = (i * A + C) mod lines with lines a power of two (full period)
__global__ void fill_chain(uint4 *buf, uint64_t lines) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < lines; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t nx = (i * 6364136223846793005ull + 1442695040888963407ull) & (lines - 1);
        buf[i * 8] = make_uint4((uint32_t)nx, (uint32_t)(nx >> 32), 0, 0);
    }
}
__global__ void chase(const uint4 *buf, uint64_t hops, uint64_t *out) {
    if (threadIdx.x != 0) return;
    uint64_t i = 12345;
    for (uint64_t h = 0; h < hops; ++h) {
        const uint4 v = buf[i * 8];
        i = (uint64_t)v.x | ((uint64_t)v.y << 32);
    }
    out[0] = i;
}
__global__ __launch_bounds__(64) void loaded(const uint4 *buf, uint64_t lines, uint32_t iters, uint64_t *out) {
    uint64_t s = mix(((uint64_t)blockIdx.x << 8 | threadIdx.x) + 0x9E37ull);
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { s = mix(s + 0x9E3779B97F4A7C15ull); v[u] = buf[(s & (lines - 1)) * 8]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u].x + v[u].z;
        s ^= acc & 1u;   // the next addresses depend on the loaded data: batches are dependent
    }
    if (acc == 0x12345678u) out[1] = acc;
}

int main(int argc, char **argv) {
    const double max_gib = argc > 1 ? atof(argv[1]) : 128.0;
    uint64_t *out; CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (double gib = 0.25; gib <= max_gib; gib *= 4) {
        const uint64_t bytes = (uint64_t)(gib * (1ull << 30)), lines = bytes / 128;
        uint4 *buf; CK(hipMalloc(&buf, bytes));
        fill_chain<<<4096, 256>>>(buf, lines); CK(hipDeviceSynchronize());
        const uint64_t hops = 200000;
        chase<<<1, 64>>>(buf, 1000, out); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); chase<<<1, 64>>>(buf, hops, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double ns_hop = ms * 1e6 / hops;
        for (uint32_t wpc : {4u, 16u, 32u}) {
            const uint32_t blocks = 256 * wpc, iters = 2000;
            loaded<<<blocks, 64>>>(buf, lines, 50, out); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); loaded<<<blocks, 64>>>(buf, lines, iters, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double req = (double)blocks * 64 * 4 * iters;
            printf("footprint %7.2f GiB: chain %6.0f ns/hop | %2u waves/CU: %6.2f G line requests/s, %6.2f us per dependent batch of 256 lines\n",
                   gib, ns_hop, wpc, req / (ms * 1e-3) / 1e9, ms * 1e3 / iters);
        }
        fflush(stdout);
        CK(hipFree(buf));
    }
    return 0;
}
