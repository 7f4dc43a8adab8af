// hbm_random.hip — what HBM delivers for the traversal's access pattern (gfx950).
// Random 128-B line reads over a large buffer, (a) as whole 128-B rows read by 8 lanes x 16 B
// (fingerprint gathers), (b) as one 8-B word per line (visited-table probes).  D independent
// loads in flight per lane.  Prints lines/s x 128 B: the ceiling the traversal kernel's real
// traffic should be priced against.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_random scripts/hbm_random.hip && /tmp/hbm_random
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

// (a) rows: lane group of 8 reads one 128-B row; D rows in flight per group
template <int D>
__global__ __launch_bounds__(256) void rows_kernel(const uint4 *buf, uint64_t n_lines, uint32_t iters, uint32_t *sink) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t grp = gid >> 3, chunk = gid & 7;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        uint4 v[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const uint64_t line = mix(grp * 0x9E3779B97F4A7C15ull + (uint64_t)it * D + d) % n_lines;
            v[d] = buf[line * 8 + chunk];
        }
#pragma unroll
        for (int d = 0; d < D; ++d) acc += __popc(v[d].x) + __popc(v[d].y) + __popc(v[d].z) + __popc(v[d].w);
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

// (b) words: every lane reads one 8-B word of its own random line; D in flight per lane
template <int D>
__global__ __launch_bounds__(256) void words_kernel(const unsigned long long *buf, uint64_t n_lines, uint32_t iters, uint32_t *sink) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        unsigned long long v[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const uint64_t r = mix(gid * 0x9E3779B97F4A7C15ull + (uint64_t)it * D + d);
            const uint64_t line = r % n_lines;
            v[d] = __hip_atomic_load(buf + line * 16 + ((r >> 60) & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) acc += v[d];
    }
    if (acc == 0x123456789ull) sink[0] = (uint32_t)acc;
}

// (c) probe + insert: every lane reads one 8-B word of a random line, then stores 8 B into the same
// line for half of them (what a visited-table probe that finds a new node does)
template <int D>
__global__ __launch_bounds__(256) void probe_insert_kernel(unsigned long long *buf, uint64_t n_lines, uint32_t iters, uint32_t *sink) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        unsigned long long v[D];
        uint64_t off[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const uint64_t r = mix(gid * 0x9E3779B97F4A7C15ull + (uint64_t)it * D + d);
            off[d] = (r % n_lines) * 16 + ((r >> 60) & 15);
            v[d] = __hip_atomic_load(buf + off[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            acc += v[d];
            if (off[d] & 1) __hip_atomic_store(buf + off[d], v[d] + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (acc == 0x123456789ull) sink[0] = (uint32_t)acc;
}

// (d) as (c) with whole aligned sectors: the lane reads SB bytes around its word (SB = 16/32/64) and
// stores all of them back — does a full-sector store avoid the read-modify-write of a partial one?
template <int SB>
__global__ __launch_bounds__(256) void sector_insert_kernel(uint4 *buf, uint64_t n_lines, uint32_t iters, uint32_t *sink) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    constexpr int Q = SB / 16;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint64_t r = mix(gid * 0x9E3779B97F4A7C15ull + (uint64_t)it);
        const uint64_t base = (r % n_lines) * 8 + (((r >> 60) & 7) / Q) * Q;   // uint4 index of the sector
        uint4 v[Q];
#pragma unroll
        for (int k = 0; k < Q; ++k) v[k] = buf[base + k];
#pragma unroll
        for (int k = 0; k < Q; ++k) acc += v[k].x;
        if (r & (1ull << 40)) {
            v[0].x += 1;
#pragma unroll
            for (int k = 0; k < Q; ++k) buf[base + k] = v[k];
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// (e) store flavours for the insert of (c): MODE 0 plain store, 1 nontemporal store, 2 no load at all
// (pure scatter of 8-B stores), 3 nontemporal load + nontemporal store
template <int MODE>
__global__ __launch_bounds__(256) void store_flavour_kernel(unsigned long long *buf, uint64_t n_lines, uint32_t iters, uint32_t *sink) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint64_t r = mix(gid * 0x9E3779B97F4A7C15ull + (uint64_t)it);
        const uint64_t off = (r % n_lines) * 16 + ((r >> 60) & 15);
        unsigned long long v = r;
        if (MODE == 0 || MODE == 1) v = __hip_atomic_load(buf + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 3) v = __builtin_nontemporal_load(buf + off);
        acc += v;
        if (off & 1) {
            if (MODE == 0 || MODE == 2) buf[off] = v + 1;
            else __builtin_nontemporal_store(v + 1, buf + off);
        }
    }
    if (acc == 0x123456789ull) sink[0] = (uint32_t)acc;
}

// (f) whole-line and half-line random writes, no reads: 8 (or 4) lanes x 16 B into one random line
template <int LANES>
__global__ __launch_bounds__(256) void line_write_kernel(uint4 *buf, uint64_t n_lines, uint32_t iters, uint32_t *sink) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t grp = gid / LANES, chunk = gid % LANES;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint64_t line = mix(grp * 0x9E3779B97F4A7C15ull + (uint64_t)it) % n_lines;
        buf[line * 8 + chunk] = make_uint4((uint32_t)it, (uint32_t)gid, 3u, 4u);
    }
    if (gid == 0xFFFFFFFFFFull) sink[0] = 1;
}

// (g) probe 8 B per lane; the inserts (half of the lanes) are done by QUADS of lanes: for each
// inserting lane of a quad, its 4 lanes read the 64-B half-line around the entry (16 B each: one
// coalesced 64-B request, an L2 hit) and write all 64 B back — a fully dirty half-line, which the
// HBM takes without a read-modify-write
__global__ __launch_bounds__(256) void quad_insert_kernel(unsigned long long *buf, uint64_t n_lines, uint32_t iters, uint32_t *sink) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, ql = lane & 3;
    unsigned long long acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint64_t r = mix(gid * 0x9E3779B97F4A7C15ull + (uint64_t)it);
        const uint64_t off = (r % n_lines) * 16 + ((r >> 60) & 15);        // entry index (8-B units)
        const unsigned long long v = __hip_atomic_load(buf + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc += v;
        const bool ins = off & 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int src = (lane & ~3) | k;
            const bool go = __shfl((int)ins, src) != 0;
            const uint64_t eo = ((uint64_t)(uint32_t)__shfl((int)(off >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)off, src);
            const unsigned long long nv = ((unsigned long long)(uint32_t)__shfl((int)(v >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)v, src);
            if (go) {
                uint4 *hl = reinterpret_cast<uint4 *>(buf + (eo & ~7ull)) + ql;   // my 16 B of the half-line
                uint4 w = *hl;
                if (((eo & 7) >> 1) == (uint64_t)ql) { if (eo & 1) { w.z = (uint32_t)nv + 1; w.w = (uint32_t)(nv >> 32); } else { w.x = (uint32_t)nv + 1; w.y = (uint32_t)(nv >> 32); } }
                *hl = w;
            }
        }
    }
    if (acc == 0x123456789ull) sink[0] = (uint32_t)acc;
}

template <typename F>
static double time_ms(F launch) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch();  // warm
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms;
}

int main(int argc, char **argv) {
    const uint64_t gib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 32;
    const uint64_t bytes = gib << 30, n_lines = bytes / 128;
    const unsigned alloc_flags = argc > 2 ? (unsigned)strtoul(argv[2], nullptr, 10) : 0u;   // 0 default, 1 fine-grained, 3 uncached
    void *buf; uint32_t *sink;
    CK(hipExtMallocWithFlags(&buf, bytes, alloc_flags)); CK(hipMalloc(&sink, 4));
    printf("allocation flags %u (0 default, 1 fine-grained, 3 uncached)\n", alloc_flags);
    CK(hipMemset(buf, 0x5A, bytes));
    printf("buffer %llu GiB, %llu lines of 128 B\n", (unsigned long long)gib, (unsigned long long)n_lines);
    const uint32_t blocks_per_cu[] = {2, 4, 8};
    for (uint32_t bpc : blocks_per_cu) {
        const uint32_t grid = 256 * bpc;   // 256 threads = 4 waves per block
        const uint32_t iters = 32768;
#define RUN_ROWS(D)                                                                                               \
        {                                                                                                         \
            double ms = time_ms([&] { rows_kernel<D><<<grid, 256>>>((const uint4 *)buf, n_lines, iters / D, sink); }); \
            double lines = (double)grid * 256 / 8 * (iters / D) * D;                                              \
            printf("rows  D=%d waves/CU=%2u: %7.1f ms  %6.2f G lines/s  %7.1f GB/s\n", D, bpc * 4, ms, lines / ms / 1e6, lines * 128 / ms / 1e6); \
        }
        RUN_ROWS(1) RUN_ROWS(2) RUN_ROWS(4) RUN_ROWS(8)
#define RUN_WORDS(D)                                                                                              \
        {                                                                                                         \
            double ms = time_ms([&] { words_kernel<D><<<grid, 256>>>((const unsigned long long *)buf, n_lines, iters / 8 / D, sink); }); \
            double lines = (double)grid * 256 * (iters / 8 / D) * D;                                              \
            printf("words D=%d waves/CU=%2u: %7.1f ms  %6.2f G lines/s  %7.1f GB/s (as 128-B lines)\n", D, bpc * 4, ms, lines / ms / 1e6, lines * 128 / ms / 1e6); \
        }
        RUN_WORDS(1) RUN_WORDS(2) RUN_WORDS(4)
#define RUN_PI(D)                                                                                                 \
        {                                                                                                         \
            double ms = time_ms([&] { probe_insert_kernel<D><<<grid, 256>>>((unsigned long long *)buf, n_lines, iters / 8 / D, sink); }); \
            double lines = (double)grid * 256 * (iters / 8 / D) * D;                                              \
            printf("probe+insert(50%%) D=%d waves/CU=%2u: %7.1f ms  %6.2f G probes/s (+ %.2f G stores/s)\n", D, bpc * 4, ms, lines / ms / 1e6, lines / 2 / ms / 1e6); \
        }
        RUN_PI(1) RUN_PI(4)
#define RUN_SI(SB)                                                                                                \
        {                                                                                                         \
            double ms = time_ms([&] { sector_insert_kernel<SB><<<grid, 256>>>((uint4 *)buf, n_lines, iters / 8, sink); }); \
            double lines = (double)grid * 256 * (iters / 8);                                                      \
            printf("sector probe+insert(50%%) %2d-B sectors waves/CU=%2u: %7.1f ms  %6.2f G probes/s (+ %.2f G stores/s)\n", SB, bpc * 4, ms, lines / ms / 1e6, lines / 2 / ms / 1e6); \
        }
        RUN_SI(16) RUN_SI(32) RUN_SI(64)
#define RUN_SF(MODE, NAME)                                                                                        \
        {                                                                                                         \
            double ms = time_ms([&] { store_flavour_kernel<MODE><<<grid, 256>>>((unsigned long long *)buf, n_lines, iters / 8, sink); }); \
            double lines = (double)grid * 256 * (iters / 8);                                                      \
            printf("store flavour %-34s waves/CU=%2u: %7.1f ms  %6.2f G lines/s (+ %.2f G stores/s)\n", NAME, bpc * 4, ms, lines / ms / 1e6, lines / 2 / ms / 1e6); \
        }
        {
            double ms = time_ms([&] { line_write_kernel<8><<<grid, 256>>>((uint4 *)buf, n_lines, iters / 8, sink); });
            double lines = (double)grid * 256 / 8 * (iters / 8);
            printf("whole-line (128 B) random writes waves/CU=%2u: %7.1f ms  %6.2f G lines/s  %7.1f GB/s\n", bpc * 4, ms, lines / ms / 1e6, lines * 128 / ms / 1e6);
            ms = time_ms([&] { line_write_kernel<4><<<grid, 256>>>((uint4 *)buf, n_lines, iters / 8, sink); });
            lines = (double)grid * 256 / 4 * (iters / 8);
            printf("half-line (64 B) random writes waves/CU=%2u: %7.1f ms  %6.2f G writes/s\n", bpc * 4, ms, lines / ms / 1e6);
            ms = time_ms([&] { line_write_kernel<2><<<grid, 256>>>((uint4 *)buf, n_lines, iters / 8, sink); });
            lines = (double)grid * 256 / 2 * (iters / 8);
            printf("sector (32 B) random writes waves/CU=%2u: %7.1f ms  %6.2f G writes/s\n", bpc * 4, ms, lines / ms / 1e6);
        }
        {
            double ms = time_ms([&] { quad_insert_kernel<<<grid, 256>>>((unsigned long long *)buf, n_lines, iters / 8, sink); });
            double lines = (double)grid * 256 * (iters / 8);
            printf("probe + quad half-line insert(50%%) waves/CU=%2u: %7.1f ms  %6.2f G probes/s (+ %.2f G inserts/s)\n", bpc * 4, ms, lines / ms / 1e6, lines / 2 / ms / 1e6);
        }
        RUN_SF(0, "load + plain store") RUN_SF(1, "load + nontemporal store") RUN_SF(2, "no load, plain store (scatter)") RUN_SF(3, "nt load + nt store")
        fflush(stdout);
    }
    CK(hipFree(buf)); CK(hipFree(sink));
    return 0;
}
