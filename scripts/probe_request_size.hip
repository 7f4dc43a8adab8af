// What does ONE probe cost the memory side?  (VERDICT r03 #7b)
// The traversal kernel's table probe is a one-lane 16-B load from a random 128-B line; bench.py's `roofline.traffic`
// priced every memory-side read request at 128 B because TCC_EA0_RDREQ_32B was 0.  This program issues a KNOWN number of
// requests in each access shape the kernel uses, one kernel name per shape, so that
//     rocprofv3 --kernel-trace --pmc FETCH_SIZE            -- scripts/bin/probe_request_size
//     rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ TCC_EA0_WRREQ_64B -- ...
//     rocprofv3 --kernel-trace --pmc WRITE_SIZE            -- ...
// give bytes per request per shape (scripts/probe_request_size.sh sums the CSVs).  Without a profiler it prints the
// request rate of every shape.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/bin/probe_request_size scripts/probe_request_size.hip
//   scripts/bin/probe_request_size [footprint GiB = 32] [waves per CU = 16]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x;
}
// LANES lanes share one random 128-B line and read BYTES each at consecutive offsets: <1,16> = the bucket probe,
// <1,4> = a one-entry probe, <4,16> = half a line (a 64-B adjacency row), <8,16> = a whole fingerprint row.
template <int LANES, int BYTES, int ACTIVE = 64>
__device__ __forceinline__ void rd_body(const uint8_t *buf, uint64_t lines, uint32_t iters, uint64_t *out) {
    const uint32_t lane = threadIdx.x, grp = lane / LANES, sub = lane % LANES;
    if (ACTIVE == 32 && (lane & 1u)) return;      // every other lane: 32 lines per instruction instead of 64
    uint64_t s = mix(((uint64_t)blockIdx.x << 8 | grp) + 0x9E37ull);
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        uint32_t got[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s = mix(s + 0x9E3779B97F4A7C15ull);
            const uint8_t *p = buf + (s & (lines - 1)) * 128 + sub * BYTES;
            if constexpr (BYTES == 16) { const uint4 v = *reinterpret_cast<const uint4 *>(p); got[u] = v.x + v.w; }
            else if constexpr (BYTES == 8) { const uint2 v = *reinterpret_cast<const uint2 *>(p); got[u] = v.x + v.y; }
            else { got[u] = *reinterpret_cast<const uint32_t *>(p); }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += got[u];
        s ^= (uint64_t)(acc == 0xDEADBEEFu);   // the next addresses wait for the data without changing
    }
    if (acc == 0x12345678u) out[1] = acc;
}
__global__ __launch_bounds__(64) void read_1lane_16B(const uint8_t *b, uint64_t l, uint32_t it, uint64_t *o) { rd_body<1, 16>(b, l, it, o); }
__global__ __launch_bounds__(64) void read_1lane_4B(const uint8_t *b, uint64_t l, uint32_t it, uint64_t *o) { rd_body<1, 4>(b, l, it, o); }
__global__ __launch_bounds__(64) void read_1lane_8B(const uint8_t *b, uint64_t l, uint32_t it, uint64_t *o) { rd_body<1, 8>(b, l, it, o); }
__global__ __launch_bounds__(64) void read_1lane_16B_32of64(const uint8_t *b, uint64_t l, uint32_t it, uint64_t *o) { rd_body<1, 16, 32>(b, l, it, o); }
__global__ __launch_bounds__(64) void read_2lanes_32B(const uint8_t *b, uint64_t l, uint32_t it, uint64_t *o) { rd_body<2, 16>(b, l, it, o); }
__global__ __launch_bounds__(64) void read_4lanes_64B(const uint8_t *b, uint64_t l, uint32_t it, uint64_t *o) { rd_body<4, 16>(b, l, it, o); }
__global__ __launch_bounds__(64) void read_8lanes_128B(const uint8_t *b, uint64_t l, uint32_t it, uint64_t *o) { rd_body<8, 16>(b, l, it, o); }

// stores: <1,4> = a table entry, <1,16> = a bucket / chunk, <4,16> = half a line, <8,16> = a whole line (scored ring, far run)
template <int LANES, int BYTES>
__device__ __forceinline__ void wr_body(uint8_t *buf, uint64_t lines, uint32_t iters) {
    const uint32_t lane = threadIdx.x, grp = lane / LANES, sub = lane % LANES;
    uint64_t s = mix(((uint64_t)blockIdx.x << 8 | grp) + 0x51ull);
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s = mix(s + 0x9E3779B97F4A7C15ull);
            uint8_t *p = buf + (s & (lines - 1)) * 128 + sub * BYTES;
            if constexpr (BYTES == 16) *reinterpret_cast<uint4 *>(p) = make_uint4((uint32_t)s, it, lane, 7u);
            else if constexpr (BYTES == 8) *reinterpret_cast<uint2 *>(p) = make_uint2((uint32_t)s, it);
            else *reinterpret_cast<uint32_t *>(p) = (uint32_t)s;
        }
    }
}
__global__ __launch_bounds__(64) void write_1lane_4B(uint8_t *b, uint64_t l, uint32_t it) { wr_body<1, 4>(b, l, it); }
__global__ __launch_bounds__(64) void write_1lane_16B(uint8_t *b, uint64_t l, uint32_t it) { wr_body<1, 16>(b, l, it); }
__global__ __launch_bounds__(64) void write_1lane_8B(uint8_t *b, uint64_t l, uint32_t it) { wr_body<1, 8>(b, l, it); }
__global__ __launch_bounds__(64) void write_2lanes_32B(uint8_t *b, uint64_t l, uint32_t it) { wr_body<2, 16>(b, l, it); }
__global__ __launch_bounds__(64) void write_4lanes_64B(uint8_t *b, uint64_t l, uint32_t it) { wr_body<4, 16>(b, l, it); }
__global__ __launch_bounds__(64) void write_8lanes_128B(uint8_t *b, uint64_t l, uint32_t it) { wr_body<8, 16>(b, l, it); }

// the traversal's mix per expansion, roughly: 1 half-line read, 10 one-lane 16-B reads, 4 whole rows, 4 one-lane 4-B stores
__global__ __launch_bounds__(64) void mix_like_an_expansion(uint8_t *buf, uint64_t lines, uint32_t iters, uint64_t *out) {
    const uint32_t lane = threadIdx.x, row = lane >> 4, gl = lane & 15u;
    uint64_t s = mix(((uint64_t)blockIdx.x << 8 | row) + 0x77ull);
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        s = mix(s + 0x9E3779B97F4A7C15ull);
        const uint32_t a = *reinterpret_cast<const uint32_t *>(buf + (s & (lines - 1)) * 128 + gl * 4);          // adjacency row
        uint64_t t = mix(s ^ ((uint64_t)gl << 40) ^ (uint64_t)(a == 0xDEADBEEFu));
        uint4 pv = make_uint4(0, 0, 0, 0);
        if (gl < 10) pv = *reinterpret_cast<const uint4 *>(buf + (t & (lines - 1)) * 128 + (t >> 61) * 16);      // ten probes
        acc += pv.x;
        uint4 v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {                                                                            // four rows (two per pass)
            const uint64_t r = mix(s + (uint64_t)u * 2 + (gl >> 3) + (uint64_t)(acc == 0xDEADBEEFu));
            v[u] = *reinterpret_cast<const uint4 *>(buf + (r & (lines - 1)) * 128 + (gl & 7u) * 16);
        }
        acc += v[0].x + v[1].y;
        if (gl < 4) *reinterpret_cast<uint32_t *>(buf + (mix(t + acc) & (lines - 1)) * 128 + 64) = acc;           // four entry stores
        s ^= (uint64_t)(acc == 0xDEADBEEFu);   // the next addresses wait for the data without changing
    }
    if (acc == 0x12345678u) out[1] = acc;
}

// The same mix with the four entry stores (a) into the line a probe of this round has just fetched, as in the kernel, and / or
// (b) as FULL 64-B sectors (four lanes x 16 B: the cheap kind of write) instead of 4-B partial writes: what would a table whose
// entry store rewrites a whole sector buy?  (The sector's other bytes would come from a probe widened to 64 B: the same request.)
template <bool W64, bool SAME_LINE>
__device__ __forceinline__ void mix_body(uint8_t *buf, uint64_t lines, uint32_t iters, uint64_t *out) {
    const uint32_t lane = threadIdx.x, row = lane >> 4, gl = lane & 15u;
    uint64_t s = mix(((uint64_t)blockIdx.x << 8 | row) + 0x77ull);
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        s = mix(s + 0x9E3779B97F4A7C15ull);
        const uint32_t a = *reinterpret_cast<const uint32_t *>(buf + (s & (lines - 1)) * 128 + gl * 4);
        uint64_t t = mix(s ^ ((uint64_t)gl << 40) ^ (uint64_t)(a == 0xDEADBEEFu));
        uint4 pv = make_uint4(0, 0, 0, 0);
        if (gl < 10) pv = *reinterpret_cast<const uint4 *>(buf + (t & (lines - 1)) * 128 + (t >> 61) * 16);
        acc += pv.x;
        uint4 v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint64_t r = mix(s + (uint64_t)u * 2 + (gl >> 3) + (uint64_t)(acc == 0xDEADBEEFu));
            v[u] = *reinterpret_cast<const uint4 *>(buf + (r & (lines - 1)) * 128 + (gl & 7u) * 16);
        }
        acc += v[0].x + v[1].y;
        if constexpr (W64) {
            const uint32_t src = (lane & ~15u) + (gl >> 2);                 // lanes 4g .. 4g+3 write the sector lane g probed
            const uint64_t tg = SAME_LINE ? __shfl(t, src) : mix(__shfl(t, src) + 1);
            *reinterpret_cast<uint4 *>(buf + (tg & (lines - 1)) * 128 + (tg >> 63) * 64 + (gl & 3u) * 16) = make_uint4(acc, it, lane, 7u);
        } else {
            const uint64_t tg = SAME_LINE ? t : mix(t + acc);
            if (gl < 4) *reinterpret_cast<uint32_t *>(buf + (tg & (lines - 1)) * 128 + (tg >> 61) * 16 + 4) = acc;
        }
        s ^= (uint64_t)(acc == 0xDEADBEEFu);
    }
    if (acc == 0x12345678u) out[1] = acc;
}
__global__ __launch_bounds__(64) void mix_store4_other_line(uint8_t *b, uint64_t l, uint32_t it, uint64_t *o) { mix_body<false, false>(b, l, it, o); }
__global__ __launch_bounds__(64) void mix_store4_probed_line(uint8_t *b, uint64_t l, uint32_t it, uint64_t *o) { mix_body<false, true>(b, l, it, o); }
__global__ __launch_bounds__(64) void mix_store64_other_line(uint8_t *b, uint64_t l, uint32_t it, uint64_t *o) { mix_body<true, false>(b, l, it, o); }
__global__ __launch_bounds__(64) void mix_store64_probed_line(uint8_t *b, uint64_t l, uint32_t it, uint64_t *o) { mix_body<true, true>(b, l, it, o); }

// What a new table entry costs: a group of 4 lanes works on one random line per step.  Lane 0 probes 16 B of it (the bucket
// probe); then either lane 0 stores 4 B into it (today's entry store: a partial write) or all four lanes read the 64-B half
// line back (an L2 hit: the probe has just fetched the line) and store the whole 64 B (REWRITE = true).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_sc1(const uint8_t *p) {    // the traversal kernel's probe load: agent scope (past the L1), 16 B
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return make_uint4(v.x, v.y, v.z, v.w);
}
template <bool REWRITE>
__device__ __forceinline__ void rmw_body(uint8_t *buf, uint64_t lines, uint32_t iters, uint64_t *out) {
    const uint32_t lane = threadIdx.x, grp = lane >> 2, sub = lane & 3u;
    uint64_t s = mix(((uint64_t)blockIdx.x << 8 | grp) + 0x33ull);
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        s = mix(s + 0x9E3779B97F4A7C15ull);
        uint8_t *line = buf + (s & (lines - 1)) * 128;
        uint4 pv = make_uint4(0, 0, 0, 0);
        if (sub == 0) pv = ld_sc1(line + 16);
        acc += pv.x;
        acc = __shfl(acc, lane & ~3u);                       // the group waits for the probe
        if constexpr (REWRITE) {
            uint4 v = ld_sc1(line + sub * 16 + (acc == 0xDEADBEEFu ? 64 : 0));
            if (sub == 1) v.y = acc + it;
            *reinterpret_cast<uint4 *>(line + sub * 16) = v;
        } else {
            if (sub == 0) *reinterpret_cast<uint32_t *>(line + 20) = acc + it;
        }
        s ^= (uint64_t)(acc == 0xDEADBEEFu);
    }
    if (acc == 0x12345678u) out[1] = acc;
}
__global__ __launch_bounds__(64) void entry_probe16_store4(uint8_t *b, uint64_t l, uint32_t it, uint64_t *o) { rmw_body<false>(b, l, it, o); }
__global__ __launch_bounds__(64) void entry_probe16_rewrite64(uint8_t *b, uint64_t l, uint32_t it, uint64_t *o) { rmw_body<true>(b, l, it, o); }

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 32.0;
    const uint32_t wpc = argc > 2 ? (uint32_t)atoi(argv[2]) : 16u;
    uint64_t lines = 1; while (lines * 2 * 128 <= (uint64_t)(gib * (1ull << 30))) lines *= 2;
    const unsigned kind = argc > 3 ? (unsigned)atoi(argv[3]) : 0u;       // 0 plain hipMalloc, 1 fine-grained, 3 uncached (hipExtMallocWithFlags)
    uint8_t *buf;
    if (kind) CK(hipExtMallocWithFlags((void **)&buf, lines * 128, kind)); else CK(hipMalloc(&buf, lines * 128));
    CK(hipMemset(buf, 1, lines * 128));
    printf("allocation kind %u (%s)\n", kind, kind == 0 ? "hipMalloc" : kind == 1 ? "hipDeviceMallocFinegrained" : kind == 3 ? "hipDeviceMallocUncached" : "?");
    uint64_t *out; CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t blocks = 256 * wpc, iters = 1000;
    float ms;
    printf("footprint %.1f GiB (%llu lines), %u wavefronts per CU, %u iterations x 4 accesses per lane\n", lines * 128 / 1073741824.0,
           (unsigned long long)lines, wpc, iters);
    fflush(stdout);
#define RUN(K, REQ_PER_LANE_ACCESS, ...)                                                                                   \
    do {                                                                                                                   \
        K<<<blocks, 64>>>(__VA_ARGS__); CK(hipDeviceSynchronize());                                                        \
        CK(hipEventRecord(e0)); K<<<blocks, 64>>>(__VA_ARGS__); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));       \
        CK(hipEventElapsedTime(&ms, e0, e1));                                                                              \
        const double req = (double)blocks * 64 * 4 * iters * (REQ_PER_LANE_ACCESS);                                        \
        printf("%-22s %8.3f ms  %14.0f line requests per launch  %7.2f G lines/s\n", #K, ms, req, req / (ms * 1e-3) / 1e9);  \
        fflush(stdout);                                                                                                    \
    } while (0)
    RUN(read_1lane_16B, 1.0, buf, lines, iters, out);
    RUN(read_1lane_4B, 1.0, buf, lines, iters, out);
    RUN(read_1lane_8B, 1.0, buf, lines, iters, out);
    RUN(read_1lane_16B_32of64, 0.5, buf, lines, iters, out);
    RUN(read_2lanes_32B, 0.5, buf, lines, iters, out);
    RUN(read_4lanes_64B, 0.25, buf, lines, iters, out);
    RUN(read_8lanes_128B, 0.125, buf, lines, iters, out);
    RUN(write_1lane_4B, 1.0, buf, lines, iters);
    RUN(write_1lane_16B, 1.0, buf, lines, iters);
    RUN(write_1lane_8B, 1.0, buf, lines, iters);
    RUN(write_2lanes_32B, 0.5, buf, lines, iters);
    RUN(write_4lanes_64B, 0.25, buf, lines, iters);
    RUN(write_8lanes_128B, 0.125, buf, lines, iters);
    RUN(entry_probe16_store4, 0.25 / 4.0, buf, lines, iters, out);       // (operations per launch: one per group of 4 lanes and iteration)
    RUN(entry_probe16_rewrite64, 0.25 / 4.0, buf, lines, iters, out);
    {
        mix_like_an_expansion<<<blocks, 64>>>(buf, lines, iters, out); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); mix_like_an_expansion<<<blocks, 64>>>(buf, lines, iters, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double rows = (double)blocks * 4 * iters;      // "expansions"
        printf("%-22s %8.3f ms  %14.0f expansion-shaped rounds (1 half line + 10 probes + 4 rows + 4 entry stores = 19 requests)  %7.2f G rounds/s = %7.2f G requests/s\n",
               "mix_like_an_expansion", ms, rows, rows / (ms * 1e-3) / 1e9, rows * 19 / (ms * 1e-3) / 1e9);
    }
#define RUNMIX(K)                                                                                                          \
    do {                                                                                                                   \
        K<<<blocks, 64>>>(buf, lines, iters, out); CK(hipDeviceSynchronize());                                             \
        CK(hipEventRecord(e0)); K<<<blocks, 64>>>(buf, lines, iters, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); \
        CK(hipEventElapsedTime(&ms, e0, e1));                                                                              \
        printf("%-26s %8.3f ms  %7.3f G rounds/s\n", #K, ms, (double)blocks * 4 * iters / (ms * 1e-3) / 1e9); fflush(stdout);  \
    } while (0)
    RUNMIX(mix_store4_other_line);
    RUNMIX(mix_store4_probed_line);
    RUNMIX(mix_store64_other_line);
    RUNMIX(mix_store64_probed_line);
    return 0;
}
