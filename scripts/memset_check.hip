// Does hipMemset / hipMemsetAsync fill ALL bytes of a > 4 GiB range on this stack, on memory that held other data before?
// (bench.py's second graph build faulted with reads of slot 0xFFFFFFFF-sized garbage: is a 6.4 GB hipMemset complete?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void fill(uint32_t *p, size_t n, uint32_t v) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v; }
__global__ void count_ne(const uint32_t *p, size_t n, uint32_t v, unsigned long long *out, unsigned long long *first) {
    unsigned long long c = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) if (p[i] != v) { c++; atomicMin(first, (unsigned long long)i); }
    if (c) atomicAdd(out, c);
}
int main() {
    const size_t bytes = 6400000000ull, n = bytes / 4;
    for (int mode = 0; mode < 2; ++mode) {
        uint32_t *p = nullptr;
        if (hipMalloc((void **)&p, bytes) != hipSuccess) { printf("malloc failed\n"); return 1; }
        fill<<<4096, 256>>>(p, n, 0x5A5A5A5Au);
        hipDeviceSynchronize();
        hipFree(p);
        if (hipMalloc((void **)&p, bytes) != hipSuccess) { printf("malloc failed\n"); return 1; }
        hipError_t e = mode == 0 ? hipMemset(p, 0xFF, bytes) : hipMemsetAsync(p, 0xFF, bytes, 0);
        hipDeviceSynchronize();
        unsigned long long *d = nullptr, h[2] = {0, ~0ull};
        hipMalloc((void **)&d, 16); hipMemcpy(d, h, 16, hipMemcpyHostToDevice);
        count_ne<<<4096, 256>>>(p, n, 0xFFFFFFFFu, d, d + 1);
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%s of %zu bytes: rc %d, words not 0xFFFFFFFF: %llu (first at word %llu = byte %llu)\n", mode == 0 ? "hipMemset" : "hipMemsetAsync", bytes, (int)e, h[0], h[0] ? h[1] : 0ull, h[0] ? h[1] * 4 : 0ull);
        hipFree(d); hipFree(p);
    }
    return 0;
}
