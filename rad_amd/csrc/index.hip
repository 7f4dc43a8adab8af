// index.hip — index object (corpus + layered adjacency in HBM), synthetic
// generators, K1 scan and K2 gather Tanimoto kernels.  gfx950 only.
#include "common.h"
#include "rows_tile.h"

#include <algorithm>
#include <new>

// ------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

void radhip_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char *radhip_last_error(void) { return g_err; }
extern "C" const char *radhip_backend_name(void) { return "hip:gfx950"; }
extern "C" int radhip_abi_version(void) { return RADHIP_ABI_VERSION; }

extern "C" int radhip_device_count(int *out_count) {
    if (!out_count) RH_FAIL(RADHIP_E_INVALID, "out_count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) n = 0;
    *out_count = n;
    return RADHIP_OK;
}

extern "C" float radhip_distance_f32(uint32_t a, uint32_t o) {
    if (o == 0) return 0.0f;
    volatile float q = (float)a / (float)o;
    return 1.0f - q;
}

extern "C" uint64_t radhip_rad_key(uint32_t a, uint32_t o, uint32_t slot, uint32_t level) {
    return rh_make_key(rh_q24(a, o), slot, level);
}
extern "C" void radhip_rad_key_decode(uint64_t key, uint32_t *slot, uint32_t *level) {
    rh_decode_key(key, slot, level);
}

// --------------------------------------------------------------- lifecycle
extern "C" int radhip_index_create(uint32_t ndim_bits, uint32_t connectivity,
                                   uint32_t connectivity_base, uint32_t expansion_add,
                                   int device, radhip_index_t **out) {
    if (!out) RH_FAIL(RADHIP_E_INVALID, "out is null");
    if (ndim_bits == 0 || ndim_bits > 2048) RH_FAIL(RADHIP_E_INVALID, "ndim_bits must be in 1..2048 (got %u)", ndim_bits);
    if (connectivity < 2 || connectivity > 64) RH_FAIL(RADHIP_E_INVALID, "connectivity must be in 2..64 (got %u)", connectivity);
    if (connectivity_base == 0) connectivity_base = 2 * connectivity;
    if (connectivity_base > 64) RH_FAIL(RADHIP_E_INVALID, "connectivity_base must be <= 64 (got %u)", connectivity_base);
    radhip_index *idx = new (std::nothrow) radhip_index();
    if (!idx) RH_FAIL(RADHIP_E_NOMEM, "out of host memory");
    idx->ndim_bits = ndim_bits;
    idx->row_bytes = (ndim_bits + 7) / 8;
    uint32_t chunks = (idx->row_bytes + 15) / 16, lpr = 1;
    while (lpr < chunks) lpr <<= 1;
    idx->lpr = lpr;
    idx->row_stride = lpr * 16;
    idx->M = connectivity;
    idx->cap0 = connectivity_base;
    idx->ef_add = expansion_add;
    idx->device = device;
    *out = idx;
    return RADHIP_OK;
}

void rh_layout_free(radhip_index *idx);   // layout.hip
static void rh_peer_release(radhip_index *idx);
#define RH_REQUIRE_OWN_CORPUS(idx) do { if ((idx)->peer) RH_FAIL(RADHIP_E_STATE, "the corpus of this index is peer-mapped (radhip_index_peer_*): it is read-only"); } while (0)

static void free_dev(radhip_index *idx) {
    if (!idx->dev_ready) return;
    (void)hipSetDevice(idx->device);
    rh_layout_free(idx);
    if (idx->peer) rh_peer_release(idx);
    else if (idx->d_fp) (void)hipFree(idx->d_fp);
    if (idx->d_levels) (void)hipFree(idx->d_levels);
    if (idx->d_adj0) (void)hipFree(idx->d_adj0);
    if (idx->d_upper_row) (void)hipFree(idx->d_upper_row);
    if (idx->d_adjU) (void)hipFree(idx->d_adjU);
    if (idx->d_top) (void)hipFree(idx->d_top);
    if (idx->stream) (void)hipStreamDestroy(idx->stream);
    idx->d_fp = nullptr; idx->d_levels = nullptr; idx->d_adj0 = nullptr;
    idx->d_upper_row = nullptr; idx->d_adjU = nullptr; idx->d_top = nullptr;
    idx->stream = nullptr;
}

extern "C" int radhip_index_destroy(radhip_index_t *idx) {
    if (!idx) return RADHIP_OK;
    free_dev(idx);
    delete idx;
    return RADHIP_OK;
}

extern "C" int radhip_index_info(const radhip_index_t *idx, radhip_index_info_t *o) {
    if (!idx || !o) RH_FAIL(RADHIP_E_INVALID, "null argument");
    memset(o, 0, sizeof *o);
    o->n = idx->has_graph ? idx->g_n : idx->n;
    if (idx->has_vectors && idx->n > o->n) o->n = idx->n;
    if (idx->sharded && idx->n_total > o->n) o->n = idx->n_total;
    o->shard_first = idx->sharded ? idx->shard_first : 0;
    o->shard_rows = idx->has_vectors ? idx->n : 0;
    o->sharded = idx->sharded ? 1 : 0;
    o->ndim_bits = idx->ndim_bits; o->row_bytes = idx->row_bytes; o->row_stride = idx->row_stride;
    o->connectivity = idx->M; o->connectivity_base = idx->cap0; o->expansion_add = idx->ef_add;
    o->max_level = idx->max_level; o->entry = idx->entry; o->n_upper_rows = idx->n_upper_rows;
    o->device_bytes = idx->device_bytes; o->device = idx->device;
    o->has_vectors = idx->has_vectors; o->has_graph = idx->has_graph;
    return RADHIP_OK;
}

static int dev_alloc(radhip_index *idx, void **p, size_t bytes) {
    if (bytes == 0) bytes = 16;
    RH_HIP(hipMalloc(p, bytes));
    idx->device_bytes += bytes;
    return RADHIP_OK;
}
static void dev_free(radhip_index *idx, void *p, size_t bytes) {
    if (!p) return;
    (void)hipFree(p);
    idx->device_bytes -= std::min<uint64_t>(idx->device_bytes, bytes ? bytes : 16);
}

int rh_alloc_graph_dev(radhip_index *idx) {
    dev_free(idx, idx->d_levels, idx->g_n); idx->d_levels = nullptr;
    // sizes of the previous graph are not tracked separately; a graph is
    // (re)loaded rarely, so the accounting is reset by the caller when needed
    if (idx->d_adj0) { (void)hipFree(idx->d_adj0); idx->d_adj0 = nullptr; }
    if (idx->d_upper_row) { (void)hipFree(idx->d_upper_row); idx->d_upper_row = nullptr; }
    if (idx->d_adjU) { (void)hipFree(idx->d_adjU); idx->d_adjU = nullptr; }
    if (idx->d_top) { (void)hipFree(idx->d_top); idx->d_top = nullptr; }
    RH_TRY(dev_alloc(idx, (void **)&idx->d_levels, idx->g_n));
    RH_TRY(dev_alloc(idx, (void **)&idx->d_adj0, idx->g_n * idx->cap0 * 4));
    RH_TRY(dev_alloc(idx, (void **)&idx->d_upper_row, idx->g_n * 4));
    RH_TRY(dev_alloc(idx, (void **)&idx->d_adjU, idx->n_upper_rows * idx->M * 4));
    idx->cap_nodes = idx->g_n;
    idx->cap_upper = idx->n_upper_rows;
    return RADHIP_OK;
}

int rh_upload_top(radhip_index *idx) {
    if (idx->d_top) { (void)hipFree(idx->d_top); idx->d_top = nullptr; }
    idx->n_top = (uint32_t)idx->h_top.size();
    RH_TRY(dev_alloc(idx, (void **)&idx->d_top, (size_t)idx->n_top * 4));
    if (idx->n_top)
        RH_HIP(hipMemcpy(idx->d_top, idx->h_top.data(), (size_t)idx->n_top * 4, hipMemcpyHostToDevice));
    return RADHIP_OK;
}

int rh_ensure_device(radhip_index *idx) {
    RH_REQUIRE_SOUND(idx);
    if (!idx->dev_ready) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
            RH_FAIL(RADHIP_E_NO_DEVICE, "no HIP device visible: librad_hip has no CPU fallback");
        if (idx->device < 0 || idx->device >= n)
            RH_FAIL(RADHIP_E_NO_DEVICE, "device %d out of range (%d visible)", idx->device, n);
        RH_HIP(hipSetDevice(idx->device));
        hipDeviceProp_t prop;
        RH_HIP(hipGetDeviceProperties(&prop, idx->device));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            RH_FAIL(RADHIP_E_NO_DEVICE, "device %d is %s; librad_hip is built for gfx950 only", idx->device, prop.gcnArchName);
        RH_HIP(hipStreamCreateWithFlags(&idx->stream, hipStreamNonBlocking));
        idx->dev_ready = true;
    }
    RH_HIP(hipSetDevice(idx->device));
    const bool uploaded = idx->h_rows_pending || (idx->has_graph && !idx->d_graph_valid);
    if (idx->h_rows_pending) {
        if (idx->d_fp) { dev_free(idx, idx->d_fp, idx->fp_cap_rows * idx->row_stride); idx->d_fp = nullptr; }
        RH_TRY(dev_alloc(idx, (void **)&idx->d_fp, idx->n * idx->row_stride));
        idx->fp_cap_rows = idx->n;
        RH_HIP(hipMemcpy(idx->d_fp, idx->h_rows.data(), idx->n * idx->row_stride, hipMemcpyHostToDevice));
        std::vector<uint8_t>().swap(idx->h_rows);
        idx->h_rows_pending = false;
    }
    if (idx->has_graph && !idx->d_graph_valid) {
        // graph came from the host (load_graph): upload it
        RH_TRY(rh_alloc_graph_dev(idx));
        RH_HIP(hipMemcpy(idx->d_levels, idx->h_levels.data(), idx->g_n, hipMemcpyHostToDevice));
        RH_HIP(hipMemcpy(idx->d_adj0, idx->h_adj0.data(), idx->g_n * idx->cap0 * 4, hipMemcpyHostToDevice));
        RH_HIP(hipMemcpy(idx->d_upper_row, idx->h_upper_row.data(), idx->g_n * 4, hipMemcpyHostToDevice));
        if (idx->n_upper_rows)
            RH_HIP(hipMemcpy(idx->d_adjU, idx->h_adjU.data(), idx->n_upper_rows * idx->M * 4, hipMemcpyHostToDevice));
        RH_TRY(rh_upload_top(idx));
        idx->d_graph_valid = true;
    }
    // (the uploads above went through the null stream; the library's stream is non-blocking: order them explicitly)
    if (uploaded) RH_HIP(hipDeviceSynchronize());
    return RADHIP_OK;
}

int rh_ensure_host_graph(radhip_index *idx) {
    RH_REQUIRE_SOUND(idx);
    if (!idx->has_graph) RH_FAIL(RADHIP_E_STATE, "no graph loaded");
    if (idx->h_graph_valid) return RADHIP_OK;
    RH_TRY(rh_ensure_device(idx));
    idx->h_levels.resize(idx->g_n);
    idx->h_adj0.resize(idx->g_n * idx->cap0);
    idx->h_upper_row.resize(idx->g_n);
    idx->h_adjU.resize(idx->n_upper_rows * idx->M);
    RH_HIP(hipMemcpy(idx->h_levels.data(), idx->d_levels, idx->g_n, hipMemcpyDeviceToHost));
    RH_HIP(hipMemcpy(idx->h_adj0.data(), idx->d_adj0, idx->g_n * idx->cap0 * 4, hipMemcpyDeviceToHost));
    RH_HIP(hipMemcpy(idx->h_upper_row.data(), idx->d_upper_row, idx->g_n * 4, hipMemcpyDeviceToHost));
    if (idx->n_upper_rows)
        RH_HIP(hipMemcpy(idx->h_adjU.data(), idx->d_adjU, idx->n_upper_rows * idx->M * 4, hipMemcpyDeviceToHost));
    idx->h_graph_valid = true;
    return RADHIP_OK;
}

// ------------------------------------------------------------------ corpus
static int load_vectors_impl(radhip_index *idx, const uint8_t *rows, uint64_t n, uint64_t first, uint64_t n_total, bool shard) {
    if (!idx || (!rows && n)) RH_FAIL(RADHIP_E_INVALID, "null argument");
    if (shard && (n == 0 || first + n > n_total)) RH_FAIL(RADHIP_E_INVALID, "rows [first, first+count) must be a non-empty range of n_total");
    std::lock_guard<std::mutex> lk(idx->mu);
    RH_REQUIRE_OWN_CORPUS(idx);
    try {
        idx->h_rows.assign((size_t)n * idx->row_stride, 0);
    } catch (...) {
        RH_FAIL(RADHIP_E_NOMEM, "out of host memory staging %llu rows", (unsigned long long)n);
    }
    for (uint64_t i = 0; i < n; ++i)   // (padding bits beyond ndim are the caller's: np.packbits zero-fills them)
        memcpy(idx->h_rows.data() + i * idx->row_stride, rows + i * idx->row_bytes, idx->row_bytes);
    idx->n = n;
    idx->sharded = shard; idx->shard_first = shard ? first : 0; idx->n_total = shard ? n_total : 0;
    idx->h_rows_pending = true;
    idx->has_vectors = true;
    idx->graph_gen++;
    return RADHIP_OK;
}
extern "C" int radhip_index_load_vectors(radhip_index_t *idx, const uint8_t *rows, uint64_t n) {
    return load_vectors_impl(idx, rows, n, 0, n, false);
}
// one rank's rows of a corpus of n_total rows: the index never holds (or stages) the rest
extern "C" int radhip_index_load_vectors_shard(radhip_index_t *idx, const uint8_t *rows, uint64_t first, uint64_t count,
                                               uint64_t n_total) {
    return load_vectors_impl(idx, rows, count, first, n_total, true);
}

// splitmix-style hashing shared (by restatement) with oracle/rad_oracle.c
__host__ __device__ __forceinline__ uint64_t syn_mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
__host__ __device__ __forceinline__ uint64_t syn_h3(uint64_t seed, uint64_t a, uint64_t b) {
    return syn_mix64(syn_mix64(seed ^ (a * 0xD6E8FEB86659FD93ULL)) + b);
}
__host__ __device__ __forceinline__ uint64_t syn_sparse(uint64_t seed, uint64_t a, uint64_t b, int k) {
    uint64_t w = ~0ULL;
    for (int t = 0; t < k; ++t) w &= syn_h3(seed + (uint64_t)t * 0x100000001B3ULL, a, b);
    return w;
}
#define SYN_CS 32ull
#define SYN_SC 64ull
#define SYN_TAG_S 0x5355504552ULL
#define SYN_TAG_CD 0x434C5544ULL
#define SYN_TAG_CA 0x434C5541ULL
#define SYN_TAG_RD 0x524F5744ULL
#define SYN_TAG_RA 0x524F5741ULL
#define SYN_TAG_G0 0x4752415048ULL

__host__ __device__ __forceinline__ uint64_t syn_nc(uint64_t n) {
    uint64_t nc = n / SYN_CS;
    return nc ? nc : 1;
}
#define SYN_TAG_H0 0x48494552ULL
#define SYN_TAG_HD 0x48494544ULL
#define SYN_TAG_HA 0x48494541ULL
#define SYN_HMUL 0x9E3779B1ULL
// mode 2: a 4-ary hierarchy with neighbourhood structure at every scale (restated in
// oracle/rad_oracle.c synth_word_h): row r sits on leaf (r * odd) mod 4^D of a tree of depth
// D = ceil(log4 n_total); every tree node keeps 15/16 of its parent's set bits and gains 1/256 of the
// clear ones, so similarity falls smoothly with the height of the lowest common ancestor.
__host__ __device__ __forceinline__ int syn_depth(uint64_t n_total) {
    int d = 1;
    while (d < 31 && (1ULL << (2 * d)) < n_total) d++;
    return d;
}
__device__ __forceinline__ uint64_t syn_word_h(uint64_t seed, uint64_t row, uint64_t n_total, uint32_t w) {
    const int D = syn_depth(n_total);
    const uint64_t leaf = (row * SYN_HMUL) & ((1ULL << (2 * D)) - 1ULL);
    uint64_t x = syn_sparse(seed ^ SYN_TAG_H0, 0, w, 4);
    for (int d = 1; d <= D; ++d) {
        const uint64_t a = leaf >> (2 * (D - d));
        const uint64_t lv = (uint64_t)d << 40;
        x = (x & ~syn_sparse((seed ^ SYN_TAG_HD) + lv, a, w, 4)) | syn_sparse((seed ^ SYN_TAG_HA) + lv, a, w, 8);
    }
    return x;
}
__device__ __forceinline__ uint64_t syn_word(uint64_t seed, uint64_t row, uint64_t n_total, uint32_t w, int mode) {
    if (mode == 0) return syn_h3(seed, row, w);
    if (mode == 2) return syn_word_h(seed, row, n_total, w);
    uint64_t nc = syn_nc(n_total);
    uint64_t c = row % nc, s = c / SYN_SC;
    uint64_t sb = syn_sparse(seed ^ SYN_TAG_S, s, w, 4);
    uint64_t cb = (sb & ~syn_sparse(seed ^ SYN_TAG_CD, c, w, 2)) | syn_sparse(seed ^ SYN_TAG_CA, c, w, 6);
    return (cb & ~syn_sparse(seed ^ SYN_TAG_RD, row, w, 3)) | syn_sparse(seed ^ SYN_TAG_RA, row, w, 6);
}

// one thread per 64-bit word of the padded row
__global__ void synth_rows_kernel(uint64_t *fp, uint64_t n, uint32_t words_per_stride, uint32_t row_bytes,
                                  uint32_t ndim_bits, uint64_t first_row, uint64_t n_total,
                                  uint64_t seed, int mode) {
    uint64_t total = n * words_per_stride;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t r = i / words_per_stride;
        uint32_t w = (uint32_t)(i % words_per_stride);
        uint64_t x = 0;
        if (w * 8u < row_bytes) {
            x = syn_word(seed, first_row + r, n_total, w, mode);
            uint32_t take = row_bytes - w * 8u;
            if (take < 8u) x &= (1ull << (take * 8u)) - 1ull;
            if ((ndim_bits % 8u) && (w * 8u + 8u >= row_bytes)) {
                // clear the bits above ndim in the last byte
                uint32_t last = row_bytes - 1u - w * 8u;
                uint64_t keep = ((1ull << (ndim_bits % 8u)) - 1ull) << (last * 8u);
                uint64_t below = last ? ((1ull << (last * 8u)) - 1ull) : 0ull;
                x &= (keep | below);
            }
        }
        fp[i] = x;
    }
}

static int synth_vectors_impl(radhip_index *idx, uint64_t n, uint64_t first_row, uint64_t n_total, uint64_t seed, int mode, bool shard) {
    if (!idx) RH_FAIL(RADHIP_E_INVALID, "null index");
    if (mode < 0 || mode > 2) RH_FAIL(RADHIP_E_INVALID, "mode must be 0, 1 or 2");
    if (first_row + n > n_total) RH_FAIL(RADHIP_E_INVALID, "rows [first, first+n) exceed n_total");
    if (shard && n == 0) RH_FAIL(RADHIP_E_INVALID, "a shard holds at least one row");
    std::lock_guard<std::mutex> lk(idx->mu);
    RH_REQUIRE_OWN_CORPUS(idx);
    idx->h_rows_pending = false;
    std::vector<uint8_t>().swap(idx->h_rows);
    RH_TRY(rh_ensure_device(idx));
    if (idx->d_fp) { dev_free(idx, idx->d_fp, idx->fp_cap_rows * idx->row_stride); idx->d_fp = nullptr; }
    RH_TRY(dev_alloc(idx, (void **)&idx->d_fp, n * idx->row_stride));
    idx->fp_cap_rows = n;
    idx->n = n;
    idx->sharded = shard; idx->shard_first = shard ? first_row : 0; idx->n_total = shard ? n_total : 0;
    uint32_t wps = idx->row_stride / 8;
    hipLaunchKernelGGL(synth_rows_kernel, dim3(256 * 16), dim3(256), 0, idx->stream, (uint64_t *)idx->d_fp, n,
                       wps, idx->row_bytes, idx->ndim_bits, first_row, n_total, seed, mode);
    RH_HIP(hipGetLastError());
    RH_HIP(hipStreamSynchronize(idx->stream));
    idx->has_vectors = true;
    idx->graph_gen++;
    return RADHIP_OK;
}
extern "C" int radhip_index_synth_vectors(radhip_index_t *idx, uint64_t n, uint64_t first_row,
                                          uint64_t n_total, uint64_t seed, int mode) {
    return synth_vectors_impl(idx, n, first_row, n_total, seed, mode, false);
}
// rows [first_row, first_row + count) of the closed-form corpus of n_total rows AS A SHARD: they keep their global
// slots (row i of the buffer is slot first_row + i), the rest of the corpus never exists on this device
extern "C" int radhip_index_synth_vectors_shard(radhip_index_t *idx, uint64_t count, uint64_t first_row,
                                                uint64_t n_total, uint64_t seed, int mode) {
    return synth_vectors_impl(idx, count, first_row, n_total, seed, mode, true);
}

// Row-sharded multi-GPU mode: this rank keeps rows [first, first + count) of the corpus it generated /
// loaded in full (the graph is built over all rows first; adjacency, levels and keys stay whole).
extern "C" int radhip_index_keep_rows(radhip_index_t *idx, uint64_t first, uint64_t count) {
    if (!idx) RH_FAIL(RADHIP_E_INVALID, "null index");
    std::lock_guard<std::mutex> lk(idx->mu);
    if (!idx->has_vectors) RH_FAIL(RADHIP_E_STATE, "no vectors loaded");
    RH_REQUIRE_OWN_CORPUS(idx);
    RH_REQUIRE_FULL_CORPUS(idx);
    if (count == 0 || first + count > idx->n) RH_FAIL(RADHIP_E_RANGE, "rows [%llu, %llu) out of range (%llu rows)",
                                                      (unsigned long long)first, (unsigned long long)(first + count),
                                                      (unsigned long long)idx->n);
    RH_TRY(rh_ensure_device(idx));
    const size_t bytes = count * idx->row_stride;
    const uint8_t *src = (const uint8_t *)idx->d_fp + first * idx->row_stride;
    uint4 *nfp = nullptr;
    if (hipMalloc((void **)&nfp, bytes) == hipSuccess) {
        if (hipMemcpyAsync(nfp, src, bytes, hipMemcpyDeviceToDevice, idx->stream) != hipSuccess || hipStreamSynchronize(idx->stream) != hipSuccess) {
            (void)hipFree(nfp); RH_FAIL(RADHIP_E_HIP, "device copy of the shard failed");
        }
        dev_free(idx, idx->d_fp, idx->fp_cap_rows * idx->row_stride);
    } else {
        // no room for the shard beside the whole corpus: the shard leaves through host memory and the corpus is
        // freed BEFORE the shard is allocated, so the peak stays at the corpus
        (void)hipGetLastError();
        std::vector<uint8_t> stage;
        try { stage.resize(bytes); } catch (...) { RH_FAIL(RADHIP_E_NOMEM, "out of host memory staging the shard (%zu bytes)", bytes); }
        RH_HIP(hipMemcpy(stage.data(), src, bytes, hipMemcpyDeviceToHost));
        dev_free(idx, idx->d_fp, idx->fp_cap_rows * idx->row_stride);
        idx->d_fp = nullptr; idx->fp_cap_rows = 0; idx->has_vectors = false;
        RH_HIP(hipMalloc((void **)&nfp, bytes));
        if (hipMemcpy(nfp, stage.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(nfp); RH_FAIL(RADHIP_E_HIP, "upload of the shard failed"); }
        idx->has_vectors = true;
    }
    idx->n_total = idx->n;
    idx->d_fp = nfp;
    idx->device_bytes += bytes;
    idx->fp_cap_rows = count;
    idx->n = count;
    idx->shard_first = first;
    idx->sharded = true;
    idx->graph_gen++;
    return RADHIP_OK;
}

extern "C" int radhip_index_read_vectors(const radhip_index_t *cidx, uint64_t first, uint64_t count,
                                         uint8_t *out_rows) {
    radhip_index *idx = const_cast<radhip_index *>(cidx);
    if (!idx || !out_rows) RH_FAIL(RADHIP_E_INVALID, "null argument");
    if (!idx->has_vectors) RH_FAIL(RADHIP_E_STATE, "no vectors loaded");
    RH_REQUIRE_FULL_CORPUS(idx);
    if (first + count > idx->n) RH_FAIL(RADHIP_E_RANGE, "rows out of range");
    std::lock_guard<std::mutex> lk(idx->mu);
    RH_TRY(rh_ensure_device(idx));
    if (idx->row_bytes == idx->row_stride) {
        RH_HIP(hipMemcpy(out_rows, (const uint8_t *)idx->d_fp + first * idx->row_stride,
                         count * idx->row_stride, hipMemcpyDeviceToHost));
    } else {
        RH_HIP(hipMemcpy2D(out_rows, idx->row_bytes, (const uint8_t *)idx->d_fp + first * idx->row_stride,
                           idx->row_stride, idx->row_bytes, count, hipMemcpyDeviceToHost));
    }
    return RADHIP_OK;
}

// ------------------------------------------------------------------- graph
static void compute_top(radhip_index *idx) {
    idx->h_top.clear();
    for (uint64_t i = 0; i < idx->g_n; ++i)
        if (idx->h_levels[i] == idx->max_level) idx->h_top.push_back((uint32_t)i);
}

extern "C" int radhip_index_load_graph(radhip_index_t *idx, uint64_t n, int32_t max_level,
                                       uint32_t entry, const int8_t *levels, const uint32_t *adj0,
                                       const uint32_t *upper_row, const uint32_t *adjU,
                                       uint64_t n_upper_rows) {
    if (!idx || !levels || !adj0 || !upper_row || (n_upper_rows && !adjU))
        RH_FAIL(RADHIP_E_INVALID, "null argument");
    if (n == 0 || n >= 0xFFFFFFFFull) RH_FAIL(RADHIP_E_INVALID, "n out of range");
    if (max_level < 0 || max_level > 15) RH_FAIL(RADHIP_E_INVALID, "max_level must be in 0..15");
    if (entry >= n) RH_FAIL(RADHIP_E_INVALID, "entry slot out of range");
    // validate: levels, row ranges, no self / duplicate / out-of-range targets,
    // every target exists on the level of the row it appears in
    for (uint64_t i = 0; i < n; ++i) {
        int lv = levels[i];
        if (lv < 0 || lv > max_level) RH_FAIL(RADHIP_E_INVALID, "levels[%llu]=%d out of range", (unsigned long long)i, lv);
        if (lv > 0 && ((uint64_t)upper_row[i] + (uint64_t)lv > n_upper_rows))
            RH_FAIL(RADHIP_E_INVALID, "upper_row[%llu] out of range", (unsigned long long)i);
        for (int l = 0; l <= lv; ++l) {
            const uint32_t cap = l == 0 ? idx->cap0 : idx->M;
            const uint32_t *row = l == 0 ? adj0 + i * idx->cap0 : adjU + ((uint64_t)upper_row[i] + (l - 1)) * idx->M;
            bool ended = false;
            for (uint32_t j = 0; j < cap; ++j) {
                uint32_t t = row[j];
                if (t == RADHIP_NO_SLOT) { ended = true; continue; }
                if (ended) RH_FAIL(RADHIP_E_INVALID, "node %llu level %d: slot after padding", (unsigned long long)i, l);
                if (t >= n || t == i) RH_FAIL(RADHIP_E_INVALID, "node %llu level %d: bad target %u", (unsigned long long)i, l, t);
                if (levels[t] < l) RH_FAIL(RADHIP_E_INVALID, "node %llu level %d: target %u absent on that level", (unsigned long long)i, l, t);
                for (uint32_t k = 0; k < j; ++k)
                    if (row[k] == t) RH_FAIL(RADHIP_E_INVALID, "node %llu level %d: duplicate target %u", (unsigned long long)i, l, t);
            }
        }
    }
    std::lock_guard<std::mutex> lk(idx->mu);
    rh_layout_invalidate(idx);
    try {
        idx->h_levels.assign(levels, levels + n);
        idx->h_adj0.assign(adj0, adj0 + n * idx->cap0);
        idx->h_upper_row.assign(upper_row, upper_row + n);
        idx->h_adjU.assign(adjU, adjU + n_upper_rows * idx->M);
    } catch (...) {
        RH_FAIL(RADHIP_E_NOMEM, "out of host memory");
    }
    idx->g_n = n; idx->max_level = max_level; idx->entry = entry; idx->n_upper_rows = n_upper_rows;
    compute_top(idx);
    idx->h_graph_valid = true;
    idx->d_graph_valid = false;
    idx->has_graph = true;
    return RADHIP_OK;
}

__host__ __device__ __forceinline__ uint64_t syn_ipow(uint64_t b, int e) {
    uint64_t r = 1;
    while (e-- > 0) r *= b;
    return r;
}
static int32_t syn_max_level(uint64_t n, uint32_t M) {
    int32_t l = 0;
    uint64_t p = 1;
    while (l < 15 && p * M < n) { p *= M; l++; }
    return l;
}

// one thread per node: level, level-0 row, upper rows (closed form)
__global__ void synth_graph_kernel(uint64_t n, uint32_t M, uint32_t cap0, int32_t L, uint64_t seed,
                                   int8_t *levels, uint32_t *adj0, uint32_t *upper_row, uint32_t *adjU) {
    const uint64_t nc = syn_nc(n);
    const uint32_t H = cap0 / 2, LK = cap0 - H, LN = LK / 2, LF = LK - LN;
    const uint64_t gs = seed ^ SYN_TAG_G0;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n;
         r += (uint64_t)gridDim.x * blockDim.x) {
        int lv;
        if (r == 0) lv = L;
        else { lv = 0; uint64_t x = r; while (lv < L && x % M == 0) { x /= M; lv++; } }
        levels[r] = (int8_t)lv;
        uint32_t *row = adj0 + r * cap0;
        uint32_t k = 0;
        const uint64_t c = r % nc, m = r / nc;
        const uint64_t cs = (n - c + nc - 1) / nc;
        for (uint32_t j = 0; j < H && j + 1 < cs; ++j) row[k++] = (uint32_t)(c + ((m + 1 + j) % cs) * nc);
        const uint64_t s0 = (c / SYN_SC) * SYN_SC;
        const uint64_t ss = (s0 + SYN_SC <= nc) ? SYN_SC : nc - s0;
        for (uint32_t jj = 0; jj < LN; ++jj) {
            uint64_t off;
            const uint64_t hh = syn_h3(gs, r, jj);
            if (ss - 1 >= LN) { uint64_t bw = (ss - 1) / LN; off = 1 + jj * bw + hh % bw; }
            else if (jj + 1 < ss) off = 1 + jj;
            else continue;
            const uint64_t c2 = s0 + ((c - s0) + off) % ss;
            const uint64_t cs2 = (n - c2 + nc - 1) / nc;
            row[k++] = (uint32_t)(c2 + ((hh >> 32) % cs2) * nc);
        }
        if (nc >= 4 * SYN_SC) {
            const uint64_t span = nc - 2 * SYN_SC + 1;
            for (uint32_t jj = 0; jj < LF; ++jj) {
                const uint64_t hh = syn_h3(gs, r, 1000 + jj);
                const uint64_t bw = span / LF;
                const uint64_t off = SYN_SC + jj * bw + hh % bw;
                const uint64_t c2 = (c + off) % nc;
                const uint64_t cs2 = (n - c2 + nc - 1) / nc;
                row[k++] = (uint32_t)(c2 + ((hh >> 32) % cs2) * nc);
            }
        }
        while (k < cap0) row[k++] = RADHIP_NO_SLOT;
        if (lv == 0) { upper_row[r] = RADHIP_NO_SLOT; continue; }
        uint64_t base = 0;
        for (int l = 1; l <= L; ++l) { uint64_t p = syn_ipow(M, l); base += (r + p - 1) / p; }
        upper_row[r] = (uint32_t)base;
        for (int l = 1; l <= lv; ++l) {
            uint32_t *ur = adjU + (base + (uint64_t)(l - 1)) * M;
            const uint64_t p = syn_ipow(M, l), kk = r / p, nl = (n + p - 1) / p;
            uint32_t u = 0;
            for (uint32_t j = 0; j < M; ++j) {
                uint64_t off;
                if (nl - 1 >= M) { uint64_t bw = (nl - 1) / M; off = 1 + j * bw + syn_h3(gs, r, 2000 + 64 * (uint64_t)l + j) % bw; }
                else if (j + 1 < nl) off = 1 + j;
                else break;
                ur[u++] = (uint32_t)(((kk + off) % nl) * p);
            }
            while (u < M) ur[u++] = RADHIP_NO_SLOT;
        }
    }
}

extern "C" int radhip_index_synth_graph(radhip_index_t *idx, uint64_t seed) {
    if (!idx) RH_FAIL(RADHIP_E_INVALID, "null index");
    if (!idx->has_vectors || idx->n == 0) RH_FAIL(RADHIP_E_STATE, "load or generate vectors first");
    std::lock_guard<std::mutex> lk(idx->mu);
    // a shard generates the graph over the WHOLE corpus (closed form over the slots: no row is read)
    const uint64_t n = idx->sharded ? idx->n_total : idx->n;
    if (n >= 0xFFFFFFFFull) RH_FAIL(RADHIP_E_INVALID, "n too large");
    RH_TRY(rh_ensure_device(idx));
    rh_layout_invalidate(idx);
    const int32_t L = syn_max_level(n, idx->M);
    uint64_t nu = 0;
    for (int l = 1; l <= L; ++l) { uint64_t p = syn_ipow(idx->M, l); nu += (n + p - 1) / p; }
    idx->g_n = n; idx->max_level = L; idx->entry = 0; idx->n_upper_rows = nu;
    RH_TRY(rh_alloc_graph_dev(idx));
    hipLaunchKernelGGL(synth_graph_kernel, dim3(256 * 8), dim3(256), 0, idx->stream, n, idx->M, idx->cap0, L,
                       seed, idx->d_levels, idx->d_adj0, idx->d_upper_row, idx->d_adjU);
    RH_HIP(hipGetLastError());
    RH_HIP(hipStreamSynchronize(idx->stream));
    // top-level nodes in closed form: multiples of M^L
    idx->h_top.clear();
    const uint64_t p = syn_ipow(idx->M, L);
    for (uint64_t r = 0; r < n; r += p) idx->h_top.push_back((uint32_t)r);
    RH_TRY(rh_upload_top(idx));
    idx->h_graph_valid = false;
    std::vector<int8_t>().swap(idx->h_levels);
    std::vector<uint32_t>().swap(idx->h_adj0);
    std::vector<uint32_t>().swap(idx->h_upper_row);
    std::vector<uint32_t>().swap(idx->h_adjU);
    idx->d_graph_valid = true;
    idx->has_graph = true;
    idx->graph_gen++;
    return RADHIP_OK;
}

extern "C" int radhip_index_read_graph(const radhip_index_t *cidx, int8_t *levels, uint32_t *adj0,
                                       uint32_t *upper_row, uint32_t *adjU) {
    radhip_index *idx = const_cast<radhip_index *>(cidx);
    if (!idx) RH_FAIL(RADHIP_E_INVALID, "null index");
    std::lock_guard<std::mutex> lk(idx->mu);
    RH_TRY(rh_ensure_host_graph(idx));
    if (levels) memcpy(levels, idx->h_levels.data(), idx->g_n);
    if (adj0) memcpy(adj0, idx->h_adj0.data(), idx->g_n * idx->cap0 * 4);
    if (upper_row) memcpy(upper_row, idx->h_upper_row.data(), idx->g_n * 4);
    if (adjU && idx->n_upper_rows) memcpy(adjU, idx->h_adjU.data(), idx->n_upper_rows * idx->M * 4);
    return RADHIP_OK;
}

extern "C" int radhip_get_neighbors(const radhip_index_t *cidx, uint32_t slot, int32_t level,
                                    uint32_t *out_slots, uint32_t cap, uint32_t *out_n) {
    radhip_index *idx = const_cast<radhip_index *>(cidx);
    if (!idx || !out_n) RH_FAIL(RADHIP_E_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    RH_TRY(rh_ensure_host_graph(idx));
    if (slot >= idx->g_n) RH_FAIL(RADHIP_E_RANGE, "node %u out of range (size %llu)", slot, (unsigned long long)idx->g_n);
    if (level < 0 || level > idx->h_levels[slot])
        RH_FAIL(RADHIP_E_RANGE, "node %u does not exist on level %d", slot, level);
    const uint32_t w = level == 0 ? idx->cap0 : idx->M;
    const uint32_t *row = level == 0 ? idx->h_adj0.data() + (uint64_t)slot * idx->cap0
                                     : idx->h_adjU.data() + ((uint64_t)idx->h_upper_row[slot] + (level - 1)) * idx->M;
    uint32_t k = 0;
    for (uint32_t j = 0; j < w && row[j] != RADHIP_NO_SLOT; ++j) {
        if (out_slots && k < cap) out_slots[k] = row[j];
        ++k;
    }
    *out_n = k;
    return RADHIP_OK;
}

extern "C" int radhip_get_top_level_nodes(const radhip_index_t *cidx, uint32_t *out_slots, uint64_t cap,
                                          uint64_t *out_n) {
    radhip_index *idx = const_cast<radhip_index *>(cidx);
    if (!idx || !out_n) RH_FAIL(RADHIP_E_INVALID, "null argument");
    if (!idx->has_graph) RH_FAIL(RADHIP_E_STATE, "no graph loaded");
    std::lock_guard<std::mutex> lk(idx->mu);
    for (uint64_t i = 0; i < idx->h_top.size() && i < cap; ++i)
        if (out_slots) out_slots[i] = idx->h_top[i];
    *out_n = idx->h_top.size();
    return RADHIP_OK;
}

static thread_local double g_last_kernel_ms = 0.0;
extern "C" double radhip_last_kernel_ms(void) { return g_last_kernel_ms; }

// ------------------------------------------------------- K1: corpus scan --
// One wave-load covers 64/LPR rows (1 KiB contiguous for 1024-bit rows): lane
// l reads the 16-B chunk (l % LPR) of row (l / LPR).  NQ query chunks stay in
// registers; and-counts are summed over the LPR lanes of a row with xor
// shuffles; or = popc(query) + popc(row) - and.
template <int LPR, int NQ>
__global__ __launch_bounds__(256) void scan_kernel(const uint4 *__restrict__ fp, uint64_t first,
                                                   uint64_t count, const uint4 *__restrict__ queries,
                                                   const uint32_t *__restrict__ qpop,
                                                   uint32_t *__restrict__ and_out,
                                                   uint32_t *__restrict__ or_out) {
    // A wave owns tiles of 64 consecutive rows: LPR wave-loads of 64/LPR rows each, all in flight
    // before the first popcount.  After the DPP sums every lane of a row group holds the row's
    // counts; lane (group, chunk) keeps the result of load #chunk, so the 64 lanes end up with the
    // 64 rows of the tile and write them with ONE coalesced 256-B store per output array.
    constexpr int RPL = 64 / LPR;            // rows per wave-load
    constexpr int BATCH = LPR < 8 ? LPR : 8; // loads in flight per lane
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t chunk = lane % LPR, grp = lane / LPR;
    uint4 q[NQ];
    uint32_t qp[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        q[i] = queries[i * LPR + chunk];
        qp[i] = qpop[i];
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
                if (r < count) {
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(&fp[(first + r) * LPR + chunk]));
                    v[u] = make_uint4(t.x, t.y, t.z, t.w);
                }
            }
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
        const uint64_t r = r0 + (uint64_t)chunk * RPL + grp;   // the row this lane kept
        if (r < count) {
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                and_out[(uint64_t)i * count + r] = keep_a[i];
                or_out[(uint64_t)i * count + r] = qp[i] + keep_rp - keep_a[i];
            }
        }
    }
}

// K1 on 1024- and 2048-bit rows with ONE ROW PER LANE (rows_tile.h rh_rows_*): used from 5 queries per pass on, where the kernel above is
// bound by its instruction stream (1090 VALU instructions per 64-row tile at 8 queries; here ~650) instead of by the rows it
// reads and the counts it writes.  Lane l of a tile ends up with row l: the stores are the same coalesced 256 B per array.
#define RH_SCAN_ROWS_WAVES 4
template <int LPR, int NQ>
__global__ __launch_bounds__(64 * RH_SCAN_ROWS_WAVES) void scan_rows_kernel(const uint4 *__restrict__ fp, uint64_t first, uint64_t count,
                                                                            const uint32_t *__restrict__ qd /* [NQ][4 * LPR] */,
                                                                            const uint32_t *__restrict__ qpop,
                                                                            uint32_t *__restrict__ and_out, uint32_t *__restrict__ or_out) {
    __shared__ uint4 s_tr[RH_SCAN_ROWS_WAVES][RH_ROWS_TR_VEC(LPR)];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t qp[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) qp[i] = qpop[i];
    const uint64_t n_tiles = (count + 63) / 64;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    uint4 nv[LPR];
    uint64_t tile = wave;
    if (tile < n_tiles) rh_rows_load<LPR>(fp, first, count, tile, lane, nv);
    for (; tile < n_tiles; tile += n_waves) {
        uint4 v[LPR];
        rh_rows_turn<LPR>(nv, s_tr[wv], lane, v);
        if (tile + n_waves < n_tiles) rh_rows_load<LPR>(fp, first, count, tile + n_waves, lane, nv);
        uint32_t rp, a[NQ];
        rh_rows_count<LPR, NQ>(v, qd, rp, a);
        const uint64_t r = tile * 64 + lane;
        if (r < count) {
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                and_out[(uint64_t)i * count + r] = a[i];
                or_out[(uint64_t)i * count + r] = qp[i] + rp - a[i];
            }
        }
    }
}

template <int LPR>
static int launch_scan(radhip_index *idx, int nq, uint64_t first, uint64_t count, const uint4 *dq,
                       const uint32_t *dqpop, uint32_t *da, uint32_t *dorr) {
    // the wavefronts stride over the tiles: the grid is what the device holds resident at once for this instantiation (2 blocks
    // per CU at 8 queries per pass of 1024-bit rows, 227 VGPRs; 3 at 4 queries: a fixed 8 per CU would leave a ragged last round)
    const uint64_t groups = (count + 255) / 256;   // 4 waves x 64-row tiles per block pass
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, idx->device);
    uint32_t grid = 1;
    // (RADHIP_SCAN_ROWS=0 keeps the row-across-eight-lanes kernel at every query count: the A/B of profiles/r04)
    static const bool rows_ok = []() { const char *e = getenv("RADHIP_SCAN_ROWS"); return !(e && e[0] == '0'); }();
    if ((LPR == 8 || LPR == 16) && nq >= 5 && rows_ok) {
        constexpr int RL = (LPR == 8 || LPR == 16) ? LPR : 8;   // (the instantiation the other widths never launch)
        const uint64_t groups_r = (count + 64ull * RH_SCAN_ROWS_WAVES - 1) / (64ull * RH_SCAN_ROWS_WAVES);
#define RH_SCAN_ROWS_CASE(NQV)                                                                  \
    case NQV: {                                                                                 \
        int nb = 0;                                                                             \
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, scan_rows_kernel<RL, NQV>, 64 * RH_SCAN_ROWS_WAVES, 0) != hipSuccess || nb < 1) { (void)hipGetLastError(); nb = 1; } \
        grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(groups_r, (uint64_t)n_cu * (uint64_t)nb)); \
        hipLaunchKernelGGL((scan_rows_kernel<RL, NQV>), dim3(grid), dim3(64 * RH_SCAN_ROWS_WAVES), 0, idx->stream, idx->d_fp, \
                           first, count, reinterpret_cast<const uint32_t *>(dq), dqpop, da, dorr); \
        break; }
        switch (nq) {
            RH_SCAN_ROWS_CASE(5) RH_SCAN_ROWS_CASE(6) RH_SCAN_ROWS_CASE(7) RH_SCAN_ROWS_CASE(8)
            default: RH_FAIL(RADHIP_E_INVALID, "internal: nq per pass must be 1..8");
        }
#undef RH_SCAN_ROWS_CASE
        RH_HIP(hipGetLastError());
        return RADHIP_OK;
    }
#define RH_SCAN_CASE(NQV)                                                                       \
    case NQV: {                                                                                 \
        int nb = 0;                                                                             \
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, scan_kernel<LPR, NQV>, 256, 0) != hipSuccess || nb < 1) { (void)hipGetLastError(); nb = 1; } \
        grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(groups, (uint64_t)n_cu * (uint64_t)nb)); \
        hipLaunchKernelGGL((scan_kernel<LPR, NQV>), dim3(grid), dim3(256), 0, idx->stream, idx->d_fp, \
                           first, count, dq, dqpop, da, dorr);                                  \
        break; }
    switch (nq) {
        RH_SCAN_CASE(1) RH_SCAN_CASE(2) RH_SCAN_CASE(3) RH_SCAN_CASE(4)
        RH_SCAN_CASE(5) RH_SCAN_CASE(6) RH_SCAN_CASE(7) RH_SCAN_CASE(8)
        default: RH_FAIL(RADHIP_E_INVALID, "internal: nq per pass must be 1..8");
    }
#undef RH_SCAN_CASE
    RH_HIP(hipGetLastError());
    return RADHIP_OK;
}

// pad host query rows to the device stride; also popcounts
static void stage_queries(const radhip_index *idx, const uint8_t *queries, uint32_t nq,
                          std::vector<uint8_t> &padded, std::vector<uint32_t> &pop) {
    padded.assign((size_t)nq * idx->row_stride, 0);
    pop.assign(nq, 0);
    for (uint32_t i = 0; i < nq; ++i) {
        memcpy(padded.data() + (size_t)i * idx->row_stride, queries + (size_t)i * idx->row_bytes, idx->row_bytes);
        uint32_t p = 0;
        for (uint32_t b = 0; b < idx->row_bytes; ++b) p += (uint32_t)__builtin_popcount(queries[(size_t)i * idx->row_bytes + b]);
        pop[i] = p;
    }
}

void rh_stage_queries(const radhip_index *idx, const uint8_t *queries, uint32_t nq,
                      std::vector<uint8_t> &padded, std::vector<uint32_t> &pop) {   // for topk.hip
    stage_queries(idx, queries, nq, padded, pop);
}

extern "C" int radhip_tanimoto_scan(radhip_index_t *idx, const uint8_t *queries, uint32_t nq,
                                    uint64_t first, uint64_t count, uint32_t *and_out, uint32_t *or_out) {
    if (!idx || !queries || !and_out || !or_out) RH_FAIL(RADHIP_E_INVALID, "null argument");
    if (!idx->has_vectors) RH_FAIL(RADHIP_E_STATE, "no vectors loaded");
    RH_REQUIRE_FULL_CORPUS(idx);
    if (first + count > idx->n) RH_FAIL(RADHIP_E_RANGE, "rows [first, first+count) out of range");
    if (nq == 0 || count == 0) return RADHIP_OK;
    std::lock_guard<std::mutex> lk(idx->mu);
    RH_TRY(rh_ensure_device(idx));
    std::vector<uint8_t> padded;
    std::vector<uint32_t> pop;
    stage_queries(idx, queries, nq, padded, pop);
    uint4 *dq = nullptr;
    uint32_t *dpop = nullptr, *da = nullptr, *dorr = nullptr;
    const uint32_t pass = std::min<uint32_t>(nq, 8);
    int rc = RADHIP_OK;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&]() { if (dq) (void)hipFree(dq); if (dpop) (void)hipFree(dpop); if (da) (void)hipFree(da); if (dorr) (void)hipFree(dorr);
                           if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); };
#define RH_G(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { radhip_set_error("%s failed: %s", #x, hipGetErrorString(e_)); cleanup(); return RADHIP_E_HIP; } } while (0)
    RH_G(hipMalloc((void **)&dq, padded.size()));
    RH_G(hipMalloc((void **)&dpop, (size_t)nq * 4));
    RH_G(hipMalloc((void **)&da, (size_t)pass * count * 4));
    RH_G(hipMalloc((void **)&dorr, (size_t)pass * count * 4));
    RH_G(hipMemcpyAsync(dq, padded.data(), padded.size(), hipMemcpyHostToDevice, idx->stream));
    RH_G(hipMemcpyAsync(dpop, pop.data(), (size_t)nq * 4, hipMemcpyHostToDevice, idx->stream));
    RH_G(hipEventCreate(&e0));
    RH_G(hipEventCreate(&e1));
    g_last_kernel_ms = 0.0;
    for (uint32_t q0 = 0; q0 < nq && rc == RADHIP_OK; q0 += pass) {
        const int k = (int)std::min<uint32_t>(pass, nq - q0);
        (void)hipEventRecord(e0, idx->stream);
        const uint4 *dqk = dq + (size_t)q0 * idx->lpr;
        switch (idx->lpr) {
            case 1: rc = launch_scan<1>(idx, k, first, count, dqk, dpop + q0, da, dorr); break;
            case 2: rc = launch_scan<2>(idx, k, first, count, dqk, dpop + q0, da, dorr); break;
            case 4: rc = launch_scan<4>(idx, k, first, count, dqk, dpop + q0, da, dorr); break;
            case 8: rc = launch_scan<8>(idx, k, first, count, dqk, dpop + q0, da, dorr); break;
            default: rc = launch_scan<16>(idx, k, first, count, dqk, dpop + q0, da, dorr); break;
        }
        if (rc != RADHIP_OK) break;
        (void)hipEventRecord(e1, idx->stream);
        RH_G(hipMemcpyAsync(and_out + (size_t)q0 * count, da, (size_t)k * count * 4, hipMemcpyDeviceToHost, idx->stream));
        RH_G(hipMemcpyAsync(or_out + (size_t)q0 * count, dorr, (size_t)k * count * 4, hipMemcpyDeviceToHost, idx->stream));
        RH_G(hipStreamSynchronize(idx->stream));
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) g_last_kernel_ms += ms;
    }
    cleanup();
    return rc;
}

// ---------------------------------------------------- K2: gather-Tanimoto --
// LPR lanes per (query, candidate) pair: random B-byte row gathers, 16 B per
// lane, so a 1024-bit row is one 128-B line fetched by 8 adjacent lanes.
// Per wave and round: W = (64 / LPR) * U consecutive pairs.  Pair ids come in and results go out
// as whole coalesced lines through LDS — 32-B partial-line stores (what "lane 0 of every
// group stores its result" makes) cost the HBM ~2 requests each and capped the first version at
// 26 G pairs/s — and the ids of the next round are fetched while this round's rows are in
// flight, so a round is one dependent HBM round trip with U row gathers in flight per lane
// (scripts/hbm_random.hip: 8 in flight at 16 waves/CU reach the 5.6 TB/s the device serves).
#ifndef RH_GATHER_BLOCKS_PER_CU
#define RH_GATHER_BLOCKS_PER_CU 6
#endif
template <int LPR>
struct GatherShape {
    static constexpr int U = LPR >= 8 ? 8 : 4;
    static constexpr int GPW = 64 / LPR;
    static constexpr int W = GPW * U;
};
template <int LPR>
__global__ __launch_bounds__(256) void gather_kernel(const uint4 *__restrict__ fp,
                                                     const uint4 *__restrict__ queries,
                                                     const uint32_t *__restrict__ qpop,
                                                     const uint32_t *__restrict__ pair_q,
                                                     const uint32_t *__restrict__ pair_slot,
                                                     uint64_t n_pairs, uint32_t *__restrict__ and_out,
                                                     uint32_t *__restrict__ or_out) {
    constexpr int U = GatherShape<LPR>::U, W = GatherShape<LPR>::W;
    __shared__ uint32_t s_slot[4][2][W], s_q[4][2][W], s_a[4][W], s_o[4][W];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int chunk = lane % LPR, grp = lane / LPR;
    const uint64_t n_rounds = (n_pairs + W - 1) / W;
    const uint64_t n_waves = (uint64_t)gridDim.x * 4;
    auto stage_ids = [&](uint64_t r, int buf) {
        for (int i = lane; i < W; i += 64) {
            const uint64_t p = r * W + i;
            const bool in = p < n_pairs;
            s_slot[wv][buf][i] = in ? pair_slot[p] : RADHIP_NO_SLOT;
            s_q[wv][buf][i] = in ? pair_q[p] : 0u;
        }
    };
    uint64_t r = (uint64_t)blockIdx.x * 4 + wv;
    int buf = 0;
    if (r < n_rounds) stage_ids(r, 0);
    while (r < n_rounds) {
        RH_WAVE_SYNC();
        uint32_t sl[U], qi[U];
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { sl[u] = s_slot[wv][buf][grp * U + u]; qi[u] = s_q[wv][buf][grp * U + u]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v[u] = make_uint4(0, 0, 0, 0);
            if (sl[u] != RADHIP_NO_SLOT) {
                v[u] = fp[(uint64_t)sl[u] * LPR + chunk];   // plain: a nontemporal load measured 10 % slower here
            }
        }
        const uint64_t rn = r + n_waves;
        if (rn < n_rounds) stage_ids(rn, buf ^ 1);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint4 q = queries[(uint64_t)qi[u] * LPR + chunk];   // L2-resident
            const uint32_t rp = rh_group_sum<LPR>(rh_popc4(v[u]));
            const uint32_t a = rh_group_sum<LPR>(rh_popc4_and(v[u], q));
            if (chunk == 0) { s_a[wv][grp * U + u] = a; s_o[wv][grp * U + u] = qpop[qi[u]] + rp - a; }
        }
        RH_WAVE_SYNC();
        for (int i = lane; i < W; i += 64) {
            const uint64_t p = r * W + i;
            if (p < n_pairs) { and_out[p] = s_a[wv][i]; or_out[p] = s_o[wv][i]; }
        }
        r = rn;
        buf ^= 1;
    }
}

extern "C" int radhip_tanimoto_gather(radhip_index_t *idx, const uint8_t *queries, uint32_t nq,
                                      const uint32_t *cand_slots, const uint64_t *cand_offsets,
                                      uint32_t *and_out, uint32_t *or_out) {
    if (!idx || !queries || !cand_offsets) RH_FAIL(RADHIP_E_INVALID, "null argument");
    if (!idx->has_vectors) RH_FAIL(RADHIP_E_STATE, "no vectors loaded");
    RH_REQUIRE_FULL_CORPUS(idx);
    const uint64_t n_pairs = nq ? cand_offsets[nq] : 0;
    if (n_pairs == 0) return RADHIP_OK;
    if (!cand_slots || !and_out || !or_out) RH_FAIL(RADHIP_E_INVALID, "null argument");
    std::vector<uint32_t> pq(n_pairs);
    for (uint32_t q = 0; q < nq; ++q) {
        if (cand_offsets[q + 1] < cand_offsets[q]) RH_FAIL(RADHIP_E_INVALID, "cand_offsets must be non-decreasing");
        for (uint64_t i = cand_offsets[q]; i < cand_offsets[q + 1]; ++i) {
            if (cand_slots[i] >= idx->n) RH_FAIL(RADHIP_E_RANGE, "candidate slot %u out of range", cand_slots[i]);
            pq[i] = q;
        }
    }
    std::lock_guard<std::mutex> lk(idx->mu);
    RH_TRY(rh_ensure_device(idx));
    std::vector<uint8_t> padded;
    std::vector<uint32_t> pop;
    stage_queries(idx, queries, nq, padded, pop);
    uint4 *dq = nullptr;
    uint32_t *dpop = nullptr, *dpq = nullptr, *dps = nullptr, *da = nullptr, *dorr = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&]() { if (dq) (void)hipFree(dq); if (dpop) (void)hipFree(dpop); if (dpq) (void)hipFree(dpq);
                           if (dps) (void)hipFree(dps); if (da) (void)hipFree(da); if (dorr) (void)hipFree(dorr);
                           if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); };
    RH_G(hipMalloc((void **)&dq, padded.size()));
    RH_G(hipMalloc((void **)&dpop, (size_t)nq * 4));
    RH_G(hipMalloc((void **)&dpq, n_pairs * 4));
    RH_G(hipMalloc((void **)&dps, n_pairs * 4));
    RH_G(hipMalloc((void **)&da, n_pairs * 4));
    RH_G(hipMalloc((void **)&dorr, n_pairs * 4));
    RH_G(hipMemcpyAsync(dq, padded.data(), padded.size(), hipMemcpyHostToDevice, idx->stream));
    RH_G(hipMemcpyAsync(dpop, pop.data(), (size_t)nq * 4, hipMemcpyHostToDevice, idx->stream));
    RH_G(hipMemcpyAsync(dpq, pq.data(), n_pairs * 4, hipMemcpyHostToDevice, idx->stream));
    RH_G(hipMemcpyAsync(dps, cand_slots, n_pairs * 4, hipMemcpyHostToDevice, idx->stream));
    const uint64_t per_block = 4ull * (64 / idx->lpr) * (idx->lpr >= 8 ? 8 : 4);   // GatherShape<LPR>::W x 4 waves
    // every wave strides over the rounds, so the grid is exactly what is resident at once (6
    // blocks = 24 waves per CU at 80 VGPRs; 4 and 8 blocks per CU measured 5-7 % slower) — a
    // larger grid would run its last blocks on a mostly idle chip
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, idx->device);
    uint32_t grid = (uint32_t)std::min<uint64_t>((n_pairs + per_block - 1) / per_block, (uint64_t)n_cu * RH_GATHER_BLOCKS_PER_CU);
    if (grid == 0) grid = 1;
    RH_G(hipEventCreate(&e0));
    RH_G(hipEventCreate(&e1));
    (void)hipEventRecord(e0, idx->stream);
    switch (idx->lpr) {
        case 1: hipLaunchKernelGGL(gather_kernel<1>, dim3(grid), dim3(256), 0, idx->stream, idx->d_fp, dq, dpop, dpq, dps, n_pairs, da, dorr); break;
        case 2: hipLaunchKernelGGL(gather_kernel<2>, dim3(grid), dim3(256), 0, idx->stream, idx->d_fp, dq, dpop, dpq, dps, n_pairs, da, dorr); break;
        case 4: hipLaunchKernelGGL(gather_kernel<4>, dim3(grid), dim3(256), 0, idx->stream, idx->d_fp, dq, dpop, dpq, dps, n_pairs, da, dorr); break;
        case 8: hipLaunchKernelGGL(gather_kernel<8>, dim3(grid), dim3(256), 0, idx->stream, idx->d_fp, dq, dpop, dpq, dps, n_pairs, da, dorr); break;
        default: hipLaunchKernelGGL(gather_kernel<16>, dim3(grid), dim3(256), 0, idx->stream, idx->d_fp, dq, dpop, dpq, dps, n_pairs, da, dorr); break;
    }
    (void)hipEventRecord(e1, idx->stream);
    RH_G(hipGetLastError());
    RH_G(hipMemcpyAsync(and_out, da, n_pairs * 4, hipMemcpyDeviceToHost, idx->stream));
    RH_G(hipMemcpyAsync(or_out, dorr, n_pairs * 4, hipMemcpyDeviceToHost, idx->stream));
    RH_G(hipStreamSynchronize(idx->stream));
    {
        float ms = 0.f;
        g_last_kernel_ms = hipEventElapsedTime(&ms, e0, e1) == hipSuccess ? ms : 0.0;
    }
    cleanup();
    return RADHIP_OK;
#undef RH_G
}

// ================================================================== peer-mapped corpus (round 4)
// BASELINE's partitioning — the fingerprint rows of one corpus sharded by contiguous slot range over the GPUs of a node —
// WITHOUT a lock step: every rank maps the row shards of all ranks into ONE contiguous virtual address range (HIP virtual
// memory management: its own physical allocation plus the peers' allocations imported through dmabuf file descriptors,
// reached over xGMI), so that `fp[slot]` is valid for every slot of the corpus and the single-GPU traversal kernel runs
// UNCHANGED: a gather of a remote row is a 128-B read over the fabric, one more hop of latency in a kernel that already
// keeps thousands of gathers in flight.  No collective, no frontier step, results bit-identical by construction (the same
// kernel reads the same bytes).  Per rank: N / G rows of HBM instead of N; per expansion ~4 x 128 B x (G - 1) / G cross the
// fabric.  The row-sharded lock-step loop (shard.hip: RCCL all-gather of the frontier candidates) stays, as north_star's
// stated exchange; this is the mode for throughput.
//   radhip_index_peer_create -> _peer_fill_synth | _peer_fill_rows -> _peer_export (fd to every peer) ->
//   _peer_import (every peer's fd) -> _peer_seal
struct RhPeerMap {
    int rank = 0, world = 0;
    uint64_t n_total = 0, rows_per_shard = 0;
    size_t shard_bytes = 0, va_bytes = 0;
    uint8_t *va = nullptr;
    std::vector<hipMemGenericAllocationHandle_t> handles;   // [world]
    std::vector<uint8_t> have, mapped;                      // [world]
    bool sealed = false;
};

static void rh_peer_release(radhip_index *idx) {
    RhPeerMap *pm = idx->peer;
    if (!pm) return;
    (void)hipDeviceSynchronize();
    for (int r = 0; r < pm->world; ++r) {
        if (pm->mapped[r]) (void)hipMemUnmap(pm->va + (size_t)r * pm->shard_bytes, pm->shard_bytes);
        if (pm->have[r]) (void)hipMemRelease(pm->handles[r]);
    }
    if (pm->va) (void)hipMemAddressFree(pm->va, pm->va_bytes);
    idx->device_bytes -= std::min<uint64_t>(idx->device_bytes, pm->shard_bytes);
    delete pm;
    idx->peer = nullptr; idx->d_fp = nullptr; idx->fp_cap_rows = 0;
}

static int peer_map_one(radhip_index *idx, int r) {
    RhPeerMap *pm = idx->peer;
    uint8_t *at = pm->va + (size_t)r * pm->shard_bytes;
    RH_HIP(hipMemMap(at, pm->shard_bytes, 0, pm->handles[r], 0));
    pm->mapped[r] = 1;
    hipMemAccessDesc d = {};
    d.location.type = hipMemLocationTypeDevice;
    d.location.id = idx->device;
    d.flags = hipMemAccessFlagsProtReadWrite;
    RH_HIP(hipMemSetAccess(at, pm->shard_bytes, &d, 1));
    return RADHIP_OK;
}

// Reserve the range for all n_total rows, create and map this rank's shard.  Rows per shard = ceil(n_total / world) rounded up
// to the allocation granularity (2 MiB = 16384 rows of 128 B), so that slot i sits at va + i * row_stride on every rank.
// Until radhip_index_peer_seal the index is a SHARD (rows [rank * rows_per_shard, ...) resident): the row-sharded loop can
// use it as it is.
extern "C" int radhip_index_peer_create(radhip_index_t *idx, int rank, int world, uint64_t n_total, uint64_t *out_rows_per_shard) {
    if (!idx) RH_FAIL(RADHIP_E_INVALID, "null index");
    if (world < 1 || rank < 0 || rank >= world || n_total == 0) RH_FAIL(RADHIP_E_INVALID, "bad rank / world / n_total");
    std::lock_guard<std::mutex> lk(idx->mu);
    RH_REQUIRE_OWN_CORPUS(idx);
    if (idx->has_vectors || idx->d_fp || idx->h_rows_pending) RH_FAIL(RADHIP_E_STATE, "the index holds a corpus already");
    RH_TRY(rh_ensure_device(idx));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = idx->device;
    prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
    size_t gran = 0;
    RH_HIP(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    // (the API's granule is 4 KiB on this stack; shards start on 2-MiB boundaries so that every shard can be backed and
    // translated in the large fragments a 2-MiB-aligned range allows)
    if (gran < ((size_t)2 << 20) && ((size_t)2 << 20) % (gran ? gran : 1) == 0) gran = (size_t)2 << 20;
    if (gran == 0 || gran % idx->row_stride) RH_FAIL(RADHIP_E_HIP, "allocation granularity %zu is not a multiple of the row stride", gran);
    const uint64_t rows_gran = gran / idx->row_stride;
    uint64_t rps = (n_total + (uint64_t)world - 1) / (uint64_t)world;
    rps = (rps + rows_gran - 1) / rows_gran * rows_gran;
    RhPeerMap *pm = new (std::nothrow) RhPeerMap();
    if (!pm) RH_FAIL(RADHIP_E_NOMEM, "out of host memory");
    pm->rank = rank; pm->world = world; pm->n_total = n_total; pm->rows_per_shard = rps;
    pm->shard_bytes = (size_t)rps * idx->row_stride;
    pm->va_bytes = pm->shard_bytes * (size_t)world;
    pm->handles.resize(world); pm->have.assign(world, 0); pm->mapped.assign(world, 0);
    idx->peer = pm;
    void *va = nullptr;
    hipError_t e = hipMemAddressReserve(&va, pm->va_bytes, gran, nullptr, 0);
    if (e != hipSuccess) { (void)hipGetLastError(); delete pm; idx->peer = nullptr; RH_FAIL(RADHIP_E_HIP, "hipMemAddressReserve(%zu) failed: %s", pm->va_bytes, hipGetErrorString(e)); }
    pm->va = (uint8_t *)va;
    e = hipMemCreate(&pm->handles[rank], pm->shard_bytes, &prop, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        const size_t sb = pm->shard_bytes;
        rh_peer_release(idx);
        RH_FAIL(e == hipErrorOutOfMemory ? RADHIP_E_NOMEM : RADHIP_E_HIP, "hipMemCreate(%zu) for this rank's shard failed: %s", sb, hipGetErrorString(e));
    }
    pm->have[rank] = 1;
    idx->device_bytes += pm->shard_bytes;
    { const int rc = peer_map_one(idx, rank); if (rc != RADHIP_OK) { rh_peer_release(idx); return rc; } }
    const uint64_t first = (uint64_t)rank * rps;
    idx->d_fp = (uint4 *)(pm->va + (size_t)rank * pm->shard_bytes);
    idx->fp_cap_rows = rps;
    idx->n = first < n_total ? std::min<uint64_t>(rps, n_total - first) : 0;
    idx->sharded = true; idx->shard_first = first; idx->n_total = n_total;
    if (out_rows_per_shard) *out_rows_per_shard = rps;
    return RADHIP_OK;
}

// this rank's rows of the closed-form corpus (the generator of radhip_index_synth_vectors), written into its shard
extern "C" int radhip_index_peer_fill_synth(radhip_index_t *idx, uint64_t seed, int mode) {
    if (!idx) RH_FAIL(RADHIP_E_INVALID, "null index");
    if (mode < 0 || mode > 2) RH_FAIL(RADHIP_E_INVALID, "mode must be 0, 1 or 2");
    std::lock_guard<std::mutex> lk(idx->mu);
    if (!idx->peer || idx->peer->sealed) RH_FAIL(RADHIP_E_STATE, "radhip_index_peer_create first (and fill before sealing)");
    RH_HIP(hipSetDevice(idx->device));
    if (idx->n) {
        hipLaunchKernelGGL(synth_rows_kernel, dim3(256 * 16), dim3(256), 0, idx->stream, (uint64_t *)idx->d_fp, idx->n,
                           idx->row_stride / 8, idx->row_bytes, idx->ndim_bits, idx->shard_first, idx->n_total, seed, mode);
        RH_HIP(hipGetLastError());
        RH_HIP(hipStreamSynchronize(idx->stream));
    }
    idx->has_vectors = true;
    idx->graph_gen++;
    return RADHIP_OK;
}
// ... or from the host: `rows` = this rank's rows [rank * rows_per_shard, ...) of the corpus, row_bytes each
extern "C" int radhip_index_peer_fill_rows(radhip_index_t *idx, const uint8_t *rows, uint64_t count) {
    if (!idx || (!rows && count)) RH_FAIL(RADHIP_E_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    if (!idx->peer || idx->peer->sealed) RH_FAIL(RADHIP_E_STATE, "radhip_index_peer_create first (and fill before sealing)");
    if (count != idx->n) RH_FAIL(RADHIP_E_INVALID, "this rank's shard holds %llu rows (got %llu)", (unsigned long long)idx->n, (unsigned long long)count);
    RH_HIP(hipSetDevice(idx->device));
    const uint64_t step = 1u << 20;
    std::vector<uint8_t> stage;
    try { stage.assign((size_t)std::min<uint64_t>(step, std::max<uint64_t>(count, 1)) * idx->row_stride, 0); } catch (...) { RH_FAIL(RADHIP_E_NOMEM, "out of host memory"); }
    for (uint64_t f = 0; f < count; f += step) {
        const uint64_t c = std::min(step, count - f);
        for (uint64_t i = 0; i < c; ++i) memcpy(stage.data() + i * idx->row_stride, rows + (f + i) * idx->row_bytes, idx->row_bytes);
        RH_HIP(hipMemcpy((uint8_t *)idx->d_fp + f * idx->row_stride, stage.data(), (size_t)c * idx->row_stride, hipMemcpyHostToDevice));
    }
    idx->has_vectors = true;
    idx->graph_gen++;
    return RADHIP_OK;
}
// a file descriptor (dmabuf) of this rank's shard for the peers: the caller passes it on (SCM_RIGHTS over a Unix socket between
// processes, as it is inside one process) and closes it
extern "C" int radhip_index_peer_export(radhip_index_t *idx, int *out_fd) {
    if (!idx || !out_fd) RH_FAIL(RADHIP_E_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    if (!idx->peer) RH_FAIL(RADHIP_E_STATE, "radhip_index_peer_create first");
    RH_HIP(hipSetDevice(idx->device));
    int fd = -1;
    RH_HIP(hipMemExportToShareableHandle(&fd, idx->peer->handles[idx->peer->rank], hipMemHandleTypePosixFileDescriptor, 0));
    *out_fd = fd;
    return RADHIP_OK;
}
// map the shard of rank `peer_rank` (its exported descriptor) at its place in this rank's range
extern "C" int radhip_index_peer_import(radhip_index_t *idx, int peer_rank, int fd) {
    if (!idx) RH_FAIL(RADHIP_E_INVALID, "null index");
    std::lock_guard<std::mutex> lk(idx->mu);
    RhPeerMap *pm = idx->peer;
    if (!pm || pm->sealed) RH_FAIL(RADHIP_E_STATE, "radhip_index_peer_create first (and import before sealing)");
    if (peer_rank < 0 || peer_rank >= pm->world || peer_rank == pm->rank) RH_FAIL(RADHIP_E_INVALID, "peer rank %d out of range", peer_rank);
    if (pm->have[peer_rank]) RH_FAIL(RADHIP_E_STATE, "the shard of rank %d is mapped already", peer_rank);
    RH_HIP(hipSetDevice(idx->device));
    RH_HIP(hipMemImportFromShareableHandle(&pm->handles[peer_rank], (void *)(uintptr_t)fd, hipMemHandleTypePosixFileDescriptor));
    pm->have[peer_rank] = 1;
    return peer_map_one(idx, peer_rank);
}
// every shard is mapped: the index holds the WHOLE corpus from here on (slot i at d_fp + i * lpr), read-only
extern "C" int radhip_index_peer_seal(radhip_index_t *idx) {
    if (!idx) RH_FAIL(RADHIP_E_INVALID, "null index");
    std::lock_guard<std::mutex> lk(idx->mu);
    RhPeerMap *pm = idx->peer;
    if (!pm) RH_FAIL(RADHIP_E_STATE, "radhip_index_peer_create first");
    if (!idx->has_vectors) RH_FAIL(RADHIP_E_STATE, "fill this rank's shard first");
    for (int r = 0; r < pm->world; ++r) if (!pm->mapped[r]) RH_FAIL(RADHIP_E_STATE, "the shard of rank %d is not mapped yet", r);
    RH_HIP(hipSetDevice(idx->device));
    RH_HIP(hipDeviceSynchronize());
    pm->sealed = true;
    idx->d_fp = (uint4 *)pm->va;
    idx->n = pm->n_total; idx->fp_cap_rows = pm->rows_per_shard * (uint64_t)pm->world;
    idx->sharded = false; idx->shard_first = 0; idx->n_total = 0;
    idx->graph_gen++;
    return RADHIP_OK;
}
// the graph of `src` (same device, same process) copied device to device into `dst` (a peer-mapped index adopts the graph of
// the index that built it without a trip through host memory)
extern "C" int radhip_index_copy_graph_from(radhip_index_t *dst, radhip_index_t *src) {
    if (!dst || !src || dst == src) RH_FAIL(RADHIP_E_INVALID, "bad argument");
    std::lock_guard<std::mutex> lk(dst < src ? dst->mu : src->mu);
    std::lock_guard<std::mutex> lk2(dst < src ? src->mu : dst->mu);
    RH_TRY(rh_ensure_device(src));
    RH_TRY(rh_ensure_device(dst));
    if (!src->has_graph || !src->d_graph_valid) RH_FAIL(RADHIP_E_STATE, "the source has no graph on its device");
    if (src->device != dst->device || src->cap0 != dst->cap0 || src->M != dst->M) RH_FAIL(RADHIP_E_INVALID, "source and destination differ in device or row widths");
    rh_layout_invalidate(dst);
    dst->g_n = src->g_n; dst->max_level = src->max_level; dst->entry = src->entry; dst->n_upper_rows = src->n_upper_rows;
    dst->has_graph = false; dst->d_graph_valid = false;
    RH_TRY(rh_alloc_graph_dev(dst));
    RH_HIP(hipMemcpy(dst->d_levels, src->d_levels, src->g_n, hipMemcpyDeviceToDevice));
    RH_HIP(hipMemcpy(dst->d_adj0, src->d_adj0, src->g_n * src->cap0 * 4, hipMemcpyDeviceToDevice));
    RH_HIP(hipMemcpy(dst->d_upper_row, src->d_upper_row, src->g_n * 4, hipMemcpyDeviceToDevice));
    if (src->n_upper_rows) RH_HIP(hipMemcpy(dst->d_adjU, src->d_adjU, src->n_upper_rows * src->M * 4, hipMemcpyDeviceToDevice));
    dst->h_top = src->h_top;
    if (dst->h_top.empty() && src->n_top) { dst->h_top.resize(src->n_top); RH_HIP(hipMemcpy(dst->h_top.data(), src->d_top, (size_t)src->n_top * 4, hipMemcpyDeviceToHost)); }
    RH_TRY(rh_upload_top(dst));
    std::vector<int8_t>().swap(dst->h_levels); std::vector<uint32_t>().swap(dst->h_adj0);
    std::vector<uint32_t>().swap(dst->h_upper_row); std::vector<uint32_t>().swap(dst->h_adjU);
    dst->h_graph_valid = false; dst->d_graph_valid = true; dst->has_graph = true;
    dst->graph_gen++;
    return RADHIP_OK;
}
