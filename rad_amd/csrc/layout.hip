// layout.hip — graph-locality layout of the slots for the grouped visited table (gfx950 host side +
// one small kernel).
//
// The traversal kernels test every neighbour of an expanded node against the traversal's visited /
// scored set (rad/visited.py:17-29, rad/scored.py:37-47 in the reference: one Redis round trip each).
// With a per-slot hash table that is one random 128-B HBM line per neighbour, and it was what bound
// the round-1 kernel (profiles/r01).  The grouped table (traverse.hip, "GT") keeps 2 bits per slot in
// 16-B chunks of 48 consecutive layout ids, 8 chunks to a line, so the probes of one adjacency row cost
// as many lines as the row's neighbours span groups of 384 ids.  This file computes the ids: a
// renumbering `lid` of the slots that keeps graph neighbours together, from the level-0 adjacency alone.
//
// Algorithm: greedy graph growing.  A block of RH_LAYOUT_GROUP ids is grown from a seed by taking, one
// node at a time, the unclaimed node with the most edges from the block so far (bucket queue by edge
// count, ties to the node that entered first); when the frontier dries up the block continues from a node
// left over by an earlier block, else from the next unclaimed slot.  Blocks are grown by several host
// threads at once (claims are atomic), so the ids are not reproducible run to run — nothing observable
// depends on them: queue keys, scored lists and every other output of a traversal use slots.
//
// Measured on the hierarchical synthetic corpus (1M rows, HNSW built with expansion_add 64, mean degree
// 7.5): neighbours of an expanded node span 3.2 groups on average, against 8.9 distinct lines for the
// per-slot table and 2.6 for the generator's own tree order (scripts/locality_sim.py).
#include "common.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <new>
#include <thread>

// ids per block: 384 = the grouped table's group (8 chunks of 48 ids to a 128-B line).  RADHIP_LAYOUT_GROUP overrides it for
// the locality-hashed bucket table, whose line holds 8 ids (experiments: profiles/r04)
static uint32_t layout_group() {
    static const uint32_t g = [] { const char *e = getenv("RADHIP_LAYOUT_GROUP"); const int v = e ? atoi(e) : 0; return (v >= 4 && v <= 4096) ? (uint32_t)v : 384u; }();
    return g;
}
#define RH_LAYOUT_GROUP (layout_group())
#define LG_MAXC 32

namespace {

struct Grower {
    const uint32_t *adj;
    uint32_t cap0;
    uint64_t n, lo, hi, scan;
    std::atomic<uint8_t> *claimed;
    std::atomic<uint64_t> *next_block;
    uint32_t *lid;
    // edge counts of the frontier of the block being grown: a small open-addressed map (node -> count)
    std::vector<uint32_t> hkey;
    std::vector<uint16_t> hcnt;
    std::vector<uint32_t> hused;
    uint32_t hmask;
    std::vector<uint32_t> bucket[LG_MAXC + 1];
    size_t bhead[LG_MAXC + 1];
    std::vector<uint32_t> carry;

    bool claim(uint32_t v) {
        uint8_t z = 0;
        return claimed[v].load(std::memory_order_relaxed) == 0 &&
               claimed[v].compare_exchange_strong(z, 1, std::memory_order_relaxed);
    }
    uint16_t &count_of(uint32_t v) {
        uint32_t i = (v * 2654435769u) & hmask;
        for (;;) {
            if (hkey[i] == v) return hcnt[i];
            if (hkey[i] == RADHIP_NO_SLOT) { hkey[i] = v; hcnt[i] = 0; hused.push_back(i); return hcnt[i]; }
            i = (i + 1u) & hmask;
        }
    }
    uint16_t count_get(uint32_t v) const {
        uint32_t i = (v * 2654435769u) & hmask;
        for (;;) {
            if (hkey[i] == v) return hcnt[i];
            if (hkey[i] == RADHIP_NO_SLOT) return 0;
            i = (i + 1u) & hmask;
        }
    }
    void reset_block() {
        for (uint32_t i : hused) hkey[i] = RADHIP_NO_SLOT;
        hused.clear();
        for (int c = 0; c <= LG_MAXC; ++c) { bucket[c].clear(); bhead[c] = 0; }
    }
    // next seed, already claimed by this thread; NO_SLOT when this thread's slot range is exhausted
    uint32_t seed() {
        while (!carry.empty()) {
            const uint32_t c = carry.back();
            carry.pop_back();
            if (claim(c)) return c;
        }
        while (scan < hi) {
            const uint32_t s = (uint32_t)scan++;
            if (claim(s)) return s;
        }
        return RADHIP_NO_SLOT;
    }
    void run() {
        hmask = (1u << 15) - 1u;    // a block's frontier stays far below 32k nodes (384 x 64 edges at most)
        hkey.assign(hmask + 1u, RADHIP_NO_SLOT);
        hcnt.assign(hmask + 1u, 0);
        scan = lo;
        for (;;) {
            uint32_t cur = seed();
            if (cur == RADHIP_NO_SLOT) return;
            const uint64_t base = next_block->fetch_add(1, std::memory_order_relaxed) * RH_LAYOUT_GROUP;
            reset_block();
            uint32_t filled = 0;
            int top = 0;
            for (;;) {
                lid[cur] = (uint32_t)(base + filled++);
                if (filled >= RH_LAYOUT_GROUP) break;
                const uint32_t *row = adj + (uint64_t)cur * cap0;
                for (uint32_t j = 0; j < cap0; ++j) {
                    const uint32_t v = row[j];
                    if (v == RADHIP_NO_SLOT) break;
                    if (claimed[v].load(std::memory_order_relaxed)) continue;
                    if (hused.size() * 2 > hmask) continue;    // map full (cannot happen at these sizes): stop counting
                    uint16_t &c = count_of(v);
                    if (c < LG_MAXC) c++;
                    bucket[c].push_back(v);
                    if (c > top) top = c;
                }
                cur = RADHIP_NO_SLOT;
                while (top > 0 && cur == RADHIP_NO_SLOT) {
                    while (bhead[top] < bucket[top].size()) {
                        const uint32_t v = bucket[top][bhead[top]++];
                        if (count_get(v) == (uint16_t)top && claim(v)) { cur = v; break; }   // stale entries are skipped
                    }
                    if (cur == RADHIP_NO_SLOT) top--;
                }
                if (cur == RADHIP_NO_SLOT) {      // frontier exhausted: the block continues from a fresh seed
                    cur = seed();
                    if (cur == RADHIP_NO_SLOT) return;
                }
            }
            // what is left of the frontier, best connected last (popped first), seeds the next blocks
            if (carry.size() > (1u << 16)) carry.erase(carry.begin(), carry.begin() + (carry.size() >> 1));
            for (int c = 1; c <= LG_MAXC; ++c)
                for (size_t i = bhead[c]; i < bucket[c].size(); ++i) {
                    const uint32_t v = bucket[c][i];
                    if (count_get(v) == (uint16_t)c && !claimed[v].load(std::memory_order_relaxed)) carry.push_back(v);
                }
        }
    }
};

}   // namespace

static void free_layout_dev(radhip_index *idx) {
    if (!idx->dev_ready) return;
    (void)hipSetDevice(idx->device);
    auto fr = [&](void *p, size_t bytes) {
        if (!p) return;
        (void)hipFree(p);
        idx->device_bytes -= std::min<uint64_t>(idx->device_bytes, bytes);
    };
    fr(idx->d_lid, idx->g_n * 4); idx->d_lid = nullptr;
    fr(idx->d_adjx0, idx->g_n * idx->cap0 * 8); idx->d_adjx0 = nullptr;
    fr(idx->d_adjxU, idx->n_upper_rows * idx->M * 8); idx->d_adjxU = nullptr;
    fr(idx->d_topx, (size_t)idx->n_top * 8); idx->d_topx = nullptr;
}

void rh_layout_invalidate(radhip_index *idx) {
    idx->graph_gen++;
    if (!idx->layout_valid && !idx->d_lid) return;
    idx->layout_valid = false;
    std::vector<uint32_t>().swap(idx->h_lid);
    free_layout_dev(idx);
}

void rh_layout_free(radhip_index *idx) { free_layout_dev(idx); }

__global__ void pair_rows_kernel(const uint32_t *__restrict__ adj, const uint32_t *__restrict__ lid, uint64_t n,
                                 uint2 *__restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t s = adj[i];
        out[i] = make_uint2(s, s == RADHIP_NO_SLOT ? RADHIP_NO_SLOT : lid[s]);
    }
}

// upload lid and build the {slot, lid} pair rows the grouped-table kernels read
static int layout_upload(radhip_index *idx) {
    RH_TRY(rh_ensure_device(idx));
    // Traversal objects bound to the grouped table hold d_lid / d_adjx0 / d_adjxU / d_topx: replacing an installed
    // layout invalidates them like any other change of the index (their run() / reset() return RADHIP_E_STATE instead
    // of reading freed arrays).  The FIRST layout of a graph frees nothing, so nobody is invalidated for it.
    if (idx->d_lid) idx->graph_gen++;
    free_layout_dev(idx);
    const uint64_t n = idx->g_n;
    auto al = [&](void **p, size_t bytes) -> int {
        RH_HIP(hipMalloc(p, bytes ? bytes : 16));
        idx->device_bytes += bytes;
        return RADHIP_OK;
    };
    RH_TRY(al((void **)&idx->d_lid, n * 4));
    RH_TRY(al((void **)&idx->d_adjx0, n * idx->cap0 * 8));
    RH_TRY(al((void **)&idx->d_adjxU, idx->n_upper_rows * idx->M * 8));
    RH_TRY(al((void **)&idx->d_topx, (size_t)idx->n_top * 8));
    RH_HIP(hipMemcpyAsync(idx->d_lid, idx->h_lid.data(), n * 4, hipMemcpyHostToDevice, idx->stream));
    hipLaunchKernelGGL(pair_rows_kernel, dim3(256 * 8), dim3(256), 0, idx->stream, idx->d_adj0, idx->d_lid,
                       n * idx->cap0, idx->d_adjx0);
    if (idx->n_upper_rows)
        hipLaunchKernelGGL(pair_rows_kernel, dim3(256 * 2), dim3(256), 0, idx->stream, idx->d_adjU, idx->d_lid,
                           idx->n_upper_rows * idx->M, idx->d_adjxU);
    if (idx->n_top)
        hipLaunchKernelGGL(pair_rows_kernel, dim3(4), dim3(256), 0, idx->stream, idx->d_top, idx->d_lid,
                           (uint64_t)idx->n_top, idx->d_topx);
    RH_HIP(hipGetLastError());
    RH_HIP(hipStreamSynchronize(idx->stream));
    return RADHIP_OK;
}

// neighbours of a row span how many groups?  (sampled; the figure the kernel choice is made on)
static void layout_quality(radhip_index *idx) {
    const uint64_t n = idx->g_n;
    const uint64_t step = n > 2000000 ? n / 1000000 : 1;
    uint64_t rows = 0, lines = 0, deg = 0;
    for (uint64_t u = 0; u < n; u += step) {
        const uint32_t *row = idx->h_adj0.data() + u * idx->cap0;
        uint32_t g[64], k = 0;
        for (uint32_t j = 0; j < idx->cap0 && row[j] != RADHIP_NO_SLOT; ++j) {
            const uint32_t gg = idx->h_lid[row[j]] / RH_LAYOUT_GROUP;
            uint32_t t = 0;
            for (; t < k; ++t) if (g[t] == gg) break;
            if (t == k) g[k++] = gg;
            deg++;
        }
        lines += k;
        rows++;
    }
    idx->layout_lines_per_row = rows ? (double)lines / (double)rows : 0.0;
    idx->layout_degree = rows ? (double)deg / (double)rows : 0.0;
}

extern "C" int radhip_index_optimize_layout(radhip_index_t *idx, uint32_t n_threads) {
    if (!idx) RH_FAIL(RADHIP_E_INVALID, "null index");
    std::lock_guard<std::mutex> lk(idx->mu);
    return rh_optimize_layout_locked(idx, n_threads);
}

int rh_optimize_layout_locked(radhip_index *idx, uint32_t n_threads) {
    if (!idx->has_graph || idx->g_n == 0) RH_FAIL(RADHIP_E_STATE, "no graph loaded");
    RH_TRY(rh_ensure_host_graph(idx));
    const uint64_t n = idx->g_n;
    if (n_threads == 0) {
        n_threads = std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) n_threads = std::min<uint32_t>(n_threads, (uint32_t)CPU_COUNT(&set));
        if (n_threads == 0) n_threads = 1;
        if (n_threads > 32) n_threads = 32;
    }
    if (n < 100000) n_threads = 1;
    const auto t0 = std::chrono::steady_clock::now();
    std::atomic<uint8_t> *claimed = nullptr;
    try {
        idx->h_lid.assign(n, RADHIP_NO_SLOT);
        claimed = new std::atomic<uint8_t>[n];
    } catch (...) { RH_FAIL(RADHIP_E_NOMEM, "out of host memory for the layout of %llu nodes", (unsigned long long)n); }
    for (uint64_t i = 0; i < n; ++i) claimed[i].store(0, std::memory_order_relaxed);
    std::atomic<uint64_t> next_block{0};
    std::vector<Grower> gs(n_threads);
    std::vector<std::thread> th;
    bool failed = false;
    for (uint32_t t = 0; t < n_threads; ++t) {
        Grower &g = gs[t];
        g.adj = idx->h_adj0.data(); g.cap0 = idx->cap0; g.n = n;
        g.lo = n * t / n_threads; g.hi = n * (t + 1) / n_threads;
        g.claimed = claimed; g.next_block = &next_block; g.lid = idx->h_lid.data();
    }
    try {
        if (n_threads == 1) gs[0].run();
        else {
            for (uint32_t t = 0; t < n_threads; ++t) th.emplace_back([&gs, t]() { gs[t].run(); });
            for (auto &x : th) x.join();
        }
    } catch (...) { failed = true; for (auto &x : th) if (x.joinable()) x.join(); }
    delete[] claimed;
    if (failed) { std::vector<uint32_t>().swap(idx->h_lid); RH_FAIL(RADHIP_E_NOMEM, "layout computation failed (out of host memory)"); }
    idx->lid_limit = next_block.load() * RH_LAYOUT_GROUP;
    if (idx->lid_limit >= 48ull * ((1ull << 25) - 2ull)) {
        std::vector<uint32_t>().swap(idx->h_lid);
        RH_FAIL(RADHIP_E_RANGE, "layout ids exceed the grouped table's 25-bit chunk tags");
    }
    layout_quality(idx);
    idx->layout_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    idx->layout_valid = true;
    if (idx->has_vectors) {   // an index that only serves adjacency reads keeps the layout on the host
        int rc = layout_upload(idx);
        if (rc != RADHIP_OK) { idx->layout_valid = false; return rc; }
    }
    return RADHIP_OK;
}

// install a given layout (tests: results must not depend on it; tools: a layout computed elsewhere).
// lid must be injective with every value < 48 * (2^25 - 2).
extern "C" int radhip_index_set_layout(radhip_index_t *idx, const uint32_t *lid) {
    if (!idx || !lid) RH_FAIL(RADHIP_E_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    if (!idx->has_graph || idx->g_n == 0) RH_FAIL(RADHIP_E_STATE, "no graph loaded");
    RH_TRY(rh_ensure_host_graph(idx));
    const uint64_t n = idx->g_n;
    uint64_t mx = 0;
    for (uint64_t i = 0; i < n; ++i) mx = std::max<uint64_t>(mx, lid[i]);
    if (mx >= 48ull * ((1ull << 25) - 2ull)) RH_FAIL(RADHIP_E_RANGE, "layout id %llu too large", (unsigned long long)mx);
    {
        std::vector<bool> seen;
        try { seen.assign(mx + 1, false); } catch (...) { RH_FAIL(RADHIP_E_NOMEM, "out of host memory"); }
        for (uint64_t i = 0; i < n; ++i) {
            if (seen[lid[i]]) RH_FAIL(RADHIP_E_INVALID, "layout is not injective (id %u twice)", lid[i]);
            seen[lid[i]] = true;
        }
    }
    try { idx->h_lid.assign(lid, lid + n); } catch (...) { RH_FAIL(RADHIP_E_NOMEM, "out of host memory"); }
    idx->lid_limit = mx + 1;
    layout_quality(idx);
    idx->layout_seconds = 0.0;
    idx->layout_valid = true;
    if (idx->has_vectors) {
        int rc = layout_upload(idx);
        if (rc != RADHIP_OK) { idx->layout_valid = false; return rc; }
    }
    return RADHIP_OK;
}

extern "C" int radhip_index_read_layout(const radhip_index_t *cidx, uint32_t *out_lid) {
    radhip_index *idx = const_cast<radhip_index *>(cidx);
    if (!idx || !out_lid) RH_FAIL(RADHIP_E_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    if (!idx->layout_valid) RH_FAIL(RADHIP_E_STATE, "the index has no layout (radhip_index_optimize_layout)");
    memcpy(out_lid, idx->h_lid.data(), idx->g_n * 4);
    return RADHIP_OK;
}

extern "C" int radhip_index_layout_info(const radhip_index_t *cidx, radhip_layout_info_t *out) {
    radhip_index *idx = const_cast<radhip_index *>(cidx);
    if (!idx || !out) RH_FAIL(RADHIP_E_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    memset(out, 0, sizeof *out);
    out->valid = idx->layout_valid ? 1 : 0;
    out->group = RH_LAYOUT_GROUP;
    out->id_limit = idx->lid_limit;
    out->groups_per_row = idx->layout_lines_per_row;
    out->degree = idx->layout_degree;
    out->seconds = idx->layout_seconds;
    return RADHIP_OK;
}
