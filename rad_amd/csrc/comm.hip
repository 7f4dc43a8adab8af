// comm.hip — RCCL communicator behind the C ABI: the exchange step of the sharded
// traversal (per-round all-gather of frontier candidate scores / scored counts over xGMI).
// One process per GPU; the ncclUniqueId is created by rank 0 and handed to the other ranks
// by the host program (bench.py: a plain TCP rendezvous, rad_amd/rendezvous.py; no torch).
#include "comm.h"

#include <algorithm>
#include <new>

#define RH_NCCL(expr)                                                                   \
    do {                                                                                \
        ncclResult_t r_ = (expr);                                                       \
        if (r_ != ncclSuccess) {                                                        \
            radhip_set_error("%s failed: %s", #expr, ncclGetErrorString(r_));           \
            return RADHIP_E_COMM;                                                       \
        }                                                                               \
    } while (0)

extern "C" int radhip_comm_unique_id(uint8_t *out128) {
    if (!out128) RH_FAIL(RADHIP_E_INVALID, "null argument");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    RH_NCCL(ncclGetUniqueId(&id));
    memcpy(out128, &id, 128);
    return RADHIP_OK;
}

extern "C" int radhip_comm_create(int rank, int world, const uint8_t *id128, int device, radhip_comm_t **out) {
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) RH_FAIL(RADHIP_E_INVALID, "bad argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n)
        RH_FAIL(RADHIP_E_NO_DEVICE, "device %d not visible (%d devices)", device, n);
    RH_HIP(hipSetDevice(device));
    radhip_comm *c = new (std::nothrow) radhip_comm();
    if (!c) RH_FAIL(RADHIP_E_NOMEM, "out of host memory");
    c->rank = rank; c->world = world; c->device = device;
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        radhip_set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, ncclGetErrorString(r));
        delete c;
        return RADHIP_E_COMM;
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        ncclCommDestroy(c->comm);
        delete c;
        RH_FAIL(RADHIP_E_HIP, "hipStreamCreate failed");
    }
    *out = c;
    return RADHIP_OK;
}

extern "C" int radhip_comm_destroy(radhip_comm_t *c) {
    if (!c) return RADHIP_OK;
    (void)hipSetDevice(c->device);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->comm) ncclCommDestroy(c->comm);
    delete c;
    return RADHIP_OK;
}

// every rank contributes `count` u64 words; recv gets world*count words in rank order
extern "C" int radhip_comm_allgather_u64(radhip_comm_t *c, const uint64_t *send, uint64_t count, uint64_t *recv) {
    if (!c || !send || !recv) RH_FAIL(RADHIP_E_INVALID, "null argument");
    if (count == 0) return RADHIP_OK;
    RH_HIP(hipSetDevice(c->device));
    const size_t sb = count * 8, rb = sb * (size_t)c->world;
    if (sb > c->cap_send) {
        if (c->d_send) (void)hipFree(c->d_send);
        c->d_send = nullptr;
        RH_HIP(hipMalloc(&c->d_send, sb));
        c->cap_send = sb;
    }
    if (rb > c->cap_recv) {
        if (c->d_recv) (void)hipFree(c->d_recv);
        c->d_recv = nullptr;
        RH_HIP(hipMalloc(&c->d_recv, rb));
        c->cap_recv = rb;
    }
    if (!c->comm) RH_FAIL(RADHIP_E_COMM, "the communicator was aborted");
    RH_HIP(hipMemcpyAsync(c->d_send, send, sb, hipMemcpyHostToDevice, c->stream));
    RH_NCCL(ncclAllGather(c->d_send, c->d_recv, count, ncclUint64, c->comm, c->stream));
    RH_HIP(hipMemcpyAsync(recv, c->d_recv, rb, hipMemcpyDeviceToHost, c->stream));
    RH_HIP(hipStreamSynchronize(c->stream));
    return RADHIP_OK;
}

// ---- device-buffer collectives of the row-sharded traversal (shard.hip): frontier candidates out with
// ncclAllGather, their scores back with ncclReduceScatter (every candidate is scored by the one rank that
// owns its row, the others contribute 0), both on the stream the kernels run on
int rh_comm_allgather_dev(radhip_comm *c, const uint32_t *d_send, uint32_t *d_recv, size_t count_u32, hipStream_t st) {
    if (c->world == 1) {
        RH_HIP(hipMemcpyAsync(d_recv, d_send, count_u32 * 4, hipMemcpyDeviceToDevice, st));
        return RADHIP_OK;
    }
    if (!c->comm) RH_FAIL(RADHIP_E_COMM, "the communicator was aborted");
    RH_NCCL(ncclAllGather(d_send, d_recv, count_u32, ncclUint32, c->comm, st));
    return RADHIP_OK;
}
int rh_comm_reduce_scatter_u32_dev(radhip_comm *c, const uint32_t *d_send, uint32_t *d_recv, size_t count_u32, hipStream_t st) {
    if (c->world == 1) {
        RH_HIP(hipMemcpyAsync(d_recv, d_send, count_u32 * 4, hipMemcpyDeviceToDevice, st));
        return RADHIP_OK;
    }
    if (!c->comm) RH_FAIL(RADHIP_E_COMM, "the communicator was aborted");
    RH_NCCL(ncclReduceScatter(d_send, d_recv, count_u32, ncclUint32, ncclSum, c->comm, st));
    return RADHIP_OK;
}

extern "C" int radhip_comm_rank(const radhip_comm_t *c) { return c ? c->rank : -1; }
extern "C" int radhip_comm_world(const radhip_comm_t *c) { return c ? c->world : -1; }

extern "C" int radhip_comm_info(const radhip_comm_t *c, radhip_comm_info_t *out) {
    if (!c || !out) RH_FAIL(RADHIP_E_INVALID, "null argument");
    memset(out, 0, sizeof *out);
    int v = 0, cnt = 0, ur = 0;
    RH_NCCL(ncclGetVersion(&v));
    if (!c->comm) RH_FAIL(RADHIP_E_COMM, "the communicator was aborted");
    RH_NCCL(ncclCommCount(c->comm, &cnt));
    RH_NCCL(ncclCommUserRank(c->comm, &ur));
    out->rccl_version = v; out->comm_count = cnt; out->comm_rank = ur; out->device = c->device;
    if (hipDeviceGetPCIBusId(out->pci_bus_id, (int)sizeof out->pci_bus_id, c->device) != hipSuccess) out->pci_bus_id[0] = 0;
    return RADHIP_OK;
}

// abort a communicator whose peers may be blocked in a collective this rank will never enter (an error inside
// radhip_shard_run): ncclCommAbort makes their pending collectives fail instead of hanging
int rh_comm_abort(radhip_comm *c) {
    if (c && c->comm) { (void)ncclCommAbort(c->comm); c->comm = nullptr; }
    return RADHIP_OK;
}

// ---- the graph of `root` on every rank, device to device over xGMI ------------------------------------
void rh_layout_invalidate(radhip_index *idx);
static int bcast_bytes(radhip_comm *c, void *p, size_t bytes, int root, hipStream_t st) {
    // pieces of 1 GiB: a level-0 adjacency of 1B nodes is 64 GB
    const size_t piece = (size_t)1 << 30;
    for (size_t off = 0; off < bytes; off += piece) {
        const size_t n = std::min(piece, bytes - off);
        RH_NCCL(ncclBroadcast((const uint8_t *)p + off, (uint8_t *)p + off, n, ncclUint8, root, c->comm, st));
    }
    return RADHIP_OK;
}

extern "C" int radhip_index_broadcast_graph(radhip_index_t *idx, radhip_comm_t *c, int root) {
    if (!idx || !c) RH_FAIL(RADHIP_E_INVALID, "null argument");
    if (root < 0 || root >= c->world) RH_FAIL(RADHIP_E_INVALID, "root %d out of range", root);
    std::lock_guard<std::mutex> lk(idx->mu);
    RH_TRY(rh_ensure_device(idx));
    if (idx->device != c->device) RH_FAIL(RADHIP_E_INVALID, "index and communicator live on different devices");
    const bool is_root = c->rank == root;
    if (is_root && (!idx->has_graph || !idx->d_graph_valid)) RH_FAIL(RADHIP_E_STATE, "the root has no graph on its device");
    if (c->world == 1) return RADHIP_OK;
    if (!c->comm) RH_FAIL(RADHIP_E_COMM, "the communicator was aborted");
    hipStream_t st = idx->stream;
    unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (is_root) {
        h[0] = idx->g_n; h[1] = (unsigned long long)(long long)idx->max_level; h[2] = idx->entry; h[3] = idx->n_upper_rows;
        h[4] = idx->n_top; h[5] = idx->cap0; h[6] = idx->M; h[7] = 0x52414447ull;
    }
    unsigned long long *dh = nullptr;
    RH_HIP(hipMalloc((void **)&dh, sizeof h));
    int rc = RADHIP_OK;
    auto fail = [&](int code) { (void)hipFree(dh); return code; };
    if (hipMemcpyAsync(dh, h, sizeof h, hipMemcpyHostToDevice, st) != hipSuccess) return fail(RADHIP_E_HIP);
    if ((rc = bcast_bytes(c, dh, sizeof h, root, st)) != RADHIP_OK) return fail(rc);
    if (hipMemcpyAsync(h, dh, sizeof h, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail(RADHIP_E_HIP);
    // Every rank checks the header against its own index BEFORE anything of the index changes, and all ranks agree on
    // accept / reject (one word, all-reduced) before the bulk broadcasts: a receiver that rejected used to return while the
    // root went on into ncclBroadcast and an unbounded wait.
    unsigned long long verdict = 0;   // 0 = accept; else the reason
    if (h[7] != 0x52414447ull) verdict = 1;
    else if (h[5] != idx->cap0 || h[6] != idx->M) verdict = 2;
    else if (!is_root && idx->sharded && idx->n_total != h[0]) verdict = 3;
    unsigned long long agreed = verdict;
    if (hipMemcpyAsync(dh, &verdict, 8, hipMemcpyHostToDevice, st) != hipSuccess) return fail(RADHIP_E_HIP);
    if (ncclAllReduce(dh, dh, 1, ncclUint64, ncclMax, c->comm, st) != ncclSuccess) { radhip_set_error("graph broadcast: ncclAllReduce of the header verdict failed"); return fail(RADHIP_E_COMM); }
    if (hipMemcpyAsync(&agreed, dh, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail(RADHIP_E_HIP);
    (void)hipFree(dh);
    if (verdict == 1) RH_FAIL(RADHIP_E_COMM, "graph broadcast: bad header from rank %d", root);
    if (verdict == 2)
        RH_FAIL(RADHIP_E_INVALID, "graph broadcast: rank %d has rows of %llu / %llu slots, this index %u / %u", root, h[5], h[6], idx->cap0, idx->M);
    if (verdict == 3)
        RH_FAIL(RADHIP_E_INVALID, "graph broadcast: the graph has %llu nodes, this shard belongs to a corpus of %llu rows", h[0], (unsigned long long)idx->n_total);
    if (agreed != 0) RH_FAIL(RADHIP_E_INVALID, "graph broadcast: another rank rejected the header of rank %d (reason %llu): nothing was broadcast", root, agreed);
    if (!is_root) {
        rh_layout_invalidate(idx);
        idx->g_n = h[0]; idx->max_level = (int32_t)(long long)h[1]; idx->entry = (uint32_t)h[2]; idx->n_upper_rows = h[3];
        idx->has_graph = false; idx->d_graph_valid = false;
        RH_TRY(rh_alloc_graph_dev(idx));
        idx->n_top = (uint32_t)h[4];
        RH_HIP(hipMalloc((void **)&idx->d_top, std::max<size_t>(idx->n_top, 4) * 4));
        idx->device_bytes += (size_t)idx->n_top * 4;
    }
    RH_TRY(bcast_bytes(c, idx->d_levels, idx->g_n, root, st));
    RH_TRY(bcast_bytes(c, idx->d_adj0, idx->g_n * idx->cap0 * 4, root, st));
    RH_TRY(bcast_bytes(c, idx->d_upper_row, idx->g_n * 4, root, st));
    if (idx->n_upper_rows) RH_TRY(bcast_bytes(c, idx->d_adjU, idx->n_upper_rows * idx->M * 4, root, st));
    if (idx->n_top) RH_TRY(bcast_bytes(c, idx->d_top, (size_t)idx->n_top * 4, root, st));
    RH_HIP(hipStreamSynchronize(st));
    if (!is_root) {
        idx->h_top.resize(idx->n_top);
        if (idx->n_top) RH_HIP(hipMemcpy(idx->h_top.data(), idx->d_top, (size_t)idx->n_top * 4, hipMemcpyDeviceToHost));
        std::vector<int8_t>().swap(idx->h_levels);
        std::vector<uint32_t>().swap(idx->h_adj0);
        std::vector<uint32_t>().swap(idx->h_upper_row);
        std::vector<uint32_t>().swap(idx->h_adjU);
        idx->h_graph_valid = false;
        idx->d_graph_valid = true;
        idx->has_graph = true;
    }
    return RADHIP_OK;
}
