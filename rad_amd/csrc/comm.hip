// comm.hip — RCCL communicator behind the C ABI: the exchange step of the sharded
// traversal (per-round all-gather of frontier candidate scores / scored counts over xGMI).
// One process per GPU; the ncclUniqueId is created by rank 0 and handed to the other ranks
// by the host program (bench.py: a plain TCP rendezvous, rad_amd/rendezvous.py; no torch).
#include "comm.h"

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
    RH_NCCL(ncclAllGather(d_send, d_recv, count_u32, ncclUint32, c->comm, st));
    return RADHIP_OK;
}
int rh_comm_reduce_scatter_u32_dev(radhip_comm *c, const uint32_t *d_send, uint32_t *d_recv, size_t count_u32, hipStream_t st) {
    if (c->world == 1) {
        RH_HIP(hipMemcpyAsync(d_recv, d_send, count_u32 * 4, hipMemcpyDeviceToDevice, st));
        return RADHIP_OK;
    }
    RH_NCCL(ncclReduceScatter(d_send, d_recv, count_u32, ncclUint32, ncclSum, c->comm, st));
    return RADHIP_OK;
}

extern "C" int radhip_comm_rank(const radhip_comm_t *c) { return c ? c->rank : -1; }
extern "C" int radhip_comm_world(const radhip_comm_t *c) { return c ? c->world : -1; }
