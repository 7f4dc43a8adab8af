// comm.h — the RCCL communicator object shared by comm.hip and shard.hip
#pragma once
#include "common.h"

#include <rccl/rccl.h>

struct radhip_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;
    void *d_send = nullptr, *d_recv = nullptr;
    size_t cap_send = 0, cap_recv = 0;
};

// device-buffer collectives on the caller's stream (no host staging, no synchronisation)
int rh_comm_allgather_dev(radhip_comm *c, const uint32_t *d_send, uint32_t *d_recv, size_t count_u32, hipStream_t st);
int rh_comm_reduce_scatter_u32_dev(radhip_comm *c, const uint32_t *d_send, uint32_t *d_recv, size_t count_u32, hipStream_t st);
// ncclCommAbort: the peers' pending collectives fail instead of hanging; the communicator is unusable afterwards
int rh_comm_abort(radhip_comm *c);
