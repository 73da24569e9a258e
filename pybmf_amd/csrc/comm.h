// Internal layout of the communicator object (comm.hip) and the primitives the sharded loop (api.hip) is written in.
#pragma once
#include "common.h"

#include <vector>

struct bmf_comm {
    int kind = 0, world = 1, rank = 0;
    void* nccl = nullptr;            // ncclComm_t (BMF_COMM_RCCL)
    bmf_allreduce_fn fn = nullptr;   // BMF_COMM_HOST
    void* user = nullptr;
    hipStream_t cs = nullptr;        // the collectives run here, fenced against the compute stream by ev[]
    hipEvent_t ev[4] = {};           // 0, 1, 3: compute -> collectives ("this buffer is complete"); 2: collectives -> compute (done)
    std::vector<hipEvent_t> tev;     // timing: per step (X^T U starts, before the wait, after the wait)
    int t_cap = 0, t_used = 0;
};

int bmf_comm_group_begin(bmf_comm* c);
int bmf_comm_group_end(bmf_comm* c);
int bmf_comm_allreduce_on(bmf_comm* c, void* buf, int64_t count, int dtype, hipStream_t s);
