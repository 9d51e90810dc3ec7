// sr_rccl.h -- the handful of RCCL entry points the strip gather needs (grouped ncclSend / ncclRecv over
// xGMI), bound at first use with dlopen: a single-GPU host never loads the collective library, and a process that already has one
// (PyTorch ships its own librccl.so) keeps using that one -- two RCCL copies in one process do not see each other's communicators.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <string>

namespace sr {

constexpr int kRcclIdBytes = 128;                 // NCCL_UNIQUE_ID_BYTES (rccl.h:40)
struct RcclId { char internal[kRcclIdBytes]; };   // ncclUniqueId, passed by value (rccl.h:43)
typedef struct ncclComm* RcclComm;                // ncclComm_t (opaque)

struct RcclApi {
    // rccl.h:187, 220, 236, 260, 700, 722, 923, 933 (ROCm 7.2); ncclInt32 == 2 (rccl.h:461)
    int (*GetUniqueId)(RcclId*);
    int (*CommInitRank)(RcclComm*, int nranks, RcclId id, int rank);
    int (*CommInitAll)(RcclComm*, int ndev, const int* devlist);
    int (*CommDestroy)(RcclComm);
    int (*Send)(const void*, size_t count, int datatype, int peer, RcclComm, hipStream_t);
    int (*Recv)(void*, size_t count, int datatype, int peer, RcclComm, hipStream_t);
    int (*GroupStart)();
    int (*GroupEnd)();
    const char* (*GetErrorString)(int);
};
constexpr int kRcclInt32 = 2;

// the process-wide binding; nullptr (and `why` filled) when librccl cannot be loaded or lacks a symbol
const RcclApi* rccl_api(std::string* why);

}  // namespace sr
