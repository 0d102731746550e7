// rccl_api.cpp -- the data-parallel gradient all-reduce straight on RCCL, behind the C ABI
// (include/taiyaki_amd_flipflop.h, "multi-GPU").
//
// Replaces the collective of the reference's DistributedDataParallel wrap
// (bin/train_flipflop.py:384-397: the gradient all-reduce inside backward) for callers that do
// not go through torch.distributed: one communicator per process (one process per GPU), one
// ncclAllReduce(SUM) of the flat fp32 gradient arena per optimiser step, enqueued on the HIP
// stream the caller names -- a side stream, so that it overlaps the RNN backward of the earlier
// layers; xGMI is point-to-point, RCCL picks ring / tree per message size.
//
// Built as its OWN shared library (libtaiyaki_amd_rccl.so): a process that already carries an
// RCCL (PyTorch bundles one) must not get a second copy through the kernel library.  The Python
// trainers use torch.distributed's ProcessGroupNCCL -- the same RCCL calls; this library is what
// a C / C++ host (or a Taiyaki build without torch.distributed) would bind.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string.h>

#include "../../include/taiyaki_amd_flipflop.h"

extern "C" {

size_t tk_rccl_unique_id_bytes(void) { return sizeof(ncclUniqueId); }

// rank 0 creates the id and hands its bytes to the other ranks (file, socket, MPI: the caller's
// business -- the reference uses a TCP store at MASTER_ADDR:MASTER_PORT, train_flipflop.py:255-268)
int tk_rccl_unique_id(void *id_out, size_t bytes) {
    if (id_out == nullptr || bytes < sizeof(ncclUniqueId)) return TK_ERR_BAD_ARG;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return TK_ERR_LAUNCH;
    memcpy(id_out, &id, sizeof(id));
    return TK_OK;
}

int tk_rccl_comm_init(void **comm_out, int nranks, const void *id_bytes, int rank) {
    if (comm_out == nullptr || id_bytes == nullptr || nranks < 1 || rank < 0 || rank >= nranks)
        return TK_ERR_BAD_ARG;
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    ncclComm_t comm;
    if (ncclCommInitRank(&comm, nranks, id, rank) != ncclSuccess) return TK_ERR_LAUNCH;
    *comm_out = comm;
    return TK_OK;
}

// SUM over the ranks, in place, on `stream`; the 1 / nranks factor is the caller's (the trainers
// fold it into the clipping pass).  Returns when the collective is ENQUEUED.
int tk_allreduce_f32_dev(void *comm, float *buf, size_t n, void *stream) {
    if (comm == nullptr || buf == nullptr) return TK_ERR_BAD_ARG;
    if (n == 0) return TK_OK;
    const ncclResult_t rc = ncclAllReduce(buf, buf, n, ncclFloat32, ncclSum, static_cast<ncclComm_t>(comm),
                                          static_cast<hipStream_t>(stream));
    return rc == ncclSuccess ? TK_OK : TK_ERR_LAUNCH;
}

// rank `root`'s buffer to every rank (the parameter broadcast that replaces the reference's
// checkpoint-file + barrier handshake, train_flipflop.py:380-392)
int tk_broadcast_f32_dev(void *comm, float *buf, size_t n, int root, void *stream) {
    if (comm == nullptr || buf == nullptr) return TK_ERR_BAD_ARG;
    if (n == 0) return TK_OK;
    const ncclResult_t rc = ncclBroadcast(buf, buf, n, ncclFloat32, root, static_cast<ncclComm_t>(comm),
                                          static_cast<hipStream_t>(stream));
    return rc == ncclSuccess ? TK_OK : TK_ERR_LAUNCH;
}

int tk_rccl_comm_destroy(void *comm) {
    if (comm == nullptr) return TK_OK;
    return ncclCommDestroy(static_cast<ncclComm_t>(comm)) == ncclSuccess ? TK_OK : TK_ERR_LAUNCH;
}

}  // extern "C"
