// Multi-GPU plumbing: one process per GPU, views shard across ranks (SimulateMultiViewDataset.java:567
// iterates independent views).  The only collective on the path is the broadcast of the
// ground-truth volume; RCCL runs it over xGMI.
#include "common.h"

#include <rccl/rccl.h>

#include <cstring>

using namespace mvsim;

static_assert(sizeof(ncclUniqueId) <= MVSIM_UNIQUE_ID_BYTES, "unique id does not fit");

#define MVSIM_NCCL(expr)                                                                     \
    do {                                                                                     \
        ncclResult_t r_ = (expr);                                                            \
        if (r_ != ncclSuccess) {                                                             \
            mvsim::set_error("%s failed: %s", #expr, ncclGetErrorString(r_));                \
            return MVSIM_ERCCL;                                                              \
        }                                                                                    \
    } while (0)

extern "C" {

int mvsim_comm_unique_id(unsigned char id[MVSIM_UNIQUE_ID_BYTES])
{
    MVSIM_CHECK_ARG(id != nullptr, "id is null");
    ncclUniqueId uid;
    MVSIM_NCCL(ncclGetUniqueId(&uid));
    std::memset(id, 0, MVSIM_UNIQUE_ID_BYTES);
    std::memcpy(id, &uid, sizeof(uid));
    return MVSIM_OK;
}

int mvsim_comm_init(mvsim_ctx* ctx, int nranks, int rank, const unsigned char id[MVSIM_UNIQUE_ID_BYTES])
{
    MVSIM_CHECK_ARG(ctx != nullptr && id != nullptr, "null pointer");
    MVSIM_CHECK_ARG(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank / nranks");
    MVSIM_CHECK_ARG(ctx->comm == nullptr, "communicator already initialised");
    MVSIM_HIP(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    MVSIM_NCCL(ncclCommInitRank(&comm, nranks, uid, rank));
    ctx->comm = comm;
    ctx->nranks = nranks;
    ctx->rank = rank;
    return MVSIM_OK;
}

int mvsim_comm_broadcast_volume(mvsim_ctx* ctx, float* vol_dev, int64_t count, int root)
{
    MVSIM_CHECK_ARG(ctx != nullptr && vol_dev != nullptr && count >= 0, "null pointer or negative count");
    MVSIM_CHECK_ARG(ctx->comm != nullptr, "communicator not initialised");
    MVSIM_CHECK_ARG(root >= 0 && root < ctx->nranks, "root out of range");
    MVSIM_HIP(hipSetDevice(ctx->device));
    MVSIM_NCCL(ncclBroadcast(vol_dev, vol_dev, (size_t)count, ncclFloat, root, (ncclComm_t)ctx->comm, ctx->stream));
    return MVSIM_OK;
}

int mvsim_comm_allreduce_sum(mvsim_ctx* ctx, float* buf_dev, int64_t count)
{
    MVSIM_CHECK_ARG(ctx != nullptr && buf_dev != nullptr && count >= 0, "null pointer or negative count");
    MVSIM_CHECK_ARG(ctx->comm != nullptr, "communicator not initialised");
    MVSIM_HIP(hipSetDevice(ctx->device));
    MVSIM_NCCL(ncclAllReduce(buf_dev, buf_dev, (size_t)count, ncclFloat, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
    return MVSIM_OK;
}

int mvsim_comm_allreduce_sum_f64(mvsim_ctx* ctx, double* value_host)
{
    MVSIM_CHECK_ARG(ctx != nullptr && value_host != nullptr, "null pointer");
    MVSIM_CHECK_ARG(ctx->comm != nullptr, "communicator not initialised");
    MVSIM_HIP(hipSetDevice(ctx->device));
    MVSIM_TRY(ctx->partials.reserve((mvsim::SUM_BLOCKS + 8) * sizeof(double)));
    double* slot = ctx->partials.as<double>() + mvsim::SUM_BLOCKS + 4;        // scratch behind [sum, corr]
    MVSIM_HIP(hipMemcpyAsync(slot, value_host, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    MVSIM_NCCL(ncclAllReduce(slot, slot, 1, ncclDouble, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
    MVSIM_HIP(hipMemcpyAsync(value_host, slot, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MVSIM_HIP(hipStreamSynchronize(ctx->stream));
    return MVSIM_OK;
}

int mvsim_comm_destroy(mvsim_ctx* ctx)
{
    if (!ctx || !ctx->comm) return MVSIM_OK;
    (void)hipSetDevice(ctx->device);
    ncclCommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->nranks = 1;
    ctx->rank = 0;
    return MVSIM_OK;
}

int mvsim_shard_views(int n_views, int nranks, int rank, int* view_idx, int max_out)
{
    if (n_views < 0 || nranks < 1 || rank < 0 || rank >= nranks) {
        mvsim::set_error("invalid argument: shard_views(n_views=%d, nranks=%d, rank=%d)", n_views, nranks, rank);
        return MVSIM_EINVAL;
    }
    int cnt = 0;
    for (int v = rank; v < n_views; v += nranks) {
        if (view_idx && cnt < max_out) view_idx[cnt] = v;
        ++cnt;
    }
    return cnt;
}

}  // extern "C"
