// Multi-GPU plumbing: one process per GPU, views shard across ranks (SimulateMultiViewDataset.java:567
// iterates independent views).  The only collective on the path is the broadcast of the
// ground-truth volume; RCCL runs it over xGMI.
#include "common.h"

#include <rccl/rccl.h>

#include <dlfcn.h>
#include <unistd.h>

#include <algorithm>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

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

// Which RCCL this library is bound to.  libmvsim.so names librccl.so.1 as a dependency; in a process that has already
// loaded another copy under that name (PyTorch ships its own in torch/lib) the dynamic loader binds to THAT copy, so the
// path and version are reported instead of assumed (bench.py prints them beside torch's own).
int mvsim_comm_library_info(char* path, size_t path_capacity, int* version)
{
    if (version) {
        int v = 0;
        MVSIM_NCCL(ncclGetVersion(&v));
        *version = v;
    }
    if (path && path_capacity > 0) {
        path[0] = 0;
        Dl_info info;
        if (dladdr(reinterpret_cast<const void*>(&ncclGetVersion), &info) && info.dli_fname)
            std::snprintf(path, path_capacity, "%s", info.dli_fname);
    }
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

// ---- broadcast=peer_copy: the same scatter + all-gather as copy-engine transfers ---------------------------------------
// RCCL's collectives are kernels: they take CUs from the views that run beside the broadcast.  This form moves the
// chunks with hipMemcpyAsync between the ranks' buffers -- mapped into each other's processes through IPC handles -- so that
// the SDMA engines carry them; RCCL is left with three 16-byte all-reduces per broadcast that act as the barriers between the
// phases (a rank's all-reduce is enqueued behind its copies, so its completion anywhere implies that every rank's copies of
// the phase before have landed).
// The peers' mappings belong to a REGISTRATION (mvsim_comm_register_volume): an explicit collective every rank calls for the
// buffer it will broadcast into, never a cache looked up by address -- an address that comes back from the allocator on one rank
// and not on another must not send one rank into a collective the others skip, and a mapping must not outlive the buffer it was
// made for (mvsim_comm_unregister_volume, mvsim_dev_free of the buffer, mvsim_comm_destroy drop it).
struct PeerMap {
    const char*        local = nullptr;      // the registered range of this rank's buffer
    size_t             bytes = 0;
    std::vector<char*> peer;                 // [rank] -> that rank's buffer in this process's address space (null for me)
    std::vector<void*> opened;               // what hipIpcCloseMemHandle must be called on
};
struct PeerCopyState {
    std::vector<PeerMap>     maps;
    std::vector<hipStream_t> streams;        // one per peer: copies to different peers run on different engines / links
    std::vector<hipEvent_t>  done;
    hipEvent_t               fork = nullptr;
    float*                   flag = nullptr; // 16 bytes for the barrier all-reduces
    void*                    xch = nullptr;  // exchange buffer for the handles: nranks * 128 bytes
    ~PeerCopyState()
    {
        for (auto& m : maps)
            for (void* o : m.opened) (void)hipIpcCloseMemHandle(o);
        for (hipStream_t s : streams) if (s) (void)hipStreamDestroy(s);
        for (hipEvent_t e : done) if (e) (void)hipEventDestroy(e);
        if (fork) (void)hipEventDestroy(fork);
        if (flag) (void)hipFree(flag);
        if (xch) (void)hipFree(xch);
    }
};

static void peer_copy_release(mvsim_ctx* ctx)
{
    delete reinterpret_cast<PeerCopyState*>(ctx->peer_copy);
    ctx->peer_copy = nullptr;
}

// the state is published to the context only once every stream, event and buffer exists (a failure half way leaves nothing behind)
static int peer_copy_state(mvsim_ctx* ctx, PeerCopyState** out)
{
    if (!ctx->peer_copy) {
        std::unique_ptr<PeerCopyState> st(new (std::nothrow) PeerCopyState());
        if (!st) { mvsim::set_error("out of host memory"); return MVSIM_ENOMEM; }
        st->streams.assign((size_t)ctx->nranks, nullptr);
        st->done.assign((size_t)ctx->nranks, nullptr);
        for (int r = 0; r < ctx->nranks; ++r) {
            if (r == ctx->rank) continue;
            MVSIM_HIP(hipStreamCreateWithFlags(&st->streams[(size_t)r], hipStreamNonBlocking));
            MVSIM_HIP(hipEventCreateWithFlags(&st->done[(size_t)r], hipEventDisableTiming));
        }
        MVSIM_HIP(hipEventCreateWithFlags(&st->fork, hipEventDisableTiming));
        MVSIM_HIP(hipMalloc(reinterpret_cast<void**>(&st->flag), 16));
        MVSIM_HIP(hipMemset(st->flag, 0, 16));
        MVSIM_HIP(hipMalloc(&st->xch, (size_t)ctx->nranks * 128));
        ctx->peer_copy = st.release();
    }
    *out = reinterpret_cast<PeerCopyState*>(ctx->peer_copy);
    return MVSIM_OK;
}

static void peer_map_close(PeerMap& m)
{
    for (void* o : m.opened) (void)hipIpcCloseMemHandle(o);
    m.opened.clear();
    m.peer.clear();
}

// local: forget every registration that meets [p, p + bytes) (bytes == 0: that starts at p or contains it)
static void peer_copy_forget(mvsim_ctx* ctx, const void* p, size_t bytes)
{
    PeerCopyState* st = reinterpret_cast<PeerCopyState*>(ctx->peer_copy);
    if (!st || !p) return;
    const char* lo = reinterpret_cast<const char*>(p);
    const char* hi = lo + (bytes ? bytes : 1);
    for (size_t i = 0; i < st->maps.size();) {
        PeerMap& m = st->maps[i];
        if (lo < m.local + m.bytes && m.local < hi) { peer_map_close(m); st->maps.erase(st->maps.begin() + (long)i); }
        else ++i;
    }
}

// every rank's view of `vol` (same call on every rank, collective): IPC handle of the allocation + offset into it, exchanged
// through the communicator itself
static int peer_copy_register(mvsim_ctx* ctx, PeerCopyState* st, float* vol, size_t bytes)
{
    peer_copy_forget(ctx, vol, bytes);                       // a registration is replaced, never reused
    struct Record { hipIpcMemHandle_t h; unsigned long long offset; int pid; char pad[128 - sizeof(hipIpcMemHandle_t) - 12]; };
    static_assert(sizeof(Record) == 128, "exchange record");
    Record mine;
    std::memset(&mine, 0, sizeof(mine));
    void* base = nullptr;
    size_t size = 0;
    MVSIM_HIP(hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t*>(&base), &size, vol));
    mine.offset = (unsigned long long)(reinterpret_cast<char*>(vol) - reinterpret_cast<char*>(base));
    MVSIM_CHECK_ARG(mine.offset + bytes <= size, "register_volume: the range leaves its allocation");
    mine.pid = (int)getpid();
    if (ctx->nranks > 1) MVSIM_HIP(hipIpcGetMemHandle(&mine.h, base));
    std::vector<Record> all((size_t)ctx->nranks);
    char* slot = reinterpret_cast<char*>(st->xch) + (size_t)ctx->rank * sizeof(Record);
    MVSIM_HIP(hipMemcpyAsync(slot, &mine, sizeof(mine), hipMemcpyHostToDevice, ctx->stream));
    MVSIM_NCCL(ncclAllGather(slot, st->xch, sizeof(Record), ncclChar, (ncclComm_t)ctx->comm, ctx->stream));
    MVSIM_HIP(hipMemcpyAsync(all.data(), st->xch, all.size() * sizeof(Record), hipMemcpyDeviceToHost, ctx->stream));
    MVSIM_HIP(hipStreamSynchronize(ctx->stream));
    PeerMap m;
    m.local = reinterpret_cast<const char*>(vol);
    m.bytes = bytes;
    m.peer.assign((size_t)ctx->nranks, nullptr);
    for (int r = 0; r < ctx->nranks; ++r) {
        if (r == ctx->rank) continue;
        if (all[(size_t)r].pid == mine.pid) {
            peer_map_close(m);
            mvsim::set_error("broadcast=peer_copy needs one process per rank (ranks %d and %d share a process: use the RCCL forms)", ctx->rank, r);
            return MVSIM_EINVAL;
        }
        void* p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, all[(size_t)r].h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            peer_map_close(m);
            mvsim::set_error("hipIpcOpenMemHandle (rank %d's buffer) failed: %s", r, hipGetErrorString(e));
            return MVSIM_EHIP;
        }
        m.opened.push_back(p);
        m.peer[(size_t)r] = reinterpret_cast<char*>(p) + all[(size_t)r].offset;
    }
    st->maps.push_back(std::move(m));
    return MVSIM_OK;
}

static PeerMap* peer_copy_find(PeerCopyState* st, const float* vol, size_t bytes)
{
    const char* p = reinterpret_cast<const char*>(vol);
    for (auto& m : st->maps)
        if (m.local == p && bytes <= m.bytes) return &m;
    return nullptr;
}

static int peer_copy_barrier(mvsim_ctx* ctx, PeerCopyState* st)
{
    MVSIM_NCCL(ncclAllReduce(st->flag, st->flag, 4, ncclFloat, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
    return MVSIM_OK;
}

// phase: the copies this rank issues (from = first float, n floats, to every rank in `to`), forked from and joined to ctx->stream
static int peer_copy_push(mvsim_ctx* ctx, PeerCopyState* st, const PeerMap& m, const float* vol, int64_t from, int64_t n, int skip_rank)
{
    if (n <= 0) return MVSIM_OK;
    MVSIM_HIP(hipEventRecord(st->fork, ctx->stream));
    for (int r = 0; r < ctx->nranks; ++r) {
        if (r == ctx->rank || r == skip_rank) continue;
        hipStream_t s = st->streams[(size_t)r];
        MVSIM_HIP(hipStreamWaitEvent(s, st->fork, 0));
        MVSIM_HIP(hipMemcpyAsync(m.peer[(size_t)r] + (size_t)from * sizeof(float), vol + from, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s));
        MVSIM_HIP(hipEventRecord(st->done[(size_t)r], s));
        MVSIM_HIP(hipStreamWaitEvent(ctx->stream, st->done[(size_t)r], 0));
    }
    return MVSIM_OK;
}

static int bcast_peer_copy(mvsim_ctx* ctx, float* vol, int64_t count, int root)
{
    PeerCopyState* st = nullptr;
    MVSIM_TRY(peer_copy_state(ctx, &st));
    PeerMap* m = peer_copy_find(st, vol, (size_t)count * sizeof(float));
    if (!m) {
        mvsim::set_error("broadcast=peer_copy: the volume is not registered on rank %d (mvsim_comm_register_volume, on every rank, first)", ctx->rank);
        return MVSIM_EINVAL;
    }
    const int n = ctx->nranks, me = ctx->rank;
    const int64_t chunk = (count / n) & ~(int64_t)15;
    const int64_t tail = count - chunk * n;
    // every rank's previous readers of `vol` are ordered on its own stream; nobody may be written into before all are there
    MVSIM_TRY(peer_copy_barrier(ctx, st));
    if (me == root) {
        // scatter: chunk r to rank r (one copy per link); the unaligned tail to everybody
        MVSIM_HIP(hipEventRecord(st->fork, ctx->stream));
        for (int r = 0; r < n; ++r) {
            if (r == me) continue;
            hipStream_t s = st->streams[(size_t)r];
            MVSIM_HIP(hipStreamWaitEvent(s, st->fork, 0));
            if (chunk > 0)
                MVSIM_HIP(hipMemcpyAsync(m->peer[(size_t)r] + (size_t)(r * chunk) * sizeof(float), vol + r * chunk, (size_t)chunk * sizeof(float),
                                         hipMemcpyDeviceToDevice, s));
            if (tail > 0)
                MVSIM_HIP(hipMemcpyAsync(m->peer[(size_t)r] + (size_t)(chunk * n) * sizeof(float), vol + chunk * n, (size_t)tail * sizeof(float),
                                         hipMemcpyDeviceToDevice, s));
            MVSIM_HIP(hipEventRecord(st->done[(size_t)r], s));
            MVSIM_HIP(hipStreamWaitEvent(ctx->stream, st->done[(size_t)r], 0));
        }
    }
    MVSIM_TRY(peer_copy_barrier(ctx, st));                   // every rank holds its own chunk
    // all-gather: my chunk to every rank but the root (which has it all); the root's own chunk goes out here as well
    MVSIM_TRY(peer_copy_push(ctx, st, *m, vol, (int64_t)me * chunk, chunk, root));
    MVSIM_TRY(peer_copy_barrier(ctx, st));                   // every chunk has landed everywhere
    return MVSIM_OK;
}

// ---- broadcast=pipelined -------------------------------------------------------------------------------------------
// The schedule (mvsim_comm_broadcast_plan, include/mvsim.h).  Peers = the ranks other than the root, peer index j = (rank - root - 1) mod N.
// The volume is cut into N - 1 chunks of whole 64-byte units (chunk j belongs to peer j), every chunk into `pieces` pieces.
//   stage s < pieces:   root -> peer j: piece s of chunk j                       (the root's N - 1 outbound links)
//   stage s >= 1:       peer j -> every other peer: piece s - 1 of chunk j       (the peer <-> peer links; received in stage s - 1)
//   last stage + 1:     the floats that do not divide into aligned chunks, root -> every peer
// A rank issues the ops of one stage inside ONE ncclGroupStart / ncclGroupEnd, so that the root's sends of piece s and the peers' exchange of
// piece s - 1 run at the same time; stages follow each other in stream order, which is what orders a piece's arrival before its forwarding.
static void bcast_plan(int n, int rank, int root, int64_t count, int pieces, std::vector<mvsim_bcast_op>& ops)
{
    ops.clear();
    if (n < 2 || count <= 0) return;
    const int np = n - 1;
    const int64_t chunk = (count / np) & ~(int64_t)15;
    pieces = chunk == 0 ? 0 : std::max(1, std::min<int>(pieces, (int)std::min<int64_t>(chunk / 16, 64)));
    auto piece_first = [&](int s) { return ((chunk / 16) * s / pieces) * 16; };     // whole 64-byte units per piece
    auto peer_rank = [&](int j) { return (root + 1 + j) % n; };
    const bool is_root = rank == root;
    const int me = is_root ? -1 : ((rank - root - 1) % n + n) % n;
    for (int s = 0; s <= pieces && pieces > 0; ++s) {
        if (s < pieces) {
            const int64_t f = piece_first(s), len = piece_first(s + 1) - f;
            if (is_root) { for (int j = 0; j < np; ++j) ops.push_back(mvsim_bcast_op{s, 0, peer_rank(j), 0, (int64_t)j * chunk + f, len}); }
            else ops.push_back(mvsim_bcast_op{s, 1, root, 0, (int64_t)me * chunk + f, len});
        }
        if (s >= 1 && !is_root) {
            const int64_t f = piece_first(s - 1), len = piece_first(s) - f;
            for (int j = 0; j < np; ++j) {
                if (j == me) continue;
                ops.push_back(mvsim_bcast_op{s, 0, peer_rank(j), 0, (int64_t)me * chunk + f, len});
                ops.push_back(mvsim_bcast_op{s, 1, peer_rank(j), 0, (int64_t)j * chunk + f, len});
            }
        }
    }
    const int64_t tail = count - chunk * np;
    if (tail > 0) {
        const int s = pieces > 0 ? pieces + 1 : 0;
        if (is_root) { for (int j = 0; j < np; ++j) ops.push_back(mvsim_bcast_op{s, 0, peer_rank(j), 0, chunk * np, tail}); }
        else ops.push_back(mvsim_bcast_op{s, 1, root, 0, chunk * np, tail});
    }
}

int mvsim_comm_broadcast_plan(int nranks, int rank, int root, int64_t count, int pieces, mvsim_bcast_op* ops, int capacity, int* n_ops)
{
    MVSIM_CHECK_ARG(nranks >= 1 && rank >= 0 && rank < nranks && root >= 0 && root < nranks && count >= 0 && pieces >= 1 && n_ops,
                    "broadcast_plan: bad rank / root / count / pieces or null n_ops");
    std::vector<mvsim_bcast_op> v;
    bcast_plan(nranks, rank, root, count, pieces, v);
    *n_ops = (int)v.size();
    if (!ops) return MVSIM_OK;
    MVSIM_CHECK_ARG(capacity >= (int)v.size(), "broadcast_plan: capacity too small");
    std::memcpy(ops, v.data(), v.size() * sizeof(mvsim_bcast_op));
    return MVSIM_OK;
}

constexpr int BCAST_PIECES = 8;

static int bcast_pipelined(mvsim_ctx* ctx, float* vol, int64_t count, int root)
{
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    std::vector<mvsim_bcast_op> ops;
    bcast_plan(ctx->nranks, ctx->rank, root, count, BCAST_PIECES, ops);
    size_t i = 0;
    while (i < ops.size()) {
        const int stage = ops[i].stage;
        ncclResult_t bad = ncclSuccess;
        MVSIM_NCCL(ncclGroupStart());
        for (; i < ops.size() && ops[i].stage == stage && bad == ncclSuccess; ++i) {
            const mvsim_bcast_op& o = ops[i];
            bad = o.kind == 0 ? ncclSend(vol + o.first, (size_t)o.count, ncclFloat, o.peer, comm, ctx->stream)
                              : ncclRecv(vol + o.first, (size_t)o.count, ncclFloat, o.peer, comm, ctx->stream);
        }
        const ncclResult_t end = ncclGroupEnd();
        if (bad != ncclSuccess || end != ncclSuccess) {
            mvsim::set_error("pipelined broadcast, stage %d: %s", stage, ncclGetErrorString(bad != ncclSuccess ? bad : end));
            return MVSIM_ERCCL;
        }
    }
    return MVSIM_OK;
}

// Broadcast of the ground truth.  xGMI is point-to-point (7 links per GPU): a ring/chain broadcast moves the whole
// volume over ONE link of every GPU (0.54 GB at 512^3: ~3.5 ms, longer than a 2.4 ms view), so the default form is
//   phase 0  scatter: root sends chunk r (count / nranks floats) to rank r -- nranks-1 concurrent sends, one per link
//   phase 1  all-gather: every rank hands its chunk to all others (ncclAllGather, in place)
//   phase 2  the few floats that do not divide into aligned chunks ride in a tiny ncclBroadcast
// which keeps all links busy in both big steps (~2 * S / (nranks * link rate)).  Option broadcast=ring selects one
// ncclBroadcast for A/B runs.  The phases are separate so that a single process driving several devices
// (mvsim_group_*) can wrap each of them in one ncclGroupStart/End over all its ranks.
static int bcast_phase(mvsim_ctx* ctx, float* vol, int64_t count, int root, int phase)
{
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    const int n = ctx->nranks, me = ctx->rank;
    const int64_t chunk = (count / n) & ~(int64_t)15;            // whole 64-byte units keep every transfer aligned
    const bool ring = ctx->opt.bcast_ring || chunk == 0;
    if (phase == 0) {
        if (ring) {
            MVSIM_NCCL(ncclBroadcast(vol, vol, (size_t)count, ncclFloat, root, comm, ctx->stream));
            return MVSIM_OK;
        }
        ncclResult_t bad = ncclSuccess;
        MVSIM_NCCL(ncclGroupStart());
        if (me == root) {
            for (int r = 0; r < n && bad == ncclSuccess; ++r)
                if (r != root) bad = ncclSend(vol + (int64_t)r * chunk, (size_t)chunk, ncclFloat, r, comm, ctx->stream);
        } else {
            bad = ncclRecv(vol + (int64_t)me * chunk, (size_t)chunk, ncclFloat, root, comm, ctx->stream);
        }
        const ncclResult_t end = ncclGroupEnd();
        if (bad != ncclSuccess || end != ncclSuccess) {
            mvsim::set_error("scatter of the ground truth failed: %s", ncclGetErrorString(bad != ncclSuccess ? bad : end));
            return MVSIM_ERCCL;
        }
        return MVSIM_OK;
    }
    if (ring) return MVSIM_OK;
    if (phase == 1) {
        MVSIM_NCCL(ncclAllGather(vol + (int64_t)me * chunk, vol, (size_t)chunk, ncclFloat, comm, ctx->stream));
        return MVSIM_OK;
    }
    const int64_t tail = count - chunk * n;
    if (tail > 0) MVSIM_NCCL(ncclBroadcast(vol + chunk * n, vol + chunk * n, (size_t)tail, ncclFloat, root, comm, ctx->stream));
    return MVSIM_OK;
}

int mvsim_comm_broadcast_volume(mvsim_ctx* ctx, float* vol_dev, int64_t count, int root)
{
    MVSIM_CHECK_ARG(ctx != nullptr && vol_dev != nullptr && count >= 0, "null pointer or negative count");
    MVSIM_CHECK_ARG(ctx->comm != nullptr, "communicator not initialised");
    MVSIM_CHECK_ARG(root >= 0 && root < ctx->nranks, "root out of range");
    MVSIM_HIP(hipSetDevice(ctx->device));
    MVSIM_TRY(mvsim::join_tail(ctx));
    if (count == 0) return MVSIM_OK;
    if (ctx->opt.bcast_peer_copy) return bcast_peer_copy(ctx, vol_dev, count, root);
    if (ctx->opt.bcast_pipelined) return bcast_pipelined(ctx, vol_dev, count, root);
    for (int phase = 0; phase < 3; ++phase) MVSIM_TRY(bcast_phase(ctx, vol_dev, count, root, phase));
    return MVSIM_OK;
}

int mvsim_comm_register_volume(mvsim_ctx* ctx, float* vol_dev, int64_t count)
{
    MVSIM_CHECK_ARG(ctx != nullptr && vol_dev != nullptr && count >= 1, "null pointer or empty volume");
    MVSIM_CHECK_ARG(ctx->comm != nullptr, "communicator not initialised");
    MVSIM_HIP(hipSetDevice(ctx->device));
    MVSIM_TRY(mvsim::join_tail(ctx));
    PeerCopyState* st = nullptr;
    MVSIM_TRY(peer_copy_state(ctx, &st));
    return peer_copy_register(ctx, st, vol_dev, (size_t)count * sizeof(float));
}

int mvsim_comm_unregister_volume(mvsim_ctx* ctx, const float* vol_dev)
{
    MVSIM_CHECK_ARG(ctx != nullptr && vol_dev != nullptr, "null pointer");
    (void)hipSetDevice(ctx->device);
    peer_copy_forget(ctx, vol_dev, 0);
    return MVSIM_OK;
}

int mvsim_comm_allreduce_sum(mvsim_ctx* ctx, float* buf_dev, int64_t count)
{
    MVSIM_CHECK_ARG(ctx != nullptr && buf_dev != nullptr && count >= 0, "null pointer or negative count");
    MVSIM_CHECK_ARG(ctx->comm != nullptr, "communicator not initialised");
    MVSIM_HIP(hipSetDevice(ctx->device));
    MVSIM_TRY(mvsim::join_tail(ctx));
    MVSIM_NCCL(ncclAllReduce(buf_dev, buf_dev, (size_t)count, ncclFloat, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
    return MVSIM_OK;
}

int mvsim_comm_allreduce_sum_f64(mvsim_ctx* ctx, double* value_host)
{
    MVSIM_CHECK_ARG(ctx != nullptr && value_host != nullptr, "null pointer");
    MVSIM_CHECK_ARG(ctx->comm != nullptr, "communicator not initialised");
    MVSIM_HIP(hipSetDevice(ctx->device));
    MVSIM_TRY(mvsim::join_tail(ctx));
    MVSIM_TRY(ctx->partials.reserve(mvsim::PARTIALS_BYTES));
    double* slot = ctx->partials.as<double>() + mvsim::SUM_BLOCKS + mvsim::SCAL_DOUBLES + 4;   // scratch behind the [sum, corr] pairs
    MVSIM_HIP(hipMemcpyAsync(slot, value_host, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    MVSIM_NCCL(ncclAllReduce(slot, slot, 1, ncclDouble, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
    MVSIM_HIP(hipMemcpyAsync(value_host, slot, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MVSIM_HIP(hipStreamSynchronize(ctx->stream));
    return MVSIM_OK;
}

int mvsim_comm_allreduce_sum_f64_dev(mvsim_ctx* ctx, double* value_dev, void* hip_stream)
{
    MVSIM_CHECK_ARG(ctx != nullptr && value_dev != nullptr, "null pointer");
    MVSIM_CHECK_ARG(ctx->comm != nullptr, "communicator not initialised");
    MVSIM_HIP(hipSetDevice(ctx->device));
    return mvsim::comm_allreduce_f64_on_stream(ctx, value_dev, hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->stream);
}

int mvsim_comm_destroy(mvsim_ctx* ctx)
{
    if (!ctx || !ctx->comm) return MVSIM_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    peer_copy_release(ctx);
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

// ---- one process driving several GPUs (a JVM is ONE process: SimulateMultiViewDataset.main's view loop, :567-613,
//      fans out over the devices of the node from a single host thread) ---------------------------------------------
struct mvsim_group {
    std::vector<mvsim_ctx*> ctx;
    std::vector<mvsim::DevBuf> gt, acq;
    int64_t dim[3] = {0, 0, 0};
    bool have_gt = false;
};

int mvsim_group_create(int ndev, const int* devices, mvsim_group** out)
{
    MVSIM_CHECK_ARG(out != nullptr, "group out pointer is null");
    *out = nullptr;
    MVSIM_CHECK_ARG(ndev >= 1 && ndev <= 64, "ndev must be in 1..64");
    mvsim_group* g = new (std::nothrow) mvsim_group();
    if (!g) { mvsim::set_error("out of host memory"); return MVSIM_ENOMEM; }
    std::vector<int> devs((size_t)ndev);
    int rc = MVSIM_OK;
    for (int i = 0; i < ndev && rc == MVSIM_OK; ++i) {
        devs[i] = devices ? devices[i] : i;
        for (int j = 0; j < i; ++j)
            if (devs[j] == devs[i]) { mvsim::set_error("invalid argument: device %d listed twice", devs[i]); rc = MVSIM_EINVAL; }
        mvsim_ctx* c = nullptr;
        if (rc == MVSIM_OK) rc = mvsim_create(devs[i], &c);
        if (rc == MVSIM_OK) g->ctx.push_back(c);
    }
    if (rc == MVSIM_OK) {
        std::vector<ncclComm_t> comms((size_t)ndev, nullptr);
        const ncclResult_t r = ncclCommInitAll(comms.data(), ndev, devs.data());
        if (r != ncclSuccess) { mvsim::set_error("ncclCommInitAll failed: %s", ncclGetErrorString(r)); rc = MVSIM_ERCCL; }
        for (int i = 0; i < ndev && rc == MVSIM_OK; ++i) { g->ctx[i]->comm = comms[i]; g->ctx[i]->nranks = ndev; g->ctx[i]->rank = i; }
    }
    if (rc != MVSIM_OK) { mvsim_group_destroy(g); return rc; }
    g->gt.resize((size_t)ndev);
    g->acq.resize((size_t)ndev);
    *out = g;
    return MVSIM_OK;
}

int mvsim_group_destroy(mvsim_group* g)
{
    if (!g) return MVSIM_OK;
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        (void)hipSetDevice(g->ctx[i]->device);
        (void)hipStreamSynchronize(g->ctx[i]->stream);
        if (i < g->gt.size()) g->gt[i].release();
        if (i < g->acq.size()) g->acq[i].release();
        mvsim_destroy(g->ctx[i]);
    }
    delete g;
    return MVSIM_OK;
}

int mvsim_group_size(const mvsim_group* g) { return g ? (int)g->ctx.size() : 0; }

mvsim_ctx* mvsim_group_ctx(mvsim_group* g, int index)
{
    if (!g || index < 0 || index >= (int)g->ctx.size()) { mvsim::set_error("invalid argument: group context index"); return nullptr; }
    return g->ctx[(size_t)index];
}

int mvsim_group_broadcast_volume(mvsim_group* g, const float* gt_host, const int64_t dim[3])
{
    MVSIM_CHECK_ARG(g != nullptr && gt_host != nullptr && dim != nullptr, "null pointer");
    MVSIM_CHECK_ARG(dim[0] >= 1 && dim[1] >= 1 && dim[2] >= 1, "dimensions must be >= 1");
    const int64_t count = dim[0] * dim[1] * dim[2];
    const size_t bytes = (size_t)count * sizeof(float);
    const int n = (int)g->ctx.size();
    for (int i = 0; i < n; ++i) {
        MVSIM_HIP(hipSetDevice(g->ctx[i]->device));
        MVSIM_TRY(g->gt[i].reserve(bytes));
    }
    MVSIM_HIP(hipSetDevice(g->ctx[0]->device));
    MVSIM_HIP(hipMemcpyAsync(g->gt[0].p, gt_host, bytes, hipMemcpyHostToDevice, g->ctx[0]->stream));
    // gt_host belongs to the caller again when this function returns (header: "keeps no reference to the pointers
    // afterwards"): the upload is waited for below, after the collectives behind it have been enqueued
    hipEvent_t uploaded = nullptr;
    if (hipEventCreateWithFlags(&uploaded, hipEventDisableTiming) != hipSuccess || hipEventRecord(uploaded, g->ctx[0]->stream) != hipSuccess) {
        // no event to wait on later: wait for the upload here, so that gt_host is the caller's again on this path too
        if (uploaded) (void)hipEventDestroy(uploaded);
        (void)hipStreamSynchronize(g->ctx[0]->stream);
        mvsim::set_error("hipEventCreate / hipEventRecord failed");
        return MVSIM_EHIP;
    }
    struct EventGuard {
        hipEvent_t e;
        ~EventGuard() { (void)hipEventSynchronize(e); (void)hipEventDestroy(e); }
    } guard{uploaded};
    for (int phase = 0; phase < 3; ++phase) {
        int rc = MVSIM_OK;
        MVSIM_NCCL(ncclGroupStart());
        for (int i = 0; i < n && rc == MVSIM_OK; ++i) {
            if (hipSetDevice(g->ctx[i]->device) != hipSuccess) { mvsim::set_error("hipSetDevice failed"); rc = MVSIM_EHIP; break; }
            rc = bcast_phase(g->ctx[i], g->gt[i].as<float>(), count, 0, phase);
        }
        const ncclResult_t end = ncclGroupEnd();
        if (rc != MVSIM_OK) return rc;
        MVSIM_NCCL(end);
    }
    for (int d = 0; d < 3; ++d) g->dim[d] = dim[d];
    g->have_gt = true;
    return MVSIM_OK;
}

int mvsim_group_simulate_views(mvsim_group* g, float* const* psf_host, const int64_t kdim[3],
                               const mvsim_view_params* params, int n_views, float* const* acq_host)
{
    MVSIM_CHECK_ARG(g != nullptr && psf_host != nullptr && kdim != nullptr && params != nullptr && acq_host != nullptr, "null pointer");
    MVSIM_CHECK_ARG(g->have_gt, "no ground truth: call mvsim_group_broadcast_volume first");
    MVSIM_CHECK_ARG(n_views >= 0, "n_views must be >= 0");
    const int n = (int)g->ctx.size();
    int rc = MVSIM_OK;
    for (int v = 0; v < n_views && rc == MVSIM_OK; ++v) {
        const int i = v % n;                                                  // view v -> device v % ndev
        mvsim_ctx* c = g->ctx[(size_t)i];
        if (!psf_host[v] || !acq_host[v] || params[v].inc < 1) { mvsim::set_error("invalid argument: view %d", v); rc = MVSIM_EINVAL; break; }
        const size_t obytes = (size_t)(g->dim[0] * g->dim[1] * mvsim_extract_nz(g->dim[2], params[v].inc)) * sizeof(float);
        if (hipSetDevice(c->device) != hipSuccess) { mvsim::set_error("hipSetDevice failed"); rc = MVSIM_EHIP; break; }
        rc = g->acq[(size_t)i].reserve(obytes);                               // stream order makes the reuse per device safe
        mvsim_view_outputs o = {nullptr, nullptr, nullptr, g->acq[(size_t)i].as<float>()};
        if (rc == MVSIM_OK) rc = mvsim_simulate_view_dev(c, g->gt[(size_t)i].as<float>(), g->dim, psf_host[v], kdim, &params[v], &o, nullptr);
        if (rc == MVSIM_OK) rc = mvsim::join_tail(c);                     // the download below reads what the tail writes
        if (rc == MVSIM_OK && hipMemcpyAsync(acq_host[v], o.acq, obytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess) {
            mvsim::set_error("download of view %d failed", v);
            rc = MVSIM_EHIP;
        }
    }
    for (int i = 0; i < n; ++i) {
        (void)hipSetDevice(g->ctx[(size_t)i]->device);
        if (hipStreamSynchronize(g->ctx[(size_t)i]->stream) != hipSuccess && rc == MVSIM_OK) { mvsim::set_error("stream synchronise failed"); rc = MVSIM_EHIP; }
    }
    return rc;
}

}  // extern "C"

namespace mvsim {
// the sum of one device-resident double over the ranks, in place, on the caller's stream (mvsim_view_slab_dev: the slab sums of a tiled
// view).  A context without a communicator, or a job of one rank, has nothing to add.
int comm_allreduce_f64_on_stream(mvsim_ctx* cc, double* value_dev, hipStream_t s)
{
    if (!cc || !cc->comm || cc->nranks <= 1) return MVSIM_OK;
    MVSIM_NCCL(ncclAllReduce(value_dev, value_dev, 1, ncclDouble, ncclSum, (ncclComm_t)cc->comm, s));
    return MVSIM_OK;
}

void comm_forget_range(mvsim_ctx* ctx, const void* p, size_t bytes) { peer_copy_forget(ctx, p, bytes); }
}  // namespace mvsim
