// collectives.cpp -- in-library cross-GPU exchange: RCCL (the ROCm build of NCCL) over xGMI, bound at run time.
//
// One process (one cge_ctx) per GPU.  The exchange steps of the path (vect_C and the landmark-pair
// matrix of the per-edge scatter, the gathers of the sharded runsplit, the bound matrix and the scalar of the diameter) are
// all-reduces of 8-byte words; with a communicator set they are issued by the library itself on the ctx stream -- no host
// synchronisation per call, no callback into the host language, so any host (Julia, C, Python) gets them.  The caller only
// distributes the 128-byte id of rank 0 (MPI, a file, torch.distributed ...).  librccl is opened with dlopen, so the
// library loads (and single-GPU runs work) on a box whose run-time lacks librccl (the BUILD needs <rccl/rccl.h> for the
// prototypes); the hook of cge_set_collectives stays available (it is what the gloo tests on the CPU use).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "common.hpp"

namespace {
struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr; // optional (the sharded ingest of the embedding)
    decltype(&ncclReduceScatter) ReduceScatter = nullptr; // optional (the N x N landmark-pair matrix by row blocks)
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};
RcclApi &rccl() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return api;
    tried = true;
    const char *names[] = {getenv("CGE_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    // A copy the process has already mapped (torch ships its own librccl) is reused: two RCCL instances in one process
    // would each bring their own topology detection and proxy threads.  Otherwise the library is loaded privately
    // (RTLD_LOCAL: its symbols must not be offered to libraries loaded later).
    for (int pass = 0; pass < 2 && !api.lib; pass++)
        for (const char *nm : names) {
            if (!nm || !*nm) continue;
            api.lib = dlopen(nm, pass == 0 ? (RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL) : (RTLD_NOW | RTLD_LOCAL));
            if (api.lib) break;
            if (pass == 1) api.err = dlerror();
        }
    if (!api.lib) return api;
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
    api.AllReduce = (decltype(api.AllReduce))dlsym(api.lib, "ncclAllReduce");
    api.AllGather = (decltype(api.AllGather))dlsym(api.lib, "ncclAllGather");
    api.ReduceScatter = (decltype(api.ReduceScatter))dlsym(api.lib, "ncclReduceScatter");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy) {
        api.err = "librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy";
        dlclose(api.lib);
        api.lib = nullptr;
    }
    return api;
}
const char *rccl_str(ncclResult_t r) {
    RcclApi &a = rccl();
    return a.GetErrorString ? a.GetErrorString(r) : "rccl error";
}
} // namespace

// all-reduce of `count` 8-byte words in place on the ctx stream: op 0 = sum of doubles, 1 = max of doubles, 2 = sum of int64
void cge_rccl_allreduce(cge_ctx *c, void *dev, i64 count, int op) {
    RcclApi &a = rccl();
    if (!a.lib || !c->rccl_comm) CGE_THROW(CGE_E_COLLECTIVE, "no RCCL communicator on this context");
    const ncclDataType_t dt = op == 2 ? ncclInt64 : ncclFloat64;
    const ncclRedOp_t ro = op == 1 ? ncclMax : ncclSum;
    const ncclResult_t r = a.AllReduce(dev, dev, (size_t)count, dt, ro, (ncclComm_t)c->rccl_comm, c->stream);
    if (r != ncclSuccess) CGE_THROW(CGE_E_COLLECTIVE, "ncclAllReduce failed: %s", rccl_str(r));
    c->stat_coll_calls++;
    c->stat_coll_bytes += 8 * count;
}

// all-gather of 8-byte words in place on the ctx stream: rank r contributes buf[r * words_per_rank, (r + 1) * words_per_rank)
// (the in-place form of ncclAllGather: sendbuff = recvbuff + rank * sendcount).  false when librccl has no ncclAllGather
// (the caller then falls back to a zero-filled integer all-reduce).
bool cge_rccl_allgather(cge_ctx *c, void *dev, i64 words_per_rank) {
    RcclApi &a = rccl();
    if (!a.lib || !c->rccl_comm) CGE_THROW(CGE_E_COLLECTIVE, "no RCCL communicator on this context");
    if (!a.AllGather) return false;
    const char *mine = (const char *)dev + (size_t)8 * words_per_rank * c->coll.rank;
    const ncclResult_t r = a.AllGather(mine, dev, (size_t)words_per_rank, ncclInt64, (ncclComm_t)c->rccl_comm, c->stream);
    if (r != ncclSuccess) CGE_THROW(CGE_E_COLLECTIVE, "ncclAllGather failed: %s", rccl_str(r));
    c->stat_coll_calls++;
    c->stat_coll_bytes += 8 * words_per_rank * c->coll.world;
    return true;
}

// reduce-scatter (sum of doubles) in place on the ctx stream: every rank holds world * words_per_rank words; afterwards rank
// r's block [r * words_per_rank, (r + 1) * words_per_rank) holds the sums of that block over the ranks (the in-place form of
// ncclReduceScatter: recvbuff = sendbuff + rank * recvcount), the other blocks are unspecified.  A row-block reduce-scatter
// of the N x N landmark-pair matrix moves (W - 1) / W of it per link instead of the 2 (W - 1) / W of an all-reduce
// (1.9 ms vs 13 ms at N = 12000 on xGMI).  false when librccl has no ncclReduceScatter.
bool cge_rccl_reduce_scatter(cge_ctx *c, void *dev, i64 words_per_rank) {
    RcclApi &a = rccl();
    if (!a.lib || !c->rccl_comm) CGE_THROW(CGE_E_COLLECTIVE, "no RCCL communicator on this context");
    if (!a.ReduceScatter) return false;
    char *mine = (char *)dev + (size_t)8 * words_per_rank * c->coll.rank;
    const ncclResult_t r = a.ReduceScatter(dev, mine, (size_t)words_per_rank, ncclFloat64, ncclSum, (ncclComm_t)c->rccl_comm, c->stream);
    if (r != ncclSuccess) CGE_THROW(CGE_E_COLLECTIVE, "ncclReduceScatter failed: %s", rccl_str(r));
    c->stat_coll_calls++;
    c->stat_coll_bytes += 8 * words_per_rank * c->coll.world;
    return true;
}

extern "C" {

int cge_rccl_unique_id(void *id_out) {
    if (!id_out) return CGE_E_ARG;
    RcclApi &a = rccl();
    if (!a.lib) return CGE_E_COLLECTIVE;
    static_assert(sizeof(ncclUniqueId) == CGE_RCCL_ID_BYTES, "cge_hip.h: CGE_RCCL_ID_BYTES");
    ncclUniqueId id;
    if (a.GetUniqueId(&id) != ncclSuccess) return CGE_E_COLLECTIVE;
    memcpy(id_out, &id, sizeof(id));
    return CGE_OK;
}

int cge_comm_init_rccl(cge_ctx *c, const void *id_in, int rank, int world) {
    if (!c || !id_in || world < 1 || rank < 0 || rank >= world) return CGE_E_ARG;
    try {
        RcclApi &a = rccl();
        if (!a.lib) CGE_THROW(CGE_E_COLLECTIVE, "librccl could not be opened: %s", a.err.c_str());
        HIP_CHECK(hipSetDevice(c->device));
        if (c->rccl_comm) { (void)a.CommDestroy((ncclComm_t)c->rccl_comm); c->rccl_comm = nullptr; }
        ncclUniqueId id;
        memcpy(&id, id_in, sizeof(id));
        ncclComm_t comm = nullptr;
        const ncclResult_t r = a.CommInitRank(&comm, world, id, rank);
        if (r != ncclSuccess) CGE_THROW(CGE_E_COLLECTIVE, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, rccl_str(r));
        c->rccl_comm = comm;
        c->coll.allreduce_f64 = nullptr;
        c->coll.user = nullptr;
        c->coll.rank = rank;
        c->coll.world = world;
        c->has_coll = world > 1; // a one-rank communicator is legal (self test) but shards nothing
    } catch (const CgeError &e) {
        c->err = e.msg;
        return e.code;
    }
    return CGE_OK;
}

int cge_comm_finalize(cge_ctx *c) {
    if (!c) return CGE_E_ARG;
    if (c->rccl_comm) {
        (void)hipStreamSynchronize(c->stream);
        RcclApi &a = rccl();
        if (a.lib) (void)a.CommDestroy((ncclComm_t)c->rccl_comm);
        c->rccl_comm = nullptr;
        c->has_coll = false;
    }
    return CGE_OK;
}

// testing hook (include/cge_hip_testing.h): host array -> device -> in-library all-reduce -> host
int cge_rccl_selftest(void *ctx, double *host_inout, int64_t count, int op) {
    cge_ctx *c = (cge_ctx *)ctx;
    if (!c || !host_inout || count <= 0) return CGE_E_ARG;
    try {
        HIP_CHECK(hipSetDevice(c->device));
        DevBuf<double> d;
        d.ensure((size_t)count);
        HIP_CHECK(hipMemcpyAsync(d.p, host_inout, sizeof(double) * count, hipMemcpyHostToDevice, c->stream));
        cge_rccl_allreduce(c, d.p, count, op);
        HIP_CHECK(hipMemcpyAsync(host_inout, d.p, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
    } catch (const CgeError &e) {
        c->err = e.msg;
        return e.code;
    }
    return CGE_OK;
}

} // extern "C"
