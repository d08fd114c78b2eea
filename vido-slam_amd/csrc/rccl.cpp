// rccl.cpp — the global BA's all-reduce issued from C++ on the context's stream (RCCL over xGMI), without a host callback.
// The sharded solve (ba.hip) sums two buffers per LM trial over the ranks — the camera diagonal blocks | bc | chi2, and the partial reduced system S | r (band layout:
// 1.6 MB at 500 keyframes) — plus two scalars.  Through the generic hook (vido_allreduce_fn; bench.py binds it to torch.distributed) every call costs a stream
// synchronisation, a Python call and a second synchronisation; here it is one ncclAllReduce enqueued behind the producing kernel.
// RCCL is resolved at run time (dlopen of the librccl the process already has — torch ships one — else the system's): the library has no link-time dependency on it,
// and a single-GPU user never loads it.
#include "common.hpp"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <cstring>

namespace {
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;
int rccl_load()
{
    if (g_rccl.lib) return VIDO_OK;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so"}) if (!h) h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);      // the copy already in the process (torch's)
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) if (!h) h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (!h) return vido_set_error(nullptr, VIDO_E_INVALID, "rccl: librccl not found (%s)", dlerror());
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy) return vido_set_error(nullptr, VIDO_E_INVALID, "rccl: symbols missing in librccl");
    g_rccl.lib = h;
    return VIDO_OK;
}
const char* rccl_err(ncclResult_t r) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "nccl error"; }
}  // namespace

extern "C" {

int vido_rccl_unique_id(uint8_t id_out[128])
{
    if (!id_out) return VIDO_E_INVALID;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    int rc = rccl_load(); if (rc != VIDO_OK) return rc;
    ncclUniqueId id; const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return vido_set_error(nullptr, VIDO_E_HIP, "ncclGetUniqueId: %s", rccl_err(r));
    memcpy(id_out, &id, 128);
    return VIDO_OK;
}

int vido_rccl_init(vido_ctx* ctx, const uint8_t id_in[128], int rank, int world)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!id_in || world < 1 || rank < 0 || rank >= world) return vido_set_error(ctx, VIDO_E_INVALID, "rccl_init: bad arguments");
    int rc = rccl_load(); if (rc != VIDO_OK) return vido_set_error(ctx, rc, "rccl_init: librccl not available");
    if (ctx->rccl_comm) { g_rccl.CommDestroy((ncclComm_t)ctx->rccl_comm); ctx->rccl_comm = nullptr; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id; memcpy(&id, id_in, 128);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = g_rccl.CommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess) return vido_set_error(ctx, VIDO_E_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, world, rccl_err(r));
    ctx->rccl_comm = comm; ctx->rccl_rank = rank; ctx->rccl_world = world;
    return VIDO_OK;
}

// a vido_allreduce_fn: pass it as `allreduce` with user = the vido_ctx* that vido_rccl_init prepared.  In place, on the context's stream, no synchronisation.
int vido_rccl_allreduce(void* user, void* dev_ptr, size_t count, int op)
{
    vido_ctx* ctx = (vido_ctx*)user;
    if (!ctx || !ctx->rccl_comm || !dev_ptr) return 1;
    const ncclResult_t r = g_rccl.AllReduce(dev_ptr, dev_ptr, count, ncclDouble, op == 0 ? ncclSum : ncclMax, (ncclComm_t)ctx->rccl_comm, ctx->stream);
    if (r != ncclSuccess) { vido_set_error(ctx, VIDO_E_HIP, "ncclAllReduce: %s", rccl_err(r)); return 1; }
    return 0;
}

int vido_rccl_destroy(vido_ctx* ctx)
{
    if (!ctx) return VIDO_E_INVALID;
    if (ctx->rccl_comm && g_rccl.CommDestroy) { hipStreamSynchronize(ctx->stream); g_rccl.CommDestroy((ncclComm_t)ctx->rccl_comm); }
    ctx->rccl_comm = nullptr;
    return VIDO_OK;
}

}  // extern "C"
