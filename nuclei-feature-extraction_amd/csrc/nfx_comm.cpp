// nfx_comm.cpp -- the gather of feature rows across GPUs below any host framework (include/nfx.h, "multi-GPU").
//
// Replaces (reference file:line): the host-side row merge `append_features` = vstack_mut under a mutex, main.rs:96-108, fed by
// one rayon worker per batch whose device is `gpus[thread_idx % gpu_count]`, utils.rs:211-224.  Here: one process per GPU,
// every rank extracts a batch-aligned index shard (no data-path collective during extraction) and its [n_local, F] f32 rows
// travel ONCE to the root rank over xGMI: grouped ncclSend / ncclRecv (ragged shards need no padding; a gather is point-to-point
// on this fabric anyway -- every peer has its own link to the root), stream-ordered on the context's stream, device pointers only.
//
// RCCL is reached through dlopen at the first nfx_comm_* call: libnfx.so keeps no link-time dependency on it (a single-GPU host
// never loads it), and a process that already holds an RCCL -- PyTorch ships one -- shares that copy instead of loading a second.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "nfx_internal.h"

namespace {

// the few RCCL types and entry points this file needs, declared locally (rccl.h: NCCL_UNIQUE_ID_BYTES 128, ncclFloat32 = 7)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[NFX_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef int ncclResult_t;
constexpr int kNcclSuccess = 0, kNcclFloat32 = 7;

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why; // set when loading failed
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl()
{
    // NFX_RCCL_PATH, when set, is the ONLY library tried (an explicit choice is not second-guessed); otherwise the usual names.
    const char *forced = getenv("NFX_RCCL_PATH");
    const char *defaults[3] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    std::string last;
    auto try_open = [&](const char *nm) {
        g_rccl.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (!g_rccl.handle) { const char *e = dlerror(); last = e ? e : "not found"; } // dlerror() clears itself: read ONCE
        return g_rccl.handle != nullptr;
    };
    if (forced && *forced) try_open(forced);
    else for (const char *nm : defaults) if (try_open(nm)) break;
    if (!g_rccl.handle) { g_rccl.why = std::string("librccl could not be loaded: ") + (last.empty() ? "not found" : last); return; }
    auto sym = [&](const char *s) { void *p = dlsym(g_rccl.handle, s); if (!p && g_rccl.why.empty()) g_rccl.why = std::string("librccl lacks ") + s; return p; };
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(sym("ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(sym("ncclGroupEnd"));
    g_rccl.Send = reinterpret_cast<decltype(g_rccl.Send)>(sym("ncclSend"));
    g_rccl.Recv = reinterpret_cast<decltype(g_rccl.Recv)>(sym("ncclRecv"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
}

const Rccl *rccl()
{
    std::call_once(g_rccl_once, load_rccl);
    return g_rccl.why.empty() ? &g_rccl : nullptr;
}

nfx_status comm_fail(nfx_ctx *ctx, nfx_status st, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    return st;
}

} // namespace

struct nfx_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

#define NFX_RCCL(ctx, R, expr)                                                                                                  \
    do {                                                                                                                        \
        const ncclResult_t r_ = (expr);                                                                                         \
        if (r_ != kNcclSuccess) return comm_fail(ctx, NFX_ERR_DEVICE, std::string(#expr) + ": " + (R)->GetErrorString(r_));      \
    } while (0)

extern "C" {

nfx_status nfx_comm_unique_id(void *id_out)
{
    if (!id_out) return NFX_ERR_INVALID_ARGUMENT;
    const Rccl *R = rccl();
    if (!R) return NFX_ERR_UNSUPPORTED;
    ncclUniqueId id;
    if (R->GetUniqueId(&id) != kNcclSuccess) return NFX_ERR_DEVICE;
    memcpy(id_out, id.internal, NFX_UNIQUE_ID_BYTES);
    return NFX_OK;
}

nfx_status nfx_comm_create(nfx_ctx *ctx, int rank, int world, const void *unique_id, nfx_comm **out)
{
    if (!ctx || !out) return NFX_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (!unique_id || world < 1 || rank < 0 || rank >= world) return comm_fail(ctx, NFX_ERR_INVALID_ARGUMENT, "nfx_comm_create: bad rank / world / id");
    const Rccl *R = rccl();
    if (!R) return comm_fail(ctx, NFX_ERR_UNSUPPORTED, g_rccl.why);
    if (hipSetDevice(ctx->device) != hipSuccess) return comm_fail(ctx, NFX_ERR_DEVICE, "hipSetDevice");
    ncclUniqueId id;
    memcpy(id.internal, unique_id, NFX_UNIQUE_ID_BYTES);
    nfx_comm *c = new nfx_comm();
    c->rank = rank; c->world = world; c->device = ctx->device;
    const ncclResult_t r = R->CommInitRank(&c->comm, world, id, rank); // blocks until all `world` ranks have called it
    if (r != kNcclSuccess) {
        delete c;
        return comm_fail(ctx, NFX_ERR_DEVICE, std::string("ncclCommInitRank: ") + R->GetErrorString(r));
    }
    *out = c;
    return NFX_OK;
}

void nfx_comm_destroy(nfx_comm *comm)
{
    if (!comm) return;
    const Rccl *R = rccl();
    if (R && comm->comm) {
        (void)hipSetDevice(comm->device);
        (void)R->CommDestroy(comm->comm);
    }
    delete comm;
}

int nfx_comm_rank(const nfx_comm *comm) { return comm ? comm->rank : -1; }
int nfx_comm_world(const nfx_comm *comm) { return comm ? comm->world : -1; }

nfx_status nfx_gather_rows(nfx_ctx *ctx, nfx_comm *comm, const float *rows, int F, const int64_t *counts, float *dst, int root)
{
    if (!ctx || !comm) return NFX_ERR_INVALID_ARGUMENT;
    if (!counts || F < 1 || root < 0 || root >= comm->world) return comm_fail(ctx, NFX_ERR_INVALID_ARGUMENT, "nfx_gather_rows: bad counts / F / root");
    for (int r = 0; r < comm->world; ++r)
        if (counts[r] < 0) return comm_fail(ctx, NFX_ERR_INVALID_ARGUMENT, "nfx_gather_rows: negative count");
    const int64_t mine = counts[comm->rank];
    if (mine > 0 && !rows) return comm_fail(ctx, NFX_ERR_INVALID_ARGUMENT, "nfx_gather_rows: null rows");
    int64_t total = 0;
    for (int r = 0; r < comm->world; ++r) total += counts[r];
    if (comm->rank == root && !dst && total > 0) return comm_fail(ctx, NFX_ERR_INVALID_ARGUMENT, "nfx_gather_rows: null dst on the root");
    if (comm->device != ctx->device) return comm_fail(ctx, NFX_ERR_INVALID_ARGUMENT, "nfx_gather_rows: communicator and context live on different devices");
    const Rccl *R = rccl();
    if (!R) return comm_fail(ctx, NFX_ERR_UNSUPPORTED, g_rccl.why);
    if (hipSetDevice(ctx->device) != hipSuccess) return comm_fail(ctx, NFX_ERR_DEVICE, "hipSetDevice");
    // one group: the root posts a receive per non-empty shard (rows land in rank = index order), every rank posts its send.
    // The root's own shard takes the same path (RCCL turns a self send/recv into a device copy), so one code path serves
    // every rank and every world size.
    NFX_RCCL(ctx, R, R->GroupStart());
    ncclResult_t first_bad = kNcclSuccess;
    if (comm->rank == root) {
        int64_t row0 = 0;
        for (int r = 0; r < comm->world; ++r) {
            if (counts[r] > 0) {
                const ncclResult_t e = R->Recv(dst + row0 * F, (size_t)counts[r] * F, kNcclFloat32, r, comm->comm, ctx->stream);
                if (e != kNcclSuccess && first_bad == kNcclSuccess) first_bad = e;
            }
            row0 += counts[r];
        }
    }
    if (mine > 0) {
        const ncclResult_t e = R->Send(rows, (size_t)mine * F, kNcclFloat32, root, comm->comm, ctx->stream);
        if (e != kNcclSuccess && first_bad == kNcclSuccess) first_bad = e;
    }
    const ncclResult_t ge = R->GroupEnd(); // the group is always closed, whatever was posted inside it
    if (first_bad != kNcclSuccess) return comm_fail(ctx, NFX_ERR_DEVICE, std::string("ncclSend/ncclRecv: ") + R->GetErrorString(first_bad));
    if (ge != kNcclSuccess) return comm_fail(ctx, NFX_ERR_DEVICE, std::string("ncclGroupEnd: ") + R->GetErrorString(ge));
    return NFX_OK;
}

} // extern "C"
