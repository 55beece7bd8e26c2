// gpe_comm.hip -- moving the neighbour segments of a sharded run between the ranks, inside the library
// ("RCCL point-to-point: ncclGroupStart(); ncclSend/ncclRecv per neighbour tile; ncclGroupEnd()").
//
// Host code only.  The exchange kernels (k_shard.hip) leave fixed-size segments in the context's send buffer; one
// grouped send / recv pair per neighbouring rank, enqueued on the context's stream, moves them over xGMI -- no
// host synchronisation, no size negotiation (both ends derive the capacities from the decomposition).  With the
// communicator inside the context, the step loop of a sharded run is gpe_shard_run: a host in any language (the
// reference's is Rust) needs no collective library binding of its own.
//
// RCCL is loaded with dlopen at the first use: libgpe.so itself does not depend on librccl (half a gigabyte that
// a single-GPU host never needs), and a process that already holds RCCL -- torch.distributed's copy has the same
// SONAME -- gets that copy, so there are never two RCCLs in one process.
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include <rccl/rccl.h>

#include "gpe_internal.h"

namespace gpe {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;               // why loading failed
};

static RcclApi *rccl_api()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
        }
        if (!api.handle) {
            const char *e = dlerror();
            api.error = std::string("librccl.so.1 not found (") + (e ? e : "dlopen failed") + ")";
            return;
        }
        struct { const char *name; void **slot; } syms[] = {
            {"ncclGetUniqueId", (void **)&api.GetUniqueId}, {"ncclCommInitRank", (void **)&api.CommInitRank},
            {"ncclCommDestroy", (void **)&api.CommDestroy}, {"ncclGroupStart", (void **)&api.GroupStart},
            {"ncclGroupEnd", (void **)&api.GroupEnd},       {"ncclSend", (void **)&api.Send},
            {"ncclRecv", (void **)&api.Recv},               {"ncclGetErrorString", (void **)&api.GetErrorString},
            {"ncclAllReduce", (void **)&api.AllReduce},
        };
        for (auto &s : syms) {
            *s.slot = dlsym(api.handle, s.name);
            if (!*s.slot) { api.error = std::string("librccl: symbol ") + s.name + " is missing"; return; }
        }
    });
    return &api;
}

static gpe_status rccl_ready(gpe_ctx *c, RcclApi **out)
{
    RcclApi *api = rccl_api();
    if (!api->error.empty()) return fail(c, GPE_ERR_UNSUPPORTED, api->error);
    *out = api;
    return GPE_OK;
}

#define GPE_NCCL(ctx, api, expr)                                                                       \
    do {                                                                                              \
        ncclResult_t _r = (expr);                                                                     \
        if (_r != ncclSuccess)                                                                        \
            return gpe::fail((ctx), GPE_ERR_HIP, std::string(#expr) + ": " + (api)->GetErrorString(_r)); \
    } while (0)

void comm_release(gpe_ctx *c)
{
    ShardState &S = c->shard;
    if (S.comm && S.comm_owned) {
        RcclApi *api = rccl_api();
        if (api->error.empty()) (void)api->CommDestroy((ncclComm_t)S.comm);
    }
    S.comm = nullptr;
    S.comm_owned = false;
    S.transport = nullptr;
    S.transport_user = nullptr;
}

static uint64_t segment_words(uint32_t cap_mig, uint32_t cap_gho)
{
    return 4ull + 6ull * cap_mig + 4ull * cap_gho;      // header, migrant rows, ghost rows (k_shard.hip)
}

// The control plane's two collectives over the communicator (gpe_shard_ctl.hip): ncclAllReduce in place, and the
// all-to-all as one group of ncclSend / ncclRecv pairs (what a rank keeps for itself is a device copy).
gpe_status rccl_all_reduce_u32(gpe_ctx *c, uint32_t *d_buf, uint64_t count, uint32_t op)
{
    RcclApi *api = nullptr;
    GPE_TRY(rccl_ready(c, &api));
    GPE_NCCL(c, api, api->AllReduce(d_buf, d_buf, count, ncclUint32, op == GPE_REDUCE_MAX ? ncclMax : ncclSum,
                                     (ncclComm_t)c->shard.comm, c->stream));
    return GPE_OK;
}

gpe_status rccl_all_to_all_u32(gpe_ctx *c, const uint32_t *d_send, const uint64_t *send_off, const uint64_t *send_cnt,
                               uint32_t *d_recv, const uint64_t *recv_off, const uint64_t *recv_cnt)
{
    RcclApi *api = nullptr;
    GPE_TRY(rccl_ready(c, &api));
    const ncclComm_t comm = (ncclComm_t)c->shard.comm;
    const uint32_t ws = c->ctl.layout.world_size, me = c->ctl.rank;
    if (send_cnt[me] != recv_cnt[me]) return fail(c, GPE_ERR_INVALID_ARG, "all_to_all: the rank's own send and receive counts differ");
    if (send_cnt[me])
        GPE_HIP(c, hipMemcpyAsync(d_recv + recv_off[me], d_send + send_off[me], send_cnt[me] * sizeof(uint32_t),
                                  hipMemcpyDeviceToDevice, c->stream));
    GPE_NCCL(c, api, api->GroupStart());
    ncclResult_t first_bad = ncclSuccess;
    for (uint32_t p = 0; p < ws; ++p) {
        if (p == me) continue;
        ncclResult_t r = ncclSuccess;
        if (send_cnt[p]) r = api->Send(d_send + send_off[p], send_cnt[p], ncclUint32, (int)p, comm, c->stream);
        if (r == ncclSuccess && recv_cnt[p]) r = api->Recv(d_recv + recv_off[p], recv_cnt[p], ncclUint32, (int)p, comm, c->stream);
        if (r != ncclSuccess && first_bad == ncclSuccess) first_bad = r;
    }
    const ncclResult_t end = api->GroupEnd();
    if (first_bad != ncclSuccess) return fail(c, GPE_ERR_HIP, std::string("ncclSend/ncclRecv: ") + api->GetErrorString(first_bad));
    if (end != ncclSuccess) return fail(c, GPE_ERR_HIP, std::string("ncclGroupEnd: ") + api->GetErrorString(end));
    return GPE_OK;
}

}  // namespace gpe

namespace gpe { gpe_status exchange_on(gpe_ctx *c, hipStream_t xs); }
using namespace gpe;

extern "C" {

gpe_status gpe_comm_probe(void)
{
    RcclApi *api = rccl_api();
    if (!api->error.empty()) return fail(nullptr, GPE_ERR_UNSUPPORTED, api->error);
    return GPE_OK;
}

gpe_status gpe_comm_unique_id(uint8_t *id128)
{
    if (!id128) return GPE_ERR_INVALID_ARG;
    RcclApi *api = nullptr;
    GPE_TRY(rccl_ready(nullptr, &api));
    static_assert(sizeof(ncclUniqueId) == GPE_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    GPE_NCCL(nullptr, api, api->GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return GPE_OK;
}

gpe_status gpe_shard_comm_init(gpe_ctx *c, const uint8_t *id128, uint32_t rank, uint32_t world_size)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    if (!id128 || world_size == 0 || rank >= world_size)
        return fail(c, GPE_ERR_INVALID_ARG, "gpe_shard_comm_init: bad argument");
    RcclApi *api = nullptr;
    GPE_TRY(rccl_ready(c, &api));
    GPE_HIP(c, hipSetDevice(c->device));
    comm_release(c);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    GPE_NCCL(c, api, api->CommInitRank(&comm, (int)world_size, id, (int)rank));
    c->shard.comm = comm;
    c->shard.comm_owned = true;
    return GPE_OK;
}

gpe_status gpe_shard_comm_attach(gpe_ctx *c, void *nccl_comm)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    if (!nccl_comm) return fail(c, GPE_ERR_INVALID_ARG, "gpe_shard_comm_attach: NULL communicator");
    RcclApi *api = nullptr;
    GPE_TRY(rccl_ready(c, &api));
    comm_release(c);
    c->shard.comm = nccl_comm;
    c->shard.comm_owned = false;
    return GPE_OK;
}

gpe_status gpe_shard_comm_destroy(gpe_ctx *c)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    if (c->stream) GPE_HIP(c, hipStreamSynchronize(c->stream));
    comm_release(c);
    return GPE_OK;
}

gpe_status gpe_shard_set_transport(gpe_ctx *c, gpe_shard_transport_fn fn, void *user)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    c->shard.transport = fn;
    c->shard.transport_user = fn ? user : nullptr;
    return GPE_OK;
}

gpe_status gpe_shard_exchange(gpe_ctx *c)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    ShardState &S = c->shard;
    if (!S.on) return fail(c, GPE_ERR_STATE, "sharded exchange not configured: call gpe_shard_configure first");
    if (!S.packed) return fail(c, GPE_ERR_STATE, "gpe_shard_exchange: nothing was packed (gpe_shard_begin / gpe_shard_step)");
    GPE_HIP(c, hipSetDevice(c->device));
    Scope s(c, "shard/exchange");
    // The stream the segments move on: the context's, or -- ShardState::overlap -- one of its own that waits for the
    // pack (the tiles along the rank's border, native_collide) and is waited for by the next unpack.
    const hipStream_t xs = S.exchange_stream(c->stream);
    const bool beside = xs != c->stream;
    if (beside && S.packed_recorded) GPE_HIP(c, hipStreamWaitEvent(xs, S.ev_packed, 0));
    gpe_status st = exchange_on(c, xs);
    if (st == GPE_OK && beside) { GPE_HIP(c, hipEventRecord(S.ev_exchanged, xs)); S.exchanged_recorded = true; }
    return st;
}

}  // extern "C"

namespace gpe {

gpe_status exchange_on(gpe_ctx *c, hipStream_t xs)
{
    ShardState &S = c->shard;
    if (S.transport) {
        const int32_t rc = S.transport(S.transport_user, S.send, S.recv, (void *)xs);
        if (rc != 0) return fail(c, GPE_ERR_HIP, "gpe_shard_exchange: the caller's transport failed");
        return GPE_OK;
    }
    if (S.slots.n_slots <= 1) return GPE_OK;                           // no neighbour: nothing to move
    // a plan made by gpe_shard_setup knows the exchange as an all-to-all: the caller's collectives or the local group
    if (c->ctl.ready && S.send == c->ctl.d_send) {
        if (c->ctl.coll_set)
            return coll_all_to_all_u32(c, c->ctl.d_send, c->ctl.x_send_off, c->ctl.x_send_cnt, c->ctl.d_recv, c->ctl.x_recv_off,
                                       c->ctl.x_recv_cnt, xs);
        if (c->ctl.group) return group_exchange_segments(c, xs);
    }
    if (!S.comm)
        return fail(c, GPE_ERR_STATE, "gpe_shard_exchange: no communicator (gpe_shard_comm_init / _attach) and no transport");
    RcclApi *api = nullptr;
    GPE_TRY(rccl_ready(c, &api));
    const ncclComm_t comm = (ncclComm_t)S.comm;
    // slots [0, n_slots - 1) are the neighbouring ranks (ascending); the last slot is this rank's own segment
    GPE_NCCL(c, api, api->GroupStart());
    ncclResult_t first_bad = ncclSuccess;
    for (uint32_t k = 0; k + 1 < S.slots.n_slots; ++k) {
        const int peer = (int)S.slots.rank[k];
        ncclResult_t r = api->Send(S.send + S.slots.send_off[k],
                                   segment_words(S.slots.send_cap_mig[k], S.slots.send_cap_gho[k]), ncclUint32, peer, comm, xs);
        if (r == ncclSuccess)
            r = api->Recv(S.recv + S.slots.recv_off[k],
                          segment_words(S.slots.recv_cap_mig[k], S.slots.recv_cap_gho[k]), ncclUint32, peer, comm, xs);
        if (r != ncclSuccess && first_bad == ncclSuccess) first_bad = r;
    }
    const ncclResult_t end = api->GroupEnd();                      // always close the group
    if (first_bad != ncclSuccess) return fail(c, GPE_ERR_HIP, std::string("ncclSend/ncclRecv: ") + api->GetErrorString(first_bad));
    if (end != ncclSuccess) return fail(c, GPE_ERR_HIP, std::string("ncclGroupEnd: ") + api->GetErrorString(end));
    return GPE_OK;
}

}  // namespace gpe

extern "C" {

gpe_status gpe_shard_run(gpe_ctx *c, float dt, uint64_t steps)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    for (uint64_t s = 0; s < steps; ++s) {
        GPE_TRY(gpe_shard_exchange(c));
        GPE_TRY(gpe_shard_step(c, dt));
    }
    return GPE_OK;
}

}  // extern "C"
