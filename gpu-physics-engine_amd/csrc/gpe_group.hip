// gpe_group.hip -- a LOCAL GROUP: several contexts of one process as the ranks of a sharded run.
//
// The reference's host is one process that owns one device (renderer/wgpu_context.rs:42-49).  On a node with several
// GPUs the smallest step from there is one process, one context per GPU, one host thread per context -- no launcher,
// no collective library.  The group carries what the ranks need from each other with hipMemcpyAsync / small kernels
// between the contexts' own buffers (contexts on one device, or on peer devices):
//   the per-step neighbour exchange   ordered by events, no stream synchronisation: two host rendezvous per step
//   all-reduce / all-to-all           for the control plane (set-up, re-sort, re-cut): rendezvous + stream synchronisation
// Every collective call blocks its thread until all ranks of the group have made it; a rank that fails calls
// gpe_local_group_abort (gpe_destroy does it for a context that is still a member), which wakes the others with an error.
#include <string.h>

#include <condition_variable>
#include <mutex>

#include "gpe_internal.h"

struct gpe_local_group {
    uint32_t ws = 0;
    std::mutex mu;
    std::condition_variable cv;
    uint32_t arrived = 0;
    uint64_t generation = 0;
    bool broken = false;
    uint32_t members = 0;
    // what every rank posts before a rendezvous
    struct Post {
        gpe_ctx *ctx = nullptr;
        const uint32_t *send = nullptr;
        const uint64_t *send_off = nullptr, *send_cnt = nullptr;
        uint32_t *buf = nullptr;
        hipEvent_t packed = nullptr, copied = nullptr;
    } post[GPE_SHARD_MAX_RANKS];
};

namespace gpe {

// all ranks arrive, all leave; false when the group was aborted
static bool rendezvous(gpe_local_group *g)
{
    std::unique_lock<std::mutex> lk(g->mu);
    if (g->broken) return false;
    const uint64_t gen = g->generation;
    if (++g->arrived == g->ws) {
        g->arrived = 0;
        ++g->generation;
        g->cv.notify_all();
        return true;
    }
    g->cv.wait(lk, [&] { return g->generation != gen || g->broken; });
    return !g->broken;
}

struct ReduceSources {
    const uint32_t *src[GPE_SHARD_MAX_RANKS];
    uint32_t n;
};

__global__ __launch_bounds__(kStreamBlock) void k_group_reduce(ReduceSources S, uint32_t *__restrict__ out, uint64_t count,
                                                               uint32_t op)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        uint32_t v = S.src[0][i];
        for (uint32_t r = 1; r < S.n; ++r) {
            const uint32_t w = S.src[r][i];
            v = op == GPE_REDUCE_MAX ? max(v, w) : v + w;
        }
        out[i] = v;
    }
}

static gpe_status group_fail(gpe_ctx *c, const char *what)
{
    return fail(c, GPE_ERR_STATE, std::string("local group: aborted (another rank failed) in ") + what);
}

gpe_status group_all_reduce_u32(gpe_ctx *c, uint32_t *d_buf, uint64_t count, uint32_t op)
{
    gpe_local_group *g = c->ctl.group;
    const uint32_t me = c->ctl.group_rank;
    // every rank reduces all ranks' buffers into a scratch of its own, then -- when nobody reads the originals any
    // more -- copies the result over its buffer
    uint32_t *tmp = nullptr;
    hipError_t e = hipMalloc((void **)&tmp, std::max<uint64_t>(count, 16) * sizeof(uint32_t));
    if (e != hipSuccess) { (void)hipGetLastError(); (void)gpe_local_group_abort(g); return fail(c, GPE_ERR_OOM, "local group: hipMalloc (all-reduce scratch)"); }
    (void)hipStreamSynchronize(c->stream);                             // my contribution is complete
    g->post[me].buf = d_buf;
    if (!rendezvous(g)) { (void)hipFree(tmp); return group_fail(c, "all_reduce"); }
    ReduceSources S;
    S.n = g->ws;
    for (uint32_t r = 0; r < g->ws; ++r) S.src[r] = g->post[r].buf;
    hipLaunchKernelGGL(k_group_reduce, dim3(stream_grid(count)), dim3(kStreamBlock), 0, c->stream, S, tmp, count, op);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { (void)gpe_local_group_abort(g); (void)hipFree(tmp); return fail(c, GPE_ERR_HIP, std::string("local group: all-reduce: ") + hipGetErrorName(e)); }
    if (!rendezvous(g)) { (void)hipFree(tmp); return group_fail(c, "all_reduce"); }
    e = hipMemcpyAsync(d_buf, tmp, count * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) { (void)gpe_local_group_abort(g); return fail(c, GPE_ERR_HIP, std::string("local group: all-reduce: ") + hipGetErrorName(e)); }
    return GPE_OK;
}

gpe_status group_all_to_all_u32(gpe_ctx *c, const uint32_t *d_send, const uint64_t *send_off, const uint64_t *send_cnt,
                                uint32_t *d_recv, const uint64_t *recv_off, const uint64_t *recv_cnt)
{
    gpe_local_group *g = c->ctl.group;
    const uint32_t me = c->ctl.group_rank;
    (void)hipStreamSynchronize(c->stream);                             // what I send is complete
    g->post[me].send = d_send; g->post[me].send_off = send_off; g->post[me].send_cnt = send_cnt;
    if (!rendezvous(g)) return group_fail(c, "all_to_all");
    hipError_t e = hipSuccess;
    bool mismatch = false;
    for (uint32_t p = 0; p < g->ws && e == hipSuccess; ++p) {
        const gpe_local_group::Post &P = g->post[p];
        if (P.send_cnt[me] != recv_cnt[p]) { mismatch = true; break; }
        if (recv_cnt[p] == 0) continue;
        e = hipMemcpyAsync(d_recv + recv_off[p], P.send + P.send_off[me], recv_cnt[p] * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (mismatch || e != hipSuccess) {
        (void)gpe_local_group_abort(g);
        return fail(c, mismatch ? GPE_ERR_INVALID_ARG : GPE_ERR_HIP,
                    mismatch ? std::string("local group: all_to_all: a peer sends another length than this rank expects")
                             : std::string("local group: all_to_all: ") + hipGetErrorName(e));
    }
    if (!rendezvous(g)) return group_fail(c, "all_to_all");            // the peers have read my send buffer
    return GPE_OK;
}

// The neighbour segments of one step: every rank pulls its neighbours' packed segments into its receive buffer.
// Ordered by events on the contexts' streams -- rank r's copies wait for the pack of each neighbour, and r's next pack
// waits until every neighbour has copied -- so the host threads only meet, they never wait for the device.
// xs: the stream the copies run on (the context's, or the exchange's own: ShardState::overlap); the pack that follows is
// always on the context's stream.
gpe_status group_exchange_segments(gpe_ctx *c, hipStream_t xs)
{
    gpe_local_group *g = c->ctl.group;
    const uint32_t me = c->ctl.group_rank;
    const ShardCtl &T = c->ctl;
    gpe_local_group::Post &mine = g->post[me];
    hipError_t e = hipSuccess;
    if (!mine.packed) e = hipEventCreateWithFlags(&mine.packed, hipEventDisableTiming);
    if (e == hipSuccess && !mine.copied) e = hipEventCreateWithFlags(&mine.copied, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(mine.packed, xs);
    mine.send = T.d_send; mine.send_off = T.x_send_off; mine.send_cnt = T.x_send_cnt;
    if (e != hipSuccess) { (void)gpe_local_group_abort(g); return fail(c, GPE_ERR_HIP, std::string("local group: exchange: ") + hipGetErrorName(e)); }
    if (!rendezvous(g)) return group_fail(c, "exchange");
    bool mismatch = false;
    for (uint32_t p = 0; p < g->ws && e == hipSuccess; ++p) {
        if (T.x_recv_cnt[p] == 0) continue;
        const gpe_local_group::Post &P = g->post[p];
        if (P.send_cnt[me] != T.x_recv_cnt[p]) { mismatch = true; break; }
        e = hipStreamWaitEvent(xs, P.packed, 0);
        if (e == hipSuccess)
            e = hipMemcpyAsync(T.d_recv + T.x_recv_off[p], P.send + P.send_off[me], T.x_recv_cnt[p] * sizeof(uint32_t),
                               hipMemcpyDeviceToDevice, xs);
    }
    if (e == hipSuccess) e = hipEventRecord(mine.copied, xs);
    if (mismatch || e != hipSuccess) {
        (void)gpe_local_group_abort(g);
        return fail(c, mismatch ? GPE_ERR_INVALID_ARG : GPE_ERR_HIP,
                    mismatch ? std::string("local group: exchange: a peer sends another length than this rank expects")
                             : std::string("local group: exchange: ") + hipGetErrorName(e));
    }
    if (!rendezvous(g)) return group_fail(c, "exchange");
    // my send buffer is rewritten by the next pack: not before every neighbour has copied its segment out of it
    for (uint32_t p = 0; p < g->ws && e == hipSuccess; ++p)
        if (T.x_send_cnt[p] != 0) e = hipStreamWaitEvent(c->stream, g->post[p].copied, 0);
    if (e != hipSuccess) { (void)gpe_local_group_abort(g); return fail(c, GPE_ERR_HIP, std::string("local group: exchange: ") + hipGetErrorName(e)); }
    return GPE_OK;
}

void group_leave(gpe_ctx *c)
{
    gpe_local_group *g = c->ctl.group;
    if (!g) return;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        gpe_local_group::Post &P = g->post[c->ctl.group_rank];
        if (P.packed) (void)hipEventDestroy(P.packed);
        if (P.copied) (void)hipEventDestroy(P.copied);
        P = gpe_local_group::Post();
        if (g->members) --g->members;
        // a member that goes away leaves a group nobody can complete a collective in
        g->broken = true;
        g->cv.notify_all();
    }
    c->ctl.group = nullptr;
}

}  // namespace gpe

using namespace gpe;

extern "C" {

gpe_status gpe_local_group_create(uint32_t world_size, gpe_local_group **out)
{
    if (!out) return fail(nullptr, GPE_ERR_INVALID_ARG, "gpe_local_group_create: out is NULL");
    *out = nullptr;
    if (world_size < 1 || world_size > GPE_SHARD_MAX_RANKS)
        return fail(nullptr, GPE_ERR_INVALID_ARG, "gpe_local_group_create: 1 .. 26 ranks");
    gpe_local_group *g = new (std::nothrow) gpe_local_group();
    if (!g) return fail(nullptr, GPE_ERR_OOM, "gpe_local_group_create: host allocation failed");
    g->ws = world_size;
    *out = g;
    return GPE_OK;
}

gpe_status gpe_local_group_destroy(gpe_local_group *g)
{
    if (!g) return GPE_OK;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        if (g->members != 0) return fail(nullptr, GPE_ERR_STATE, "gpe_local_group_destroy: contexts are still members (destroy them first)");
    }
    delete g;
    return GPE_OK;
}

gpe_status gpe_local_group_join(gpe_ctx *c, gpe_local_group *g, uint32_t rank)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    if (!g || rank >= g->ws) return fail(c, GPE_ERR_INVALID_ARG, "gpe_local_group_join: bad group or rank");
    if (c->ctl.group) return fail(c, GPE_ERR_STATE, "gpe_local_group_join: the context is a member of a group already");
    std::lock_guard<std::mutex> lk(g->mu);
    if (g->post[rank].ctx) return fail(c, GPE_ERR_STATE, "gpe_local_group_join: the rank is taken");
    // contexts on different devices read each other's buffers: peer access, as far as the devices grant it (the copies
    // themselves work without; the reduction kernels need it)
    for (uint32_t r = 0; r < g->ws; ++r) {
        const gpe_ctx *o = g->post[r].ctx;
        if (!o || o->device == c->device) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, c->device, o->device) == hipSuccess && can) {
            (void)hipSetDevice(c->device);
            if (hipDeviceEnablePeerAccess(o->device, 0) != hipSuccess) (void)hipGetLastError();
            (void)hipSetDevice(o->device);
            if (hipDeviceEnablePeerAccess(c->device, 0) != hipSuccess) (void)hipGetLastError();
        }
    }
    (void)hipSetDevice(c->device);
    g->post[rank].ctx = c;
    ++g->members;
    c->ctl.group = g;
    c->ctl.group_rank = rank;
    return GPE_OK;
}

gpe_status gpe_local_group_abort(gpe_local_group *g)
{
    if (!g) return GPE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(g->mu);
    g->broken = true;
    g->cv.notify_all();
    return GPE_OK;
}

}  // extern "C"
