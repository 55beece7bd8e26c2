// k_shard.hip -- device-resident halo exchange of a sharded run (the reference is single-device).
//
// One process per GPU owns the particles whose home block (8x8 cells) lies in its rectangle of the world.
// Every step each rank must (a) hand particles that left its rectangle to their new owner (migrants) and
// (b) give every neighbour a copy of the particles within one block of that neighbour's rectangle (ghosts: the
// dependency cone of the four colour passes).  gpu-physics-engine_amd/sharded.py has the general
// formulation in torch (two host round trips per step); this file is the same protocol with NO host round trip:
//
//   pack     one pass over the owned particles: rows for each neighbour go into that neighbour's segment of
//            the send buffer, [n_mig, n_gho, -, -][migrant rows: x y px py r key][ghost rows: x y r key];
//            migrants are also listed as holes
//   (the host framework moves the segments: RCCL send/recv over xGMI, fixed sizes, so nothing to wait for)
//   unpack   holes are filled from the tail of the owned range, arriving migrants are appended to it, ghosts
//            follow; the new counts stay ON THE DEVICE (and are mirrored to pinned host memory)
//   step     the ordinary native step; the host passes an upper bound of the particle count, the kernels read
//            the true one (k_native_hash pads [n, bound) with a key that sorts behind every block)
//
// The order of the particles inside a rank is free (members of a cell are ordered by their ORDER KEY = index in
// the unsharded system), so compaction may move any survivor into any hole.
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>

#include "gpe_internal.h"

namespace gpe {

struct ShardArrays {
    float2 *pos, *prev;
    float *radius;
    uint32_t *gid;
};

__global__ void k_shard_zero(uint32_t *__restrict__ send, ShardSlots S, uint32_t *__restrict__ counts)
{
    const uint32_t t = threadIdx.x;
    if (t < S.n_slots * kSegHeader) send[S.send_off[t / kSegHeader] + (t % kSegHeader)] = 0;
    if (t == 0) counts[kShardHoles] = 0;                               // (the set in use)
}

__global__ __launch_bounds__(kStreamBlock) void k_shard_pack(ShardArrays A, uint64_t n_bound, float cell_size, PackArgs P)
{
    const uint64_t n_owned = P.counts[kShardOwned];
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t lim = n_owned < n_bound ? n_owned : n_bound;
    const uint64_t rounds = (lim + stride - 1) / stride;
    for (uint64_t r = 0; r < rounds; ++r) {
        const uint64_t i = r * stride + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        const bool mine = i < lim;
        const uint64_t ic = mine ? i : 0;
        // (the rows need radius, key and -- for migrants -- the previous position: loaded for every lane, three cached
        // streams beside the positions; the first pack of a run only, the steps pack in their tiles)
        pack_particle(P, mine, (uint32_t)ic, A.pos[ic], A.prev[ic], A.radius[ic], A.gid[ic], cell_size);
    }
}

// ONE launch consumes what arrived.  Every workgroup derives the same plan from what no workgroup writes during the
// launch -- the counts of the set in use, the received headers, the capacities: the owned range shrinks by the holes and
// grows by the arriving migrants, the ghosts follow -- and copies its share of the rows whose destination lies BEYOND the
// old owned range.  Workgroup 0 alone touches the old range: it compacts it (holes below the new end take the survivors
// from behind it), then writes the few rows that land in the part of the old range the compaction has vacated, resets
// the send headers for the next pack, and writes the new counts into the OTHER set, which the kernels behind this one
// are pointed at (ShardState::parity).  (Rounds 1-3: a one-workgroup plan kernel, then a rows kernel.)
struct UnpackPlan {
    uint32_t mig_off[kShardMaxSlots], mig_cnt[kShardMaxSlots], gho_off[kShardMaxSlots], gho_cnt[kShardMaxSlots];
    uint32_t n0, base, n_owned, total, err;
};

__device__ __forceinline__ void unpack_rows_of(const ShardArrays &A, const ShardSlots &S, const uint32_t *send,
                                               const uint32_t *recv, const UnpackPlan &U, uint32_t t0, uint32_t stride,
                                               uint32_t lo, uint32_t hi)
{
    // rows whose destination d satisfies lo <= d < hi
    for (uint32_t s = 0; s < S.n_slots; ++s) {
        const bool self = s + 1 == S.n_slots;
        const uint32_t *seg = self ? send + S.send_off[s] : recv + S.recv_off[s];
        const uint32_t cap_mig = self ? S.send_cap_mig[s] : S.recv_cap_mig[s];
        if (!self) {
            const uint32_t cnt = U.mig_cnt[s], off = U.mig_off[s];
            const uint32_t j0 = lo > off ? lo - off : 0u, j1 = hi > off ? min(cnt, hi - off) : 0u;
            for (uint32_t j = j0 + t0; j < j1; j += stride) {
                const uint32_t *row = seg + kSegHeader + (uint64_t)j * kMigWords;
                A.pos[off + j] = make_float2(__uint_as_float(row[0]), __uint_as_float(row[1]));
                A.prev[off + j] = make_float2(__uint_as_float(row[2]), __uint_as_float(row[3]));
                A.radius[off + j] = __uint_as_float(row[4]);
                A.gid[off + j] = row[5];
            }
        }
        const uint32_t cnt = U.gho_cnt[s], off = U.gho_off[s];
        const uint32_t j0 = lo > off ? lo - off : 0u, j1 = hi > off ? min(cnt, hi - off) : 0u;
        for (uint32_t j = j0 + t0; j < j1; j += stride) {
            const uint32_t *row = seg + kSegHeader + (uint64_t)cap_mig * kMigWords + (uint64_t)j * kGhoWords;
            A.pos[off + j] = make_float2(__uint_as_float(row[0]), __uint_as_float(row[1]));
            A.radius[off + j] = __uint_as_float(row[2]);
            A.gid[off + j] = row[3];
        }
    }
}

constexpr int kUnpackBlock = 1024, kUnpackGrid = 32;
__global__ __launch_bounds__(kUnpackBlock) void k_shard_unpack(ShardArrays A, ShardSlots S, uint32_t *send,
                                                               const uint32_t *__restrict__ recv,
                                                               const uint32_t *__restrict__ counts_old,
                                                               uint32_t *__restrict__ counts_new,
                                                               uint32_t *__restrict__ err_word,
                                                               uint32_t *__restrict__ done_ticket,
                                                               uint32_t *__restrict__ host_counts,
                                                               uint32_t *__restrict__ holes,
                                                               uint8_t *__restrict__ hole_flag,
                                                               uint32_t *__restrict__ fill_src,
                                                               uint32_t *__restrict__ fill_dst, uint32_t holes_cap,
                                                               uint64_t capacity)
{
    __shared__ UnpackPlan U;
    __shared__ uint32_t s_ns, s_nh;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) {
        s_ns = 0; s_nh = 0;
        const uint32_t n0 = counts_old[kShardOwned];
        uint32_t m = counts_old[kShardHoles];
        if (m > holes_cap) m = holes_cap;
        if (m > n0) m = n0;
        uint32_t err = 0;
        uint64_t o = n0 - m;
        U.n0 = n0; U.base = n0 - m;
        for (uint32_t s = 0; s + 1 < S.n_slots; ++s) {                 // neighbours; the last slot is this rank
            uint32_t c = recv[S.recv_off[s] + 0];
            if (c > S.recv_cap_mig[s]) { c = S.recv_cap_mig[s]; err |= kShardErrRecvOverflow; }
            if (o + c > capacity) { c = 0; err |= kShardErrCapacity; }
            U.mig_off[s] = (uint32_t)o;
            U.mig_cnt[s] = c;
            o += c;
        }
        U.n_owned = (uint32_t)o;
        for (uint32_t s = 0; s < S.n_slots; ++s) {
            const bool self = s + 1 == S.n_slots;
            uint32_t c = self ? send[S.send_off[s] + 1] : recv[S.recv_off[s] + 1];
            const uint32_t cap = self ? S.send_cap_gho[s] : S.recv_cap_gho[s];
            if (c > cap) { c = cap; err |= self ? kShardErrSendOverflow : kShardErrRecvOverflow; }
            if (o + c > capacity) { c = 0; err |= kShardErrCapacity; }
            U.gho_off[s] = (uint32_t)o;
            U.gho_cnt[s] = c;
            o += c;
        }
        U.total = (uint32_t)o;
        U.err = err;
    }
    __syncthreads();
    const uint32_t n0 = U.n0, base = U.base;
    // rows that land beyond the old owned range: everybody's job
    unpack_rows_of(A, S, send, recv, U, blockIdx.x * blockDim.x + tid, gridDim.x * blockDim.x, n0, 0xFFFFFFFFu);
    if (blockIdx.x != 0) {
        // The segment headers have been consumed -- by the plan of EVERY workgroup: the last one to get here resets them
        // for the next pack (the tiles of the coming step append to them), which then needs no launch of its own for that.
        __syncthreads();
        __shared__ uint32_t s_last;
        if (tid == 0) {
            __threadfence();
            s_last = atomicAdd(done_ticket, 1u) == gridDim.x - 1u ? 1u : 0u;
        }
        __syncthreads();
        if (s_last) {
            if (tid < S.n_slots * kSegHeader) send[S.send_off[tid / kSegHeader] + (tid % kSegHeader)] = 0;
            if (tid == 0) *done_ticket = 0u;
        }
        return;
    }
    // ---- workgroup 0: compaction of the old range, then the rows that land in it ----
    const uint32_t m = n0 - base;
    for (uint32_t t = tid; t < m; t += blockDim.x) {
        const uint32_t slot = base + t;
        if (!hole_flag[slot]) fill_src[atomicAdd(&s_ns, 1u)] = slot;
        const uint32_t h = holes[t];
        if (h < base) fill_dst[atomicAdd(&s_nh, 1u)] = h;
    }
    __syncthreads();
    const uint32_t moves = min(s_ns, s_nh);                            // equal by construction
    for (uint32_t t = tid; t < moves; t += blockDim.x) {
        const uint32_t a = fill_src[t], b = fill_dst[t];
        A.pos[b] = A.pos[a]; A.prev[b] = A.prev[a]; A.radius[b] = A.radius[a]; A.gid[b] = A.gid[a];
    }
    for (uint32_t t = tid; t < m; t += blockDim.x) hole_flag[holes[t]] = 0;
    __syncthreads();                                                   // (the moves have read the tail: it may be overwritten now)
    unpack_rows_of(A, S, send, recv, U, tid, blockDim.x, base, n0);
    __syncthreads();
    {
        __shared__ uint32_t s_last0;
        if (tid == 0) {
            __threadfence();
            s_last0 = atomicAdd(done_ticket, 1u) == gridDim.x - 1u ? 1u : 0u;
        }
        __syncthreads();
        if (s_last0) {                                                 // (see above: the last workgroup resets the headers)
            if (tid < S.n_slots * kSegHeader) send[S.send_off[tid / kSegHeader] + (tid % kSegHeader)] = 0;
            if (tid == 0) *done_ticket = 0u;
        }
    }
    if (tid == 0) {
        const uint32_t err = U.err | ((s_ns != s_nh) ? kShardErrHoles : 0u);
        const uint32_t epoch = counts_old[kShardEpoch] + 1u;
        counts_new[kShardOwned] = U.n_owned;
        counts_new[kShardTotal] = U.total;
        counts_new[kShardEpoch] = epoch;
        counts_new[kShardHoles] = 0;
        if (err) atomicOr(err_word, err);
        // pinned mirror: the epoch last, so a host that sees it also sees the counts of that epoch or newer
        __hip_atomic_store(&host_counts[kShardOwned], U.n_owned, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&host_counts[kShardTotal], U.total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&host_counts[kShardError], *err_word | err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&host_counts[kShardEpoch], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
std::string shard_error_text(uint32_t flags)
{
    char msg[320];
    snprintf(msg, sizeof(msg), "sharded exchange failed (flags 0x%x:%s%s%s%s)", flags,
             (flags & (kShardErrSendOverflow | kShardErrRecvOverflow)) ? " a neighbour segment overflowed" : "",
             (flags & kShardErrNoSlot) ? " a particle moved more than one block in a step" : "",
             (flags & kShardErrCapacity) ? " particle capacity exceeded" : "",
             (flags & kShardErrHoles) ? " too many migrants in one step" : "");
    return msg;
}

void shard_release(gpe_ctx *c)
{
    ShardState &S = c->shard;
    if (S.counts) (void)hipFree(S.counts);
    if (S.holes) (void)hipFree(S.holes);
    if (S.fill_src) (void)hipFree(S.fill_src);
    if (S.fill_dst) (void)hipFree(S.fill_dst);
    if (S.hole_flag) (void)hipFree(S.hole_flag);
    if (S.host_counts) (void)hipHostFree(S.host_counts);
    for (hipEvent_t e : S.fence) if (e) (void)hipEventDestroy(e);
    if (S.ev_packed) (void)hipEventDestroy(S.ev_packed);
    if (S.ev_exchanged) (void)hipEventDestroy(S.ev_exchanged);
    if (S.xstream) (void)hipStreamDestroy(S.xstream);
    S = ShardState();
}

static ShardArrays shard_arrays(gpe_ctx *c)
{
    ShardArrays A;
    A.pos = c->pos; A.prev = c->prev; A.radius = c->radius; A.gid = c->order_keys;
    return A;
}

static gpe_status shard_ready(gpe_ctx *c, bool need_active)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    if (c->n == 0 || !c->pos) return fail(c, GPE_ERR_STATE, "no particles: call gpe_set_particles first");
    if (!c->shard.on) return fail(c, GPE_ERR_STATE, "sharded exchange not configured: call gpe_shard_configure first");
    if (need_active && !c->shard.active) return fail(c, GPE_ERR_STATE, "call gpe_shard_begin first");
    GPE_HIP(c, hipSetDevice(c->device));
    return GPE_OK;
}

static gpe_status ensure_flag_capacity(gpe_ctx *c)
{
    ShardState &S = c->shard;
    if (S.flag_cap >= c->cap) return GPE_OK;
    if (S.hole_flag) GPE_HIP(c, hipFree(S.hole_flag));
    S.hole_flag = nullptr;
    GPE_HIP(c, hipMalloc((void **)&S.hole_flag, c->cap + 64));
    GPE_HIP(c, hipMemsetAsync(S.hole_flag, 0, c->cap + 64, c->stream));
    S.flag_cap = c->cap;
    return GPE_OK;
}

gpe_status shard_ensure_flag_capacity(gpe_ctx *c) { return ensure_flag_capacity(c); }

// what a pack needs (PackArgs): the plan's tables, the segments, the hole list; `on` and the ring / safe boxes are for the
// tiles of the native step and only set when the plan told the rank's rectangle
void shard_pack_args(gpe_ctx *c, PackArgs *P)
{
    const ShardState &S = c->shard;
    *P = PackArgs();
    P->owner = S.owner; P->dest_mask = S.dest_mask; P->blocks_x = S.blocks_x; P->blocks_y = S.blocks_y;
    P->my_rank = S.my_rank;
    P->send = S.send; P->counts = S.counts_now(); P->err = S.counts + kShardError; P->holes = S.holes; P->hole_flag = S.hole_flag;
    P->holes_cap = (uint32_t)S.holes_cap;
    P->slots = S.slots;
    if (S.have_rect && S.active) {
        // Who can need packing: a particle whose NEW block is owned by another rank or borders one, i.e. lies in the
        // outermost block ring of the rectangle or beyond; one cell of margin covers the rounding of the box's edges.
        const int x0 = S.rect[0], y0 = S.rect[1], x1 = S.rect[2], y1 = S.rect[3];
        const bool nb_l = x0 > 0, nb_r = x1 < S.blocks_x, nb_d = y0 > 0, nb_u = y1 < S.blocks_y;
        const float cs = c->cell_size, big = 3.0e38f;
        P->safe_x0 = nb_l ? (float)((x0 + 1) * 8 + 1) * cs : -big;
        P->safe_y0 = nb_d ? (float)((y0 + 1) * 8 + 1) * cs : -big;
        P->safe_x1 = nb_r ? (float)((x1 - 1) * 8 - 1) * cs : big;
        P->safe_y1 = nb_u ? (float)((y1 - 1) * 8 - 1) * cs : big;
        P->on = 1u;
    }
}

static gpe_status launch_pack(gpe_ctx *c, bool reset_headers)
{
    ShardState &S = c->shard;
    Scope s(c, "shard/pack");
    if (reset_headers) {                                               // the first pack of a run; later ones find
        hipLaunchKernelGGL(k_shard_zero, dim3(1), dim3(64), 0, c->stream, S.send, S.slots, S.counts_now());   // them reset by the unpack
        GPE_HIP(c, hipGetLastError());
    }
    const uint64_t bound = std::min<uint64_t>(c->cap, c->n);          // owned <= total <= bound
    PackArgs P;
    shard_pack_args(c, &P);
    hipLaunchKernelGGL(k_shard_pack, dim3(stream_grid(bound)), dim3(kStreamBlock), 0, c->stream, shard_arrays(c), bound,
                       c->cell_size, P);
    GPE_HIP(c, hipGetLastError());
    S.packed = true;
    if (S.overlap) { GPE_HIP(c, hipEventRecord(S.ev_packed, c->stream)); S.packed_recorded = true; }
    return GPE_OK;
}

static gpe_status launch_unpack(gpe_ctx *c)
{
    ShardState &S = c->shard;
    // (the segments arrive on the exchange's stream)
    if (S.overlap && S.exchanged_recorded) GPE_HIP(c, hipStreamWaitEvent(c->stream, S.ev_exchanged, 0));
    Scope s(c, "shard/unpack");
    uint32_t *old_set = S.counts_now();
    S.parity ^= 1u;                                                    // the kernels behind this launch read the new set
    hipLaunchKernelGGL(k_shard_unpack, dim3(kUnpackGrid), dim3(kUnpackBlock), 0, c->stream, shard_arrays(c), S.slots, S.send,
                       S.recv, old_set, S.counts_now(), S.counts + kShardError, S.counts + kShardDoneTicket, S.host_counts, S.holes, S.hole_flag,
                       S.fill_src, S.fill_dst, (uint32_t)S.holes_cap, c->cap);
    GPE_HIP(c, hipGetLastError());
    S.packed = false;
    return GPE_OK;
}

// Upper bound of the device-side particle total, from the pinned mirror (it lags by the steps in flight; the
// fence in gpe_shard_step bounds that, the slack covers the drift of the ghost population meanwhile).
static uint64_t shard_bound(gpe_ctx *c)
{
    ShardState &S = c->shard;
    const uint32_t epoch = __atomic_load_n(&S.host_counts[kShardEpoch], __ATOMIC_ACQUIRE);
    if ((int32_t)(epoch - S.begin_epoch) <= 0) return c->cap;          // no unpack of this run has landed yet
    const uint64_t total = S.host_counts[kShardTotal];
    const uint64_t slack = std::max<uint64_t>(16384, total / 32);
    return std::min<uint64_t>(c->cap, total + slack);
}

// A rank that particles pile up on (gravity) outgrows its buffers: when the mirrored total (it lags by < ~64 steps)
// passes 3/4 of the capacity, read the exact counts and reallocate at 1.5 x.  Called right behind the unpack (no
// migrant is flagged then); purely local -- no collective, the other ranks do not notice.
static gpe_status shard_grow_if_needed(gpe_ctx *c)
{
    ShardState &S = c->shard;
    const uint32_t epoch = __atomic_load_n(&S.host_counts[kShardEpoch], __ATOMIC_ACQUIRE);
    if ((int32_t)(epoch - S.begin_epoch) <= 0) return GPE_OK;
    if ((uint64_t)S.host_counts[kShardTotal] * 4 < c->cap * 3) return GPE_OK;
    uint32_t w[5] = {0, 0, 0, 0, 0};
    GPE_HIP(c, hipMemcpyAsync(w, S.counts_now(), sizeof(w), hipMemcpyDeviceToHost, c->stream));
    GPE_HIP(c, hipMemcpyAsync(&w[kShardError], S.counts + kShardError, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    if (w[kShardError]) return fail(c, GPE_ERR_UNSUPPORTED, shard_error_text(w[kShardError]));
    const uint64_t total = w[kShardTotal];
    if (total * 10 < c->cap * 7) return GPE_OK;                        // the mirror was ahead of a shrinking count
    const uint64_t want = total + total / 2 + 4096;
    if (want > (1ull << 30) - 1) return fail(c, GPE_ERR_INVALID_ARG, "sharded run: 4n must fit in u32");
    c->n = std::min<uint64_t>(c->cap, std::max<uint64_t>(total, 1));  // what grow_for_shard carries over
    GPE_TRY(grow_for_shard(c, want));
    GPE_TRY(ensure_flag_capacity(c));
    return reconfigure_native(c);
}

}  // namespace gpe

using namespace gpe;

extern "C" {

gpe_status gpe_shard_configure(gpe_ctx *c, const gpe_shard_plan *p)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    constexpr uint32_t kPlanV1 = (uint32_t)offsetof(gpe_shard_plan, own_x0);     // the plan before it told the rectangle
    if (!p || (p->struct_size != sizeof(gpe_shard_plan) && p->struct_size != kPlanV1))
        return fail(c, GPE_ERR_INVALID_ARG, "gpe_shard_configure: bad plan");
    if (c->n == 0 || !c->pos) return fail(c, GPE_ERR_STATE, "no particles: call gpe_set_particles first");
    if (p->n_slots < 1 || p->n_slots > (uint32_t)kShardMaxSlots || p->world_size > 26 || p->rank >= p->world_size ||
        !p->d_owner_of_block || !p->d_dest_mask_of_block || !p->d_send || !p->d_recv || p->blocks_x <= 0 || p->blocks_y <= 0 ||
        p->slot_rank[p->n_slots - 1] != p->rank)
        return fail(c, GPE_ERR_INVALID_ARG, "gpe_shard_configure: bad plan (slots = neighbours ascending, then this rank)");
    GPE_HIP(c, hipSetDevice(c->device));
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    ShardState &S = c->shard;
    S.my_rank = p->rank;
    S.blocks_x = p->blocks_x; S.blocks_y = p->blocks_y;
    S.owner = p->d_owner_of_block; S.dest_mask = p->d_dest_mask_of_block;
    S.send = p->d_send; S.recv = p->d_recv;
    S.have_rect = false;
    if (p->struct_size == sizeof(gpe_shard_plan) && p->own_x1 > p->own_x0 && p->own_y1 > p->own_y0) {
        if (p->own_x0 < 0 || p->own_y0 < 0 || p->own_x1 > p->blocks_x || p->own_y1 > p->blocks_y)
            return fail(c, GPE_ERR_INVALID_ARG, "gpe_shard_configure: the rank's rectangle lies outside the block grid");
        S.rect[0] = p->own_x0; S.rect[1] = p->own_y0; S.rect[2] = p->own_x1; S.rect[3] = p->own_y1;
        S.have_rect = true;
    }
    memset(&S.slots, 0, sizeof(S.slots));
    S.slots.n_slots = p->n_slots;
    for (int r = 0; r < 32; ++r) S.slots.slot_of_rank[r] = -1;
    for (uint32_t s = 0; s < p->n_slots; ++s) {
        if (p->slot_rank[s] >= p->world_size) return fail(c, GPE_ERR_INVALID_ARG, "gpe_shard_configure: slot rank out of range");
        S.slots.rank[s] = p->slot_rank[s];
        S.slots.slot_of_rank[p->slot_rank[s]] = (int8_t)s;
        S.slots.send_off[s] = p->send_off[s]; S.slots.send_cap_mig[s] = p->send_cap_mig[s]; S.slots.send_cap_gho[s] = p->send_cap_gho[s];
        S.slots.recv_off[s] = p->recv_off[s]; S.slots.recv_cap_mig[s] = p->recv_cap_mig[s]; S.slots.recv_cap_gho[s] = p->recv_cap_gho[s];
    }
    if (!S.counts) {
        GPE_HIP(c, hipMalloc((void **)&S.counts, 2 * kShardSetWords * sizeof(uint32_t)));
        GPE_HIP(c, hipMemset(S.counts, 0, 2 * kShardSetWords * sizeof(uint32_t)));
        GPE_HIP(c, hipHostMalloc((void **)&S.host_counts, 64, hipHostMallocDefault));
        memset(S.host_counts, 0, 64);
    }
    uint64_t want = 0;
    for (uint32_t s = 0; s < p->n_slots; ++s) want += p->send_cap_mig[s];
    want = std::max<uint64_t>(want, 1024);
    if (S.holes_cap < want) {
        uint32_t **bufs[3] = {&S.holes, &S.fill_src, &S.fill_dst};
        for (uint32_t **b : bufs) {
            if (*b) GPE_HIP(c, hipFree(*b));
            *b = nullptr;
            GPE_HIP(c, hipMalloc((void **)b, (want + 16) * sizeof(uint32_t)));
        }
        S.holes_cap = want;
    }
    GPE_TRY(ensure_flag_capacity(c));
    // the exchange beside the step: a stream of its own (above the step's priority: its few small operations should not
    // queue behind two thousand tiles) and the two events that order it with the step
    if ((c->cfg.flags & GPE_FLAG_SHARD_OVERLAP) != 0 && !S.xstream) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        hipError_t e = hipStreamCreateWithPriority(&S.xstream, hipStreamNonBlocking, hi);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&S.ev_packed, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&S.ev_exchanged, hipEventDisableTiming);
        if (e != hipSuccess) {                                         // (without: everything in order on the one stream)
            (void)hipGetLastError();
            if (S.ev_packed) (void)hipEventDestroy(S.ev_packed);
            if (S.ev_exchanged) (void)hipEventDestroy(S.ev_exchanged);
            if (S.xstream) (void)hipStreamDestroy(S.xstream);
            S.xstream = nullptr; S.ev_packed = nullptr; S.ev_exchanged = nullptr;
        }
    }
    S.overlap = S.xstream != nullptr;
    S.packed_recorded = S.exchanged_recorded = false;
    const bool was_on = S.on;
    S.on = true;
    S.active = false;
    S.packed = false;
    if (!was_on) GPE_TRY(reconfigure_native(c));                       // the padding key needs its bits (native_configure)
    return GPE_OK;
}

gpe_status gpe_shard_begin(gpe_ctx *c)
{
    GPE_TRY(shard_ready(c, false));
    ShardState &S = c->shard;
    if (!c->use_order_keys || !c->order_keys) return fail(c, GPE_ERR_STATE, "gpe_shard_begin: order keys are off");
    GPE_TRY(ensure_flag_capacity(c));
    // the host's counts are exact here (after set_particles / gpe_shard_counts): ghosts are dropped
    c->n = c->n_owned;
    GPE_HIP(c, hipStreamSynchronize(c->stream));                       // every earlier unpack has landed
    const uint32_t epoch = S.host_counts[kShardEpoch] + 1u;            // the mirror counts from here
    // (the sticky error word is word kShardError of set 0 whichever set is in use: a new run starts without one)
    const uint32_t init[5] = {(uint32_t)c->n_owned, (uint32_t)c->n_owned, 0u, epoch, 0u};
    S.parity = 0;
    GPE_HIP(c, hipMemcpyAsync(S.counts, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    S.begin_epoch = epoch;
    S.active = true;
    S.steps = 0;
    if (S.xstream) GPE_HIP(c, hipStreamSynchronize(S.xstream));
    S.packed_recorded = S.exchanged_recorded = false;
    c->native.sort_state_valid = false;       // the caller has just re-sorted / re-dealt the particles: a new grouping
    return launch_pack(c, true);
}

gpe_status gpe_shard_unpack(gpe_ctx *c)
{
    GPE_TRY(shard_ready(c, true));
    if (!c->shard.packed) return fail(c, GPE_ERR_STATE, "gpe_shard_unpack: nothing was packed");
    return launch_unpack(c);
}

gpe_status gpe_shard_step(gpe_ctx *c, float dt)
{
    GPE_TRY(shard_ready(c, true));
    ShardState &S = c->shard;
    if (!S.packed) return fail(c, GPE_ERR_STATE, "gpe_shard_step: exchange the segments packed by the previous call first");
    if ((S.steps & 31u) == 0) {                                        // the host runs at most ~64 steps ahead
        const int slot = (int)((S.steps >> 5) & 1u);
        if (S.armed[slot]) (void)hipEventSynchronize(S.fence[slot]);
        if (!S.fence[slot] && hipEventCreateWithFlags(&S.fence[slot], hipEventDisableTiming) != hipSuccess) S.fence[slot] = nullptr;
        if (S.fence[slot]) S.armed[slot] = hipEventRecord(S.fence[slot], c->stream) == hipSuccess;
    }
    ++S.steps;
    // sampled profiling (gpe_set_profiling(k > 1)): the exchange kernels are bracketed on the same steps as the
    // step's own kernels (step_for_shard advances the sample counter)
    const bool sampled = c->profile_every > 1;
    const bool on = sampled ? (c->profile_step % c->profile_every) == 0 : c->profiling;
    c->profiling = on;
    gpe_status st = launch_unpack(c);
    if (st == GPE_OK) st = shard_grow_if_needed(c);
    c->n = shard_bound(c);
    if (st == GPE_OK) st = step_for_shard(c, dt);
    c->profiling = on;
    // the tiles of the step packed their own particles as they wrote them back (native_collide, PackArgs.on); a plan
    // that does not tell the rank's rectangle leaves it to the pack kernel
    if (st == GPE_OK) {
        if (S.have_rect) S.packed = true;
        else st = launch_pack(c, false);
    }
    if (sampled) c->profiling = true;
    return st;
}

gpe_status gpe_shard_peek(gpe_ctx *c, uint64_t *n_owned, uint64_t *n_total)
{
    // the pinned mirror as it stands: no synchronisation, the values lag by the steps in flight (at most ~64)
    if (!c) return GPE_ERR_INVALID_ARG;
    if (!c->shard.on || !c->shard.host_counts) return fail(c, GPE_ERR_STATE, "sharded exchange not configured");
    const uint32_t epoch = __atomic_load_n(&c->shard.host_counts[kShardEpoch], __ATOMIC_ACQUIRE);
    const bool fresh = c->shard.active && (int32_t)(epoch - c->shard.begin_epoch) > 0;
    if (n_owned) *n_owned = fresh ? c->shard.host_counts[kShardOwned] : c->n_owned;
    if (n_total) *n_total = fresh ? c->shard.host_counts[kShardTotal] : c->n_owned;
    return GPE_OK;
}

gpe_status gpe_shard_counts(gpe_ctx *c, uint64_t *n_owned, uint64_t *n_total, int32_t leave)
{
    GPE_TRY(shard_ready(c, true));
    ShardState &S = c->shard;
    uint32_t w[5] = {0, 0, 0, 0, 0};
    if (S.xstream) GPE_HIP(c, hipStreamSynchronize(S.xstream));       // (an exchange nobody unpacked: nothing of it is needed)
    GPE_HIP(c, hipMemcpyAsync(w, S.counts_now(), sizeof(w), hipMemcpyDeviceToHost, c->stream));
    GPE_HIP(c, hipMemcpyAsync(&w[kShardError], S.counts + kShardError, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    if (n_owned) *n_owned = w[kShardOwned];
    if (n_total) *n_total = w[kShardTotal];
    if (w[kShardError]) return fail(c, GPE_ERR_UNSUPPORTED, shard_error_text(w[kShardError]));
    if (leave) {
        // back to host-side counts (the caller re-sorts or reads the owned range): ghosts are dropped
        if (w[kShardOwned] == 0 || w[kShardOwned] > c->cap) return fail(c, GPE_ERR_STATE, "gpe_shard_counts: bad owned count");
        c->n_owned = w[kShardOwned];
        c->n = c->n_owned;
        S.active = false;
        S.packed = false;
    }
    return GPE_OK;
}

}  // extern "C"
