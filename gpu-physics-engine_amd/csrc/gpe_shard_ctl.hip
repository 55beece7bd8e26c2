// gpe_shard_ctl.hip -- the control plane of a sharded run behind the C-ABI (include/gpe.h, "sharded control plane").
//
// The reference steps one device (state.rs:115-131: re-sort gate, then Grid::update / solve_collisions / integrate).
// Across ranks the same schedule needs four things the step kernels do not do themselves, all of them here so that a
// host in any language drives a sharded run through gpe.h alone:
//   layout    the world cut into px x py rectangles of 8x8-cell blocks; owner / destination tables; neighbours;
//             segment capacities                                                   (host arithmetic, no GPU)
//   set-up    cell size of the whole system, tile grid cut to the rank, tables and segments on the device,
//             gpe_shard_configure; both ends of every neighbour pair checked against each other
//   re-sort   ParticleSort::sort (particle_sort.rs:58-69) made global: the new index of a particle is its position in
//             the single-device sorted order = particles of all ranks in earlier Morton blocks + its position inside
//             its block; one all-reduce of the Morton-block histogram
//   re-cut    rectangles re-cut at the particle quantiles when the load has drifted apart; one all-to-all of rows
// Collectives come from the RCCL communicator inside the library, a local group (several contexts in one process) or
// the caller (gpe_shard_set_collectives): two primitives, an all-reduce and an all-to-all over device words.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "gpe_internal.h"

namespace gpe {

constexpr int kBlock = 8;                      // cells per block edge: the granule of ownership and of the block table

// ---------------------------------------------------------------------------------------------------------
// layout: host arithmetic
// ---------------------------------------------------------------------------------------------------------
static uint32_t host_split16(uint32_t n)
{
    uint32_t x = n & 0x0000FFFFu;
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}

// Morton blocks (64 consecutive home-cell keys = one 8x8-cell block) of the cell box
static uint64_t morton_entries(const gpe_shard_layout &L)
{
    const uint32_t max_key = host_split16((uint32_t)(L.cells_x - 1)) | (host_split16((uint32_t)(L.cells_y - 1)) << 1);
    return (uint64_t)(max_key >> 6) + 1;
}

static bool layout_valid(const gpe_shard_layout *L)
{
    return L && L->struct_size == sizeof(gpe_shard_layout) && L->world_size >= 1 && L->world_size <= GPE_SHARD_MAX_RANKS &&
           L->px >= 1 && L->py >= 1 && L->px * L->py == L->world_size && L->blocks_x >= (int32_t)L->px &&
           L->blocks_y >= (int32_t)L->py && L->xcuts[0] == 0 && L->ycuts[0] == 0 && L->xcuts[L->px] == L->blocks_x &&
           L->ycuts[L->py] == L->blocks_y;
}

static void rect_blocks(const gpe_shard_layout &L, uint32_t rank, int &x0, int &y0, int &x1, int &y1)
{
    const uint32_t i = rank % L.px, j = rank / L.px;
    x0 = L.xcuts[i]; x1 = L.xcuts[i + 1]; y0 = L.ycuts[j]; y1 = L.ycuts[j + 1];
}

// owner[block] and dest_mask[block] (the ranks whose rectangle lies within one block, the owner excluded)
static void build_tables(const gpe_shard_layout &L, std::vector<uint8_t> &owner, std::vector<uint32_t> &mask)
{
    const int bx = L.blocks_x, by = L.blocks_y;
    std::vector<uint8_t> col(bx), row(by);
    for (uint32_t i = 0; i < L.px; ++i) for (int x = L.xcuts[i]; x < L.xcuts[i + 1]; ++x) col[x] = (uint8_t)i;
    for (uint32_t j = 0; j < L.py; ++j) for (int y = L.ycuts[j]; y < L.ycuts[j + 1]; ++y) row[y] = (uint8_t)j;
    owner.assign((size_t)bx * by, 0);
    mask.assign((size_t)bx * by, 0);
    for (int y = 0; y < by; ++y)
        for (int x = 0; x < bx; ++x) owner[(size_t)y * bx + x] = (uint8_t)(row[y] * L.px + col[x]);
    for (int y = 0; y < by; ++y)
        for (int x = 0; x < bx; ++x) {
            uint32_t m = 0;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int xx = x + dx, yy = y + dy;
                    if (xx < 0 || yy < 0 || xx >= bx || yy >= by) continue;
                    m |= 1u << owner[(size_t)yy * bx + xx];
                }
            mask[(size_t)y * bx + x] = m & ~(1u << owner[(size_t)y * bx + x]);
        }
}

// ranks whose rectangle lies within one block of `rank`'s, ascending
static std::vector<uint32_t> neighbours(const gpe_shard_layout &L, const std::vector<uint32_t> &mask, uint32_t rank)
{
    int x0, y0, x1, y1;
    rect_blocks(L, rank, x0, y0, x1, y1);
    uint32_t m = 0;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) m |= mask[(size_t)y * L.blocks_x + x];
    std::vector<uint32_t> out;
    for (uint32_t p = 0; p < L.world_size; ++p)
        if (((m >> p) & 1u) && p != rank) out.push_back(p);
    return out;
}

// how many of src's blocks lie within one block of dst's rectangle (their particles are dst's ghosts)
static uint64_t border_blocks(const gpe_shard_layout &L, const std::vector<uint32_t> &mask, uint32_t src, uint32_t dst)
{
    int x0, y0, x1, y1;
    rect_blocks(L, src, x0, y0, x1, y1);
    uint64_t n = 0;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) n += (mask[(size_t)y * L.blocks_x + x] >> dst) & 1u;
    return n;
}

// (migrant rows, ghost rows) of the segment src -> dst: three times the mean population of src's blocks bordering
// dst, plus slack.  A pure function of the layout, per_block and scale: the sender and the receiver size it alike.
static void segment_caps(const gpe_shard_layout &L, const std::vector<uint32_t> &mask, uint32_t src, uint32_t dst,
                         double per_block, double scale, uint32_t &cap_mig, uint32_t &cap_gho)
{
    const uint64_t gho = (uint64_t)(((double)border_blocks(L, mask, src, dst) * per_block * 3.0 + 2048.0) * scale);
    cap_mig = (uint32_t)(gho / 4 + (uint64_t)(512.0 * scale) + 1);
    cap_gho = (uint32_t)(gho + 1);
}

static uint64_t segment_words(uint32_t cap_mig, uint32_t cap_gho) { return 4ull + 6ull * cap_mig + 4ull * cap_gho; }

static int min_region_blocks(const gpe_shard_layout &L)
{
    int m = 1 << 30;
    for (uint32_t i = 0; i < L.px; ++i) m = std::min(m, L.xcuts[i + 1] - L.xcuts[i]);
    for (uint32_t j = 0; j < L.py; ++j) m = std::min(m, L.ycuts[j + 1] - L.ycuts[j]);
    return m;
}

// ---------------------------------------------------------------------------------------------------------
// collectives: dispatch and the small host-side forms
// ---------------------------------------------------------------------------------------------------------
static int transport_kind(const gpe_ctx *c)
{
    if (c->ctl.coll_set) return 3;
    if (c->ctl.group) return 2;
    if (c->shard.comm) return 1;
    return 0;
}

gpe_status coll_all_reduce_u32(gpe_ctx *c, uint32_t *d_buf, uint64_t count, uint32_t op)
{
    if (count == 0) return GPE_OK;
    if (c->ctl.coll_set) {
        if (!c->ctl.coll.all_reduce_u32) return fail(c, GPE_ERR_STATE, "sharded run: the caller's collectives have no all_reduce_u32");
        if (c->ctl.coll.all_reduce_u32(c->ctl.coll.user, d_buf, count, op, (void *)c->stream) != 0)
            return fail(c, GPE_ERR_HIP, "sharded run: the caller's all_reduce_u32 failed");
        return GPE_OK;
    }
    if (c->ctl.group) return group_all_reduce_u32(c, d_buf, count, op);
    if (c->shard.comm) return rccl_all_reduce_u32(c, d_buf, count, op);
    if (c->ctl.ready ? c->ctl.layout.world_size == 1 : false) return GPE_OK;
    return fail(c, GPE_ERR_STATE, "sharded run: no collectives (gpe_shard_comm_init / _attach, gpe_local_group_join or "
                                  "gpe_shard_set_collectives)");
}

gpe_status coll_all_to_all_u32(gpe_ctx *c, const uint32_t *d_send, const uint64_t *send_off, const uint64_t *send_cnt,
                               uint32_t *d_recv, const uint64_t *recv_off, const uint64_t *recv_cnt, hipStream_t on)
{
    if (c->ctl.coll_set) {
        if (!c->ctl.coll.all_to_all_u32) return fail(c, GPE_ERR_STATE, "sharded run: the caller's collectives have no all_to_all_u32");
        if (c->ctl.coll.all_to_all_u32(c->ctl.coll.user, d_send, send_off, send_cnt, d_recv, recv_off, recv_cnt, (void *)(on ? on : c->stream)) != 0)
            return fail(c, GPE_ERR_HIP, "sharded run: the caller's all_to_all_u32 failed");
        return GPE_OK;
    }
    if (c->ctl.group) return group_all_to_all_u32(c, d_send, send_off, send_cnt, d_recv, recv_off, recv_cnt);
    if (c->shard.comm) return rccl_all_to_all_u32(c, d_send, send_off, send_cnt, d_recv, recv_off, recv_cnt);
    return fail(c, GPE_ERR_STATE, "sharded run: no collectives (gpe_shard_comm_init / _attach, gpe_local_group_join or "
                                  "gpe_shard_set_collectives)");
}

// all-reduce of a few host words (<= kCtlSmallWords) through the device scratch; synchronises
static gpe_status host_all_reduce(gpe_ctx *c, uint32_t *vals, uint64_t count, uint32_t op, uint32_t world_size)
{
    if (world_size <= 1) return GPE_OK;
    if (count > kCtlSmallWords) return fail(c, GPE_ERR_INVALID_ARG, "sharded run: small all-reduce too long");
    ShardCtl &T = c->ctl;
    if (!T.d_small) GPE_HIP(c, hipMalloc((void **)&T.d_small, kCtlSmallWords * sizeof(uint32_t)));
    GPE_HIP(c, hipMemcpyAsync(T.d_small, vals, count * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    GPE_TRY(coll_all_reduce_u32(c, T.d_small, count, op));
    GPE_HIP(c, hipMemcpyAsync(vals, T.d_small, count * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    return GPE_OK;
}

// Did every rank get here without an error?  A rank that fails alone would leave its peers waiting in the next
// collective: every rank reports, the maximum decides, all fail together.
static gpe_status agree(gpe_ctx *c, gpe_status mine, uint32_t world_size, const char *what)
{
    uint32_t bad = mine == GPE_OK ? 0u : 1u;
    const std::string why = c->last_error;
    const gpe_status st = host_all_reduce(c, &bad, 1, GPE_REDUCE_MAX, world_size);
    if (st != GPE_OK) return st;
    if (mine != GPE_OK) { c->last_error = why; return mine; }
    if (bad) return fail(c, GPE_ERR_STATE, std::string("sharded run: another rank failed in ") + what);
    return GPE_OK;
}

// ---------------------------------------------------------------------------------------------------------
// device kernels of the control plane
// ---------------------------------------------------------------------------------------------------------
// Re-sort: the owned particles are in (home-cell key, old global index) order.  hist[mb] += members of Morton block mb
// (one atomic per run of equal blocks inside a wave), first[mb] = local position of its first member.
__global__ __launch_bounds__(kStreamBlock) void k_ctl_block_hist(const uint32_t *__restrict__ keys, uint64_t n,
                                                                  uint32_t *__restrict__ hist, uint32_t *__restrict__ first)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t rounds = (n + stride - 1) / stride;
    for (uint64_t r = 0; r < rounds; ++r) {
        const uint64_t i = r * stride + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        const bool in = i < n;
        const uint32_t mb = in ? keys[i] >> 6 : 0xFFFFFFFFu;
        const uint32_t before = (in && i > 0) ? keys[i - 1] >> 6 : 0xFFFFFFFFu;
        if (in && (i == 0 || before != mb)) first[mb] = (uint32_t)i;     // the head of the block's run (runs are contiguous)
        // run heads inside the wave: a lane whose left neighbour (lane - 1) holds another block, or lane 0
        const uint32_t left = (uint32_t)__shfl_up((int)mb, 1, 64);
        const bool head = in && (lane_id() == 0 || left != mb);
        const uint64_t heads = ballot64(head);
        const uint64_t valid = ballot64(in);
        if (head) {
            // members of this run inside the wave: up to the next head (or the end of the valid lanes)
            const uint64_t above = heads & ~((2ull << lane_id()) - 1ull);
            const int end = above ? (int)__builtin_ctzll(above) : (int)__popcll(valid);
            atomicAdd(&hist[mb], (uint32_t)(end - lane_id()));
        }
    }
}

// new global index = particles of all ranks in earlier Morton blocks + position inside the block
__global__ __launch_bounds__(kStreamBlock) void k_ctl_assign(const uint32_t *__restrict__ keys, uint64_t n,
                                                             const uint32_t *__restrict__ incl,
                                                             const uint32_t *__restrict__ first,
                                                             uint32_t *__restrict__ gid)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t mb = keys[i] >> 6;
        const uint32_t base = mb ? incl[mb - 1] : 0u;
        gid[i] = base + ((uint32_t)i - first[mb]);
    }
}

__global__ __launch_bounds__(kStreamBlock) void k_ctl_copy_iota(const uint32_t *__restrict__ src, uint32_t *__restrict__ keys,
                                                                uint32_t *__restrict__ vals, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        keys[i] = src[i];
        vals[i] = (uint32_t)i;
    }
}

// Re-cut: particles per block column / block row (hx[0..bx), hy[0..by) behind it), LDS histograms per workgroup.
constexpr int kCutBins = 12288;                // block columns + rows (48 KB of LDS); larger block grids are not re-cut
__global__ __launch_bounds__(1024) void k_ctl_axis_hist(const float2 *__restrict__ pos, uint64_t n, float cell_size,
                                                        int32_t bx, int32_t by, uint32_t *__restrict__ out)
{
    __shared__ uint32_t s_h[kCutBins];
    const int bins = bx + by;
    for (int i = threadIdx.x; i < bins; i += blockDim.x) s_h[i] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float2 p = pos[i];
        int x = cell_coord(p.x, cell_size) >> 3, y = cell_coord(p.y, cell_size) >> 3;
        x = min(max(x, 0), bx - 1);
        y = min(max(y, 0), by - 1);
        atomicAdd(&s_h[x], 1u);
        atomicAdd(&s_h[bx + y], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += blockDim.x)
        if (s_h[i]) atomicAdd(&out[i], s_h[i]);
}

// destination rank of every owned particle under the NEW owner table (sort key), iota payload, and the count per rank
__global__ __launch_bounds__(kStreamBlock) void k_ctl_dest(const float2 *__restrict__ pos, uint64_t n, float cell_size,
                                                           const uint8_t *__restrict__ owner, int32_t bx, int32_t by,
                                                           uint32_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                           uint32_t *__restrict__ counts)
{
    __shared__ uint32_t s_c[32];
    if (threadIdx.x < 32) s_c[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float2 p = pos[i];
        int x = cell_coord(p.x, cell_size) >> 3, y = cell_coord(p.y, cell_size) >> 3;
        x = min(max(x, 0), bx - 1);
        y = min(max(y, 0), by - 1);
        const uint32_t d = owner[(size_t)y * bx + x];
        keys[i] = d;
        vals[i] = (uint32_t)i;
        atomicAdd(&s_c[d & 31u], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 32 && s_c[threadIdx.x]) atomicAdd(&counts[threadIdx.x], s_c[threadIdx.x]);
}

// rows on the move: x y prev_x prev_y r key, gathered through `order` (NULL: identity)
__global__ __launch_bounds__(kStreamBlock) void k_ctl_rows_pack(const float2 *__restrict__ pos, const float2 *__restrict__ prev,
                                                                const float *__restrict__ radius, const uint32_t *__restrict__ gid,
                                                                const uint32_t *__restrict__ order, uint64_t n,
                                                                uint32_t *__restrict__ rows)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const uint32_t i = order ? order[j] : (uint32_t)j;
        const float2 p = pos[i], q = prev[i];
        uint32_t *row = rows + j * 6;
        row[0] = __float_as_uint(p.x); row[1] = __float_as_uint(p.y);
        row[2] = __float_as_uint(q.x); row[3] = __float_as_uint(q.y);
        row[4] = __float_as_uint(radius[i]); row[5] = gid[i];
    }
}

__global__ __launch_bounds__(kStreamBlock) void k_ctl_rows_unpack(const uint32_t *__restrict__ rows, uint64_t n,
                                                                  float2 *__restrict__ pos, float2 *__restrict__ prev,
                                                                  float *__restrict__ radius, uint32_t *__restrict__ gid)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const uint32_t *row = rows + j * 6;
        pos[j] = make_float2(__uint_as_float(row[0]), __uint_as_float(row[1]));
        prev[j] = make_float2(__uint_as_float(row[2]), __uint_as_float(row[3]));
        radius[j] = __uint_as_float(row[4]);
        gid[j] = row[5];
    }
}

// ---------------------------------------------------------------------------------------------------------
// set-up pieces
// ---------------------------------------------------------------------------------------------------------
void ctl_release(gpe_ctx *c)
{
    ShardCtl &T = c->ctl;
    void *bufs[] = {T.d_owner, T.d_mask, T.d_send, T.d_recv, T.d_small, T.d_hist, T.d_first, T.d_rows_send, T.d_rows_recv};
    for (void *b : bufs) if (b) (void)hipFree(b);
    T = ShardCtl();
}

template <typename T>
static gpe_status ensure_words(gpe_ctx *c, T **buf, uint64_t *cap, uint64_t want, uint64_t unit = 1)
{
    if (*cap >= want && *buf) return GPE_OK;
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    if (*buf) GPE_HIP(c, hipFree(*buf));
    *buf = nullptr; *cap = 0;
    hipError_t e = hipMalloc((void **)buf, std::max<uint64_t>(want, 16) * unit * sizeof(T) + 64);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(c, e == hipErrorOutOfMemory ? GPE_ERR_OOM : GPE_ERR_HIP, std::string("sharded run: hipMalloc: ") + hipGetErrorName(e));
    }
    *cap = want;
    return GPE_OK;
}

// the active cell box of a rank: its own blocks plus the one-block ghost ring
static void active_cells(const gpe_shard_layout &L, uint32_t rank, int32_t box[4])
{
    int x0, y0, x1, y1;
    rect_blocks(L, rank, x0, y0, x1, y1);
    box[0] = std::max(0, (x0 - 1) * kBlock);
    box[1] = std::max(0, (y0 - 1) * kBlock);
    box[2] = std::min(L.cells_x - 1, (x1 + 1) * kBlock - 1);
    box[3] = std::min(L.cells_y - 1, (y1 + 1) * kBlock - 1);
}

// particles per owned block on the most crowded rank: every rank gets the same number (the maximum of non-negative
// floats is the maximum of their bit patterns)
static gpe_status densest_rank_per_block(gpe_ctx *c, double *out)
{
    const ShardCtl &T = c->ctl;
    int x0, y0, x1, y1;
    rect_blocks(T.layout, T.rank, x0, y0, x1, y1);
    const float d = (float)((double)c->n_owned / (double)std::max(1, (x1 - x0) * (y1 - y0)));
    uint32_t bits;
    memcpy(&bits, &d, 4);
    GPE_TRY(host_all_reduce(c, &bits, 1, GPE_REDUCE_MAX, T.layout.world_size));
    float m;
    memcpy(&m, &bits, 4);
    *out = (double)m;
    return GPE_OK;
}

// tables to the device, tile grid cut to the rank, neighbour segments sized, gpe_shard_configure.  Collective.
static gpe_status plan_exchange(gpe_ctx *c, bool new_layout)
{
    ShardCtl &T = c->ctl;
    const gpe_shard_layout &L = T.layout;
    const uint32_t ws = L.world_size, rank = T.rank;
    std::vector<uint8_t> owner;
    std::vector<uint32_t> mask;
    build_tables(L, owner, mask);
    const uint64_t blocks = (uint64_t)L.blocks_x * L.blocks_y;
    gpe_status st = GPE_OK;
    double per_block = 0.0;
    GPE_TRY(densest_rank_per_block(c, &per_block));
    T.planned_per_block = per_block;
    const std::vector<uint32_t> nb = neighbours(L, mask, rank);
    gpe_shard_plan plan;
    memset(&plan, 0, sizeof(plan));
    plan.struct_size = sizeof(plan);
    plan.rank = rank; plan.world_size = ws; plan.n_slots = (uint32_t)nb.size() + 1;
    plan.blocks_x = L.blocks_x; plan.blocks_y = L.blocks_y;
    uint64_t so = 0, ro = 0;
    std::vector<uint32_t> sent(ws * ws, 0);            // [src * ws + dst] = words src sends to dst (this rank's row)
    if (nb.size() > 8) st = fail(c, GPE_ERR_UNSUPPORTED, "sharded run: more than 8 neighbouring ranks");
    else if (min_region_blocks(L) < 2 && ws > 1) st = fail(c, GPE_ERR_UNSUPPORTED, "sharded run: every rectangle must be at least two blocks wide");
    if (st == GPE_OK) {
        memset(T.x_send_off, 0, sizeof(T.x_send_off)); memset(T.x_send_cnt, 0, sizeof(T.x_send_cnt));
        memset(T.x_recv_off, 0, sizeof(T.x_recv_off)); memset(T.x_recv_cnt, 0, sizeof(T.x_recv_cnt));
        uint64_t self_gho = 0;
        for (size_t s = 0; s < nb.size(); ++s) {
            const uint32_t p = nb[s];
            uint32_t cm, cg;
            segment_caps(L, mask, rank, p, per_block, T.scale, cm, cg);
            plan.slot_rank[s] = p; plan.send_off[s] = (uint32_t)so; plan.send_cap_mig[s] = cm; plan.send_cap_gho[s] = cg;
            T.x_send_off[p] = so; T.x_send_cnt[p] = segment_words(cm, cg);
            sent[rank * ws + p] = (uint32_t)segment_words(cm, cg);
            so += segment_words(cm, cg);
            self_gho += cm;
            segment_caps(L, mask, p, rank, per_block, T.scale, cm, cg);
            plan.recv_off[s] = (uint32_t)ro; plan.recv_cap_mig[s] = cm; plan.recv_cap_gho[s] = cg;
            T.x_recv_off[p] = ro; T.x_recv_cnt[p] = segment_words(cm, cg);
            ro += segment_words(cm, cg);
        }
        if (so + segment_words(0, (uint32_t)self_gho) + 16 > 0xFFFFFFFFull || ro + 16 > 0xFFFFFFFFull)
            st = fail(c, GPE_ERR_UNSUPPORTED, "sharded run: neighbour segments beyond 2^32 words");
        else {
            const size_t s = nb.size();                 // this rank's own segment: migrants that stay behind as ghosts; not sent
            plan.slot_rank[s] = rank; plan.send_off[s] = (uint32_t)so; plan.send_cap_mig[s] = 0; plan.send_cap_gho[s] = (uint32_t)self_gho;
            plan.recv_off[s] = (uint32_t)ro; plan.recv_cap_mig[s] = 0; plan.recv_cap_gho[s] = 0;
            T.n_neighbours = (uint32_t)nb.size();
            st = ensure_words(c, &T.d_send, &T.send_cap, so + segment_words(0, (uint32_t)self_gho) + 16);
            if (st == GPE_OK) st = ensure_words(c, &T.d_recv, &T.recv_cap, ro + 16);
            if (st == GPE_OK) st = ensure_words(c, &T.d_owner, &T.tables_cap, blocks);
            if (st == GPE_OK && (!T.d_mask || new_layout)) {
                if (T.d_mask) (void)hipFree(T.d_mask);
                T.d_mask = nullptr;
                if (hipMalloc((void **)&T.d_mask, std::max<uint64_t>(blocks, 16) * sizeof(uint32_t) + 64) != hipSuccess) {
                    (void)hipGetLastError();
                    st = fail(c, GPE_ERR_OOM, "sharded run: hipMalloc (destination masks)");
                }
            }
        }
    }
    GPE_TRY(agree(c, st, ws, "the planning of the neighbour segments"));
    // both ends of every pair must agree on the segment lengths: a send and a receive of different lengths wait for
    // each other instead of failing (RCCL) or scramble the rows
    if (ws > 1) {
        GPE_TRY(host_all_reduce(c, sent.data(), (uint64_t)ws * ws, GPE_REDUCE_SUM, ws));
        for (uint32_t p = 0; p < ws && st == GPE_OK; ++p)
            if (sent[p * ws + rank] != (uint32_t)T.x_recv_cnt[p]) {
                char msg[256];
                snprintf(msg, sizeof(msg), "sharded run: rank %u and rank %u disagree on the segment sizes (%u words sent, %llu "
                         "expected) -- the ranks were configured differently", p, rank, sent[p * ws + rank],
                         (unsigned long long)T.x_recv_cnt[p]);
                st = fail(c, GPE_ERR_INVALID_ARG, msg);
            }
        GPE_TRY(agree(c, st, ws, "the comparison of the segment sizes (the ranks disagree on the segment sizes)"));
    }
    GPE_HIP(c, hipMemcpyAsync(T.d_owner, owner.data(), blocks, hipMemcpyHostToDevice, c->stream));
    GPE_HIP(c, hipMemcpyAsync(T.d_mask, mask.data(), blocks * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    GPE_HIP(c, hipMemsetAsync(T.d_send, 0, T.send_cap * sizeof(uint32_t), c->stream));
    GPE_HIP(c, hipMemsetAsync(T.d_recv, 0, T.recv_cap * sizeof(uint32_t), c->stream));
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    plan.d_owner_of_block = T.d_owner; plan.d_dest_mask_of_block = T.d_mask;
    plan.d_send = T.d_send; plan.d_recv = T.d_recv;
    {
        int x0, y0, x1, y1;
        rect_blocks(L, rank, x0, y0, x1, y1);
        plan.own_x0 = x0; plan.own_y0 = y0; plan.own_x1 = x1; plan.own_y1 = y1;
    }
    if (new_layout) {
        int32_t box[4];
        active_cells(L, rank, box);
        GPE_TRY(gpe_set_active_cells(c, box[0], box[1], box[2], box[3]));
    }
    return gpe_shard_configure(c, &plan);
}

// every particle to its owner, ghosts dropped, counts back on the host
static gpe_status go_home(gpe_ctx *c)
{
    ShardCtl &T = c->ctl;
    if (T.home) return GPE_OK;
    if (!c->shard.active) {
        c->n = c->n_owned;
        GPE_TRY(gpe_shard_begin(c));
    }
    GPE_TRY(gpe_shard_exchange(c));
    GPE_TRY(gpe_shard_unpack(c));
    uint64_t no = 0, nt = 0;
    GPE_TRY(gpe_shard_counts(c, &no, &nt, 1));
    T.n_ghost = 0;
    T.home = true;
    return GPE_OK;
}

static gpe_status resort_local(gpe_ctx *c)
{
    ShardCtl &T = c->ctl;
    const uint64_t n = c->n_owned;
    c->n = n;
    // 1. into the order of the old global indices (the tie order of the key sort that follows)
    hipLaunchKernelGGL(k_ctl_copy_iota, dim3(stream_grid(n)), dim3(kStreamBlock), 0, c->stream, c->order_keys,
                       c->home_cell_ids, c->particle_ids, n);
    GPE_HIP(c, hipGetLastError());
    GPE_TRY(sort_reserve(c, n));
    GPE_TRY(sort_pairs(c, c->home_cell_ids, c->particle_ids, n));
    GPE_TRY(launch_rearrange(c, c->pos, c->prev, c->radius, c->particle_ids, n, c->pos_copy, c->prev_copy, c->radius_copy));
    std::swap(c->pos, c->pos_copy);
    std::swap(c->prev, c->prev_copy);
    std::swap(c->radius, c->radius_copy);
    // 2. the reference's re-sort on the owned particles: K1, stable sort by home-cell key, K4
    GPE_TRY(resort_for_shard(c));
    // 3. new global indices
    const uint64_t entries = morton_entries(T.layout);
    if (T.hist_cap < entries) {
        uint64_t cap1 = T.hist_cap, cap2 = T.hist_cap;
        GPE_TRY(ensure_words(c, &T.d_hist, &cap1, entries + 1));
        GPE_TRY(ensure_words(c, &T.d_first, &cap2, entries + 1));
        T.hist_cap = entries;
    }
    GPE_HIP(c, hipMemsetAsync(T.d_hist, 0, entries * sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(k_ctl_block_hist, dim3(stream_grid(n)), dim3(kStreamBlock), 0, c->stream, c->home_cell_ids, n,
                       T.d_hist, T.d_first);
    GPE_HIP(c, hipGetLastError());
    if (T.layout.world_size > 1) GPE_TRY(coll_all_reduce_u32(c, T.d_hist, entries, GPE_REDUCE_SUM));
    GPE_TRY(scan_reserve(c, entries));
    GPE_TRY(inclusive_scan(c, T.d_hist, entries));
    hipLaunchKernelGGL(k_ctl_assign, dim3(stream_grid(n)), dim3(kStreamBlock), 0, c->stream, c->home_cell_ids, n, T.d_hist,
                       T.d_first, c->order_keys);
    GPE_HIP(c, hipGetLastError());
    ++T.resorts;
    return GPE_OK;
}

static gpe_status recut_home(gpe_ctx *c, float above, int32_t *recut)
{
    ShardCtl &T = c->ctl;
    const gpe_shard_layout &L = T.layout;
    const uint32_t ws = L.world_size, rank = T.rank;
    if (recut) *recut = 0;
    if (!(above > 0.0f) || ws == 1) return GPE_OK;
    const uint64_t n = c->n_owned;
    std::vector<uint32_t> owned(ws, 0);
    owned[rank] = (uint32_t)n;
    GPE_TRY(host_all_reduce(c, owned.data(), ws, GPE_REDUCE_SUM, ws));
    uint64_t total = 0, most = 0;
    for (uint32_t v : owned) { total += v; most = std::max<uint64_t>(most, v); }
    if ((double)most <= (double)above * ((double)total / (double)ws)) return GPE_OK;
    // particle counts per block column / row over all ranks
    const int bins = L.blocks_x + L.blocks_y;
    if (bins > kCutBins) return GPE_OK;
    uint64_t cap = T.rows_send_cap;
    GPE_TRY(ensure_words(c, &T.d_rows_send, &cap, std::max<uint64_t>((uint64_t)bins, n * 6)));
    T.rows_send_cap = cap;
    GPE_HIP(c, hipMemsetAsync(T.d_rows_send, 0, (size_t)bins * sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(k_ctl_axis_hist, dim3(std::min(256, stream_grid(n, 1024))), dim3(1024), 0, c->stream, c->pos, n,
                       c->cell_size, L.blocks_x, L.blocks_y, T.d_rows_send);
    GPE_HIP(c, hipGetLastError());
    GPE_TRY(coll_all_reduce_u32(c, T.d_rows_send, (uint64_t)bins, GPE_REDUCE_SUM));
    std::vector<uint32_t> h32(bins);
    GPE_HIP(c, hipMemcpyAsync(h32.data(), T.d_rows_send, (size_t)bins * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    std::vector<uint64_t> hx(h32.begin(), h32.begin() + L.blocks_x), hy(h32.begin() + L.blocks_x, h32.end());
    int32_t xc[GPE_SHARD_MAX_RANKS + 1], yc[GPE_SHARD_MAX_RANKS + 1];
    GPE_TRY(gpe_shard_quantile_cuts(hx.data(), (uint32_t)L.blocks_x, L.px, 2, xc));
    GPE_TRY(gpe_shard_quantile_cuts(hy.data(), (uint32_t)L.blocks_y, L.py, 2, yc));
    if (memcmp(xc, L.xcuts, (L.px + 1) * sizeof(int32_t)) == 0 && memcmp(yc, L.ycuts, (L.py + 1) * sizeof(int32_t)) == 0)
        return GPE_OK;
    gpe_shard_layout NL;
    GPE_TRY(gpe_shard_layout_build(L.world_width, L.world_height, L.cell_size, ws, L.px, L.py, xc, yc, &NL));
    // new owner of every particle, particles grouped by it
    std::vector<uint8_t> owner;
    std::vector<uint32_t> mask;
    build_tables(NL, owner, mask);
    const uint64_t blocks = (uint64_t)NL.blocks_x * NL.blocks_y;
    GPE_TRY(ensure_words(c, &T.d_owner, &T.tables_cap, blocks));
    GPE_HIP(c, hipMemcpyAsync(T.d_owner, owner.data(), blocks, hipMemcpyHostToDevice, c->stream));
    if (!T.d_small) GPE_HIP(c, hipMalloc((void **)&T.d_small, kCtlSmallWords * sizeof(uint32_t)));
    GPE_HIP(c, hipMemsetAsync(T.d_small, 0, 32 * sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(k_ctl_dest, dim3(stream_grid(n)), dim3(kStreamBlock), 0, c->stream, c->pos, n, c->cell_size, T.d_owner,
                       NL.blocks_x, NL.blocks_y, c->home_cell_ids, c->particle_ids, T.d_small);
    GPE_HIP(c, hipGetLastError());
    uint32_t cnt32[32];
    GPE_HIP(c, hipMemcpyAsync(cnt32, T.d_small, sizeof(cnt32), hipMemcpyDeviceToHost, c->stream));
    GPE_TRY(sort_reserve(c, n));
    GPE_TRY(sort_pairs(c, c->home_cell_ids, c->particle_ids, n));
    hipLaunchKernelGGL(k_ctl_rows_pack, dim3(stream_grid(n)), dim3(kStreamBlock), 0, c->stream, c->pos, c->prev, c->radius,
                       c->order_keys, c->particle_ids, n, T.d_rows_send);
    GPE_HIP(c, hipGetLastError());
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    // the count matrix, then the rows themselves
    std::vector<uint32_t> mat(ws * ws, 0);
    for (uint32_t p = 0; p < ws; ++p) mat[rank * ws + p] = cnt32[p];
    GPE_TRY(host_all_reduce(c, mat.data(), (uint64_t)ws * ws, GPE_REDUCE_SUM, ws));
    uint64_t so[GPE_SHARD_MAX_RANKS], sc[GPE_SHARD_MAX_RANKS], ro[GPE_SHARD_MAX_RANKS], rc[GPE_SHARD_MAX_RANKS];
    uint64_t o = 0, n_new = 0;
    for (uint32_t p = 0; p < ws; ++p) { so[p] = o; sc[p] = 6ull * cnt32[p]; o += sc[p]; }
    for (uint32_t p = 0; p < ws; ++p) { ro[p] = 6ull * n_new; rc[p] = 6ull * mat[p * ws + rank]; n_new += mat[p * ws + rank]; }
    gpe_status st = GPE_OK;
    if (n_new == 0) st = fail(c, GPE_ERR_UNSUPPORTED, "sharded run: this rank owns no particle after the re-cut");
    cap = T.rows_recv_cap;
    if (st == GPE_OK) st = ensure_words(c, &T.d_rows_recv, &cap, n_new * 6);
    T.rows_recv_cap = cap;
    if (st == GPE_OK && n_new + 2 > c->cap) {
        const uint64_t need = (uint64_t)((double)n_new * 1.3) + 4096;
        c->n = std::max<uint64_t>(1, std::min<uint64_t>(n, c->cap));
        st = grow_for_shard(c, (uint64_t)((double)need * 1.25) + 4096);
    }
    GPE_TRY(agree(c, st, ws, "the re-cut"));
    GPE_TRY(coll_all_to_all_u32(c, T.d_rows_send, so, sc, T.d_rows_recv, ro, rc));
    hipLaunchKernelGGL(k_ctl_rows_unpack, dim3(stream_grid(n_new)), dim3(kStreamBlock), 0, c->stream, T.d_rows_recv, n_new,
                       c->pos, c->prev, c->radius, c->order_keys);
    GPE_HIP(c, hipGetLastError());
    c->n = c->n_owned = n_new;
    T.layout = NL;
    ++T.recuts;
    GPE_TRY(plan_exchange(c, true));
    if (recut) *recut = 1;
    return GPE_OK;
}

}  // namespace gpe

using namespace gpe;

extern "C" {

gpe_status gpe_shard_layout_build(float world_width, float world_height, float cell_size, uint32_t world_size,
                                  uint32_t px, uint32_t py, const int32_t *xcuts, const int32_t *ycuts,
                                  gpe_shard_layout *out)
{
    if (!out) return fail(nullptr, GPE_ERR_INVALID_ARG, "gpe_shard_layout_build: out is NULL");
    if (world_size < 1 || world_size > GPE_SHARD_MAX_RANKS)
        return fail(nullptr, GPE_ERR_INVALID_ARG, "gpe_shard_layout_build: 1 .. 26 ranks (destination masks are 26 bits wide)");
    if (!(cell_size > 0.0f) || !(world_width > 0.0f) || !(world_height > 0.0f))
        return fail(nullptr, GPE_ERR_INVALID_ARG, "gpe_shard_layout_build: world and cell size must be positive");
    gpe_shard_layout L;
    memset(&L, 0, sizeof(L));
    L.struct_size = sizeof(L);
    L.world_size = world_size;
    L.world_width = world_width; L.world_height = world_height; L.cell_size = cell_size;
    // the same arithmetic as native_configure: largest home coordinate = floor(world / cell), f32
    const float fx = floorf(world_width / cell_size), fy = floorf(world_height / cell_size);
    if (!(fx >= 0.0f) || !(fy >= 0.0f) || fx > 65000.0f || fy > 65000.0f)
        return fail(nullptr, GPE_ERR_INVALID_ARG, "gpe_shard_layout_build: more than 65000 cells along an axis");
    L.cells_x = (int32_t)fx + 1; L.cells_y = (int32_t)fy + 1;
    L.blocks_x = (L.cells_x + kBlock - 1) / kBlock; L.blocks_y = (L.cells_y + kBlock - 1) / kBlock;
    if (px == 0 && py == 0) {                                          // as square as possible, px <= py
        px = (uint32_t)floor(sqrt((double)world_size));
        while (px > 1 && world_size % px) --px;
        py = world_size / px;
    }
    if (px == 0 || py == 0 || px * py != world_size)
        return fail(nullptr, GPE_ERR_INVALID_ARG, "gpe_shard_layout_build: process grid does not match world_size");
    if ((int32_t)px > L.blocks_x || (int32_t)py > L.blocks_y)
        return fail(nullptr, GPE_ERR_INVALID_ARG, "gpe_shard_layout_build: world too small for this many ranks");
    L.px = px; L.py = py;
    // equal widths unless given: round half to even, like the planning arithmetic this replaces
    for (uint32_t i = 0; i <= px; ++i) L.xcuts[i] = xcuts ? xcuts[i] : (int32_t)nearbyint((double)i * (double)L.blocks_x / (double)px);
    for (uint32_t j = 0; j <= py; ++j) L.ycuts[j] = ycuts ? ycuts[j] : (int32_t)nearbyint((double)j * (double)L.blocks_y / (double)py);
    bool ok = L.xcuts[0] == 0 && L.ycuts[0] == 0 && L.xcuts[px] == L.blocks_x && L.ycuts[py] == L.blocks_y;
    for (uint32_t i = 0; i < px; ++i) ok = ok && L.xcuts[i + 1] > L.xcuts[i];
    for (uint32_t j = 0; j < py; ++j) ok = ok && L.ycuts[j + 1] > L.ycuts[j];
    if (!ok) return fail(nullptr, GPE_ERR_INVALID_ARG, "gpe_shard_layout_build: cuts must rise from 0 to the block count");
    *out = L;
    return GPE_OK;
}

gpe_status gpe_shard_layout_owner_of(const gpe_shard_layout *L, const float *pos_xy, uint64_t n, uint8_t *owner_out)
{
    if (!layout_valid(L) || (!pos_xy && n) || (!owner_out && n))
        return fail(nullptr, GPE_ERR_INVALID_ARG, "gpe_shard_layout_owner_of: bad argument");
    std::vector<uint8_t> col(L->blocks_x), row(L->blocks_y);
    for (uint32_t i = 0; i < L->px; ++i) for (int x = L->xcuts[i]; x < L->xcuts[i + 1]; ++x) col[x] = (uint8_t)i;
    for (uint32_t j = 0; j < L->py; ++j) for (int y = L->ycuts[j]; y < L->ycuts[j + 1]; ++y) row[y] = (uint8_t)j;
    const float cs = L->cell_size;
    for (uint64_t i = 0; i < n; ++i) {
        // floor(p / cell) >> 3 in f32, clamped to the block grid (k_shard_pack / k_shard_classify)
        const float fx = floorf(pos_xy[2 * i] / cs), fy = floorf(pos_xy[2 * i + 1] / cs);
        int64_t cx = fx != fx ? 0 : (fx >= 2147483648.0f ? 0x7fffffffLL : (fx <= -2147483648.0f ? -0x80000000LL : (int64_t)fx));
        int64_t cy = fy != fy ? 0 : (fy >= 2147483648.0f ? 0x7fffffffLL : (fy <= -2147483648.0f ? -0x80000000LL : (int64_t)fy));
        int64_t bx = cx >> 3, by = cy >> 3;
        bx = std::min<int64_t>(std::max<int64_t>(bx, 0), L->blocks_x - 1);
        by = std::min<int64_t>(std::max<int64_t>(by, 0), L->blocks_y - 1);
        owner_out[i] = (uint8_t)(row[by] * L->px + col[bx]);
    }
    return GPE_OK;
}

gpe_status gpe_shard_quantile_cuts(const uint64_t *hist, uint32_t bins, uint32_t parts, uint32_t min_width, int32_t *cuts_out)
{
    if (!hist || !cuts_out || parts < 1 || parts > GPE_SHARD_MAX_RANKS || bins < parts)
        return fail(nullptr, GPE_ERR_INVALID_ARG, "gpe_shard_quantile_cuts: bad argument");
    int64_t nb = bins, mw = min_width;
    if ((int64_t)parts * mw > nb) mw = std::max<int64_t>(1, nb / parts);
    std::vector<double> cum(bins);
    double run = 0.0;
    for (uint32_t i = 0; i < bins; ++i) { run += (double)hist[i]; cum[i] = run; }
    const double total = bins ? cum[bins - 1] : 0.0;
    cuts_out[0] = 0;
    for (uint32_t i = 1; i < parts; ++i) {
        int64_t cut;
        if (total > 0.0) {
            const double want = total * (double)i / (double)parts;
            cut = (int64_t)(std::lower_bound(cum.begin(), cum.end(), want) - cum.begin()) + 1;   // searchsorted(side="left") + 1
        } else {
            cut = (int64_t)nearbyint((double)i * (double)nb / (double)parts);
        }
        cut = std::max<int64_t>(cut, (int64_t)cuts_out[i - 1] + mw);
        cut = std::min<int64_t>(cut, nb - (int64_t)(parts - i) * mw);
        cuts_out[i] = (int32_t)cut;
    }
    cuts_out[parts] = (int32_t)nb;
    return GPE_OK;
}

gpe_status gpe_shard_set_collectives(gpe_ctx *c, const gpe_shard_collectives *coll)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    if (!coll) { c->ctl.coll_set = false; return GPE_OK; }
    if (coll->struct_size != sizeof(gpe_shard_collectives)) return fail(c, GPE_ERR_INVALID_ARG, "gpe_shard_set_collectives: bad struct_size");
    c->ctl.coll = *coll;
    c->ctl.coll_set = true;
    return GPE_OK;
}

gpe_status gpe_shard_set_particles(gpe_ctx *c, const float *pos_xy, const float *prev_xy, const float *radius,
                                   const uint32_t *order_key, uint64_t n, uint64_t capacity)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    if (!order_key) return fail(c, GPE_ERR_INVALID_ARG, "gpe_shard_set_particles: order_key is NULL");
    GPE_TRY(gpe_set_particles(c, pos_xy, prev_xy, radius, n));
    GPE_TRY(gpe_use_order_keys(c, 1));
    if (capacity == 0) capacity = (uint64_t)((double)n * 1.3) + 4096;
    GPE_TRY(gpe_reserve(c, std::max(capacity, n)));
    GPE_HIP(c, hipMemcpyAsync(c->order_keys, order_key, n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    c->ctl.ready = false;
    c->ctl.home = false;
    return GPE_OK;
}

gpe_status gpe_shard_setup(gpe_ctx *c, const gpe_shard_layout *layout, uint32_t rank, float capacity_scale)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    if (!layout_valid(layout) || rank >= layout->world_size)
        return fail(c, GPE_ERR_INVALID_ARG, "gpe_shard_setup: bad layout or rank");
    if (c->n == 0 || !c->pos) return fail(c, GPE_ERR_STATE, "no particles: call gpe_shard_set_particles first");
    if (!(capacity_scale > 0.0f)) capacity_scale = 1.0f;
    GPE_HIP(c, hipSetDevice(c->device));
    ShardCtl &T = c->ctl;
    const uint32_t ws = layout->world_size;
    if (ws > 1 && transport_kind(c) == 0)
        return fail(c, GPE_ERR_STATE, "gpe_shard_setup: no collectives (gpe_shard_comm_init / _attach, gpe_local_group_join or "
                                      "gpe_shard_set_collectives)");
    T.layout = *layout;
    T.rank = rank;
    T.scale = capacity_scale;
    T.ready = true;
    T.home = false;
    T.resorts = 0; T.steps = 0; T.recuts = 0; T.n_ghost = 0;
    // The cell size is 2.2 x the largest radius of the WHOLE system (grid.rs:159-161); a context only saw its own
    // particles.  Every rank takes the maximum over the ranks (Grid::new's max_obj_radius) and checks that the layout
    // was cut with that cell size -- a rank on a different grid would exchange nonsense.
    float mine = fabsf(c->max_radius);
    uint32_t words[2];
    memcpy(&words[0], &mine, 4);
    // (every rank must have been handed the same layout: its digest travels with the radius)
    uint32_t digest = 2166136261u;
    for (size_t i = 0; i < sizeof(gpe_shard_layout); ++i) digest = (digest ^ ((const uint8_t *)layout)[i]) * 16777619u;
    words[1] = digest;
    uint32_t lo[2] = {~words[0], ~words[1]};
    GPE_TRY(host_all_reduce(c, words, 2, GPE_REDUCE_MAX, ws));
    GPE_TRY(host_all_reduce(c, lo, 2, GPE_REDUCE_MAX, ws));
    gpe_status st = GPE_OK;
    if (words[1] != (uint32_t)~lo[1]) st = fail(c, GPE_ERR_INVALID_ARG, "gpe_shard_setup: the ranks were handed different layouts");
    float gmax;
    memcpy(&gmax, &words[0], 4);
    if (st == GPE_OK && gmax != fabsf(c->grid_max_radius)) st = gpe_grid_set_max_radius(c, gmax);
    if (st == GPE_OK && c->cell_size != layout->cell_size) {
        char msg[256];
        snprintf(msg, sizeof(msg), "gpe_shard_setup: the layout was cut with cell size %.9g, the system's is %.9g (2.2 x the "
                 "largest radius over all ranks)", (double)layout->cell_size, (double)c->cell_size);
        st = fail(c, GPE_ERR_INVALID_ARG, msg);
    }
    if (st == GPE_OK) st = gpe_use_order_keys(c, 1);
    GPE_TRY(agree(c, st, ws, "gpe_shard_setup"));
    return plan_exchange(c, true);
}

gpe_status gpe_shard_get_layout(const gpe_ctx *c, gpe_shard_layout *out)
{
    if (!c || !out) return GPE_ERR_INVALID_ARG;
    if (!c->ctl.ready) return fail(const_cast<gpe_ctx *>(c), GPE_ERR_STATE, "gpe_shard_get_layout: call gpe_shard_setup first");
    *out = c->ctl.layout;
    return GPE_OK;
}

static gpe_status ctl_ready(gpe_ctx *c)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    if (!c->ctl.ready || !c->shard.on) return fail(c, GPE_ERR_STATE, "sharded run: call gpe_shard_setup first");
    GPE_HIP(c, hipSetDevice(c->device));
    return GPE_OK;
}

gpe_status gpe_shard_resort(gpe_ctx *c)
{
    GPE_TRY(ctl_ready(c));
    GPE_TRY(go_home(c));
    return resort_local(c);
}

gpe_status gpe_shard_recut(gpe_ctx *c, float above, int32_t *recut)
{
    GPE_TRY(ctl_ready(c));
    GPE_TRY(go_home(c));
    return recut_home(c, above, recut);
}

gpe_status gpe_shard_run_scheduled(gpe_ctx *c, float dt, uint64_t steps, uint64_t resort_every, int32_t resort_first)
{
    GPE_TRY(ctl_ready(c));
    ShardCtl &T = c->ctl;
    uint64_t s = 0;
    while (s < steps) {
        const bool resort = (s == 0 && resort_first) || (resort_every && s > 0 && (s % resort_every) == 0);
        if (resort) {
            GPE_TRY(go_home(c));
            int32_t recut = 0;
            GPE_TRY(recut_home(c, 1.25f, &recut));
            GPE_TRY(resort_local(c));
            // the scene may have piled up on some ranks since the segments were sized: re-plan (collectively)
            if (!recut) {
                double d = 0.0;
                GPE_TRY(densest_rank_per_block(c, &d));
                if (d > 1.5 * T.planned_per_block) GPE_TRY(plan_exchange(c, false));
            }
        }
        if (!c->shard.active) {
            c->n = c->n_owned;
            GPE_TRY(gpe_shard_begin(c));
        }
        const uint64_t nxt = resort_every ? std::min(steps, (s / resort_every + 1) * resort_every) : steps;
        T.home = false;
        GPE_TRY(gpe_shard_run(c, dt, nxt - s));
        T.steps += nxt - s;
        s = nxt;
    }
    return GPE_OK;
}

gpe_status gpe_shard_download_owned(gpe_ctx *c, uint32_t *order_key_out, float *pos_xy_out, float *prev_xy_out,
                                    uint64_t capacity, uint64_t *n_owned)
{
    if (!c || !n_owned) return GPE_ERR_INVALID_ARG;
    if (c->n == 0 || !c->pos) return fail(c, GPE_ERR_STATE, "no particles");
    GPE_HIP(c, hipSetDevice(c->device));
    uint64_t no = c->n_owned, nt = c->n_owned;
    if (c->shard.on && c->shard.active) GPE_TRY(gpe_shard_counts(c, &no, &nt, 0));
    else GPE_TRY(gpe_sync(c));
    c->ctl.n_ghost = nt - no;
    *n_owned = no;
    const uint64_t k = std::min(no, capacity);
    if (order_key_out) GPE_HIP(c, hipMemcpyAsync(order_key_out, c->order_keys, k * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    if (pos_xy_out) GPE_HIP(c, hipMemcpyAsync(pos_xy_out, c->pos, k * sizeof(float2), hipMemcpyDeviceToHost, c->stream));
    if (prev_xy_out) GPE_HIP(c, hipMemcpyAsync(prev_xy_out, c->prev, k * sizeof(float2), hipMemcpyDeviceToHost, c->stream));
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    return GPE_OK;
}

gpe_status gpe_shard_get_stats(gpe_ctx *c, gpe_shard_stats *out)
{
    if (!c || !out) return GPE_ERR_INVALID_ARG;
    if (out->struct_size == 0 || out->struct_size > sizeof(gpe_shard_stats))
        return fail(c, GPE_ERR_INVALID_ARG, "gpe_shard_get_stats: bad struct_size");
    gpe_shard_stats s;
    memset(&s, 0, sizeof(s));
    s.struct_size = out->struct_size;
    s.recuts = c->ctl.recuts; s.resorts = c->ctl.resorts; s.steps = c->ctl.steps;
    s.n_owned = c->n_owned; s.n_ghost = c->ctl.n_ghost;
    s.n_neighbours = c->ctl.n_neighbours;
    s.transport = (uint32_t)transport_kind(c);
    memcpy(out, &s, out->struct_size);
    return GPE_OK;
}

}  // extern "C"
