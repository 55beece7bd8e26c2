// gpe_api.hip -- the extern "C" boundary of include/gpe.h: context, buffers, step ordering,
// downloads, profiling.  All device work goes to one in-order hipStream per context.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>

#include "gpe_internal.h"

namespace gpe {

static std::mutex g_err_mu;
static std::string g_last_error;

gpe_status fail(gpe_ctx *ctx, gpe_status code, const std::string &msg)
{
    if (ctx) ctx->last_error = msg;
    std::lock_guard<std::mutex> lk(g_err_mu);
    g_last_error = msg;
    return code;
}

// ---- profiling scopes ---------------------------------------------------------------------------
static hipEvent_t take_event(gpe_ctx *c)
{
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

static int stat_index(gpe_ctx *c, const char *name)
{
    for (size_t i = 0; i < c->stats.size(); ++i)
        if (c->stats[i].name == name) return (int)i;
    ScopeStat s;
    s.name = name;
    c->stats.push_back(s);
    return (int)c->stats.size() - 1;
}

Scope::Scope(gpe_ctx *ctx, const char *name) : ctx_(ctx)
{
    if (!ctx_ || !ctx_->profiling) return;
    stat_ = stat_index(ctx_, name);
    start_ = take_event(ctx_);
    if (start_) (void)hipEventRecord(start_, ctx_->stream);
}

Scope::~Scope()
{
    if (!ctx_ || !start_) return;
    hipEvent_t stop = take_event(ctx_);
    if (!stop) { ctx_->event_pool.push_back(start_); return; }
    (void)hipEventRecord(stop, ctx_->stream);
    PendingEvent p;
    p.stat = stat_;
    p.start = start_;
    p.stop = stop;
    ctx_->pending.push_back(p);
}

constexpr size_t kTraceCap = 1u << 16;

static void resolve_pending(gpe_ctx *c)
{
    for (const PendingEvent &p : c->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
            c->stats[p.stat].total_ms += ms;
            c->stats[p.stat].calls += 1;
            float t0 = 0.f;
            if (c->trace_origin && hipEventElapsedTime(&t0, c->trace_origin, p.start) == hipSuccess) {
                if (c->trace.size() >= kTraceCap) c->trace.erase(c->trace.begin(), c->trace.begin() + kTraceCap / 2);
                TraceEvent e;
                e.stat = p.stat; e.start_ms = t0; e.dur_ms = ms;
                c->trace.push_back(e);
            }
        }
        c->event_pool.push_back(p.start);
        c->event_pool.push_back(p.stop);
    }
    c->pending.clear();
}

// ---- buffers --------------------------------------------------------------------------------------
template <typename T>
static gpe_status dev_alloc(gpe_ctx *c, T **p, uint64_t count)
{
    *p = nullptr;
    hipError_t e = hipMalloc((void **)p, std::max<uint64_t>(count, 4) * sizeof(T) + 64);
    if (e != hipSuccess) (void)hipGetLastError();   // the failure is reported here: do not leave it for the next launch check
    if (e == hipErrorOutOfMemory) return fail(c, GPE_ERR_OOM, "hipMalloc: out of device memory");
    if (e != hipSuccess) return fail(c, GPE_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorName(e));
    return GPE_OK;
}

template <typename T>
static void dev_free(T *&p)
{
    if (p) (void)hipFree(p);
    p = nullptr;
}

static void free_particle_buffers(gpe_ctx *c)
{
    dev_free(c->pos); dev_free(c->prev); dev_free(c->radius);
    dev_free(c->pos_copy); dev_free(c->prev_copy); dev_free(c->radius_copy);
    dev_free(c->home_cell_ids); dev_free(c->particle_ids);
    dev_free(c->cell_ids); dev_free(c->object_ids);
    dev_free(c->chunk_obj_count); dev_free(c->collision_cells); dev_free(c->indirect_args);
    dev_free(c->order_keys);
    c->cap = 0;
}

__global__ void k_fill_u32(uint32_t *p, uint64_t n, uint32_t v)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}
__global__ void k_iota_u32(uint32_t *p, uint64_t lo, uint64_t hi)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += stride)
        p[i] = (uint32_t)i;
}

static gpe_status fill_u32(gpe_ctx *c, uint32_t *p, uint64_t n, uint32_t v)
{
    if (n == 0) return GPE_OK;
    hipLaunchKernelGGL(k_fill_u32, dim3(stream_grid(n)), dim3(kStreamBlock), 0, c->stream, p, n, v);
    GPE_HIP(c, hipGetLastError());
    return GPE_OK;
}

static uint64_t total_cell_ids(const gpe_ctx *c) { return c->n * GPE_MAX_CELLS_PER_OBJECT; }
static uint64_t num_chunks(const gpe_ctx *c)
{
    return (total_cell_ids(c) + GPE_COUNTING_CHUNK_SIZE - 1) / GPE_COUNTING_CHUNK_SIZE;
}

// The reference's grid / collision-cell buffers (4 entries per particle: 52 B per particle) and the sort
// partners for 4N pairs (32 B per particle) are touched by the compat kernels only.  A COMPAT context allocates
// them with the particles; a NATIVE context when something first asks for them: a per-module entry point
// (gpe_grid_*, gpe_build_collision_cells, gpe_solve_collisions), array access, gpe_set_mode(COMPAT), or a step
// the native kernels hand over (particles outside the box, over-dense windows) -- the one place where that costs
// an allocation on the step path, once.  At 100 M particles that is 8.4 GB of 15 that stay unallocated.
static gpe_status alloc_grid_buffers(gpe_ctx *c, uint64_t cap)
{
    gpe_status st = dev_alloc(c, &c->cell_ids, cap * 4);
    if (st == GPE_OK) st = dev_alloc(c, &c->object_ids, cap * 4);
    if (st == GPE_OK) st = dev_alloc(c, &c->chunk_obj_count, cap);
    if (st == GPE_OK) st = dev_alloc(c, &c->collision_cells, cap * 4);
    if (st == GPE_OK) st = dev_alloc(c, &c->indirect_args, 4);
    if (st == GPE_OK) st = sort_reserve(c, cap * 4);
    if (st != GPE_OK) {
        dev_free(c->cell_ids); dev_free(c->object_ids); dev_free(c->chunk_obj_count);
        dev_free(c->collision_cells); dev_free(c->indirect_args);
    }
    return st;
}

// Allocate every particle-count-dependent buffer for `cap` particles (State::new, state.rs:34-70).
static gpe_status alloc_particle_buffers(gpe_ctx *c, uint64_t cap, bool with_grid)
{
    GPE_TRY(dev_alloc(c, &c->pos, cap));
    GPE_TRY(dev_alloc(c, &c->prev, cap));
    GPE_TRY(dev_alloc(c, &c->radius, cap));
    GPE_TRY(dev_alloc(c, &c->pos_copy, cap));
    GPE_TRY(dev_alloc(c, &c->prev_copy, cap));
    GPE_TRY(dev_alloc(c, &c->radius_copy, cap));
    GPE_TRY(dev_alloc(c, &c->home_cell_ids, cap));
    GPE_TRY(dev_alloc(c, &c->particle_ids, cap));
    GPE_TRY(dev_alloc(c, &c->order_keys, cap));
    c->cap = cap;
    if (with_grid) GPE_TRY(alloc_grid_buffers(c, cap));
    else GPE_TRY(sort_reserve(c, cap));                                // the Morton re-sort's N pairs
    GPE_TRY(scan_reserve(c, cap));
    return GPE_OK;
}

// Initial values of the index buffers for particles [lo, hi).
static gpe_status init_index_buffers(gpe_ctx *c, uint64_t lo, uint64_t hi)
{
    if (hi <= lo) return GPE_OK;
    const uint64_t cnt = hi - lo;
    GPE_TRY(fill_u32(c, c->home_cell_ids + lo, cnt, kUnused));              // particle_system.rs:130-133
    hipLaunchKernelGGL(k_iota_u32, dim3(stream_grid(cnt)), dim3(kStreamBlock), 0, c->stream,
                       c->particle_ids, lo, hi);                            // particle_sort.rs:30
    GPE_HIP(c, hipGetLastError());
    if (!c->cell_ids) return GPE_OK;                                        // not allocated yet: need_grid_buffers
    GPE_TRY(fill_u32(c, c->cell_ids + 4 * lo, 4 * cnt, kUnused));           // grid.rs:80-83
    GPE_TRY(fill_u32(c, c->object_ids + 4 * lo, 4 * cnt, 0u));              // grid.rs:85-89
    GPE_TRY(fill_u32(c, c->collision_cells + 4 * lo, 4 * cnt, kUnused));    // collision_cell_buffers.rs:23-27
    GPE_TRY(fill_u32(c, c->chunk_obj_count + lo, cnt, 0u));                 // collision_cell_buffers.rs:17-21
    return GPE_OK;
}

// Every user of the grid / collision-cell buffers calls this first (see alloc_grid_buffers).
static gpe_status need_grid_buffers(gpe_ctx *c)
{
    if (c->cell_ids || c->cap == 0) return GPE_OK;
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    GPE_TRY(alloc_grid_buffers(c, c->cap));
    const uint64_t cnt = c->n;
    GPE_TRY(fill_u32(c, c->cell_ids, 4 * cnt, kUnused));
    GPE_TRY(fill_u32(c, c->object_ids, 4 * cnt, 0u));
    GPE_TRY(fill_u32(c, c->collision_cells, 4 * cnt, kUnused));
    GPE_TRY(fill_u32(c, c->chunk_obj_count, cnt, 0u));
    return GPE_OK;
}

// Reallocate every particle-count-dependent buffer for `cap` particles, keeping the contents of the
// first c->n (GpuBuffer::push grows x2 with a device copy, utils/gpu_buffer.rs:49-87).  Synchronises.
// Transactional: the new set is allocated beside the old one and takes its place only when every allocation and
// copy has succeeded; on failure the new set is freed and the context is exactly as before.
struct ParticleBufferSet {
    float2 *pos = nullptr, *prev = nullptr, *pos_copy = nullptr, *prev_copy = nullptr;
    float *radius = nullptr, *radius_copy = nullptr;
    uint32_t *home_cell_ids = nullptr, *particle_ids = nullptr, *cell_ids = nullptr, *object_ids = nullptr,
             *chunk_obj_count = nullptr, *collision_cells = nullptr, *indirect_args = nullptr, *order_keys = nullptr;
    uint64_t cap = 0;
};
static ParticleBufferSet take_buffers(gpe_ctx *c)
{
    ParticleBufferSet b;
    b.pos = c->pos; b.prev = c->prev; b.pos_copy = c->pos_copy; b.prev_copy = c->prev_copy;
    b.radius = c->radius; b.radius_copy = c->radius_copy;
    b.home_cell_ids = c->home_cell_ids; b.particle_ids = c->particle_ids; b.cell_ids = c->cell_ids;
    b.object_ids = c->object_ids; b.chunk_obj_count = c->chunk_obj_count; b.collision_cells = c->collision_cells;
    b.indirect_args = c->indirect_args; b.order_keys = c->order_keys;
    b.cap = c->cap;
    c->pos = c->prev = c->pos_copy = c->prev_copy = nullptr;
    c->radius = c->radius_copy = nullptr;
    c->home_cell_ids = c->particle_ids = c->cell_ids = c->object_ids = nullptr;
    c->chunk_obj_count = c->collision_cells = c->indirect_args = c->order_keys = nullptr;
    c->cap = 0;
    return b;
}
static void put_buffers(gpe_ctx *c, const ParticleBufferSet &b)
{
    c->pos = b.pos; c->prev = b.prev; c->pos_copy = b.pos_copy; c->prev_copy = b.prev_copy;
    c->radius = b.radius; c->radius_copy = b.radius_copy;
    c->home_cell_ids = b.home_cell_ids; c->particle_ids = b.particle_ids; c->cell_ids = b.cell_ids;
    c->object_ids = b.object_ids; c->chunk_obj_count = b.chunk_obj_count; c->collision_cells = b.collision_cells;
    c->indirect_args = b.indirect_args; c->order_keys = b.order_keys;
    c->cap = b.cap;
}

static gpe_status copy_into_new_buffers(gpe_ctx *c, const ParticleBufferSet &old, uint64_t old_n)
{
    if (!old.pos || old_n == 0) return GPE_OK;
#define GPE_COPY_OLD(field, count)                                                                     \
    GPE_HIP(c, hipMemcpyAsync(c->field, old.field, (count) * sizeof(*c->field), hipMemcpyDeviceToDevice, c->stream))
    GPE_COPY_OLD(pos, old_n); GPE_COPY_OLD(prev, old_n); GPE_COPY_OLD(radius, old_n);
    GPE_COPY_OLD(home_cell_ids, old_n); GPE_COPY_OLD(particle_ids, old_n);
    if (old.cell_ids) {
        GPE_COPY_OLD(cell_ids, 4 * old_n); GPE_COPY_OLD(object_ids, 4 * old_n);
        GPE_COPY_OLD(chunk_obj_count, old_n); GPE_COPY_OLD(collision_cells, 4 * old_n);
        GPE_COPY_OLD(indirect_args, 3);
    }
    if (old.order_keys) GPE_COPY_OLD(order_keys, old_n);
#undef GPE_COPY_OLD
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    return GPE_OK;
}

static gpe_status grow_particle_buffers(gpe_ctx *c, uint64_t cap)
{
    const uint64_t old_n = c->n;
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    const ParticleBufferSet old = take_buffers(c);                     // the context now holds no particle buffer
    gpe_status st = alloc_particle_buffers(c, cap, old.cell_ids != nullptr || c->cfg.mode != GPE_MODE_NATIVE);
    if (st == GPE_OK) st = copy_into_new_buffers(c, old, old_n);
    if (st != GPE_OK) {
        const std::string why = c->last_error;
        free_particle_buffers(c);                                      // whatever part of the new set exists
        put_buffers(c, old);                                           // the context is as it was
        c->last_error = why;
        return st;
    }
    float2 *f2[] = {old.pos, old.prev, old.pos_copy, old.prev_copy};
    for (float2 *p : f2) if (p) (void)hipFree(p);
    float *f1[] = {old.radius, old.radius_copy};
    for (float *p : f1) if (p) (void)hipFree(p);
    uint32_t *u[] = {old.home_cell_ids, old.particle_ids, old.cell_ids, old.object_ids, old.chunk_obj_count,
                     old.collision_cells, old.indirect_args, old.order_keys};
    for (uint32_t *p : u) if (p) (void)hipFree(p);
    return GPE_OK;
}

static float max_abs_radius(const float *radius, uint64_t n, float start)
{
    // particle_system.rs:51: the radius of largest magnitude (the element itself, sign kept; of several elements
    // of that magnitude the last one -- Rust's max_by -- which decides the sign of cell_size for radii like [2, -2])
    float best = start;
    for (uint64_t i = 0; i < n; ++i)
        if (!(fabsf(radius[i]) < fabsf(best))) best = radius[i];      // ties: the LAST element, as Iterator::max_by returns
    return best;
}

static void refresh_cell_size(gpe_ctx *c)
{
    c->cell_size = c->grid_max_radius * c->cfg.cell_size_multiplier;        // grid.rs:159-161
}

static gpe_status need_particles(gpe_ctx *c)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    if (c->n == 0 || !c->pos) return fail(c, GPE_ERR_STATE, "no particles: call gpe_set_particles first");
    return GPE_OK;
}

// ---- step pieces ------------------------------------------------------------------------------------
static gpe_status do_resort(gpe_ctx *c)
{
    // particle_sort.rs:58-69
    GPE_TRY(launch_home_cell_ids(c, c->pos, c->n, c->cell_size, c->home_cell_ids, c->particle_ids));
    {
        Scope s(c, "Particle sort");   // particle_sort.rs:64
        GPE_TRY(sort_pairs(c, c->home_cell_ids, c->particle_ids, c->n));
    }
    GPE_TRY(launch_rearrange(c, c->pos, c->prev, c->radius, c->particle_ids, c->n, c->pos_copy,
                             c->prev_copy, c->radius_copy));
    // particle_rearrange.rs:205-238 copies the copy set back; swapping the two sets is equivalent
    std::swap(c->pos, c->pos_copy);
    std::swap(c->prev, c->prev_copy);
    std::swap(c->radius, c->radius_copy);
    return GPE_OK;
}

static gpe_status do_grid_sort(gpe_ctx *c)
{
    GPE_TRY(need_grid_buffers(c));
    Scope s(c, "Sort map");   // grid.rs:329
    return sort_pairs(c, c->cell_ids, c->object_ids, total_cell_ids(c));
}

static gpe_status do_build_collision_cells(gpe_ctx *c)
{
    // collision_cell_builder.rs:211-236
    GPE_TRY(need_grid_buffers(c));
    GPE_TRY(launch_count_chunks(c, c->cell_ids, total_cell_ids(c), c->chunk_obj_count));
    {
        Scope s(c, "Collision cell prefix sum");   // collision_cell_builder.rs:227
        GPE_TRY(inclusive_scan(c, c->chunk_obj_count, num_chunks(c)));
    }
    GPE_TRY(launch_build_collision_cells(c, c->cell_ids, total_cell_ids(c), c->chunk_obj_count,
                                         num_chunks(c), c->collision_cells, c->indirect_args));
    return GPE_OK;
}

static gpe_status do_solve_colors(gpe_ctx *c)
{
    GPE_TRY(need_grid_buffers(c));
    for (uint32_t color = 1; color <= 4; ++color)   // collision_solver.rs:224
        GPE_TRY(launch_solve_color(c, c->collision_cells, c->chunk_obj_count, num_chunks(c), c->cell_ids,
                                   c->object_ids, total_cell_ids(c), c->pos, c->radius, c->cfg.stiffness,
                                   color));
    return GPE_OK;
}

static gpe_status do_step_scoped(gpe_ctx *c, float dt, uint32_t flags)
{
    // state.rs:115-131
    if (flags & GPE_STEP_RESORT) GPE_TRY(do_resort(c));                          // :122-125
    const bool native = native_should_run(c);
    if (c->use_order_keys && !native)
        return fail(c, GPE_ERR_UNSUPPORTED,
                    "order keys (sharded run) need the native pipeline: mode NATIVE, particles inside the world "
                    "box, bounded density");
    (native ? c->native.native_steps : c->native.compat_steps) += 1;
    if (native) {
        // grid update + collision solve as N-key sort + LDS cell windows (k_native.hip); the resolved
        // positions land in the scratch set, which then becomes the live one.  The integration (:130) is
        // applied as the tiles write their particles back -- same arithmetic, one pass over memory less.
        const VerletParams vp = verlet_params(c, dt);
        GPE_TRY(native_collide(c, c->pos, c->pos_copy, &vp));
        std::swap(c->pos, c->pos_copy);
        return GPE_OK;
    }
    GPE_TRY(need_grid_buffers(c));
    GPE_TRY(launch_build_cell_ids(c, c->pos, c->radius, c->n, c->cell_size, c->cell_ids,
                                  c->object_ids));                               // :126 Grid::update
    GPE_TRY(do_grid_sort(c));
    GPE_TRY(do_build_collision_cells(c));                                        // :127
    GPE_TRY(do_solve_colors(c));
    GPE_TRY(launch_verlet(c, c->pos, c->prev, c->radius, c->n_owned, dt));       // :130
    return GPE_OK;
}

// Sampled profiling (gpe_set_profiling(ctx, k > 1)): only every k-th step records its scopes -- an event pair
// per kernel costs more than some of the kernels at small particle counts.
static gpe_status do_step(gpe_ctx *c, float dt, uint32_t flags)
{
    if (c->profile_every > 1) c->profiling = (c->profile_step++ % c->profile_every) == 0;
    const gpe_status st = do_step_scoped(c, dt, flags);
    if (c->profile_every > 1) c->profiling = true;
    return st;
}

gpe_status step_for_shard(gpe_ctx *c, float dt) { return do_step(c, dt, 0u); }
gpe_status resort_for_shard(gpe_ctx *c) { return do_resort(c); }
gpe_status grow_for_shard(gpe_ctx *c, uint64_t capacity) { return grow_particle_buffers(c, capacity); }

}  // namespace gpe

using namespace gpe;

// =====================================================================================================
// extern "C"
// =====================================================================================================
extern "C" {

uint32_t gpe_abi_version(void) { return GPE_ABI_VERSION; }

gpe_status gpe_config_default(gpe_config *cfg)
{
    if (!cfg) return GPE_ERR_INVALID_ARG;
    memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = (uint32_t)sizeof(gpe_config);
    cfg->device = -1;
    cfg->world_width = 3048.0f;          // state.rs:35
    cfg->world_height = 1048.0f;
    cfg->gravity_x = 0.0f;               // particle_integration.wgsl:21
    cfg->gravity_y = 0.0f;
    cfg->cell_size_multiplier = 2.2f;    // grid.rs:20
    cfg->stiffness = 0.6f;               // collision_solver.wgsl:2
    cfg->mouse_strength = 150.0f;        // particle_integration.wgsl:22
    cfg->mode = GPE_MODE_NATIVE;         // (falls back to the COMPAT kernels by itself: gpe_get_pipeline_info)
    cfg->profiling = 0;
    cfg->flags = 0;
    return GPE_OK;
}

const char *gpe_last_error(const gpe_ctx *ctx)
{
    if (ctx) return ctx->last_error.c_str();
    std::lock_guard<std::mutex> lk(g_err_mu);
    static thread_local std::string copy;
    copy = g_last_error;
    return copy.c_str();
}

gpe_status gpe_create(const gpe_config *cfg, gpe_ctx **out)
{
    if (!out) return fail(nullptr, GPE_ERR_INVALID_ARG, "gpe_create: out is NULL");
    *out = nullptr;
    gpe_config local;
    gpe_config_default(&local);
    if (cfg) {
        if (cfg->struct_size == 0 || cfg->struct_size > sizeof(gpe_config))
            return fail(nullptr, GPE_ERR_INVALID_ARG, "gpe_create: bad gpe_config.struct_size");
        memcpy(&local, cfg, cfg->struct_size);
        local.struct_size = (uint32_t)sizeof(gpe_config);
    }
    if (local.mode != GPE_MODE_COMPAT && local.mode != GPE_MODE_NATIVE)
        return fail(nullptr, GPE_ERR_INVALID_ARG, "gpe_create: unknown mode");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, GPE_ERR_NO_DEVICE,
                    "gpe_create: no HIP device visible (this library has no CPU fallback)");
    int dev = local.device;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    }
    if (dev >= count) return fail(nullptr, GPE_ERR_INVALID_ARG, "gpe_create: device ordinal out of range");
    gpe_ctx *c = new (std::nothrow) gpe_ctx();
    if (!c) return fail(nullptr, GPE_ERR_OOM, "gpe_create: host allocation failed");
    c->cfg = local;
    c->device = dev;
    c->profiling = local.profiling != 0;
    c->profile_every = local.profiling;
    c->use_onesweep = (local.flags & GPE_FLAG_SAFE_SORT) == 0;
    if ((e = hipSetDevice(dev)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) {
        std::string m = std::string("gpe_create: ") + hipGetErrorName(e);
        delete c;
        return fail(nullptr, GPE_ERR_HIP, m);
    }
    c->stream = c->own_stream;
    *out = c;
    return GPE_OK;
}

gpe_status gpe_destroy(gpe_ctx *c)
{
    if (!c) return GPE_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    resolve_pending(c);
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    if (c->trace_origin) (void)hipEventDestroy(c->trace_origin);
    free_particle_buffers(c);
    sort_release(c);
    scan_release(c);
    onesweep_release(c);
    native_release(c);
    group_leave(c);
    ctl_release(c);
    comm_release(c);
    shard_release(c);
    // a stream lent by gpe_set_stream belongs to the caller (a host framework may still hold buffers and
    // events that name it): only the library's own stream is destroyed
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return GPE_OK;
}

// Device-side error words (sticky): reported at the synchronising entry points.
static gpe_status check_device_errors(gpe_ctx *c)
{
    uint32_t words[3] = {0, 0, 0};
    if (c->shard.counts)
        GPE_HIP(c, hipMemcpyAsync(&words[2], c->shard.counts + kShardError, 4, hipMemcpyDeviceToHost, c->stream));
    if (c->native.tile_ctl)
        GPE_HIP(c, hipMemcpyAsync(&words[0], c->native.tile_ctl + 8, 4, hipMemcpyDeviceToHost, c->stream));
    if (c->os_ws.ctl)
        GPE_HIP(c, hipMemcpyAsync(&words[1], c->os_ws.ctl + 4, 4, hipMemcpyDeviceToHost, c->stream));
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    if (words[1]) return fail(c, GPE_ERR_HIP, "radix sort: decoupled look-back timed out");
    if (words[0] & 2u)
        return fail(c, GPE_ERR_UNSUPPORTED,
                    "native collide: a region of 24x24 cells holds more particles than the LDS cell window "
                    "takes; results of that step are unresolved there -- use GPE_MODE_COMPAT for this scene");
    if (words[0] & 5u)
        return fail(c, GPE_ERR_STATE, "native collide: a particle left the world box between steps");
    if (words[0] & 16u)
        return fail(c, GPE_ERR_STATE, "sharded run: the device-side particle count passed the host's bound");
    if (words[2])
        return fail(c, GPE_ERR_UNSUPPORTED, shard_error_text(words[2]));
    return GPE_OK;
}

// (Re)derive the native pipeline's cell box after anything it depends on changed.
static gpe_status reconfigure(gpe_ctx *c)
{
    if (c->cfg.mode == GPE_MODE_NATIVE && c->n > 0) return native_configure(c);
    c->native.eligible = false;
    return GPE_OK;
}

extern "C++" {
namespace gpe {
gpe_status reconfigure_native(gpe_ctx *c) { return reconfigure(c); }
}
}

gpe_status gpe_sync(gpe_ctx *c)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    if (c->shard.xstream) GPE_HIP(c, hipStreamSynchronize(c->shard.xstream));   // (a sharded run's exchange stream)
    return check_device_errors(c);
}

gpe_status gpe_get_pipeline_info(gpe_ctx *c, gpe_pipeline_info *info)
{
    if (!c || !info) return GPE_ERR_INVALID_ARG;
    if (info->struct_size == 0 || info->struct_size > sizeof(gpe_pipeline_info))
        return fail(c, GPE_ERR_INVALID_ARG, "gpe_get_pipeline_info: bad struct_size");
    gpe_pipeline_info out;
    memset(&out, 0, sizeof(out));
    out.struct_size = info->struct_size;
    const NativeState &N = c->native;
    if (c->cfg.mode != GPE_MODE_NATIVE) { out.pipeline = GPE_PIPELINE_COMPAT; out.reason = GPE_REASON_MODE_COMPAT; }
    else if (c->n == 0) { out.pipeline = GPE_PIPELINE_COMPAT; out.reason = GPE_REASON_NO_PARTICLES; }
    else if (N.eligible || ((N.force || c->use_order_keys) && N.in_box)) { out.pipeline = GPE_PIPELINE_NATIVE; out.reason = GPE_REASON_NONE; }
    else { out.pipeline = GPE_PIPELINE_COMPAT; out.reason = N.reason; }
    out.sort_passes = (uint32_t)N.passes;
    out.native_steps = N.native_steps;
    out.compat_steps = N.compat_steps;
    if (N.host_stat) {
        out.window_max = N.host_stat[kStatWindowMax];
        out.arena_slots = N.host_stat[kStatArena]; out.overflow_tiles = N.host_stat[kStatOverflow];
        out.overflow_subtiles = N.host_stat[kStatSubTiles]; out.overflow_spills = N.host_stat[kStatSpills];
    }
    if (N.tile_ctl) {
        uint32_t sorts = 0, seen = 0;
        GPE_HIP(c, hipMemcpyAsync(&sorts, N.tile_ctl + kNativeCtlSorts, sizeof(sorts), hipMemcpyDeviceToHost, c->stream));
        GPE_HIP(c, hipMemcpyAsync(&seen, N.tile_ctl + kNativeCtlSortsSeen, sizeof(seen), hipMemcpyDeviceToHost, c->stream));
        GPE_HIP(c, hipStreamSynchronize(c->stream));
        out.native_sorts = sorts;
        out.roster_stamp = seen;
    }
    memcpy(info, &out, info->struct_size);
    return GPE_OK;
}

gpe_status gpe_set_mode(gpe_ctx *c, uint32_t mode)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    if (mode != GPE_MODE_COMPAT && mode != GPE_MODE_NATIVE) return fail(c, GPE_ERR_INVALID_ARG, "unknown mode");
    c->cfg.mode = mode;
    if (mode == GPE_MODE_COMPAT) GPE_TRY(need_grid_buffers(c));
    return reconfigure(c);
}

// ---- particles -----------------------------------------------------------------------------------------
gpe_status gpe_set_particles(gpe_ctx *c, const float *pos_xy, const float *prev_xy, const float *radius,
                             uint64_t n)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    if (!pos_xy || !radius || n == 0) return fail(c, GPE_ERR_INVALID_ARG, "gpe_set_particles: NULL array or n == 0");
    if (n > (1ull << 30) - 1) return fail(c, GPE_ERR_INVALID_ARG, "gpe_set_particles: 4n must fit in u32");
    GPE_HIP(c, hipSetDevice(c->device));
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    if (n > c->cap) {
        free_particle_buffers(c);
        gpe_status s = alloc_particle_buffers(c, n, c->cfg.mode != GPE_MODE_NATIVE);
        if (s != GPE_OK) { free_particle_buffers(c); c->n = 0; return s; }
    }
    c->n = n;
    c->n_owned = n;
    GPE_HIP(c, hipMemcpyAsync(c->pos, pos_xy, n * sizeof(float2), hipMemcpyHostToDevice, c->stream));
    GPE_HIP(c, hipMemcpyAsync(c->prev, prev_xy ? prev_xy : pos_xy, n * sizeof(float2), hipMemcpyHostToDevice,
                              c->stream));
    GPE_HIP(c, hipMemcpyAsync(c->radius, radius, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    GPE_TRY(init_index_buffers(c, 0, n));
    c->max_radius = max_abs_radius(radius, n, radius[0]);
    c->grid_max_radius = c->max_radius;       // Grid::new (grid.rs:66-71)
    refresh_cell_size(c);
    GPE_HIP(c, hipStreamSynchronize(c->stream));   // the host arrays may be released on return
    return reconfigure(c);
}

gpe_status gpe_add_particles(gpe_ctx *c, const float *pos_xy, const float *radius, uint64_t n_add)
{
    GPE_TRY(need_particles(c));
    if (!pos_xy || !radius) return fail(c, GPE_ERR_INVALID_ARG, "gpe_add_particles: NULL array");
    if (n_add == 0) return GPE_OK;
    const uint64_t old_n = c->n, new_n = c->n + n_add;
    if (new_n > (1ull << 30) - 1) return fail(c, GPE_ERR_INVALID_ARG, "gpe_add_particles: 4n must fit in u32");
    GPE_HIP(c, hipSetDevice(c->device));
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    if (new_n > c->cap) GPE_TRY(grow_particle_buffers(c, std::max<uint64_t>(new_n, c->cap * 2)));
    GPE_HIP(c, hipMemcpyAsync(c->pos + old_n, pos_xy, n_add * sizeof(float2), hipMemcpyHostToDevice, c->stream));
    GPE_HIP(c, hipMemcpyAsync(c->prev + old_n, pos_xy, n_add * sizeof(float2), hipMemcpyHostToDevice, c->stream));
    GPE_HIP(c, hipMemcpyAsync(c->radius + old_n, radius, n_add * sizeof(float), hipMemcpyHostToDevice, c->stream));
    c->n = new_n;
    c->n_owned = new_n;
    GPE_TRY(init_index_buffers(c, old_n, new_n));
    // particle_system.rs:198: max_radius = max(max_radius, r)
    for (uint64_t i = 0; i < n_add; ++i) c->max_radius = fmaxf(c->max_radius, radius[i]);
    c->grid_max_radius = c->max_radius;   // Grid::refresh_grid (grid.rs:266)
    refresh_cell_size(c);
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    return reconfigure(c);
}

gpe_status gpe_len(const gpe_ctx *c, uint64_t *n)
{
    if (!c || !n) return GPE_ERR_INVALID_ARG;
    *n = c->n;
    return GPE_OK;
}

gpe_status gpe_max_radius(const gpe_ctx *c, float *r)
{
    if (!c || !r) return GPE_ERR_INVALID_ARG;
    *r = c->max_radius;
    return GPE_OK;
}

gpe_status gpe_set_mouse(gpe_ctx *c, int32_t pressed, float x, float y)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    c->mouse_pressed = pressed ? 1 : 0;
    c->mouse_x = x;
    c->mouse_y = y;
    return GPE_OK;
}

gpe_status gpe_set_world(gpe_ctx *c, float w, float h)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    c->cfg.world_width = w;
    c->cfg.world_height = h;
    return reconfigure(c);
}

gpe_status gpe_set_gravity(gpe_ctx *c, float gx, float gy)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    c->cfg.gravity_x = gx;
    c->cfg.gravity_y = gy;
    return GPE_OK;
}

gpe_status gpe_morton_resort(gpe_ctx *c)
{
    GPE_TRY(need_particles(c));
    GPE_HIP(c, hipSetDevice(c->device));
    return do_resort(c);
}

gpe_status gpe_integrate(gpe_ctx *c, float dt)
{
    GPE_TRY(need_particles(c));
    GPE_HIP(c, hipSetDevice(c->device));
    return launch_verlet(c, c->pos, c->prev, c->radius, c->n_owned, dt);
}

// ---- grid ---------------------------------------------------------------------------------------------
float gpe_compute_cell_size(float max_obj_radius) { return max_obj_radius * 2.2f; }

gpe_status gpe_grid_set_max_radius(gpe_ctx *c, float r)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    c->grid_max_radius = r;
    refresh_cell_size(c);
    return reconfigure(c);
}

gpe_status gpe_cell_size(const gpe_ctx *c, float *cs)
{
    if (!c || !cs) return GPE_ERR_INVALID_ARG;
    *cs = c->cell_size;
    return GPE_OK;
}

gpe_status gpe_grid_build(gpe_ctx *c)
{
    GPE_TRY(need_particles(c));
    GPE_HIP(c, hipSetDevice(c->device));
    GPE_TRY(need_grid_buffers(c));
    return launch_build_cell_ids(c, c->pos, c->radius, c->n, c->cell_size, c->cell_ids, c->object_ids);
}

gpe_status gpe_grid_sort(gpe_ctx *c)
{
    GPE_TRY(need_particles(c));
    GPE_HIP(c, hipSetDevice(c->device));
    return do_grid_sort(c);
}

gpe_status gpe_grid_update(gpe_ctx *c)
{
    GPE_TRY(gpe_grid_build(c));
    return do_grid_sort(c);
}

// ---- physics ---------------------------------------------------------------------------------------------
gpe_status gpe_build_collision_cells(gpe_ctx *c)
{
    GPE_TRY(need_particles(c));
    GPE_HIP(c, hipSetDevice(c->device));
    return do_build_collision_cells(c);
}

gpe_status gpe_solve_collisions(gpe_ctx *c)
{
    GPE_TRY(gpe_build_collision_cells(c));
    return do_solve_colors(c);
}

// ---- step ------------------------------------------------------------------------------------------------
gpe_status gpe_step(gpe_ctx *c, float dt, uint32_t flags)
{
    GPE_TRY(need_particles(c));
    GPE_HIP(c, hipSetDevice(c->device));
    return do_step(c, dt, flags);
}

gpe_status gpe_run(gpe_ctx *c, float dt, uint64_t steps, uint64_t resort_every, int32_t resort_first)
{
    GPE_TRY(need_particles(c));
    GPE_HIP(c, hipSetDevice(c->device));
    // The host may run at most ~64 steps ahead of the device: bounds the queue and the lag of the
    // device-side statistics the step policy reads (native_should_run).
    hipEvent_t fence[2] = {nullptr, nullptr};
    bool armed[2] = {false, false};
    gpe_status rc = GPE_OK;
    for (uint64_t s = 0; s < steps && rc == GPE_OK; ++s) {
        if ((s & 31u) == 0 && steps > 64) {
            const int slot = (int)((s >> 5) & 1u);
            if (armed[slot]) (void)hipEventSynchronize(fence[slot]);
            if (!fence[slot] && hipEventCreateWithFlags(&fence[slot], hipEventDisableTiming) != hipSuccess)
                fence[slot] = nullptr;
            if (fence[slot]) armed[slot] = hipEventRecord(fence[slot], c->stream) == hipSuccess;
        }
        const bool resort = (s == 0 && resort_first) || (resort_every && s > 0 && (s % resort_every) == 0);
        rc = do_step(c, dt, resort ? GPE_STEP_RESORT : 0u);
    }
    for (hipEvent_t e : fence) if (e) (void)hipEventDestroy(e);
    return rc;
}

// ---- downloads ---------------------------------------------------------------------------------------------
static gpe_status locate(gpe_ctx *c, gpe_array what, const void **ptr, uint64_t *bytes, bool sizes_only = false)
{
    const uint64_t n = c->n;
    if (sizes_only) {}
    else if (what == GPE_CELL_IDS || what == GPE_OBJECT_IDS || what == GPE_COLLISION_CELLS ||
        what == GPE_NUM_COLLISION_CELLS || what == GPE_CHUNK_OBJ_COUNT || what == GPE_INDIRECT_ARGS)
        GPE_TRY(need_grid_buffers(c));
    switch (what) {
        case GPE_POS: *ptr = c->pos; *bytes = n * 8; break;
        case GPE_PREV: *ptr = c->prev; *bytes = n * 8; break;
        case GPE_RADIUS: *ptr = c->radius; *bytes = n * 4; break;
        case GPE_HOME_CELL_IDS: *ptr = c->home_cell_ids; *bytes = n * 4; break;
        case GPE_PARTICLE_IDS: *ptr = c->particle_ids; *bytes = n * 4; break;
        case GPE_CELL_IDS: *ptr = c->cell_ids; *bytes = n * 16; break;
        case GPE_OBJECT_IDS: *ptr = c->object_ids; *bytes = n * 16; break;
        case GPE_COLLISION_CELLS: *ptr = c->collision_cells; *bytes = n * 16; break;
        case GPE_NUM_COLLISION_CELLS:
            *ptr = c->chunk_obj_count ? c->chunk_obj_count + (num_chunks(c) - 1) : nullptr;
            *bytes = 4;
            break;
        case GPE_CHUNK_OBJ_COUNT: *ptr = c->chunk_obj_count; *bytes = num_chunks(c) * 4; break;
        case GPE_INDIRECT_ARGS: *ptr = c->indirect_args; *bytes = 12; break;
        case GPE_ORDER_KEYS: *ptr = c->order_keys; *bytes = n * 4; break;
        default: return fail(c, GPE_ERR_INVALID_ARG, "unknown gpe_array");
    }
    return GPE_OK;
}

gpe_status gpe_array_bytes(const gpe_ctx *c, gpe_array what, uint64_t *bytes)
{
    if (!c || !bytes) return GPE_ERR_INVALID_ARG;
    const void *p;
    return locate(const_cast<gpe_ctx *>(c), what, &p, bytes, true);
}

gpe_status gpe_device_ptr(gpe_ctx *c, gpe_array what, void **device_ptr, uint64_t *bytes)
{
    GPE_TRY(need_particles(c));
    if (!device_ptr) return fail(c, GPE_ERR_INVALID_ARG, "device_ptr is NULL");
    const void *p;
    uint64_t b;
    GPE_TRY(locate(c, what, &p, &b));
    *device_ptr = const_cast<void *>(p);
    if (bytes) *bytes = b;
    return GPE_OK;
}

gpe_status gpe_download(gpe_ctx *c, gpe_array what, void *dst, uint64_t bytes)
{
    GPE_TRY(need_particles(c));
    if (!dst) return fail(c, GPE_ERR_INVALID_ARG, "gpe_download: dst is NULL");
    const void *p;
    uint64_t b;
    GPE_TRY(locate(c, what, &p, &b));
    if (bytes != b) return fail(c, GPE_ERR_INVALID_ARG, "gpe_download: byte count does not match the array");
    GPE_HIP(c, hipSetDevice(c->device));
    GPE_HIP(c, hipMemcpyAsync(dst, p, b, hipMemcpyDeviceToHost, c->stream));
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    return check_device_errors(c);
}

// ---- primitives ---------------------------------------------------------------------------------------------
gpe_status gpe_buffer_alloc(gpe_ctx *c, uint64_t bytes, void **device_ptr)
{
    if (!c || !device_ptr) return GPE_ERR_INVALID_ARG;
    GPE_HIP(c, hipSetDevice(c->device));
    *device_ptr = nullptr;
    hipError_t e = hipMalloc(device_ptr, std::max<uint64_t>(bytes, 16) + 64);
    if (e == hipErrorOutOfMemory) return fail(c, GPE_ERR_OOM, "gpe_buffer_alloc: out of device memory");
    if (e != hipSuccess) return fail(c, GPE_ERR_HIP, std::string("gpe_buffer_alloc: ") + hipGetErrorName(e));
    return GPE_OK;
}

gpe_status gpe_buffer_free(gpe_ctx *c, void *device_ptr)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    if (!device_ptr) return GPE_OK;
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    GPE_HIP(c, hipFree(device_ptr));
    return GPE_OK;
}

gpe_status gpe_buffer_upload(gpe_ctx *c, void *device_ptr, const void *src, uint64_t bytes)
{
    if (!c || (!device_ptr && bytes) || (!src && bytes)) return GPE_ERR_INVALID_ARG;
    if (bytes == 0) return GPE_OK;
    GPE_HIP(c, hipMemcpyAsync(device_ptr, src, bytes, hipMemcpyHostToDevice, c->stream));
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    return GPE_OK;
}

gpe_status gpe_buffer_download(gpe_ctx *c, const void *device_ptr, void *dst, uint64_t bytes)
{
    if (!c || (!device_ptr && bytes) || (!dst && bytes)) return GPE_ERR_INVALID_ARG;
    if (bytes == 0) return GPE_OK;
    GPE_HIP(c, hipMemcpyAsync(dst, device_ptr, bytes, hipMemcpyDeviceToHost, c->stream));
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    return GPE_OK;
}

gpe_status gpe_sort_pairs_u32(gpe_ctx *c, uint32_t *d_keys, uint32_t *d_payload, uint64_t n)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    if (n == 0) return GPE_OK;
    if (!d_keys || !d_payload) return fail(c, GPE_ERR_INVALID_ARG, "gpe_sort_pairs_u32: NULL buffer");
    if (n > 0xffffffffull) return fail(c, GPE_ERR_INVALID_ARG, "gpe_sort_pairs_u32: n must be < 2^32");
    GPE_HIP(c, hipSetDevice(c->device));
    GPE_TRY(sort_reserve(c, n));
    return sort_pairs(c, d_keys, d_payload, n);
}

gpe_status gpe_sort_histogram_u32(gpe_ctx *c, const uint32_t *d_keys, uint64_t n, uint32_t shift,
                                  uint32_t *d_hist256)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    if (!d_hist256 || (!d_keys && n) || shift > 24) return fail(c, GPE_ERR_INVALID_ARG, "gpe_sort_histogram_u32: bad argument");
    GPE_HIP(c, hipSetDevice(c->device));
    return sort_histogram(c, d_keys, n, shift, d_hist256);
}

gpe_status gpe_sort_scatter_pass_u32(gpe_ctx *c, const uint32_t *ka, const uint32_t *va, uint32_t *kb,
                                     uint32_t *vb, uint64_t n, uint32_t shift)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    if (n == 0) return GPE_OK;
    if (!ka || !va || !kb || !vb || shift > 24) return fail(c, GPE_ERR_INVALID_ARG, "gpe_sort_scatter_pass_u32: bad argument");
    GPE_HIP(c, hipSetDevice(c->device));
    GPE_TRY(sort_reserve(c, n));
    return sort_scatter_pass(c, ka, va, kb, vb, n, shift);
}

gpe_status gpe_inclusive_scan_u32(gpe_ctx *c, uint32_t *d_data, uint64_t n)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    if (n == 0) return GPE_OK;
    if (!d_data) return fail(c, GPE_ERR_INVALID_ARG, "gpe_inclusive_scan_u32: NULL buffer");
    GPE_HIP(c, hipSetDevice(c->device));
    GPE_TRY(scan_reserve(c, n));
    return inclusive_scan(c, d_data, n);
}

// ---- sharding support -------------------------------------------------------------------------------------
gpe_status gpe_reserve(gpe_ctx *c, uint64_t capacity)
{
    GPE_TRY(need_particles(c));
    if (capacity > (1ull << 30) - 1) return fail(c, GPE_ERR_INVALID_ARG, "gpe_reserve: 4n must fit in u32");
    GPE_HIP(c, hipSetDevice(c->device));
    if (capacity > c->cap) {
        GPE_TRY(grow_particle_buffers(c, capacity));
        GPE_TRY(reconfigure(c));
    }
    return GPE_OK;
}

gpe_status gpe_capacity(const gpe_ctx *c, uint64_t *capacity)
{
    if (!c || !capacity) return GPE_ERR_INVALID_ARG;
    *capacity = c->cap;
    return GPE_OK;
}

gpe_status gpe_set_counts(gpe_ctx *c, uint64_t n_total, uint64_t n_owned)
{
    GPE_TRY(need_particles(c));
    if (n_total == 0 || n_total > c->cap || n_owned > n_total)
        return fail(c, GPE_ERR_INVALID_ARG, "gpe_set_counts: need 0 < n_owned <= n_total <= capacity");
    c->n = n_total;
    c->n_owned = n_owned;
    return GPE_OK;
}

gpe_status gpe_use_order_keys(gpe_ctx *c, int32_t enable)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    c->use_order_keys = enable != 0;
    return GPE_OK;
}

gpe_status gpe_set_active_cells(gpe_ctx *c, int32_t cx0, int32_t cy0, int32_t cx1, int32_t cy1)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    if (cx1 < cx0 || cy1 < cy0) return fail(c, GPE_ERR_INVALID_ARG, "gpe_set_active_cells: empty box");
    c->active_box[0] = cx0; c->active_box[1] = cy0; c->active_box[2] = cx1; c->active_box[3] = cy1;
    c->has_active_box = true;
    if (c->n == 0 || !c->pos) return GPE_OK;
    GPE_HIP(c, hipSetDevice(c->device));
    return reconfigure(c);                     // the block box (sort keys, block table) follows the active box
}

gpe_status gpe_stream_handle(gpe_ctx *c, void **hip_stream)
{
    if (!c || !hip_stream) return GPE_ERR_INVALID_ARG;
    *hip_stream = (void *)c->stream;
    return GPE_OK;
}

gpe_status gpe_set_stream(gpe_ctx *c, void *hip_stream)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    GPE_HIP(c, hipSetDevice(c->device));
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    resolve_pending(c);                                   // event pairs recorded on the stream being left
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    if (c->trace_origin) (void)hipEventRecord(c->trace_origin, c->stream);
    for (int i = 0; i < 2; ++i) c->shard.armed[i] = false;
    return GPE_OK;
}

gpe_status gpe_refresh(gpe_ctx *c)
{
    GPE_TRY(need_particles(c));
    GPE_HIP(c, hipSetDevice(c->device));
    return reconfigure(c);
}

gpe_status gpe_shard_classify(gpe_ctx *c, const uint8_t *d_owner_of_block, const uint32_t *d_dest_mask_of_block,
                              int32_t blocks_x, int32_t blocks_y, uint32_t my_rank, uint32_t *d_out_index,
                              uint32_t *d_out_info, uint32_t *d_out_count, uint64_t out_capacity)
{
    GPE_TRY(need_particles(c));
    if (!d_owner_of_block || !d_dest_mask_of_block || !d_out_index || !d_out_info || !d_out_count || blocks_x <= 0 ||
        blocks_y <= 0)
        return fail(c, GPE_ERR_INVALID_ARG, "gpe_shard_classify: bad argument");
    GPE_HIP(c, hipSetDevice(c->device));
    return launch_shard_classify(c, d_owner_of_block, d_dest_mask_of_block, blocks_x, blocks_y, my_rank, d_out_index,
                                 d_out_info, d_out_count, out_capacity);
}

// ---- profiling ---------------------------------------------------------------------------------------------
static gpe_status mark_trace_origin(gpe_ctx *c)
{
    if (!c->trace_origin) GPE_HIP(c, hipEventCreate(&c->trace_origin));
    GPE_HIP(c, hipEventRecord(c->trace_origin, c->stream));
    return GPE_OK;
}

gpe_status gpe_set_profiling(gpe_ctx *c, uint32_t on)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    c->profiling = on != 0;
    c->profile_every = on;
    c->profile_step = 0;
    if (on && !c->trace_origin) GPE_TRY(mark_trace_origin(c));
    return GPE_OK;
}

gpe_status gpe_reset_timings(gpe_ctx *c)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    resolve_pending(c);
    c->stats.clear();
    c->trace.clear();
    return mark_trace_origin(c);
}

gpe_status gpe_get_trace(gpe_ctx *c, gpe_trace_event *out, uint32_t *count)
{
    if (!c || !count) return GPE_ERR_INVALID_ARG;
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    resolve_pending(c);
    const uint32_t avail = (uint32_t)c->trace.size();
    if (out) {
        const uint32_t m = std::min(avail, *count);
        for (uint32_t i = 0; i < m; ++i) {
            const TraceEvent &e = c->trace[avail - m + i];               // the newest m, oldest first
            memset(&out[i], 0, sizeof(gpe_trace_event));
            strncpy(out[i].name, c->stats[e.stat].name.c_str(), sizeof(out[i].name) - 1);
            out[i].start_ms = e.start_ms;
            out[i].duration_ms = e.dur_ms;
        }
    }
    *count = avail;
    return GPE_OK;
}

gpe_status gpe_get_timings(gpe_ctx *c, gpe_timing *out, uint32_t *count)
{
    if (!c || !count) return GPE_ERR_INVALID_ARG;
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    resolve_pending(c);
    const uint32_t avail = (uint32_t)c->stats.size();
    if (out) {
        const uint32_t m = std::min(avail, *count);
        for (uint32_t i = 0; i < m; ++i) {
            memset(&out[i], 0, sizeof(gpe_timing));
            strncpy(out[i].name, c->stats[i].name.c_str(), sizeof(out[i].name) - 1);
            out[i].total_ms = c->stats[i].total_ms;
            out[i].calls = c->stats[i].calls;
        }
    }
    *count = avail;
    return GPE_OK;
}

}  // extern "C"
