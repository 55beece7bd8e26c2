/*
 * gpe.h -- C-ABI of the MI355X-native particle step (libgpe.so).
 *
 * Drop-in boundary for the per-timestep particle pipeline of MarcVivas/gpu-physics-engine:
 * everything State::update() (src/state.rs:115-131) calls except the camera update, i.e. the
 * module API of src/particles, src/grid, src/physics and the two GPU primitives in src/utils.
 * The reference exports no FFI of its own (Cargo.toml:6-7 builds cdylib+rlib with nothing
 * extern "C"), so each entry point below names the Rust method a host shim would forward to it.
 * Citations are relative to /root/reference/src.  INTEGRATION.md shows the Rust-side binding.
 *
 * Conventions
 *  - plain pointers and sizes only; positions are interleaved (x,y) f32 pairs (glam::Vec2).
 *  - the library owns all device memory; the caller owns every host array passed in or out.
 *  - every function returns gpe_status (0 = OK, negative = error) and never unwinds;
 *    gpe_last_error() gives the message of the last failure on that context (or globally when
 *    ctx is NULL).  The reference unwrap()s/panics instead (gpu_buffer.rs:266-268).
 *  - one in-order hipStream per context; calls on one context are not re-entrant.
 *    gpe_step/gpe_run and the per-module calls are asynchronous; gpe_download/gpe_sync block.
 */
#ifndef GPE_H
#define GPE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPE_ABI_VERSION 1u
#define GPE_UNUSED_CELL_ID 0xffffffffu    /* grid/grid.rs:22  UNUSED_CELL_ID            */
#define GPE_MAX_CELLS_PER_OBJECT 4u       /* grid/grid.rs:18  MAX_CELLS_PER_OBJECT      */
#define GPE_COUNTING_CHUNK_SIZE 4u        /* physics/collision_cell_builder.rs:13       */

typedef struct gpe_ctx gpe_ctx;
typedef int32_t gpe_status;

enum {
    GPE_OK = 0,
    GPE_ERR_INVALID_ARG = -1,
    GPE_ERR_HIP = -2,          /* a HIP runtime call failed (message has the hipError name)  */
    GPE_ERR_OOM = -3,
    GPE_ERR_STATE = -4,        /* call order violated (e.g. step before set_particles)       */
    GPE_ERR_UNSUPPORTED = -5,
    GPE_ERR_NO_DEVICE = -6     /* no gfx950 device visible: there is NO CPU fallback         */
};

/* Which kernels gpe_step() runs.  Both produce identical positions
 * (tests/test_gpu_native.py::test_native_equals_compat_1m, _100m; each against the oracle: test_gpu_parity_step.py).
 * COMPAT materialises the reference's own intermediate buffers every step (4N (cell,object)
 * pairs, chunk counts, collision-cell list) -- needed for bit-exact comparison with the
 * reference tests.  NATIVE is the MI355X design: N-key sort + LDS-staged cell windows. */
enum { GPE_MODE_COMPAT = 0, GPE_MODE_NATIVE = 1 };

enum { GPE_STEP_RESORT = 1u };            /* gpe_step flags: Morton re-sort first (state.rs:122) */

typedef struct gpe_config {
    uint32_t struct_size;          /* = sizeof(gpe_config), for ABI growth                    */
    int32_t  device;               /* HIP device ordinal, -1 = current device                 */
    float    world_width;          /* state.rs:35  (3048)                                     */
    float    world_height;         /* state.rs:35  (1048)                                     */
    float    gravity_x;            /* particle_integration.wgsl:21 FORCE_OF_GRAVITY (0,0)     */
    float    gravity_y;
    float    cell_size_multiplier; /* grid.rs:20  CELL_SIZE_MULTIPLIER = 2.2                  */
    float    stiffness;            /* collision_solver.wgsl:2  STIFFNESS = 0.6                */
    float    mouse_strength;       /* particle_integration.wgsl:22  = 150                     */
    uint32_t mode;                 /* GPE_MODE_*                                              */
    uint32_t profiling;            /* as gpe_set_profiling: 0 off, 1 every scope, k every k-th step */
    uint32_t flags;                /* GPE_FLAG_* (0 = the defaults); was reserved[0]          */
    uint32_t reserved[4];
} gpe_config;

/* gpe_config.flags -- switches for tests and measurements; the reference has no counterpart (state.rs:34-70 builds one
 * pipeline).  None of them changes a result bit. */
enum {
    GPE_FLAG_NATIVE_FORCE = 1u,      /* never hand an over-dense scene to the COMPAT kernels (windows then spill)   */
    GPE_FLAG_SORT_EVERY_STEP = 2u,   /* NATIVE: run the radix passes every step instead of when a particle has left */
                                     /* the reach of the kept block table (what rounds 1-2 did; for A/B timing)     */
    GPE_FLAG_NATIVE_STATS = 4u,      /* print the tile statistics to stderr every 128 steps                          */
    GPE_FLAG_SAFE_SORT = 8u,         /* per-module sorts by the communication-free reduce-then-scan radix sort       */
                                     /* (k_radix_sort.hip) instead of onesweep: an in-GPU cross-check                */
    GPE_FLAG_COUNTING_SORT_TILES = 16u, /* NATIVE: the dense launch builds its member lists by a counting sort      */
                                     /* (rounds 1-2) instead of direct cell slots; for A/B timing                    */
    GPE_FLAG_XCD_EIGHTHS = 64u,      /* NATIVE: every XCD works through one contiguous eighth of the tile rows       */
                                     /* (rounds 1-3) instead of interleaved bands of rows; for A/B timing            */
    GPE_FLAG_NO_HALF_TILES = 128u,   /* NATIVE: tiles the direct-slot launch hands on go straight to the 16x16 / 8x8 */
                                     /* windows (rounds 1-3), not first through 32x16 direct-slot halves -- neither  */
                                     /* the half-tile launch nor the dense launch's front workgroups; A/B timing     */
    GPE_FLAG_SHARD_OVERLAP = 256u,   /* sharded runs: the neighbour exchange on a stream of its own beside the       */
                                     /* interior tiles, the tiles along the rank's border first (off by default: on  */
                                     /* one GPU the two cross-stream waits cost more than the exchange they hide)    */
    GPE_FLAG_FUSED_HISTOGRAMS = 512u /* NATIVE: the hash kernel counts the radix digits every step (rounds 1-3)      */
                                     /* instead of a gated launch counting them when a sort is due; for A/B timing   */
};

/* Fills *cfg with the reference's compile-time constants. */
gpe_status gpe_config_default(gpe_config *cfg);

/* State::new (state.rs:34-70) minus window/renderer: creates the HIP context (stream, events).
 * Fails with GPE_ERR_NO_DEVICE when no GPU is visible. */
gpe_status gpe_create(const gpe_config *cfg, gpe_ctx **out);
gpe_status gpe_destroy(gpe_ctx *ctx);
const char *gpe_last_error(const gpe_ctx *ctx);
uint32_t gpe_abi_version(void);

/* ---- particles (src/particles/particle_system.rs) ----------------------------------------- */
/* ParticleSystem::new_from_buffers (:49-99) / generate_initial_particles (:102-161).
 * prev_xy == NULL => previous = current (zero velocity, :126-127).  Also (re)creates the Grid
 * (grid.rs:74-149) and CollisionSystem (collision_system.rs:14-22) buffers for n particles and
 * sets max_radius = the radius of largest magnitude (:51). */
gpe_status gpe_set_particles(gpe_ctx *ctx, const float *pos_xy, const float *prev_xy,
                             const float *radius, uint64_t n);
/* ParticleSystem::add_particles (:163-220) + Grid::refresh_grid (grid.rs:265-291) +
 * CollisionSystem::refresh (collision_system.rs:24-28): append n particles (prev = pos),
 * grow every dependent buffer (amortised x2 like gpu_buffer.rs:54-56), recompute cell size. */
gpe_status gpe_add_particles(gpe_ctx *ctx, const float *pos_xy, const float *radius, uint64_t n);
/* ParticleSystem::len (:275) / get_max_radius (:291) */
gpe_status gpe_len(const gpe_ctx *ctx, uint64_t *n);
gpe_status gpe_max_radius(const gpe_ctx *ctx, float *r);
/* ParticleSystem::sort_by_cell_id (:236-243) -> ParticleSort::sort (particle_sort.rs:58-69):
 * K1 home cell ids, stable sort of (home cell, particle id), K4 rearrange.  The live and copy
 * sets are swapped instead of copied back (particle_rearrange.rs:205-238). */
gpe_status gpe_morton_resort(gpe_ctx *ctx);
/* ParticleSystem::update_positions (:245-247) -> K12 verlet_integration */
gpe_status gpe_integrate(gpe_ctx *ctx, float dt);
/* mouse_click_callback / mouse_move_callback (:221-227, particle_integration.rs:176-185) */
gpe_status gpe_set_mouse(gpe_ctx *ctx, int32_t pressed, float x, float y);
/* world size used by K12's wall clamp (ParticleIntegration::new, particle_integration.rs:34-48;
 * new_from_buffers hard-codes 1920x1080, particle_system.rs:86) */
gpe_status gpe_set_world(gpe_ctx *ctx, float width, float height);
gpe_status gpe_set_gravity(gpe_ctx *ctx, float gx, float gy);

/* ---- grid (src/grid/grid.rs) ---------------------------------------------------------------- */
/* Grid::compute_cell_size (:159-161) */
float gpe_compute_cell_size(float max_obj_radius);
/* Grid::new_without_camera(ctx, max_obj_radius, &particles) (:74): override the radius the cell
 * size is derived from (default: the particle system's max radius, Grid::new :66-71). */
gpe_status gpe_grid_set_max_radius(gpe_ctx *ctx, float max_obj_radius);
gpe_status gpe_cell_size(const gpe_ctx *ctx, float *cell_size);     /* Grid::cell_size (:163) */
gpe_status gpe_grid_build(gpe_ctx *ctx);     /* Grid::build_cell_ids (:296-306), K5          */
gpe_status gpe_grid_sort(gpe_ctx *ctx);      /* Grid::sort_map (:310-312), 4N-pair sort      */
gpe_status gpe_grid_update(gpe_ctx *ctx);    /* Grid::update (:322-332) = build + sort       */

/* ---- physics (src/physics/collision_system.rs) ---------------------------------------------- */
/* CollisionSystem::solve_collisions (:30-39): collision-cell list (K6, scan, K10) then the four
 * colour passes (K11).  Operates on the pair list left by gpe_grid_update. */
gpe_status gpe_solve_collisions(gpe_ctx *ctx);
/* CollisionCellBuilder::build_collision_cells alone (collision_cell_builder.rs:211-236) */
gpe_status gpe_build_collision_cells(gpe_ctx *ctx);

/* ---- step (src/state.rs:115-131) ------------------------------------------------------------ */
/* One State::update(): [re-sort] -> grid update -> solve collisions -> integrate. */
gpe_status gpe_step(gpe_ctx *ctx, float dt, uint32_t flags);
/* `steps` updates with no host synchronisation in between; re-sorts on the first step when
 * resort_first != 0 (particle_system.rs:45) and then every resort_every steps (0 = never;
 * the reference re-sorts every 4 s of wall clock, particle_system.rs:13-14,229-231). */
gpe_status gpe_run(gpe_ctx *ctx, float dt, uint64_t steps, uint64_t resort_every, int32_t resort_first);
gpe_status gpe_sync(gpe_ctx *ctx);
gpe_status gpe_set_mode(gpe_ctx *ctx, uint32_t mode);

/* Which kernels does the next gpe_step() run, and if not the NATIVE ones, why not?  The reference has one pipeline
 * and no switch (state.rs:34-70); here mode NATIVE (the default) falls back to the COMPAT kernels -- same bits -- for
 * the inputs listed below, and this call is how a host sees it.  Synchronises (it reads a device counter). */
enum { GPE_PIPELINE_COMPAT = 0, GPE_PIPELINE_NATIVE = 1 };
enum {
    GPE_REASON_NONE = 0,             /* the NATIVE kernels run                                                     */
    GPE_REASON_MODE_COMPAT = 1,      /* gpe_config.mode / gpe_set_mode asked for COMPAT                            */
    GPE_REASON_NO_PARTICLES = 2,     /* nothing to step yet                                                        */
    GPE_REASON_OUT_OF_BOX = 3,       /* a particle lay outside [0, world] when the context was configured          */
    GPE_REASON_GRID_TOO_WIDE = 4,    /* more than 65000 cells along an axis (16-bit cell coordinates)              */
    GPE_REASON_TABLE_TOO_LARGE = 5,  /* more than 2^27 8x8-cell blocks (a block table above 1 GiB)                 */
    GPE_REASON_DENSE_WINDOWS = 6     /* a 24x24-cell window holds more than 16384 particles (24576 to enter): held */
                                     /* on COMPAT, probed every 256 steps, returns by itself when it thins out     */
};
typedef struct gpe_pipeline_info {
    uint32_t struct_size;        /* in: sizeof(gpe_pipeline_info)                                                  */
    uint32_t pipeline;           /* GPE_PIPELINE_*: what the next step runs                                        */
    uint32_t reason;             /* GPE_REASON_*                                                                   */
    uint32_t sort_passes;        /* NATIVE: 8-bit radix passes of one sort of the block keys                       */
    uint64_t native_steps;       /* steps run on the NATIVE kernels since gpe_create                               */
    uint64_t compat_steps;       /* ... on the COMPAT kernels                                                      */
    uint64_t native_sorts;       /* NATIVE steps whose radix passes ran (the others reused the kept block table)   */
    uint32_t window_max;         /* largest 24x24-cell window population last reported by the tiles                */
    uint32_t roster_stamp;       /* the sort count the tile rosters are checked against (the tiles' own copy of    */
                                 /* native_sorts, low 32 bits: the two are equal or the rosters would be stale)    */
    /* what the tiles of the last reported NATIVE step did (lagged by the steps in flight; a host that fills in a   */
    /* struct_size up to roster_stamp gets the fields above only):                                                   */
    uint32_t overflow_tiles;     /* 32x32 tiles handed to the over-capacity launch                                  */
    uint32_t overflow_subtiles;  /* 16x16 quarters of those redone as four 8x8 tiles                                */
    uint32_t overflow_spills;    /* 8x8 tiles staged in the global spill arena                                      */
    uint32_t arena_slots;        /* spill-arena slots handed out                                                    */
} gpe_pipeline_info;
gpe_status gpe_get_pipeline_info(gpe_ctx *ctx, gpe_pipeline_info *info);

/* ---- downloads (GpuBuffer::download, utils/gpu_buffer.rs:96-175) ---------------------------- */
typedef enum gpe_array {
    GPE_POS = 0,               /* current_positions   f32[2n]   particle_system.rs:258-265        */
    GPE_PREV = 1,              /* previous_positions  f32[2n]                                     */
    GPE_RADIUS = 2,            /* radii               f32[n]                                      */
    GPE_HOME_CELL_IDS = 3,     /* u32[n]   ParticleSystem::download_home_cell_ids (:250)          */
    GPE_PARTICLE_IDS = 4,      /* u32[n]   ParticleSystem::download_particle_ids (:254)           */
    GPE_CELL_IDS = 5,          /* u32[4n]  Grid::download_cell_ids (grid.rs:314)                  */
    GPE_OBJECT_IDS = 6,        /* u32[4n]  Grid::download_object_ids (grid.rs:318)                */
    GPE_COLLISION_CELLS = 7,   /* u32[4n]  CollisionSystem::download_collision_cells (:41)        */
    GPE_NUM_COLLISION_CELLS = 8, /* u32[1] last element of the scanned chunk counts               */
    GPE_CHUNK_OBJ_COUNT = 9,   /* u32[n]   CollisionCellBuilder::chunk_obj_count (scanned)        */
    GPE_INDIRECT_ARGS = 10,    /* u32[3]   collision_cell_builder.wgsl:96-109                     */
    GPE_ORDER_KEYS = 11        /* u32[n]   sharded runs: global object index of each local particle */
} gpe_array;
/* Blocks until the stream is idle, then copies exactly `bytes` (must equal the array's size). */
gpe_status gpe_download(gpe_ctx *ctx, gpe_array what, void *dst, uint64_t bytes);
gpe_status gpe_array_bytes(const gpe_ctx *ctx, gpe_array what, uint64_t *bytes);
/* Render hand-off (particle_drawer.wgsl:11-13 reads these three as storage buffers): the device
 * pointer stays valid until the next set/add_particles or morton_resort. */
gpe_status gpe_device_ptr(gpe_ctx *ctx, gpe_array what, void **device_ptr, uint64_t *bytes);

/* ---- GPU primitives (src/utils/radix_sort, src/utils/prefix_sum) ----------------------------- */
/* GpuBuffer<u32> stand-in for the primitive tests (utils/gpu_buffer.rs:31-47,96-175). */
gpe_status gpe_buffer_alloc(gpe_ctx *ctx, uint64_t bytes, void **device_ptr);
gpe_status gpe_buffer_free(gpe_ctx *ctx, void *device_ptr);
gpe_status gpe_buffer_upload(gpe_ctx *ctx, void *device_ptr, const void *src, uint64_t bytes);
gpe_status gpe_buffer_download(gpe_ctx *ctx, const void *device_ptr, void *dst, uint64_t bytes);
/* GPUSorter::sort (radix_sort.rs:199-217): stable ascending sort of n (u32 key, u32 payload)
 * pairs, result in the caller's buffers.  Device pointers. */
gpe_status gpe_sort_pairs_u32(gpe_ctx *ctx, uint32_t *d_keys, uint32_t *d_payload, uint64_t n);
/* GPUSorter::build_histogram (radix_sort.rs:180-188): 256-bin histogram of (key >> shift) & 255
 * over all n keys (the reference keeps one per workgroup; with n <= 11520 there is one). */
gpe_status gpe_sort_histogram_u32(gpe_ctx *ctx, const uint32_t *d_keys, uint64_t n, uint32_t shift,
                                  uint32_t *d_hist256);
/* GPUSorter::scatter (radix_sort.rs:190-198): ONE stable pass on the 8-bit digit at `shift`,
 * from (keys_a, payload_a) into (keys_b, payload_b). */
gpe_status gpe_sort_scatter_pass_u32(gpe_ctx *ctx, const uint32_t *d_keys_a, const uint32_t *d_payload_a,
                                     uint32_t *d_keys_b, uint32_t *d_payload_b, uint64_t n, uint32_t shift);
/* PrefixSum::execute (prefix_sum.rs:143-160): in-place inclusive u32 scan, wrap-around add. */
gpe_status gpe_inclusive_scan_u32(gpe_ctx *ctx, uint32_t *d_data, uint64_t n);

/* ---- sharding support (the reference is single-device, so no counterpart there) ---- */
/* One context per GPU holds the particles whose home cell that rank owns, followed by ghost copies of
 * the neighbours' particles within 5 cells of its region (the dependency cone of the four colour passes).
 * Ghosts take part in collisions but are not integrated; their results are discarded.  The host side
 * (gpu-physics-engine_amd/sharded.py) moves migrants and ghosts between ranks over RCCL point-to-point. */
/* Grow every buffer to hold `capacity` particles, keeping the current ones. */
gpe_status gpe_reserve(gpe_ctx *ctx, uint64_t capacity);
gpe_status gpe_capacity(const gpe_ctx *ctx, uint64_t *capacity);
/* The first n_owned of the n_total resident particles are this rank's own (integrated by K12); the rest
 * are ghosts written by the caller through gpe_device_ptr(GPE_POS / GPE_RADIUS / GPE_ORDER_KEYS). */
gpe_status gpe_set_counts(gpe_ctx *ctx, uint64_t n_total, uint64_t n_owned);
/* Order the members of a cell by GPE_ORDER_KEYS[local index] (the particle's index in the unsharded
 * system) instead of by local index, so a sharded run reproduces the single-device pair order. */
gpe_status gpe_use_order_keys(gpe_ctx *ctx, int32_t enable);
/* Cells [cx0..cx1] x [cy0..cy1] contain every resident particle: the native tile grid is cut to it. */
gpe_status gpe_set_active_cells(gpe_ctx *ctx, int32_t cx0, int32_t cy0, int32_t cx1, int32_t cy1);
/* The context's hipStream_t, so that a host framework can enqueue its own packing / exchange work in
 * order with the library's kernels. */
gpe_status gpe_stream_handle(gpe_ctx *ctx, void **hip_stream);
/* Run on a stream the CALLER owns (hipStream_t; it must outlive the context or the next gpe_set_stream).  A host
 * framework that allocates, frees or communicates on the context's stream (torch's caching allocators and
 * ProcessGroupNCCL remember the stream of every buffer they handle) lends its own stream instead of borrowing
 * the library's: the library never destroys a borrowed stream.  NULL returns to the library's own stream.
 * Synchronises the stream in use before switching. */
gpe_status gpe_set_stream(gpe_ctx *ctx, void *hip_stream);
/* Re-derive the native pipeline's configuration after the caller changed particles in place. */
gpe_status gpe_refresh(gpe_ctx *ctx);
/* For every owned particle whose 8x8-cell block (row-major blocks_x x blocks_y over the world) is owned by
 * another rank (a migrant) and/or borders other ranks (a ghost for them), append (local index, info):
 * info bits 0-25 = the block's destination-rank mask, bits 26-30 = 1 + the block's owner when that is not
 * my_rank (at most 26 ranks).  *d_out_count (device, zeroed by the caller) receives the number appended
 * (entries beyond out_capacity are dropped). */
gpe_status gpe_shard_classify(gpe_ctx *ctx, const uint8_t *d_owner_of_block, const uint32_t *d_dest_mask_of_block,
                              int32_t blocks_x, int32_t blocks_y, uint32_t my_rank, uint32_t *d_out_index,
                              uint32_t *d_out_info, uint32_t *d_out_count, uint64_t out_capacity);

/* Device-resident exchange (k_shard.hip): the same protocol without a host round trip per step.  The caller owns
 * two device buffers of u32 words and moves the neighbour segments between ranks (RCCL send/recv, fixed sizes):
 * after pack, segment s of d_send goes to rank slot_rank[s] and lands in that rank's d_recv segment for this rank.
 * Segment = [n_migrants, n_ghosts, 0, 0][cap_mig rows of 6 words: x y prev_x prev_y r key][cap_gho rows of 4 words:
 * x y r key].  Slots are the neighbouring ranks in ascending order followed by this rank (its own segment of
 * d_send holds the migrants that stay behind as ghosts; it is not sent).  Both sides must agree on the capacities.
 * Needs order keys, the native pipeline and an active box; every rank region at least two blocks wide. */
typedef struct gpe_shard_plan {
    uint32_t struct_size;          /* = sizeof(gpe_shard_plan)                                   */
    uint32_t rank, world_size, n_slots;
    int32_t  blocks_x, blocks_y;   /* block grid of the whole world (tables below)               */
    const uint8_t  *d_owner_of_block;
    const uint32_t *d_dest_mask_of_block;
    uint32_t slot_rank[9];
    uint32_t send_off[9], send_cap_mig[9], send_cap_gho[9];   /* word offsets into d_send, rows   */
    uint32_t recv_off[9], recv_cap_mig[9], recv_cap_gho[9];   /* word offsets into d_recv, rows   */
    uint32_t *d_send, *d_recv;
    int32_t  own_x0, own_y0, own_x1, own_y1;   /* the rank's rectangle in blocks, half-open (all 0: not told).  Told, the  */
                                               /* tiles of the step pack their own particles as they write them back and   */
                                               /* no pack kernel runs (struct_size without these four is accepted too)     */
} gpe_shard_plan;
gpe_status gpe_shard_configure(gpe_ctx *ctx, const gpe_shard_plan *plan);
/* Counts go to the device (owned = total = the host's owned count, ghosts dropped); packs the first segments. */
gpe_status gpe_shard_begin(gpe_ctx *ctx);
/* Consume d_recv: fill the migrants' holes, append arriving migrants, then ghosts.  No host sync. */
gpe_status gpe_shard_unpack(gpe_ctx *ctx);
/* gpe_shard_unpack + one step (State::update without re-sort) + pack of the next segments.  No host sync. */
gpe_status gpe_shard_step(gpe_ctx *ctx, float dt);
/* The counts as last mirrored to pinned host memory: no synchronisation, they lag by the steps in flight (<= ~64).
 * For capacity planning (grow the buffers before the device-side total reaches the capacity). */
gpe_status gpe_shard_peek(gpe_ctx *ctx, uint64_t *n_owned, uint64_t *n_total);
/* Synchronises and returns the device-side counts; leave != 0 also returns the context to host-side counts
 * (owned particles only), e.g. before a Morton re-sort or a download.  Reports exchange errors. */
gpe_status gpe_shard_counts(gpe_ctx *ctx, uint64_t *n_owned, uint64_t *n_total, int32_t leave);

/* Moving the packed segments between the ranks, inside the library: RCCL point-to-point over xGMI.  One grouped
 * ncclSend / ncclRecv pair per neighbouring rank on the context's stream (fixed sizes, nothing to wait for on the
 * host), so a host in any language drives a sharded run with gpe_shard_run alone between two re-sorts.
 * librccl.so.1 is loaded at the first of these calls (dlopen: a process that already holds RCCL, e.g. under
 * torch.distributed, shares its copy). */
#define GPE_COMM_ID_BYTES 128u
/* ncclGetUniqueId: call on one rank, hand the 128 bytes to every rank (any side channel). */
gpe_status gpe_comm_unique_id(uint8_t *id128);
/* ncclCommInitRank on the context's device: collective over the world_size ranks of the decomposition
 * (rank numbers = gpe_shard_plan.rank / slot_rank).  The communicator belongs to the context. */
gpe_status gpe_shard_comm_init(gpe_ctx *ctx, const uint8_t *id128, uint32_t rank, uint32_t world_size);
/* Use a communicator the caller created (ncclComm_t); the caller keeps ownership. */
gpe_status gpe_shard_comm_attach(gpe_ctx *ctx, void *nccl_comm);
gpe_status gpe_shard_comm_destroy(gpe_ctx *ctx);
/* Any other transport (tests: gloo through host memory): fn must enqueue / perform the transfer of every neighbour
 * segment of d_send into the peers' d_recv in order with hip_stream and return 0.  NULL removes it. */
typedef int32_t (*gpe_shard_transport_fn)(void *user, const uint32_t *d_send, uint32_t *d_recv, void *hip_stream);
gpe_status gpe_shard_set_transport(gpe_ctx *ctx, gpe_shard_transport_fn fn, void *user);
/* Send the segments packed by gpe_shard_begin / gpe_shard_step and receive the neighbours' (communicator or transport). */
gpe_status gpe_shard_exchange(gpe_ctx *ctx);
/* `steps` x (gpe_shard_exchange, gpe_shard_step): the step loop of a sharded run between two re-sorts, no host
 * synchronisation (the host stays at most ~64 steps ahead of the device). */
gpe_status gpe_shard_run(gpe_ctx *ctx, float dt, uint64_t steps);
/* Do librccl.so.1 and every entry point used above resolve on this machine?  (No GPU needed.) */
gpe_status gpe_comm_probe(void);

/* ---- sharded control plane: decomposition, set-up, global re-sort, load re-cut, the scheduled run -----------------
 * Everything a host needs to drive a sharded run through this header alone (no Python, no torch): the reference's
 * State::update schedule (state.rs:115-131: re-sort gate, then the step) across ranks.  One context per rank -- one
 * process per GPU, or one thread per context inside one process (gpe_local_group_*).  The collective calls below
 * (marked so) must be made by every rank, in the same order. */
#define GPE_SHARD_MAX_RANKS 26u
/* The world cut into px x py rectangles of 8x8-cell blocks (rank = j * px + i owns block columns xcuts[i]..xcuts[i+1],
 * block rows ycuts[j]..ycuts[j+1]).  A plain value: every rank builds the same one from the same arguments. */
typedef struct gpe_shard_layout {
    uint32_t struct_size;                       /* = sizeof(gpe_shard_layout)                                  */
    uint32_t world_size, px, py;
    float    world_width, world_height, cell_size;
    int32_t  cells_x, cells_y;                  /* home cell columns / rows: floor(world / cell) + 1           */
    int32_t  blocks_x, blocks_y;                /* 8x8-cell blocks                                             */
    int32_t  xcuts[27];                         /* px + 1 entries rising from 0 to blocks_x (27 = MAX_RANKS + 1) */
    int32_t  ycuts[27];                         /* py + 1 entries rising from 0 to blocks_y                    */
} gpe_shard_layout;
/* Host only (no GPU needed).  px = py = 0: the process grid as square as possible, px <= py.  xcuts / ycuts NULL: equal
 * widths; else px + 1 / py + 1 entries (what gpe_shard_recut derives from the particle quantiles). */
gpe_status gpe_shard_layout_build(float world_width, float world_height, float cell_size, uint32_t world_size,
                                  uint32_t px, uint32_t py, const int32_t *xcuts, const int32_t *ycuts,
                                  gpe_shard_layout *out);
/* Owner rank of each of n host positions (interleaved x,y; the kernels' own f32 arithmetic: floor(p / cell) >> 3,
 * clamped to the block grid): how a host deals the initial particles to the ranks. */
gpe_status gpe_shard_layout_owner_of(const gpe_shard_layout *layout, const float *pos_xy, uint64_t n, uint8_t *owner_out);
/* Cut `bins` block columns (rows) into `parts` runs of about equal particle count, every run at least min_width blocks
 * wide: cuts_out receives parts + 1 entries.  Pure function (every rank derives the same cuts from the same histogram). */
gpe_status gpe_shard_quantile_cuts(const uint64_t *hist, uint32_t bins, uint32_t parts, uint32_t min_width, int32_t *cuts_out);

/* What the control plane needs from "the other ranks", when it is not the in-library RCCL communicator
 * (gpe_shard_comm_init / _attach) or a local group (gpe_local_group_join): two collectives over device memory, both
 * ordered with hip_stream; they may block the host.  Return 0 on success. */
enum { GPE_REDUCE_SUM = 0, GPE_REDUCE_MAX = 1 };
typedef struct gpe_shard_collectives {
    uint32_t struct_size;
    uint32_t reserved;
    void *user;
    /* in-place all-reduce of `count` u32 words at d_buf over all ranks (op: GPE_REDUCE_*) */
    int32_t (*all_reduce_u32)(void *user, uint32_t *d_buf, uint64_t count, uint32_t op, void *hip_stream);
    /* for every rank r: send_count[r] words at d_send + send_off[r] go to rank r, which finds them at its d_recv +
     * recv_off[this rank]; recv_count[r] words arrive from rank r.  (ncclSend / ncclRecv pairs in one group.) */
    int32_t (*all_to_all_u32)(void *user, const uint32_t *d_send, const uint64_t *send_off, const uint64_t *send_count,
                              uint32_t *d_recv, const uint64_t *recv_off, const uint64_t *recv_count, void *hip_stream);
} gpe_shard_collectives;
/* NULL: back to the in-library communicator.  The struct is copied. */
gpe_status gpe_shard_set_collectives(gpe_ctx *ctx, const gpe_shard_collectives *coll);

/* Several contexts of ONE process as the ranks of a sharded run: one host thread per context (every collective call
 * blocks until all ranks have made it), segments and collectives moved by hipMemcpyAsync / kernels between the
 * contexts' buffers (same device, or peers).  What a single-process host -- the reference's State owns one device,
 * renderer/wgpu_context.rs:42-49 -- grows into on a multi-GPU node without a collective library. */
typedef struct gpe_local_group gpe_local_group;
gpe_status gpe_local_group_create(uint32_t world_size, gpe_local_group **out);
gpe_status gpe_local_group_destroy(gpe_local_group *group);
/* The context becomes rank `rank` of the group: its collectives and its segment transport are the group's. */
gpe_status gpe_local_group_join(gpe_ctx *ctx, gpe_local_group *group, uint32_t rank);
/* A rank that failed leaves the others waiting in a collective: this wakes them all with an error. */
gpe_status gpe_local_group_abort(gpe_local_group *group);

/* ParticleSystem::new_from_buffers (particle_system.rs:49-99) for one rank: its n owned particles with their global
 * indices (order_key[i] = the particle's index in the unsharded system), buffers sized for `capacity` particles
 * (0: 1.3 n + 4096; ghosts and arrivals need room).  prev_xy NULL: previous = current. */
gpe_status gpe_shard_set_particles(gpe_ctx *ctx, const float *pos_xy, const float *prev_xy, const float *radius,
                                   const uint32_t *order_key, uint64_t n, uint64_t capacity);
/* COLLECTIVE.  Makes the context rank `rank` of `layout`: agrees on the cell size of the whole system (2.2 x the
 * largest radius over all ranks, grid.rs:159-161) and checks the layout was cut with it, switches the order keys on,
 * cuts the tile grid to the rank's rectangle + ghost ring, sizes the neighbour segments from the densest rank's block
 * population x capacity_scale (1.0; both ends of a pair get the same numbers, and the set-up compares them: a send and
 * a receive of different lengths would wait for ever), allocates the tables and segment buffers inside the library and
 * configures the device-resident exchange (gpe_shard_configure).  Needs collectives: a communicator, a local group or
 * gpe_shard_set_collectives; every rectangle at least two blocks wide, at most 8 neighbours. */
gpe_status gpe_shard_setup(gpe_ctx *ctx, const gpe_shard_layout *layout, uint32_t rank, float capacity_scale);
/* The layout in use (gpe_shard_recut replaces it). */
gpe_status gpe_shard_get_layout(const gpe_ctx *ctx, gpe_shard_layout *out);
/* COLLECTIVE.  ParticleSort::sort (particle_sort.rs:58-69) across the ranks: every particle goes home to its owner
 * (exchange + unpack, ghosts dropped), then K1 + stable sort by home-cell key + K4 on the owned particles in the order
 * of their old global indices, and the NEW global indices -- the particle's position in the single-device sorted order --
 * from one all-reduce of the histogram over Morton blocks (a block of 64 consecutive keys = one 8x8-cell block = one
 * owner): index = particles of all ranks in earlier blocks + position inside the block.  Leaves the context with
 * host-side counts; the next gpe_shard_run_scheduled / gpe_shard_begin starts the device-resident loop again. */
gpe_status gpe_shard_resort(gpe_ctx *ctx);
/* COLLECTIVE; call where gpe_shard_resort may be called (it is called BY gpe_shard_run_scheduled before its re-sorts).
 * When the most loaded rank owns more than `above` x the mean (1.25; <= 0: never): new cuts at the particle quantiles of
 * the all-reduced block-column / block-row histograms, every particle moved to its new owner (one all-to-all), tables,
 * tile grid and segments re-planned.  Results do not depend on the cuts.  *recut (may be NULL) = 1 when they changed. */
gpe_status gpe_shard_recut(gpe_ctx *ctx, float above, int32_t *recut);
/* COLLECTIVE.  State::update (state.rs:115-131) `steps` times for this rank: a re-sort (gpe_shard_recut at 1.25, then
 * gpe_shard_resort) before step 0 when resort_first != 0 and before every resort_every-th step (0: never), and
 * gpe_shard_run for the steps in between -- no host synchronisation except at the re-sorts. */
gpe_status gpe_shard_run_scheduled(gpe_ctx *ctx, float dt, uint64_t steps, uint64_t resort_every, int32_t resort_first);
/* The rank's owned particles as host arrays (synchronises; the counts come from the device): up to `capacity` of
 * them, *n_owned receives their number.  order_key_out / pos_xy_out / prev_xy_out may each be NULL. */
gpe_status gpe_shard_download_owned(gpe_ctx *ctx, uint32_t *order_key_out, float *pos_xy_out, float *prev_xy_out,
                                    uint64_t capacity, uint64_t *n_owned);
/* Counters of the control plane since gpe_shard_setup. */
typedef struct gpe_shard_stats {
    uint32_t struct_size;
    uint32_t recuts;             /* gpe_shard_recut calls that changed the cuts                          */
    uint64_t resorts;            /* gpe_shard_resort calls                                               */
    uint64_t steps;              /* steps run by gpe_shard_run_scheduled                                 */
    uint64_t n_owned, n_ghost;   /* as of the last synchronising call                                    */
    uint32_t n_neighbours;
    uint32_t transport;          /* 0 none, 1 RCCL inside the library, 2 local group, 3 caller callbacks */
} gpe_shard_stats;
gpe_status gpe_shard_get_stats(gpe_ctx *ctx, gpe_shard_stats *out);

/* ---- profiling (wgpu_profiler scopes threaded through every reference call) ------------------ */
typedef struct gpe_timing {
    char     name[64];    /* the reference's scope label, e.g. "Sort map" (grid.rs:329); kernel-level
                             entries are "<scope>/<kernel>"                                   */
    double   total_ms;    /* sum over calls since the last gpe_reset_timings                  */
    uint64_t calls;
} gpe_timing;
/* on = 0: off; 1: every scope of every call; k > 1: the scopes of every k-th step only (sampled --
   an event pair per kernel costs about as much as a small kernel). */
gpe_status gpe_set_profiling(gpe_ctx *ctx, uint32_t on);
gpe_status gpe_reset_timings(gpe_ctx *ctx);
/* Synchronises, then writes up to *count entries; *count receives the number available. */
gpe_status gpe_get_timings(gpe_ctx *ctx, gpe_timing *out, uint32_t *count);

/* The reference's `--features benchmark` build writes every finished frame's scopes as a Chrome trace
 * (state.rs:108-112, wgpu_profiler::chrometrace): one entry per recorded scope instance, start relative to the
 * last gpe_reset_timings (or to the gpe_set_profiling call that switched profiling on).  The newest 65536
 * instances are kept.  Same calling convention as gpe_get_timings; gpu-physics-engine_amd/engine.py
 * (Context.write_chrome_trace) turns them into the JSON chrome://tracing loads. */
typedef struct gpe_trace_event {
    char   name[64];
    double start_ms;
    double duration_ms;
} gpe_trace_event;
gpe_status gpe_get_trace(gpe_ctx *ctx, gpe_trace_event *out, uint32_t *count);

#ifdef __cplusplus
}
#endif
#endif /* GPE_H */
