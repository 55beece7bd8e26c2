// gpe_internal.h -- shared declarations of the gfx950 implementation behind include/gpe.h.
// HIP for CDNA4 only: wave64, no portability layers.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/gpe.h"

namespace gpe {

constexpr uint32_t kUnused = GPE_UNUSED_CELL_ID;
constexpr int kWave = 64;
// Streaming kernels cap their grid and grid-stride the rest (256 CUs x 8 blocks).
constexpr int kStreamBlock = 256;
constexpr int kMaxStreamGrid = 256 * 8;

inline int stream_grid(uint64_t items, int per_block = kStreamBlock)
{
    uint64_t g = (items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > (uint64_t)kMaxStreamGrid) g = kMaxStreamGrid;
    return (int)g;
}

// ------------------------------------------------------------------------------------------------
// device helpers shared by several kernels (each restates a WGSL helper; citations are into
// /root/reference/src)
// ------------------------------------------------------------------------------------------------
// grid.wgsl:101-108 / home_cell_ids.wgsl:38-45
__device__ __forceinline__ uint32_t split_by_bits(uint32_t n)
{
    uint32_t x = n & 0x0000FFFFu;
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}
// grid.wgsl:112-114 ; u32(i32) is a bit cast
__device__ __forceinline__ uint32_t morton_encode(int32_t x, int32_t y)
{
    return split_by_bits((uint32_t)x) | (split_by_bits((uint32_t)y) << 1);
}
// collision_solver.wgsl:123-130
__device__ __forceinline__ uint32_t unsplit_by_bits(uint32_t n)
{
    uint32_t x = n & 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu;
    x = (x | (x >> 8)) & 0x0000FFFFu;
    return x;
}
// collision_solver.wgsl:55-58
__device__ __forceinline__ uint32_t cell_color(uint32_t cell_hash)
{
    return 1u + (unsplit_by_bits(cell_hash) & 1u) + (unsplit_by_bits(cell_hash >> 1) & 1u) * 2u;
}
// WGSL i32(f32): truncate, saturate, NaN -> 0.  v_cvt_i32_f32 does exactly that on gfx950, but
// the C++ cast is undefined out of range, so spell it out (the compiler folds it back).
__device__ __forceinline__ int32_t f32_to_i32_sat(float f)
{
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 0x7fffffff;
    if (f <= -2147483648.0f) return (int32_t)0x80000000;
    return (int32_t)f;
}
// grid.wgsl:53 / home_cell_ids.wgsl:27 : vec2<i32>(floor(pos / cell_size)) -- a true division.
__device__ __forceinline__ int32_t cell_coord(float p, float cell_size)
{
    return f32_to_i32_sat(floorf(p / cell_size));
}
// WGSL clamp(e, lo, hi) = min(max(e, lo), hi), written with compares so NaN/-0 behave as in the oracle
__device__ __forceinline__ float clamp_f(float x, float lo, float hi)
{
    float m = (x > lo) ? x : lo;
    return (m < hi) ? m : hi;
}
// grid.wgsl:117-129 is_obj_in_cell
__device__ __forceinline__ bool is_obj_in_cell(float px, float py, float sq_radius, int32_t cx,
                                               int32_t cy, float cs)
{
    float lo_x = (float)cx * cs, lo_y = (float)cy * cs;
    float hi_x = lo_x + cs, hi_y = lo_y + cs;
    float qx = clamp_f(px, lo_x, hi_x), qy = clamp_f(py, lo_y, hi_y);
    float dx = px - qx, dy = py - qy;
    float dist_sq = dx * dx + dy * dy;
    return dist_sq < sq_radius;
}

// ------------------------------------------------------------------------------------------------
// wave64 / block helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// The lanes of the wave for which `p` holds, as a mask.  (The builtin on the predicate itself: HIP's __ballot(int) turns
// the lane mask the compare has just produced into a 0 / 1 register and compares that again -- two VALU instructions per
// vote, and the tile kernels vote ~40 times per workgroup pass.)
__device__ __forceinline__ uint64_t ballot64(const bool p) { return __builtin_amdgcn_ballot_w64(p); }
// ... and back: the per-lane predicate of a (wave-uniform) lane mask.  Costs nothing: selects and branches take the
// mask as it is.  Votes on single comparisons combined by scalar ANDs, turned back into a predicate here, replace
// `a && b && c` where the combined predicate is also voted on (pair_response, k_native.hip).
__device__ __forceinline__ bool lanes_of(const uint64_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

// lane LANE of `v` = the wave-uniform `value`: one v_writelane_b32
template <int LANE>
__device__ __forceinline__ void write_lane(int &v, const int value)
{
    asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(value), "n"(LANE));
}

// number of set bits of `m` strictly below this lane
__device__ __forceinline__ uint32_t popc_below_lane(uint64_t m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// inclusive scan across the 64 lanes of a wave: six DPP adds (row shifts by 1, 2, 4, 8 inside the 16-lane rows, then
// the row totals broadcast from lane 15 into rows 1 and 3 and from lane 31 into rows 2 and 3) instead of six
// ds_bpermute round trips through the LDS crossbar with a select each.  A source lane outside the row / wave reads
// as 0 (bound_ctrl), which is the identity of the sum.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add(uint32_t v)
{
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);
}
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
    v = dpp_add<0x111, 0xF>(v);        // row_shr:1
    v = dpp_add<0x112, 0xF>(v);        // row_shr:2
    v = dpp_add<0x114, 0xF>(v);        // row_shr:4
    v = dpp_add<0x118, 0xF>(v);        // row_shr:8
    v = dpp_add<0x142, 0xA>(v);        // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xC>(v);        // row_bcast:31 into rows 2 and 3
    return v;
}
// sum over the 64 lanes, the same in every lane
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(v), 63);
}

// exclusive scan of one value per thread over a 256-thread block; `total` receives the block sum.
// s_w: >= 4 words of LDS scratch.  Contains __syncthreads (all 256 threads must call it).
__device__ __forceinline__ uint32_t block256_exclusive_scan(uint32_t v, uint32_t *s_w, uint32_t *total)
{
    const int lane = lane_id();
    const int w = (int)(threadIdx.x >> 6);
    uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    uint32_t w0 = s_w[0], w1 = s_w[1], w2 = s_w[2], w3 = s_w[3];
    uint32_t base = (w > 0 ? w0 : 0u) + (w > 1 ? w1 : 0u) + (w > 2 ? w2 : 0u);
    if (total) *total = w0 + w1 + w2 + w3;
    __syncthreads();
    return base + inc - v;
}

// LDS words shared by the lanes of ONE wave between barriers-free steps (per-wave counters, match tables).
// Not `volatile`: volatile accesses are left in the generic address space (flat_load/flat_store ... sc0 sc1
// followed by s_waitcnt vmcnt(0)), several times the cost of ds_read/ds_write.  Relaxed wavefront-scope atomics
// compile to plain ds_* instructions; wave_lds_order() keeps the compiler from moving them across each other
// (the LDS itself executes a wave's instructions in order).
template <typename T>
__device__ __forceinline__ T wave_lds_load(const T *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
template <typename T>
__device__ __forceinline__ void wave_lds_store(T *p, T v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
__device__ __forceinline__ void wave_lds_order()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// LDS histogram add for one wave-round: one atomic when the whole wave holds one digit (the common
// case for the high digits of nearly sorted keys), else one atomic per lane.
__device__ __forceinline__ void hist_add(uint32_t *s_hist, uint32_t d, bool valid)
{
    // Only scalar work on the critical path (ballots, one v_readlane): either the whole wave holds one
    // digit value -- one atomic -- or every lane issues its own fire-and-forget LDS atomic (the LDS
    // serialises lanes that hit the same bin; measured cheaper than peeling the values off with ballots:
    // the hash kernel went from 0.44 to 0.70 ms at 100 M particles with a four-value peel).
    const uint64_t m = ballot64(valid);
    if (m == 0) return;                                        // wave-uniform
    const int first = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(m));
    const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, first);
    if (ballot64(valid && d != d0) == 0) {
        if (lane_id() == first) atomicAdd(&s_hist[d0], (uint32_t)__popcll(m));
        return;
    }
    if (valid) atomicAdd(&s_hist[d], 1u);
}

struct VerletParams {
    float dt_squared;
    float world_w, world_h;
    float acc_x, acc_y;          // FORCE_OF_GRAVITY (particle_integration.wgsl:21)
    uint32_t mouse_pressed;
    float mouse_x, mouse_y, mouse_strength;
};

// particle_integration.wgsl:34-76 for one particle
__device__ __forceinline__ void verlet_one(float cx, float cy, float qx, float qy, float r,
                                           const VerletParams &P, float &nx, float &ny)
{
    float vx = cx - qx, vy = cy - qy;                       // :40
    float ax = P.acc_x, ay = P.acc_y;                       // :42
    if (P.mouse_pressed == 1u) {                            // :44
        float dx = P.mouse_x - cx, dy = P.mouse_y - cy;     // :46
        float len = sqrtf(dx * dx + dy * dy);               // :50 normalize()
        ax = ax + (dx / len) * P.mouse_strength;            // :50,53
        ay = ay + (dy / len) * P.mouse_strength;
    }
    nx = (cx + vx) + ax * P.dt_squared;                     // :59
    ny = (cy + vy) + ay * P.dt_squared;
    nx = clamp_f(nx, r, P.world_w - r);                     // :70
    ny = clamp_f(ny, r, P.world_h - r);                     // :71
}

// ------------------------------------------------------------------------------------------------
// host-side context
// ------------------------------------------------------------------------------------------------
struct ScopeStat {
    std::string name;
    double total_ms = 0.0;
    uint64_t calls = 0;
};
struct PendingEvent {
    int stat;
    hipEvent_t start, stop;
};
struct TraceEvent {                  // one resolved scope instance (gpe_get_trace)
    int stat;
    double start_ms, dur_ms;         // start relative to the trace origin (gpe_set_profiling / gpe_reset_timings)
};

struct ScanWorkspace {           // reduce-then-scan tile sums, one array per recursion level
    std::vector<uint32_t *> level;
    std::vector<uint64_t> cap;
};

struct SortWorkspace {
    uint32_t *keys_b = nullptr, *vals_b = nullptr;   // ping-pong partners, cap entries each
    uint64_t cap = 0;
    uint32_t *counts = nullptr;                      // [256][tiles] per-tile digit counts
    uint64_t counts_cap = 0;
    uint32_t *hist4 = nullptr;                       // 4 x 256 global digit histograms
};

struct OnesweepWorkspace {
    uint64_t *status = nullptr;      // [tiles][256] {epoch:30, flag:2, value:32}
    uint64_t status_cap = 0;
    uint32_t *hist4 = nullptr;       // 4 x 256 digit histograms
    uint32_t *bases4 = nullptr;      // 4 x 256 exclusive digit bases
    uint32_t *hist_plain = nullptr;  // 4 x 256: histograms of the sorts whose producer brings none
    uint32_t *ctl = nullptr;         // [0..3] tile tickets per pass, [4] error word
    uint32_t epoch = 0;
    bool hist_clean = false;         // the set the next native step adds into is zero: the native step keeps it so
    uint32_t hist_set = 0;           // which of the two sets of copies that is
};

// Straggler lists, ghost lists and rosters are indexed by the tile's position in the TILE BOX of the run: the whole cell
// box, or a sharded rank's active box (so that their memory follows the rank's share of the world, not the world).
struct TileBox {
    int32_t x0 = 0, y0 = 0, nx = 0, ny = 0;     // first tile (32x32 cells), tiles per row / column
    __host__ __device__ __forceinline__ bool holds(int tx, int ty) const { return tx >= x0 && ty >= y0 && tx < x0 + nx && ty < y0 + ny; }
    __host__ __device__ __forceinline__ uint32_t index(int tx, int ty) const { return (uint32_t)(ty - y0) * (uint32_t)nx + (uint32_t)(tx - x0); }
};
// pinned host words the native kernels report to (NativeState::host_stat; read by the step policy and by
// gpe_get_pipeline_info with a lag of the steps in flight)
constexpr int kStatWindowMax = 0;              // largest 24x24-cell window population of the last native step
constexpr int kStatArena = 1;                  // spill-arena slots handed out by the last native step
constexpr int kStatProbe = 2;                  // window population measured by the last asynchronous probe + 1 (0: none yet)
constexpr int kStatOverflow = 3;               // 32x32 tiles over capacity in the last native step
constexpr int kStatSubTiles = 4, kStatSpills = 5;   // quarters redone as 8x8 tiles / 8x8 tiles through the arena (diagnostics)
constexpr int kStatSorts = 6;                  // running count of steps whose radix passes ran (tile_ctl[kCtlSorts]), lagged
constexpr int kStatOverflowNew = 7;            // ... of kStatOverflow, the tiles the dense launch handed on itself (the others were known: hinted)
constexpr int kStatHalvesOver = 8;             // halves handed on to the over-capacity launch in the last native step
constexpr int kNativeCtlSorts = 14;         // tile_ctl word: running count of steps whose radix passes ran (k_native.hip kCtlSorts)
constexpr int kNativeCtlSortsSeen = 38;     // its copy in the line the tiles only read (k_native.hip kCtlSortsSeen)
// Native (N-key sort + LDS cell windows) pipeline state
struct NativeState {
    bool eligible = false;           // every particle inside the world box, grid small enough, windows not over-dense
    bool in_box = false;             // the configuration-time box check passed (density is the only possible obstacle)
    int32_t gx = 0, gy = 0;          // home cell columns / rows: cx in [0,gx), cy in [0,gy)
    int passes = 4;                  // radix passes needed for morton(gx-1, gy-1)
    uint32_t table_entries = 0;      // 8x8-cell block table length
    uint2 *block_table = nullptr;    // [morton >> 6] = (start, end) in the sorted order
    uint64_t table_cap = 0;
    uint32_t *keys = nullptr, *ids = nullptr;       // N each: hash output / sort ping
    uint32_t *keys_b = nullptr, *ids_b = nullptr;   // sort pong
    uint64_t cap = 0;
    uint32_t *codes = nullptr;       // per particle: cell relative to its sorted block | neighbour overlap mask | straggler (k_native_hash)
    uint32_t *gkeys = nullptr, *gids = nullptr, *gkeys_b = nullptr, *gids_b = nullptr;   // sharded runs: the ghosts' sort
    uint64_t gcap = 0;
    uint2 *gtable = nullptr;         // ... and their block table
    uint32_t *ghist = nullptr;       // ... and its digit histograms (two sets of kHistCopies copies, like os_ws.hist4)
    uint32_t ghist_set = 0;
    uint64_t gtable_cap = 0;
    const uint32_t *gsorted_ids_now = nullptr;   // the ghosts' indices in block order, this step (NULL: no ghost table)
    uint32_t *exc_count = nullptr;   // straggler lists: [2][exc_tiles] counts, then [2][exc_tiles][16] entries (one allocation)
    uint2 *exc_entry = nullptr;
    uint64_t exc_tiles = 0, exc_cap = 0;
    uint4 *roster_hdr = nullptr;           // tile rosters (k_native.hip, CollideArgs)
    uint32_t *roster_ids = nullptr;
    uint64_t roster_cap = 0;
    TileBox tb;                      // the tiles the straggler lists, ghost lists and rosters are kept for
    uint32_t *gho_count = nullptr;   // sharded runs: ghost lists, [2][exc_tiles] counts then [2][exc_tiles][kGhostSlots] ids
    uint64_t gho_cap = 0;
    const uint32_t *gho_count_now = nullptr, *gho_entry_now = nullptr, *ghost_sort_now = nullptr;   // what this step's tiles read
    const uint32_t *exc_count_now = nullptr;   // the set the current step's tiles read (NULL: the step sorts anyway)
    const uint2 *exc_entry_now = nullptr;
    uint32_t *sorted_key = nullptr;  // per particle: the block key it had when the radix passes last ran (written by their
                                     // first pass); the sorted ids and the block table describe that grouping
    bool sort_state_valid = false;   // sorted_key / sorted ids / block table belong to the current particle set and box
    uint64_t sorted_n = 0;           // ... of this many particles
    uint32_t sort_hold = 0;          // steps left that sort unconditionally (the scene sorted on most steps anyway)
    uint32_t watch_steps = 0, watch_sorts = 0;   // the passes' counter over the current 64-step window
    bool watch_valid = false;
    uint32_t calm_steps = 0;         // steps without an over-capacity tile or a crowded window while `crowded`
    bool crowded = false;            // many tiles run over the direct-slot form: the dense launch uses the counting-sort form
    bool hist_fused = false;         // the hash kernel counts the radix digits (most recent steps sorted), not the gated launch
    uint32_t hist_watch_steps = 0, hist_watch_sorts = 0;
    uint32_t quiet_steps = 0;        // native steps since the tiles last reported an over-capacity 32x32 tile (lagged)
    uint32_t step_seq = 0;           // native_prepare_step calls: its parity selects the per-step control words
    uint32_t collide_seq = 0;        // native_collide calls: numbers the dense launches for the tile hints (k_native.hip kCtlHints)
    uint32_t hint_quiet = 0xFFFFFFFFu; // native steps since a tile last ran over, hinted ones included (lagged): the dense launch's front workgroups
    uint32_t new_streak = 0;         // consecutive native steps whose (lagged) list 1 was not empty
    uint32_t dense_quiet = 0;        // native steps since list 1 or list 2 last had an entry (lagged): the over-capacity launch's grid
    const uint32_t *fresh_word = nullptr;   // tile_ctl word the tiles of the current step read (did the passes run?)
    uint32_t reason = GPE_REASON_NO_PARTICLES;   // why the native kernels do not run (GPE_REASON_*), NONE when they do
    uint64_t native_steps = 0, compat_steps = 0;
    bool always_sort = false;        // GPE_FLAG_SORT_EVERY_STEP (A/B measurements, tests): sort every step as rounds 1-2 did
    int32_t blocks_x = 0, blocks_y = 0;   // 8x8-cell blocks of the block box: table index = (by - by0) * blocks_x + (bx - bx0)
    int32_t bx0 = 0, by0 = 0;        // first block of the box (sharded runs: the rank's active box; else 0, 0)
    uint32_t *tile_ctl = nullptr;    // device control words (k_native.hip kCtl*)
    uint32_t *overflow1 = nullptr;   // packed (ty << 16 | tx) of 32x32 tiles over capacity
    uint64_t overflow_cap = 0;
    void *arena = nullptr;           // global spill arena for those tiles' particle arrays (37 B per slot)
    uint64_t arena_cap = 0;          // slots
    bool force = false;              // GPE_FLAG_NATIVE_FORCE (tests): no hand-over to the compat kernels
    bool print_stats = false;        // GPE_FLAG_NATIVE_STATS: print the step statistics every 128 steps
    uint32_t stat_calls = 0;
    uint32_t *host_stat = nullptr;   // pinned, 16 words (k_native.hip kStat*): window maximum, arena use, probe answer, overflow tiles
    uint32_t window_max = 0;         // the same, measured synchronously at configuration time
    bool dense_hold = false;         // left the native path because windows were filling up
    uint32_t steps_since_check = 0;
};

// Device-resident halo exchange of a sharded run (k_shard.hip): particle counts live on the device, the host
// only keeps an upper bound.  Slots = the neighbouring ranks in ascending order, then this rank itself.
constexpr int kShardMaxSlots = 9;
// counts[] words.  Two sets of them (kShardSetWords apart), alternating with every unpack: the unpack kernel reads the
// old set in all its workgroups while one of them writes the new one.  The error word is sticky and shared: always
// word kShardError of set 0.
constexpr int kShardOwned = 0, kShardTotal = 1, kShardError = 2, kShardEpoch = 3, kShardHoles = 4;
constexpr int kShardSetWords = 8;
constexpr int kShardDoneTicket = 5;              // set 0: workgroups of the unpack launch that are through (the last one resets the send headers)
struct ShardSlots {                  // passed to the kernels by value
    uint32_t n_slots;
    uint32_t rank[kShardMaxSlots];
    uint32_t send_off[kShardMaxSlots], send_cap_mig[kShardMaxSlots], send_cap_gho[kShardMaxSlots];
    uint32_t recv_off[kShardMaxSlots], recv_cap_mig[kShardMaxSlots], recv_cap_gho[kShardMaxSlots];
    int8_t slot_of_rank[32];
};
constexpr uint32_t kShardErrSendOverflow = 1u;   // more rows for a neighbour than its segment takes
constexpr uint32_t kShardErrNoSlot = 2u;         // a particle needs a rank that is not a neighbour (moved > 1 block)
constexpr uint32_t kShardErrCapacity = 4u;       // owned + ghosts exceed the particle capacity
constexpr uint32_t kShardErrHoles = 8u;          // more migrants in one step than the hole list takes
constexpr uint32_t kShardErrRecvOverflow = 16u;  // a received header claims more rows than the segment holds
constexpr int kSegHeader = 4;                    // words: [n_migrants, n_ghosts, 0, 0]
constexpr int kMigWords = 6, kGhoWords = 4;      // migrant row: x y prev_x prev_y r key; ghost row: x y r key

// rows of one (slot, kind): a wave-aggregated append; every lane of the wave must call it
__device__ __forceinline__ uint32_t wave_append(uint32_t *counter, bool want)
{
    const uint64_t m = ballot64(want);
    if (m == 0) return 0xFFFFFFFFu;
    const int leader = (int)__builtin_ctzll(m);
    uint32_t base = 0;
    if (lane_id() == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader, 64);
    return want ? base + popc_below_lane(m) : 0xFFFFFFFFu;
}

// What a particle's NEW position means for the neighbours (k_shard.hip, "pack"): the owner of its block takes it over
// when that is another rank (a migrant: listed as a hole here), and every rank within one block of it gets a ghost copy.
// Shared by the pack kernel (k_shard_pack: all owned particles, the first pack of a run) and by the tiles of the
// native step, which pack their own particles as they write them back (k_native.hip: the pack kernel then never runs).
struct PackArgs {
    const uint8_t *owner = nullptr;              // gpe_shard_plan tables over the global block grid
    const uint32_t *dest_mask = nullptr;
    int32_t blocks_x = 0, blocks_y = 0;
    uint32_t my_rank = 0;
    uint32_t *send = nullptr, *counts = nullptr, *holes = nullptr;   // counts: the set in use (kShardOwned, kShardHoles)
    uint32_t *err = nullptr;                     // the sticky error word
    uint8_t *hole_flag = nullptr;
    uint32_t holes_cap = 0;
    // the tiles: a new position inside the `safe` box (world units: the rank's rectangle shrunk by one block + one cell on
    // every side that has a neighbour) concerns no other rank -- its block is this rank's and borders nobody
    float safe_x0 = 0.f, safe_y0 = 0.f, safe_x1 = 0.f, safe_y1 = 0.f;
    uint32_t on = 0;                             // 1: pack; 2: this launch runs behind the exchange -- a particle that would
                                                 // have to be packed is an error (it moved more than a block in one step)
    ShardSlots slots;
};

// mine: this lane holds an owned particle (local index i, new position p, previous position q).  Every lane of the
// wave must call it (ballots).
__device__ __forceinline__ void pack_particle(const PackArgs &P, const bool mine, const uint32_t i, const float2 p,
                                              const float2 q, const float rad, const uint32_t key, const float cell_size)
{
    uint32_t gho = 0;
    int mig = -1;
    if (mine) {
        int bx = cell_coord(p.x, cell_size) >> 3, by = cell_coord(p.y, cell_size) >> 3;
        bx = min(max(bx, 0), P.blocks_x - 1);
        by = min(max(by, 0), P.blocks_y - 1);
        const uint32_t b = (uint32_t)by * (uint32_t)P.blocks_x + (uint32_t)bx;
        const uint32_t owner = P.owner[b];
        gho = P.dest_mask[b] & 0x03FFFFFFu;             // ranks within one block of b, its owner excluded
        if (owner != P.my_rank) mig = (int)owner;       // b's owner takes the particle over
    }
    if (ballot64(gho != 0 || mig >= 0) == 0) return;    // interior wave
    uint32_t err = 0;
    {
        const uint32_t h = wave_append(&P.counts[kShardHoles], mig >= 0);
        if (mig >= 0) {
            if (h < P.holes_cap) { P.holes[h] = i; P.hole_flag[i] = 1; } else err |= kShardErrHoles;
            if (P.slots.slot_of_rank[mig & 31] < 0) err |= kShardErrNoSlot;
        }
    }
    uint32_t served = 0;
    for (uint32_t s = 0; s < P.slots.n_slots; ++s) {
        const uint32_t rk = P.slots.rank[s];
        uint32_t *seg = P.send + P.slots.send_off[s];
        const bool wm = mig == (int)rk;
        const bool wg = ((gho >> rk) & 1u) != 0;
        served |= wg ? (1u << rk) : 0u;
        const uint32_t rm = wave_append(seg + 0, wm);
        if (wm) {
            if (rm < P.slots.send_cap_mig[s]) {
                uint32_t *row = seg + kSegHeader + (uint64_t)rm * kMigWords;
                row[0] = __float_as_uint(p.x); row[1] = __float_as_uint(p.y);
                row[2] = __float_as_uint(q.x); row[3] = __float_as_uint(q.y);
                row[4] = __float_as_uint(rad); row[5] = key;
            } else err |= kShardErrSendOverflow;
        }
        const uint32_t rg = wave_append(seg + 1, wg);
        if (wg) {
            if (rg < P.slots.send_cap_gho[s]) {
                uint32_t *row = seg + kSegHeader + (uint64_t)P.slots.send_cap_mig[s] * kMigWords + (uint64_t)rg * kGhoWords;
                row[0] = __float_as_uint(p.x); row[1] = __float_as_uint(p.y);
                row[2] = __float_as_uint(rad); row[3] = key;
            } else err |= kShardErrSendOverflow;
        }
    }
    if (gho & ~served) err |= kShardErrNoSlot;
    if (err) atomicOr(P.err, err);
}

struct ShardState {
    bool on = false;                 // gpe_shard_configure was called
    bool active = false;             // between gpe_shard_begin and gpe_shard_counts: counts are on the device
    bool packed = false;             // the send buffer holds this step's rows
    bool have_rect = false;          // the plan told the rank's rectangle: the tiles of the step pack (no pack kernel)
    int32_t rect[4] = {0, 0, 0, 0};  // own rectangle in blocks, half-open: x0, y0, x1, y1
    uint32_t my_rank = 0;
    int32_t blocks_x = 0, blocks_y = 0;          // global block grid of the decomposition
    const uint8_t *owner = nullptr;              // caller's device tables (gpe_shard_plan)
    const uint32_t *dest_mask = nullptr;
    uint32_t *send = nullptr, *recv = nullptr;   // caller's device buffers
    ShardSlots slots;
    uint32_t *counts = nullptr;      // device, two sets of kShardSetWords words (kShard*)
    uint32_t parity = 0;             // the set in use: counts + parity * kShardSetWords (flips with every unpack)
    uint32_t *counts_now() const { return counts + parity * kShardSetWords; }
    uint32_t *host_counts = nullptr; // pinned mirror written by the unpack kernel
    uint32_t *holes = nullptr, *fill_src = nullptr, *fill_dst = nullptr;   // device, holes_cap each
    uint8_t *hole_flag = nullptr;    // device, one byte per particle slot
    uint64_t holes_cap = 0, flag_cap = 0;
    uint32_t begin_epoch = 0;        // epoch value at the last gpe_shard_begin
    uint64_t steps = 0;
    hipEvent_t fence[2] = {nullptr, nullptr};
    bool armed[2] = {false, false};
    // Exchange beside the step (round 4): the segments move on a stream of their own as soon as the tiles along the rank's
    // border have packed (ev_packed, recorded on the context's stream), while the interior tiles are still being resolved;
    // the next unpack waits for ev_exchanged.  overlap == false: everything on the context's stream, in order (rounds 1-3).
    bool overlap = false;
    hipStream_t xstream = nullptr;
    hipEvent_t ev_packed = nullptr, ev_exchanged = nullptr;
    bool packed_recorded = false, exchanged_recorded = false;
    hipStream_t exchange_stream(hipStream_t main) const { return overlap && xstream ? xstream : main; }
    // moving the segments between ranks (gpe_comm.hip): an RCCL communicator or a caller-supplied transport
    void *comm = nullptr;            // ncclComm_t
    bool comm_owned = false;
    gpe_shard_transport_fn transport = nullptr;
    void *transport_user = nullptr;
};

// The sharded run's control plane (gpe_shard_ctl.hip): the decomposition this context is a rank of, the tables and
// segment buffers it owns, and who carries its collectives.
struct ShardCtl {
    bool ready = false;              // gpe_shard_setup has run
    bool home = false;               // every particle sits on its owner, counts on the host, ghosts dropped
    gpe_shard_layout layout;
    uint32_t rank = 0;
    float scale = 1.0f;              // capacity_scale of the neighbour segments
    double planned_per_block = 0.0;  // block population the segments were sized for
    uint8_t *d_owner = nullptr;      // library-owned device tables (gpe_shard_plan)
    uint32_t *d_mask = nullptr;
    uint64_t tables_cap = 0;         // blocks
    uint32_t *d_send = nullptr, *d_recv = nullptr;
    uint64_t send_cap = 0, recv_cap = 0;          // words allocated
    // the neighbour exchange spelled as an all-to-all (word offsets / counts per rank; zero for non-neighbours)
    uint64_t x_send_off[GPE_SHARD_MAX_RANKS] = {}, x_send_cnt[GPE_SHARD_MAX_RANKS] = {};
    uint64_t x_recv_off[GPE_SHARD_MAX_RANKS] = {}, x_recv_cnt[GPE_SHARD_MAX_RANKS] = {};
    uint32_t n_neighbours = 0;
    gpe_shard_collectives coll;      // caller-supplied collectives
    bool coll_set = false;
    gpe_local_group *group = nullptr;   // local group this context is a rank of
    uint32_t group_rank = 0;
    uint32_t *d_small = nullptr;     // device scratch for the small host-side collectives (kCtlSmallWords)
    uint32_t *d_hist = nullptr, *d_first = nullptr;   // re-sort: Morton-block histogram / first local index
    uint64_t hist_cap = 0;
    uint32_t *d_rows_send = nullptr, *d_rows_recv = nullptr;   // re-cut: particle rows on the move
    uint64_t rows_send_cap = 0, rows_recv_cap = 0;
    uint64_t resorts = 0, steps = 0, n_ghost = 0;
    uint32_t recuts = 0;
};
constexpr uint64_t kCtlSmallWords = 4096;

}  // namespace gpe

struct gpe_ctx {
    gpe_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;        // the stream every kernel, copy and RCCL call of this context goes to
    hipStream_t own_stream = nullptr;    // created by gpe_create; `stream` is either this or one lent by gpe_set_stream
    std::string last_error;

    uint64_t n = 0;          // particles taking part in collisions (owned + ghosts in a sharded run)
    uint64_t n_owned = 0;    // the first n_owned are integrated by K12 (== n unless sharded)
    uint64_t cap = 0;        // allocated particle capacity
    float max_radius = 0.f;  // ParticleSystem::max_radius
    float grid_max_radius = 0.f;
    float cell_size = 0.f;
    int32_t mouse_pressed = 0;
    float mouse_x = 0.f, mouse_y = 0.f;

    // particle_buffers.rs:4-10, two sets (particle_system.rs:17-18); colours are render-only
    float2 *pos = nullptr, *prev = nullptr;
    float *radius = nullptr;
    float2 *pos_copy = nullptr, *prev_copy = nullptr;
    float *radius_copy = nullptr;
    uint32_t *home_cell_ids = nullptr, *particle_ids = nullptr;   // particle_sort.rs:29-33
    // grid.rs:36-40 GridBuffers (4 slots per particle)
    uint32_t *cell_ids = nullptr, *object_ids = nullptr;
    // collision_cell_buffers.rs:6-10
    uint32_t *chunk_obj_count = nullptr;   // ceil(4n/4) = n entries
    uint32_t *collision_cells = nullptr;   // 4n entries
    uint32_t *indirect_args = nullptr;     // 3 entries (+1: K)
    // sharded runs: global object index of every local particle (in-cell order), active cell box
    uint32_t *order_keys = nullptr;
    bool use_order_keys = false;
    bool has_active_box = false;
    int32_t active_box[4] = {0, 0, 0, 0};   // cx0, cy0, cx1, cy1 (inclusive) holding this rank's particles

    gpe::SortWorkspace sort_ws;
    gpe::ScanWorkspace scan_ws;
    gpe::OnesweepWorkspace os_ws;
    gpe::NativeState native;
    gpe::ShardState shard;
    gpe::ShardCtl ctl;
    bool use_onesweep = true;        // GPE_SORT=safe selects the reduce-then-scan sort

    // profiling
    bool profiling = false;          // scopes record events now
    uint32_t profile_every = 0;      // 0 off, 1 every call, k > 1: every k-th step (gpe_set_profiling)
    uint64_t profile_step = 0;
    std::vector<gpe::ScopeStat> stats;
    std::vector<gpe::PendingEvent> pending;
    std::vector<hipEvent_t> event_pool;
    hipEvent_t trace_origin = nullptr;            // recorded when profiling is switched on / timings are reset
    std::vector<gpe::TraceEvent> trace;           // the last kTraceCap resolved scopes, oldest first
};

namespace gpe {

// error plumbing -----------------------------------------------------------------------------
gpe_status fail(gpe_ctx *ctx, gpe_status code, const std::string &msg);
#define GPE_HIP(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            (void)hipGetLastError(); /* reported here: not left for the next launch check */    \
            return gpe::fail((ctx), _e == hipErrorOutOfMemory ? GPE_ERR_OOM : GPE_ERR_HIP,      \
                             std::string(#expr) + ": " + hipGetErrorName(_e) + " (" +           \
                                 hipGetErrorString(_e) + ")");                                  \
        }                                                                                       \
    } while (0)
#define GPE_TRY(expr)                                                                           \
    do {                                                                                        \
        gpe_status _s = (expr);                                                                 \
        if (_s != GPE_OK) return _s;                                                            \
    } while (0)

// profiling scope: a hipEvent pair on ctx->stream when ctx->profiling, else nothing.
class Scope {
   public:
    Scope(gpe_ctx *ctx, const char *name);
    ~Scope();

   private:
    gpe_ctx *ctx_;
    int stat_ = -1;
    hipEvent_t start_ = nullptr;
};

// kernel launchers (one translation unit each) --------------------------------------------------
// particles
gpe_status launch_home_cell_ids(gpe_ctx *c, const float2 *pos, uint64_t n, float cell_size,
                                uint32_t *home, uint32_t *ids);
gpe_status launch_rearrange(gpe_ctx *c, const float2 *pos, const float2 *prev, const float *radius,
                            const uint32_t *ids, uint64_t n, float2 *pos_out, float2 *prev_out,
                            float *radius_out);
gpe_status launch_verlet(gpe_ctx *c, float2 *pos, float2 *prev, const float *radius, uint64_t n,
                         float dt);
// grid
gpe_status launch_build_cell_ids(gpe_ctx *c, const float2 *pos, const float *radius, uint64_t n,
                                 float cell_size, uint32_t *cell_ids, uint32_t *object_ids);
// sort / scan
gpe_status sort_reserve(gpe_ctx *c, uint64_t n);
gpe_status sort_pairs(gpe_ctx *c, uint32_t *keys, uint32_t *vals, uint64_t n);
gpe_status sort_histogram(gpe_ctx *c, const uint32_t *keys, uint64_t n, uint32_t shift, uint32_t *hist256);
gpe_status sort_scatter_pass(gpe_ctx *c, const uint32_t *ka, const uint32_t *va, uint32_t *kb,
                             uint32_t *vb, uint64_t n, uint32_t shift);
gpe_status scan_reserve(gpe_ctx *c, uint64_t n);
gpe_status inclusive_scan(gpe_ctx *c, uint32_t *data, uint64_t n);
void sort_release(gpe_ctx *c);
void scan_release(gpe_ctx *c);
// onesweep (k_onesweep.hip)
constexpr int kHistCopies = 8;   // digit-histogram copies the hash kernel's workgroups flush into (same-line atomics serialise)
VerletParams verlet_params(const gpe_ctx *c, float dt);
gpe_status onesweep_reserve(gpe_ctx *c, uint64_t n);
void onesweep_release(gpe_ctx *c);
gpe_status onesweep_zero_hist(gpe_ctx *c);
// A sort that the device may skip (the native step, k_native.hip): every pass returns at once when *need == 0.
// The first pass copies the keys in input order to key_copy and resets the block table (table_pairs uint4 entries),
// the last one sets *fresh and counts the sort.  Passed by value to the pass kernel.
struct OnesweepGate {
    const uint32_t *need = nullptr;
    uint32_t *key_copy = nullptr;           // first pass: key_copy[i] = (key % key_blocks_x) | (key / key_blocks_x) << 16
    uint32_t key_blocks_x = 1;             //   (the block's coordinates in the block box: what the hash's drift test needs)
    uint64_t key_div_magic = 0;            //   ceil(2^40 / key_blocks_x): key / key_blocks_x = key * magic >> 40, exact for key * blocks_x < 2^40
    uint4 *table_reset = nullptr;
    uint64_t table_pairs = 0;
    uint32_t *fresh = nullptr;
    uint32_t *sorts = nullptr;
    uint32_t *sorts_seen = nullptr;        // last pass: copy of the new *sorts (a word nobody updates atomically while tiles read it)
    int ticket_base = 0;                   // tile tickets at ctl[ticket_base + pass]
    const uint32_t *count_now = nullptr;   // first pass: *sorted_count = *count_now (NULL: n) -- the particles the grouping covers
    uint32_t *sorted_count = nullptr;
};
gpe_status onesweep_sort(gpe_ctx *c, uint32_t *keys, uint32_t *vals, uint32_t *keys_b, uint32_t *vals_b,
                         uint64_t n, int passes, bool hist_ready, bool iota_vals, uint32_t **out_keys,
                         uint32_t **out_vals, bool bases_ready = false, uint2 *table = nullptr,
                         uint32_t table_entries = 0,    // table: the last pass also fills the native block table
                         const uint32_t *hist_src = nullptr,   // bases_ready: histogram copies the passes scan themselves
                         const OnesweepGate *gate = nullptr);
// native pipeline (k_native.hip)
gpe_status native_configure(gpe_ctx *c);
bool native_should_run(gpe_ctx *c);
gpe_status launch_shard_classify(gpe_ctx *c, const uint8_t *owner_of_block, const uint32_t *dest_mask_of_block,
                                 int32_t blocks_x, int32_t blocks_y, uint32_t my_rank, uint32_t *out_index,
                                 uint32_t *out_info, uint32_t *out_count, uint64_t out_capacity);
void native_release(gpe_ctx *c);
void shard_release(gpe_ctx *c);
void comm_release(gpe_ctx *c);
void ctl_release(gpe_ctx *c);
void shard_pack_args(gpe_ctx *c, PackArgs *P);
// collectives of a sharded run, carried by (in this order) the caller's callbacks, the local group, the RCCL communicator
gpe_status coll_all_reduce_u32(gpe_ctx *c, uint32_t *d_buf, uint64_t count, uint32_t op);
gpe_status coll_all_to_all_u32(gpe_ctx *c, const uint32_t *d_send, const uint64_t *send_off, const uint64_t *send_cnt,
                               uint32_t *d_recv, const uint64_t *recv_off, const uint64_t *recv_cnt,
                               hipStream_t on = nullptr);   // (on: the stream handed to the caller's callback; default the context's)
// the same over the RCCL communicator (gpe_comm.hip) and over a local group (gpe_group.hip)
gpe_status rccl_all_reduce_u32(gpe_ctx *c, uint32_t *d_buf, uint64_t count, uint32_t op);
gpe_status rccl_all_to_all_u32(gpe_ctx *c, const uint32_t *d_send, const uint64_t *send_off, const uint64_t *send_cnt,
                               uint32_t *d_recv, const uint64_t *recv_off, const uint64_t *recv_cnt);
gpe_status group_all_reduce_u32(gpe_ctx *c, uint32_t *d_buf, uint64_t count, uint32_t op);
gpe_status group_all_to_all_u32(gpe_ctx *c, const uint32_t *d_send, const uint64_t *send_off, const uint64_t *send_cnt,
                                uint32_t *d_recv, const uint64_t *recv_off, const uint64_t *recv_cnt);
gpe_status group_exchange_segments(gpe_ctx *c, hipStream_t xs);   // the per-step neighbour exchange, event-ordered (no stream sync)
void group_leave(gpe_ctx *c);
gpe_status resort_for_shard(gpe_ctx *c);          // ParticleSort::sort on the owned particles (gpe_api.hip do_resort)
gpe_status shard_ensure_flag_capacity(gpe_ctx *c);
std::string shard_error_text(uint32_t flags);
gpe_status step_for_shard(gpe_ctx *c, float dt);           // one ordinary step (gpe_api.hip do_step)
gpe_status grow_for_shard(gpe_ctx *c, uint64_t capacity);  // reallocate the particle buffers, keeping the first c->n
gpe_status reconfigure_native(gpe_ctx *c);
// verlet != nullptr: K12 is applied to the first n_owned particles as they are written back (pos_out = integrated
// position, prev = resolved position) -- the separate integration launch is then skipped
gpe_status native_collide(gpe_ctx *c, const float2 *pos_in, float2 *pos_out, const VerletParams *verlet = nullptr);
// collision cells + solver
gpe_status launch_count_chunks(gpe_ctx *c, const uint32_t *cell_ids, uint64_t total, uint32_t *chunk_counts);
gpe_status launch_build_collision_cells(gpe_ctx *c, const uint32_t *cell_ids, uint64_t total,
                                        const uint32_t *scanned, uint64_t num_chunks,
                                        uint32_t *collision_cells, uint32_t *indirect_args);
gpe_status launch_solve_color(gpe_ctx *c, const uint32_t *collision_cells, const uint32_t *scanned,
                              uint64_t num_chunks, const uint32_t *cell_ids, const uint32_t *object_ids,
                              uint64_t total, float2 *pos, const float *radius, float stiffness,
                              uint32_t color);

}  // namespace gpe
