// k_native.hip -- the MI355X-native collision pipeline (GPE_MODE_NATIVE).
//
// The reference sorts 4N (cell, object) pairs every step and resolves collisions cell by cell through
// global memory in four colour launches (~520 B/particle/step).  Here:
//
//   hash     key[i] = row-major index of particle i's 8x8-cell block, code[i] = its cell inside the block + the
//            overlap mask of its 8 neighbour cells; the four radix histograms and digit bases (fused); resets
//            the block table and the per-step control words
//   sort     onesweep over N (key, id) pairs (k_onesweep.hip); its last pass also fills the block table:
//            (start, end) of every block in the sorted order
//   collide  one workgroup per 32x32-cell tile: look the blocks of tile +- 8 cells up, keep the particles
//            whose home cell lies within 5 cells of the tile in LDS (positions, radii, ids gathered through
//            the sorted ids), rebuild the reference's per-cell member lists (home + phantom cells, members
//            in ascending object index), run ALL FOUR colour passes there, write back the tile's own
//            particles with the Verlet integration (K12) applied.
//
// Exactness: cell membership is frozen from the step-start positions
// (grid.wgsl:39-97), pairs of a cell run sequentially in ascending object index on live positions
// (collision_solver.wgsl:66-118), colours run 1..4.  A colour-k cell is exact when every cell that
// shares a particle with it was exact in colours < k; with the tile's own particles needing cells
// within +-1, the cells needed at colour k lie within +-(5-k) of the tile and their members' home
// cells within +-5 (the cell window; the block lookup reaches +-8).  Halo cells are recomputed redundantly by the neighbouring tiles (same
// inputs, same operation order => same bits), so no inter-tile communication and no global colour
// barrier is needed.  Positions are double-buffered (read pos_in, write pos_out) because neighbours
// read a tile's step-start positions while it writes its results.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>

#include "gpe_internal.h"

#ifndef GPE_NAT_THREADS
#define GPE_NAT_THREADS 512
#endif
#ifndef GPE_P5_PRIO
#define GPE_P5_PRIO 0
#endif

namespace gpe {

constexpr int kNatThreads = GPE_NAT_THREADS;
constexpr int kNatWaves = kNatThreads / 64;

// exclusive scan of one value per thread over the whole workgroup (kNatWaves waves)
__device__ __forceinline__ uint32_t nat_block_exclusive_scan(uint32_t v, uint32_t *s_w, uint32_t *total)
{
    const int lane = lane_id();
    const int w = (int)(threadIdx.x >> 6);
    const uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    uint32_t base = 0, all = 0;
#pragma unroll
    for (int i = 0; i < kNatWaves; ++i) {
        const uint32_t x = s_w[i];
        if (i < w) base += x;
        all += x;
    }
    if (total) *total = all;
    // no barrier behind the reads of s_w: its next writer (this function again) sits behind other barriers
    return base + inc - v;
}
constexpr int kHalo = 8;                       // cells = one 8x8 block: the granule of the block table lookups
// Cells a tile really needs around itself (tiles start at even cell coordinates, T is even).  Colour - 1 =
// (x & 1) + 2 (y & 1) (collision_solver.wgsl:55-58), a particle spans at most 2 x 2 cells, and a cell only depends on
// cells of EARLIER colours that share a member with it -- its horizontal, vertical or diagonal neighbours, by parity:
//   colour 4 (odd, odd)   <- colour 3 left / right, colour 2 below / above, colour 1 diagonal
//   colour 3 (even, odd)  <- colour 2 diagonal, colour 1 below / above
//   colour 2 (odd, even)  <- colour 1 left / right
// The tile's own particles sit in cells within 1 of the tile [x0, x1] x [y0, y1] (x0, y0 even; x1, y1 odd), so the
// cells that must be exact are
//   colour 4: x in [x0-1, x1]   y in [y0-1, y1]        colour 3: x in [x0-2, x1+1] y in [y0-1, y1]
//   colour 2: x in [x0-3, x1+2] y in [y0-2, y1+1]      colour 1: x in [x0-4, x1+3] y in [y0-2, y1+1]
// and their members' home cells lie within [x0-5, x1+4] x [y0-3, y1+2]: the cell window.  (A square window of
// tile +- 5 holds 16 % more cells, and the square colour zones 12 % more cells to resolve.)
constexpr int kConeLeft = 4, kConeRight = 3, kConeDown = 2, kConeUp = 1;   // colour-1 cells beyond the tile's edges
constexpr uint32_t kErrOutOfBox = 1u;          // a particle outside the configured cell box
constexpr uint32_t kErrTileOverflow = 2u;      // an 8x8 tile region over LDS capacity (no result!)
constexpr uint32_t kErrBoundExceeded = 16u;    // sharded run: the device-side particle count passed the host's bound
// tile_ctl words (the first four are cleared every step, the error word is sticky)
constexpr int kCtlOverflow1 = 0;               // 32x32 tiles over capacity this step
constexpr int kCtlOverflow2 = 1;               // 32x16 halves of those the half-tile launch could not take either
constexpr int kCtlHalfTicket = 7;              // next work item of the half-tile launch
constexpr int kCtlWindowMax = 2;               // largest 24x24-cell window population seen this step
constexpr int kCtlArena = 3;                   // particles handed out of the global spill arena this step
constexpr int kCtlOverflowTicket = 4;          // next work item of the over-capacity launch
constexpr int kCtlSubTiles = 5;                // 16x16 quarters redone as four 8x8 tiles this step
constexpr int kCtlSpills = 6;                  // 8x8 tiles staged in the global spill arena this step
constexpr int kCtlPerStepWords = 8;            // words [0, 8) are cleared every step
// HINTS: a tile the direct-slot dense launch hands on is, as a rule, over capacity on the next step too (a clump lives for
// hundreds of steps).  It registers itself for the next collide launch -- hints[next parity][k], k from tile_ctl[kCtlHints
// + next parity], and that launch's number in the fourth word of its roster header -- and there the FIRST kHintMax * 2
// workgroups of the dense launch redo it as two 32x16 halves while the others resolve their tiles: nothing waits behind
// the dense launch for it (the half-tile launch there cost the 1 M step ~20 us from step ~1000 of the benchmark run on).
// The tile's own workgroup sees the number in the header it loads anyway and returns; the half workgroup registers the
// tile again, kHintAge launches long -- then the tile tries itself once (it may fit again) and, if it still runs over, is
// registered afresh by the launch that takes it off list 1.  Launches are
// numbered by native_collide itself (not by step: a host may collide twice on one grid); the list a launch has used is
// cleared behind it by its over-capacity launch.  Exact whatever the lists hold: a tile is skipped by its own workgroup
// iff its header carries this launch's number or the next one's (registered again already, by a front workgroup -- or by
// itself, which is past the test), and a front workgroup takes a listed tile iff the header carries one of the two.
constexpr int kCtlHints = 11;                  // [launch parity] tiles registered for the launch of that parity
constexpr int kCtlHintsSeen = 13;              // hinted tiles of the last launch (statistics)
constexpr uint32_t kHintMax = 64;
constexpr uint32_t kHintAge = 240;             // a hint entry: age << 22 | ty << 11 | tx (at most 2048 tiles per axis: 16-bit cells)
constexpr int kCtlError = 8;                   // sticky
// Two words each, indexed by the parity of the step (native_prepare_step counts them): a step's hash kernel clears
// the NEXT step's word while its own is being set, so no workgroup of a launch races with another's reset.
// (The words the tiles only READ -- fresh, sorted count -- live in a 128-byte line of their own, the L2's granule: the
// first line holds the words every tile and every work item of the over-capacity launch hammers with atomics (overflow
// count, work ticket, window maximum, arena), and a load from a line under atomic fire queues behind them: with
// `fresh` next to the ticket P0 of a tile took 11.5 k instead of 5.9 k cycles and the over-capacity launch 6.3 instead
// of 4.1 ms at step 1000 of the 100 M soak.)
constexpr int kCtlNeedSort = 34;               // [parity] the hash found a particle outside the drift its code can express
constexpr int kCtlFresh = 36;                  // [parity] the radix passes ran: the block table describes THIS step's positions
constexpr int kCtlSorts = kNativeCtlSorts;     // steps whose radix passes ran (running count, gpe_get_pipeline_info)
constexpr int kCtlStragglers0 = 9, kCtlStragglers1 = 15;   // [parity] stragglers found by the step's hash so far
constexpr int kCtlSortedCount = 32;            // particles the kept grouping covers (written by the first radix pass)
constexpr int kCtlSortsSeen = kNativeCtlSortsSeen;   // copy of kCtlSorts in the line the tiles only read (written by the last radix pass)
constexpr int kCtlGhostSort = 40;              // [parity] sharded: a tile's ghost list ran over -- the ghosts' radix passes run and the tiles look the ghosts up in their block table
constexpr int kCtlWords = 64;                  // tile_ctl is this long (two 128-byte lines)
// How far a particle may have left the 8x8-cell block it was sorted into (cells beyond the block's extent, per
// direction) and still be found by every tile that needs it.  A tile looks up the blocks of tile +- 8 cells but keeps
// only the window [x0-5, x1+4] x [y0-3, y1+2] (kCone* + 1): a kept particle that moved right by dr cells comes from a
// block that starts at >= (x0-5) - 7 - dr, inside the lookup (>= x0-15, block-aligned: x0-8) while dr <= 3; moved left:
// its block starts at <= x1+4+dl <= x1+8 while dl <= 4; up: >= (y0-3) - 7 - du >= y0-15 while du <= 5; down: <= y1+2+dd
// <= y1+8 while dd <= 6.
constexpr int kDriftLeft = 4, kDriftRight = 3, kDriftDown = 6, kDriftUp = 5;
// The code word of a particle (k_native_hash -> tiles): its cell MOD 128 per axis (a tile's lookup region plus the drift
// spans fewer than 128 cells, so the value names one cell of it) | the overlap mask of the 8 neighbour cells | straggler.
constexpr uint32_t kCodeCellMask = 127u;
constexpr int kCodeYShift = 7, kCodeOverlapShift = 14;
// A particle beyond that reach (a straggler: in a cloud without damping a few particles are always fast) does not
// force a sort by itself: the hash kernel hands it, with its cell, to every 32x32 tile whose cell window holds it
// (at most four), kExcSlots per tile, and marks its code so that the tiles skip it in the old block's list.  Only a
// tile's list running over raises kCtlNeedSort.  Two sets of lists, by step parity (reset like the control words).
constexpr uint32_t kExcSlots = 16;
constexpr uint32_t kCodeStraggler = 1u << 22;
// Sharded runs: the ghosts (copies of the neighbours' particles, new every step) reach the tiles the same way -- the hash
// kernel lists every ghost for the 32x32 tiles whose window holds its cell, kGhostSlots per tile (a tile on the rank's
// border sees ~60-150 at the benchmark density).  Only when a list runs over do the ghosts get sorted into a block
// table of their own (rounds 1-3 did that every step: two radix launches).
constexpr uint32_t kGhostSlots = 256;
static_assert(kDriftRight <= kHalo - (kConeLeft + 1) && kDriftLeft <= kHalo - (kConeRight + 1), "x drift inside the lookup slack");
static_assert(kDriftUp <= kHalo - (kConeDown + 1) && kDriftDown <= kHalo - (kConeUp + 1), "y drift inside the lookup slack");
static_assert(64 + 2 * kHalo + kDriftLeft + kDriftRight < 128 && 32 + 2 * kHalo + kDriftDown + kDriftUp < 128, "a lookup region + drift names every cell mod 128 once");
// window coordinate of the cell a code names, for a window whose first cell is o (any sign): in [0, 128)
__device__ __forceinline__ int code_window_x(uint32_t code, int o) { return (int)((code - (uint32_t)o) & kCodeCellMask); }
__device__ __forceinline__ int code_window_y(uint32_t code, int o) { return (int)(((code >> kCodeYShift) - (uint32_t)o) & kCodeCellMask); }
// (the pinned host words the kernels report to -- kStat* -- are declared in gpe_internal.h: gpe_get_pipeline_info reads them too)
constexpr uint64_t kArenaBytesPerSlot = 37;     // px, py, rad, id, hm (4 B each), 4 member entries (16 B), block (1 B)
constexpr uint64_t kArenaMaxSlots = 1ull << 30; // the arena's slot numbers are 32 bit; 40 GB of the 288 GB
// tile sizes (cells) and LDS capacities (particles staged per region)
#ifndef GPE_CAP_MAIN
#define GPE_CAP_MAIN 1192
#endif
constexpr int kTileMain = 32, kCapMain = GPE_CAP_MAIN;
// The sub-tile windows take the same LDS as the main one, so the launch for over-capacity tiles also runs four
// workgroups per CU (with 1920 / 2048-particle windows it ran two: half the waves to hide latency with).
// (Round 4: 1600-particle sub-tile windows, three workgroups per CU at 80 VGPRs, looked-up capacity 8 x 512 -- against
// 1200 / four per CU / 64 VGPRs / 6 x 512: -1.4 % at step 1250 and -7.5 % at step 2000 of the 100 M soak; 2000-particle
// windows at two per CU and 128 VGPRs, which spill nothing: +5 % / -7 %.  profiles/r04/soak_marks_overflow_windows.txt)
#ifndef GPE_CAP_MID
#define GPE_CAP_MID 1600
#endif
#ifndef GPE_CAP_SMALL
#define GPE_CAP_SMALL 1600
#endif
#ifndef GPE_QMAX_MID
#define GPE_QMAX_MID 8
#endif
#ifndef GPE_OVF_WAVES
#define GPE_OVF_WAVES 6                        // waves per SIMD the over-capacity launch is compiled for (512 VGPRs / this)
#endif
constexpr int kTileMid = 16, kCapMid = GPE_CAP_MID;
constexpr int kTileSmall = 8, kCapSmall = GPE_CAP_SMALL;
// sharded (order-key) instantiations: 4 more bytes per slot for the local index, so 1024 slots in the same LDS
#ifndef GPE_CAP_ORD
#define GPE_CAP_ORD 1024
#endif
constexpr int kCapOrd = GPE_CAP_ORD;
// population of a 3x3-block (24x24-cell) window = what an 8x8 sub-tile looks up (it keeps ~56 % of it):
constexpr uint32_t kWindowReport = 512;     // tiles report windows above this population
// Windows beyond the 8x8 sub-tile's 1536 slots take their arrays from the spill arena; with the cells of up to
// 64 members resolved by whole waves that is faster than the compat kernels for the piles gravity builds (4 M
// particles, windows up to ~2900, cells up to ~53 members: 2.2 ms/step against 3.75).  Far denser blobs (mouse
// attraction: thousands per cell) are one-lane O(n^2) work that the overlapping windows would repeat: those leave
// the native path.
#ifndef GPE_WINDOW_HANDOVER
#define GPE_WINDOW_HANDOVER 16384
#endif
constexpr uint32_t kWindowHandover = GPE_WINDOW_HANDOVER;              // above it the context leaves the native path
constexpr uint32_t kWindowEligible = GPE_WINDOW_HANDOVER * 3 / 2;      // a scene whose windows exceed this never enters it

// ---------------------------------------------------------------------------------------------------
// hash: R pos 8 B, W key 4 B per particle; fused 4-digit histogram for the onesweep passes.
// (home_cell_ids.wgsl:24-31 computes the Morton id of the home cell; the key kept here is the row-major index
// of its 8x8-cell block; the particle id is implicit in the first pass.)
// The workgroup that flushes its histogram last also turns the histograms into the digit bases of the four
// passes and resets the tile tickets (k_os_prepare's job): one launch less on the step path.
// ---------------------------------------------------------------------------------------------------
// Which of the 8 neighbour cells does the particle overlap?  Bit k = the k-th neighbour of the reference's scan
// (grid.wgsl:68-90: y outer, x inner, centre skipped).  is_obj_in_cell (grid.wgsl:117-129) per axis: the
// clamped offset of neighbour column i / row j does not depend on the other axis, so the 8 tests share 3 + 3
// squared offsets (same operations and order as dot(d, d) = d.x*d.x + d.y*d.y).
__device__ __forceinline__ uint32_t neighbour_overlap_mask(float2 p, float r, int32_t cx, int32_t cy, float cell_size)
{
    const float sq = r * r;
    float sx[3], sy[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float lo_x = (float)(cx + i - 1) * cell_size, lo_y = (float)(cy + i - 1) * cell_size;
        // clamp as one v_med3_f32: equal to clamp_f for every non-NaN p (a zero of either sign squares
        // to +0 below), and for a NaN p the difference is NaN whatever the clamp returns
        const float dx = p.x - __builtin_amdgcn_fmed3f(p.x, lo_x, lo_x + cell_size);
        const float dy = p.y - __builtin_amdgcn_fmed3f(p.y, lo_y, lo_y + cell_size);
        sx[i] = dx * dx;
        sy[i] = dy * dy;
    }
    uint32_t over = 0;
    int k = 0;
#pragma unroll
    for (int y = -1; y <= 1; ++y) {
#pragma unroll
        for (int x = -1; x <= 1; ++x) {
            if (x == 0 && y == 0) continue;
            over |= (sx[x + 1] + sy[y + 1] < sq) ? (1u << k) : 0u;
            ++k;
        }
    }
    return over;
}

// Every workgroup flushes up to 256 bins per digit with device-scope atomics (they resolve beyond the per-XCD
// L2s), so the workgroups are large (1024 lanes).  The kernel is bound by instruction issue and latency, not by
// HBM: two workgroups per CU (64 VGPRs: two positions per lane in flight instead of eight) and up to 2048 of them
// take it from 544 to 454 us at 100 M particles and from 98 to 84 us at 16 M, the extra flushes included
// (profiles/r02/ab_hash_occupancy.txt).
// Sharded runs with the counts on the device (k_shard.hip): the first *owned particles are the rank's own and take
// part in the kept grouping; the ghosts behind them change every step, so they are grouped by a small sort of their
// own every step (gkeys / gids, g_bound pairs, the ghost block table).  All NULL / 0 otherwise.
struct HashGhosts {
    const uint32_t *owned = nullptr;
    uint32_t *gkeys = nullptr, *gids = nullptr;
    uint64_t g_bound = 0;
    uint4 *gtable2 = nullptr;
    uint64_t gtable_pairs = 0;
    const uint32_t *sorted_count = nullptr;    // tile_ctl[kCtlSortedCount]
    uint32_t *ghist_now = nullptr, *ghist_next = nullptr;   // the ghost sort's digit histograms (kHistCopies copies, two sets)
    // ghost lists (kGhostSlots ids per tile of the tile box): this step's, and the next step's counts to reset
    uint32_t *gl_count = nullptr, *gl_entry = nullptr, *gl_count_next = nullptr;
    uint32_t *ghost_sort = nullptr, *ghost_sort_next = nullptr;   // tile_ctl[kCtlGhostSort + parity], ... of the next step
};
constexpr int kHashBlock = 1024;
constexpr int kHashBatch = 2;                  // positions loaded per lane before any of them is ranked
constexpr int kHashGridMax = 2048;
// GHOSTS: a sharded run with its counts on the device (HashGhosts); the ordinary instantiation carries none of that code.
template <bool GHOSTS>
__global__ __launch_bounds__(kHashBlock, 8) void k_native_hash(const float2 *__restrict__ pos,
                                                            const float *__restrict__ radius, uint64_t n,
                                                            const uint32_t *__restrict__ n_valid_ptr,
                                                            float cell_size, int32_t gx, int32_t gy,
                                                            int32_t bx0, int32_t by0, int32_t blocks_x,
                                                            int32_t blocks_y, uint32_t pad_key,
                                                            uint32_t *__restrict__ keys,
                                                            uint32_t *__restrict__ codes, int digits,
                                                            uint32_t *hist4, uint32_t *__restrict__ hist_next,
                                                            uint32_t *__restrict__ os_ctl, uint32_t *tile_ctl,
                                                            uint4 *__restrict__ table2, uint64_t table_pairs,
                                                            uint32_t *__restrict__ host_stat,
                                                            const uint32_t *__restrict__ sorted_key, uint32_t parity,
                                                            uint64_t div_magic, uint32_t *__restrict__ exc_count,
                                                            uint2 *__restrict__ exc_entry,
                                                            uint32_t *__restrict__ exc_count_next, TileBox tb,
                                                            uint32_t straggler_limit, uint32_t fuse_always, HashGhosts G)
{
    // sorted_key[i] = the block key particle i had when the radix passes last ran (they keep it up to date,
    // k_onesweep.hip): the sorted ids and the block table still describe THAT grouping.  As long as every particle
    // is within kDrift* cells of its old block, the tiles find it through the old table (they look up more than they
    // keep) and this step needs no sort: the code carries the particle's cell RELATIVE to the old block.  Particles
    // beyond that go to per-tile straggler lists (kExcSlots); a list running over raises tile_ctl[kCtlNeedSort +
    // parity]; the radix passes that follow look at the word and return at once when it is 0.
    // sorted_key == NULL: always sort (first step, sharded runs, one-pass sorts).
    (void)div_magic;                                                   // (the division moved to the radix pass that writes sorted_key)
    __shared__ uint32_t s_hist[4 * 256];
    __shared__ uint32_t s_ghist[GHOSTS ? 4 * 256 : 1];                 // (the ghosts' keys, sharded runs)
    __shared__ uint32_t s_sort_known;                                  // a wave of this workgroup has seen the straggler limit passed
    s_hist[threadIdx.x] = 0;
    if (GHOSTS) s_ghist[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_sort_known = 0u;
    // What a step accumulates into is reset here instead of by a launch of its own (a launch costs ~6 us, 5 % of
    // the step at 1 M particles): the block table (filled two kernels later), and -- workgroup 0 -- the per-step
    // control words, after handing the previous step's window statistic to the host (pinned memory), the radix
    // passes' tile tickets, and the digit histograms the NEXT step's hash adds into (two sets, alternating steps:
    // this step's set is read by the radix passes that follow, nobody touches the other one meanwhile).
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < table_pairs; i += (uint64_t)gridDim.x * blockDim.x)
        table2[i] = make_uint4(0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u);      // (first, one past last) = (max, 0): empty
    if (GHOSTS)
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < G.gtable_pairs; i += (uint64_t)gridDim.x * blockDim.x)
        G.gtable2[i] = make_uint4(0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u);   // the ghosts' block table: rebuilt every step
    if (exc_count_next) {                                              // the next step's straggler lists
        const uint64_t nt = (uint64_t)tb.nx * (uint64_t)tb.ny;
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nt; i += (uint64_t)gridDim.x * blockDim.x)
            exc_count_next[i] = 0;
    }
    if (GHOSTS && G.gl_count_next) {                                   // ... and ghost lists
        const uint64_t nt = (uint64_t)tb.nx * (uint64_t)tb.ny;
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nt; i += (uint64_t)gridDim.x * blockDim.x)
            G.gl_count_next[i] = 0;
    }
    if (blockIdx.x == 0 && threadIdx.x < kCtlPerStepWords) {
        if (host_stat) {                                               // last step's statistics (kStat*), lagged
            if (threadIdx.x == kCtlWindowMax) host_stat[kStatWindowMax] = tile_ctl[kCtlWindowMax];
            if (threadIdx.x == kCtlArena) host_stat[kStatArena] = tile_ctl[kCtlArena];
            if (threadIdx.x == kCtlOverflow1) {
                // (+ the hinted tiles of the last launch)
                const uint32_t fresh_over = tile_ctl[kCtlOverflow1];
                host_stat[kStatOverflow] = fresh_over + tile_ctl[kCtlHintsSeen];
                host_stat[kStatOverflowNew] = fresh_over;
            }
            if (threadIdx.x == kCtlOverflow2) host_stat[kStatHalvesOver] = tile_ctl[kCtlOverflow2];
            if (threadIdx.x == kCtlSubTiles) host_stat[kStatSubTiles] = tile_ctl[kCtlSubTiles];
            if (threadIdx.x == kCtlSpills) host_stat[kStatSpills] = tile_ctl[kCtlSpills];
            if (threadIdx.x == 7) host_stat[kStatSorts] = tile_ctl[kCtlSorts];
        }
        tile_ctl[threadIdx.x] = 0;
    }
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < kHistCopies * 4 * 256; i += kHashBlock) hist_next[i] = 0;
        if (GHOSTS && G.ghist_next)
            for (int i = threadIdx.x; i < kHistCopies * 4 * 256; i += kHashBlock) G.ghist_next[i] = 0;
        if (threadIdx.x < 16) os_ctl[threadIdx.x] = 0;                 // tile tickets (owned [0..3], ghosts [8..11]) + error word [4]
        if (threadIdx.x == 0) {                                        // the next step's words; this step's if it must sort
            tile_ctl[kCtlNeedSort + (parity ^ 1u)] = 0;
            tile_ctl[kCtlFresh + (parity ^ 1u)] = 0;
            tile_ctl[parity ? kCtlStragglers0 : kCtlStragglers1] = 0;
            if (GHOSTS && G.ghost_sort_next) *G.ghost_sort_next = 0u;
            if (GHOSTS && G.gkeys && !G.gl_count) atomicOr(G.ghost_sort, 1u);   // (no lists: the ghosts are always sorted)
            if (!sorted_key) atomicOr(&tile_ctl[kCtlNeedSort + parity], 1u);
        }
    }
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t rounds = (n + stride - 1) / stride;
    // Sharded runs keep the particle count on the device (the exchange changes it every step without a host
    // round trip): the host passes an upper bound n, slots [nv, n) are padding that sorts behind every block.
    uint64_t nv = n;
    if (n_valid_ptr) {
        nv = *n_valid_ptr;
        if (nv > n) { nv = n; if (threadIdx.x == 0) atomicOr(&tile_ctl[kCtlError], kErrBoundExceeded); }
    }
    // owned particles [0, n_own), ghosts [n_own, nv); indices from sorted_cnt on are not in the kept grouping
    uint64_t n_own = nv;
    if (GHOSTS && G.owned) n_own = min((uint64_t)*G.owned, nv);
    const uint64_t sorted_cnt = (sorted_key && G.sorted_count) ? (uint64_t)*G.sorted_count : ~0ull;
    if (GHOSTS && G.gkeys) {
        // ghost slots behind the last one the particle loop below reaches: padding (sorts behind every block)
        const uint64_t first = n - n_own;
        for (uint64_t j = first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < G.g_bound; j += (uint64_t)gridDim.x * blockDim.x) {
            G.gkeys[j] = pad_key; G.gids[j] = 0u;
        }
        if (nv - n_own > G.g_bound && threadIdx.x == 0) atomicOr(&tile_ctl[kCtlError], kErrBoundExceeded);
    }
    bool oob = false, drifted = false;
    bool sort_known = sorted_key == nullptr;                           // (wave-uniform) the radix passes will run this step
    // The digit histograms of the owned particles' keys feed the radix passes -- which run on ~1 % of the steps of a run
    // that keeps its block table (7-12 % at 100 M in free fall).  Such a run leaves them to a gated launch of its own
    // behind this kernel (k_native_hist_gated: it returns at once unless need_sort was raised); only a step that sorts
    // anyway (no kept table: first step, sort_hold, GPE_FLAG_SORT_EVERY_STEP) counts them here, fused.
    const bool fuse_hist = sorted_key == nullptr || fuse_always != 0u;   // (fuse_always: GPE_FLAG_FUSED_HISTOGRAMS, rounds 1-3)
    for (uint64_t r0 = 0; r0 < rounds; r0 += kHashBatch) {
        float2 p[kHashBatch];
        float rad[kHashBatch];
        uint64_t idx[kHashBatch];
        uint32_t okey[kHashBatch];
#pragma unroll
        for (int u = 0; u < kHashBatch; ++u) {                        // the loads of a batch are in flight together
            idx[u] = (r0 + u) * stride + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
            const bool in = r0 + u < rounds && idx[u] < nv;
            p[u] = in ? pos[idx[u]] : make_float2(0.f, 0.f);
            rad[u] = in ? radius[idx[u]] : 0.f;
            okey[u] = (in && sorted_key) ? sorted_key[idx[u]] : 0u;
        }
#pragma unroll
        for (int u = 0; u < kHashBatch; ++u) {
            if (r0 + u >= rounds) break;                               // wave-uniform
            const bool valid = idx[u] < n;
            uint32_t key = pad_key;
            if (valid && idx[u] >= nv) {
                keys[idx[u]] = pad_key; codes[idx[u]] = 0u;
                if (GHOSTS && G.gkeys && idx[u] - n_own < G.g_bound) { G.gkeys[idx[u] - n_own] = pad_key; G.gids[idx[u] - n_own] = 0u; }
            }
            const bool ghost = GHOSTS && idx[u] >= n_own;
            uint32_t gkey = 0;
            bool gvalid = false;
            if (idx[u] < nv) {
                const int32_t cx = cell_coord(p[u].x, cell_size), cy = cell_coord(p[u].y, cell_size);
                // the particle's 8x8-cell block, row-major over the block box (0 for a particle outside it: flagged)
                const int32_t lbx = (cx >> 3) - bx0, lby = (cy >> 3) - by0;
                const bool out = (cx < 0) | (cx >= gx) | (cy < 0) | (cy >= gy) | (lbx < 0) | (lbx >= blocks_x) |
                                 (lby < 0) | (lby >= blocks_y);
                oob |= out;
                // (24-bit multiply-add: both factors are below 2^13 for an in-box cell; a full-rate instruction where
                // the 32-bit multiply takes four issue slots -- the kernel is bound by issue at 100 M, not by HBM)
                key = out ? 0u : __umul24((uint32_t)lby, (uint32_t)blocks_x) + (uint32_t)lbx;
                if (GHOSTS && ghost && G.gkeys) {
                    // a ghost: grouped by the ghosts' own sort; in the owned particles' sort it is padding
                    const uint64_t j = idx[u] - n_own;
                    if (j < G.g_bound) { G.gkeys[j] = key; G.gids[j] = (uint32_t)idx[u]; gkey = key; gvalid = true; }
                    key = pad_key;
                }
                keys[idx[u]] = key;
                // The particle's cell relative to the first cell of the block it was SORTED into: is it still within the
                // reach of the tiles that find it through that block?
                bool straggler = false;
                if (sorted_key && !(GHOSTS && ghost && G.gkeys)) {
                    // (sorted_key holds the block's coordinates in the block box, x | y << 16: the radix pass that
                    // copies the keys divides once per sort, k_onesweep.hip, instead of this kernel every step)
                    const uint32_t obx = okey[u] & 0xFFFFu, oby = okey[u] >> 16;
                    const int32_t relx = cx - (int32_t)((obx + (uint32_t)bx0) << 3);
                    const int32_t rely = cy - (int32_t)((oby + (uint32_t)by0) << 3);
                    // out of reach of its old block, or not in the kept grouping at all (an arrival of a sharded run,
                    // filed behind the particles the last sort covered): handed to the tiles directly
                    straggler = !out && ((relx < -kDriftLeft) | (relx > 7 + kDriftRight) | (rely < -kDriftDown) | (rely > 7 + kDriftUp) |
                                         (idx[u] >= sorted_cnt));
                }
                // what the tiles need to file the particle: its cell (mod 128) and its phantom cells.
                // Computed once here instead of by each of the ~2.25 tiles that stage the particle.
                codes[idx[u]] = ((uint32_t)cx & kCodeCellMask) | (((uint32_t)cy & kCodeCellMask) << kCodeYShift) |
                                (neighbour_overlap_mask(p[u], rad[u], cx, cy, cell_size) << kCodeOverlapShift) |
                                (straggler ? kCodeStraggler : 0u);
                // Stragglers are meant to be the few fast particles of a hot cloud.  When the whole cloud moves (free
                // fall) a large share of the particles runs out of reach within a step or two: handing millions of
                // them over, at up to four atomics each, costs more than the sort that makes them ordinary again.
                // Once the sort is known to run, handing stragglers over is wasted work (the tiles then ignore the lists):
                // a crushed scene makes a third of the particles stragglers, and their atomics -- one per wave on the
                // counter, up to four per particle on the lists -- took longer than the tiles (100 M soak, step 1000:
                // 22 ms/step with them, 16 without).  A wave learns it from the counter itself (its own add comes back
                // above the limit) or from its workgroup (an LDS word): no extra global round trip in calm scenes.
                const uint64_t ms = ballot64(straggler);
                bool route = false;
                if (ms != 0 && !sort_known) {                            // (wave-uniform)
                    sort_known = wave_lds_load(&s_sort_known) != 0u;     // (another wave of the workgroup has learnt it)
                    if (!sort_known) {
                        const int leader = (int)__builtin_ctzll(ms);
                        uint32_t before = 0;
                        if (lane_id() == leader)
                            before = atomicAdd(&tile_ctl[parity ? kCtlStragglers1 : kCtlStragglers0], (uint32_t)__popcll(ms));
                        before = (uint32_t)__builtin_amdgcn_readlane((int)before, leader);
                        if (before + (uint32_t)__popcll(ms) > straggler_limit) {
                            sort_known = true;
                            if (lane_id() == leader) { wave_lds_store(&s_sort_known, 1u); atomicOr(&tile_ctl[kCtlNeedSort + parity], 1u); }
                        } else route = true;
                    }
                }
                if (straggler && route) {
                    // every 32x32 tile whose window [32 tx - 5, 32 tx + 35] x [32 ty - 3, 32 ty + 33] holds the cell
                    // (sub-tiles of the over-capacity launch read their parent's list: their windows lie inside its)
                    const int tx0 = max(tb.x0, (cx - (kConeRight + 1)) >> 5), tx1 = min(tb.x0 + tb.nx - 1, (cx + kConeLeft + 1) >> 5);
                    const int ty0 = max(tb.y0, (cy - (kConeUp + 1)) >> 5), ty1 = min(tb.y0 + tb.ny - 1, (cy + kConeDown + 1) >> 5);
                    for (int ty = ty0; ty <= ty1; ++ty)
                        for (int tx = tx0; tx <= tx1; ++tx) {
                            const uint32_t t = tb.index(tx, ty);
                            const uint32_t slot = atomicAdd(&exc_count[t], 1u);
                            if (slot < kExcSlots) exc_entry[(uint64_t)t * kExcSlots + slot] = make_uint2((uint32_t)idx[u], (uint32_t)cx | ((uint32_t)cy << 16));
                            else { drifted = true; atomicOr(&tile_ctl[kCtlNeedSort + parity], 1u); }   // a list ran over: sort (and tell the others)
                        }
                }
            }
            if (fuse_hist) {                                           // (uniform)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (q < digits) hist_add(s_hist + q * 256, (key >> (8 * q)) & 255u, valid);
            }
            if (GHOSTS && G.gkeys) {                                   // (uniform)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (q < digits) hist_add(s_ghist + q * 256, (gkey >> (8 * q)) & 255u, gvalid);
            }
        }
    }
    if (GHOSTS && G.gl_count) {
        // The ghosts once more, for the tiles: every ghost is listed for each 32x32 tile whose window holds its cell (as
        // the stragglers above).  A loop of its own over the few thousand ghosts -- inside the particle loop the routing
        // pushed the kernel over its 64 registers.
        for (uint64_t i = n_own + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
            const float2 p = pos[i];
            const int32_t cx = cell_coord(p.x, cell_size), cy = cell_coord(p.y, cell_size);
            if ((cx < 0) | (cx >= gx) | (cy < 0) | (cy >= gy)) continue;
            const int tx0 = max(tb.x0, (cx - (kConeRight + 1)) >> 5), tx1 = min(tb.x0 + tb.nx - 1, (cx + kConeLeft + 1) >> 5);
            const int ty0 = max(tb.y0, (cy - (kConeUp + 1)) >> 5), ty1 = min(tb.y0 + tb.ny - 1, (cy + kConeDown + 1) >> 5);
            for (int ty = ty0; ty <= ty1; ++ty)
                for (int tx = tx0; tx <= tx1; ++tx) {
                    const uint32_t t = tb.index(tx, ty);
                    const uint32_t slot = atomicAdd(&G.gl_count[t], 1u);
                    if (slot < kGhostSlots) G.gl_entry[(uint64_t)t * kGhostSlots + slot] = (uint32_t)i;
                    else atomicOr(G.ghost_sort, 1u);                   // a list ran over: the ghosts' sort runs this step
                }
        }
    }
    if (GHOSTS && G.gkeys && blockIdx.x == 0 && threadIdx.x < (uint32_t)digits) {
        // the ghost sort covers g_bound slots: those behind the ghosts hold the padding key
        const uint64_t ng = min(nv - n_own, G.g_bound);
        atomicAdd(&s_ghist[threadIdx.x * 256 + ((pad_key >> (8 * threadIdx.x)) & 255u)], (uint32_t)(G.g_bound - ng));
    }
    if (oob) atomicOr(&tile_ctl[kCtlError], kErrOutOfBox);
    if (ballot64(drifted) != 0 && lane_id() == 0) atomicOr(&tile_ctl[kCtlNeedSort + parity], 1u);
    __syncthreads();
    // Flush: fire-and-forget device-scope atomics (the kernel boundary makes them visible); the radix passes turn the
    // histograms into digit bases themselves (k_os_pass, hist_src), so no workgroup waits for the others here.
    if ((threadIdx.x & 1u) == 0) {
        // two neighbouring bins per 64-bit atomic (no bin reaches 2^32, so nothing carries into the upper one)
        const uint32_t lo = s_hist[threadIdx.x], hi = s_hist[threadIdx.x + 1];   // index = digit * 256 + bin
        if (fuse_hist && (lo | hi))
            __hip_atomic_fetch_add(
                reinterpret_cast<unsigned long long *>(&hist4[(blockIdx.x % kHistCopies) * 1024 + threadIdx.x]),
                (unsigned long long)lo | ((unsigned long long)hi << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (GHOSTS && G.ghist_now) {
            const uint32_t glo = s_ghist[threadIdx.x], ghi = s_ghist[threadIdx.x + 1];
            if (glo | ghi)
                __hip_atomic_fetch_add(
                    reinterpret_cast<unsigned long long *>(&G.ghist_now[(blockIdx.x % kHistCopies) * 1024 + threadIdx.x]),
                    (unsigned long long)glo | ((unsigned long long)ghi << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// The digit histograms of the keys when the hash kernel has left them out (a run that keeps its block table): gated like
// the radix passes behind it -- every workgroup looks at need_sort first and returns at once when no sort is due.  Same
// layout as the hash's flush (kHistCopies copies of 4 x 256 bins, two bins per 64-bit atomic).
constexpr int kHistGatedBlock = 1024, kHistGatedGridMax = 2048;
__global__ __launch_bounds__(kHistGatedBlock) void k_native_hist_gated(const uint32_t *__restrict__ keys, uint64_t n, int digits,
                                                                       uint32_t *__restrict__ hist4,
                                                                       const uint32_t *__restrict__ need)
{
    if (*need == 0u) return;
    __shared__ uint32_t s_hist[4 * 256];
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t rounds = (n + stride - 1) / stride;
    for (uint64_t r = 0; r < rounds; ++r) {
        const uint64_t idx = r * stride + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        const bool valid = idx < n;
        const uint32_t key = valid ? keys[idx] : 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < digits) hist_add(s_hist + q * 256, (key >> (8 * q)) & 255u, valid);
    }
    __syncthreads();
    if ((threadIdx.x & 1u) == 0) {
        const uint32_t lo = s_hist[threadIdx.x], hi = s_hist[threadIdx.x + 1];
        if (lo | hi)
            __hip_atomic_fetch_add(
                reinterpret_cast<unsigned long long *>(&hist4[(blockIdx.x % kHistCopies) * 1024 + threadIdx.x]),
                (unsigned long long)lo | ((unsigned long long)hi << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Largest particle count of any 3x3-block (24x24-cell) window: what the smallest cell window must hold.
// Configuration-time check over the whole table (the step path gets the same number from the tiles).
__global__ __launch_bounds__(kStreamBlock) void k_native_window_max(const uint2 *__restrict__ table,
                                                                    uint32_t entries, int32_t blocks_x,
                                                                    int32_t blocks_y,
                                                                    uint32_t *__restrict__ out_max)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t best = 0;
    for (uint64_t mb = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; mb < entries; mb += stride) {
        const int bx = (int)(mb % (uint32_t)blocks_x), by = (int)(mb / (uint32_t)blocks_x);
        uint32_t sum = 0;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int x = bx + dx, y = by + dy;
                if (x < 0 || y < 0 || x >= blocks_x || y >= blocks_y) continue;
                const uint32_t m = (uint32_t)(y * blocks_x + x);
                if (m < entries) { const uint2 se = table[m]; sum += se.y > se.x ? se.y - se.x : 0u; }
            }
        best = max(best, sum);
    }
    for (int d = 32; d >= 1; d >>= 1) best = max(best, (uint32_t)__shfl_xor((int)best, d, 64));
    if (lane_id() == 0 && best) atomicMax(out_max, best);
}

// the asynchronous probe's last link: the measured maximum (+ 1, so that 0 means "no answer yet") goes to the host
__global__ void k_native_publish_probe(uint32_t *__restrict__ tile_ctl, uint32_t *__restrict__ host_stat)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        __hip_atomic_store(&host_stat[kStatProbe], tile_ctl[kCtlWindowMax] + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        tile_ctl[kCtlWindowMax] = 0;
    }
}

// every particle inside the cell box?  (configuration-time check, not on the step path)
__global__ __launch_bounds__(kStreamBlock) void k_native_check_box(const float2 *__restrict__ pos, uint64_t n,
                                                                   const uint32_t *__restrict__ n_valid_ptr,
                                                                   float cell_size, int32_t gx, int32_t gy,
                                                                   uint32_t *__restrict__ flag)
{
    if (n_valid_ptr && *n_valid_ptr < n) n = *n_valid_ptr;           // sharded run: the count lives on the device
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    bool oob = false;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float2 p = pos[i];
        const int32_t cx = cell_coord(p.x, cell_size), cy = cell_coord(p.y, cell_size);
        oob |= (cx < 0) | (cx >= gx) | (cy < 0) | (cy >= gy);
    }
    if (oob) atomicOr(flag, kErrOutOfBox);
}

// ---------------------------------------------------------------------------------------------------
// table: the sort key is the particle's 8x8-cell block (row-major index over the box).  table[b] = (first,
// one-past-last) position of the block's particles in the sorted order, (0xFFFFFFFF, 0) for an empty block.
// It is filled by the last radix pass (k_onesweep.hip, k_os_pass): no launch of its own.
// ---------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------
// collide
// ---------------------------------------------------------------------------------------------------
#ifndef GPE_QMAX_MAIN_VALUE
#define GPE_QMAX_MAIN_VALUE 3                  // looked-up particles per thread of a 32x32 tile's workgroup
#endif
struct CollideArgs {
    const float2 *pos_in;
    const float *radius;
    float2 *pos_out;
    const uint32_t *sorted_ids;
    const uint32_t *codes;       // per particle: cell mod 128 (7 + 7 bits) | neighbour overlap mask (8 bits) | kCodeStraggler
    const uint2 *gtable;         // sharded runs: (start, end) of every block among the GHOSTS, sorted every step (else NULL)
    const uint32_t *gsorted_ids; // ... and their particle indices in that order
    const uint32_t *exc_count;   // stragglers handed to each 32x32 tile this step (NULL: none, the run always sorts)
    const uint2 *exc_entry;      // kExcSlots x (particle, cell x | y << 16) per tile
    TileBox tb;                  // the tiles those lists, the ghost lists and the rosters are kept for
    const uint32_t *gho_count;   // sharded runs: ghosts listed for each tile this step (NULL: none / not sharded)
    const uint32_t *gho_entry;   // kGhostSlots x particle index per tile
    const uint32_t *ghost_sort;  // tile_ctl[kCtlGhostSort + parity]: != 0 when a ghost list ran over this step -- the
                                 // ghosts are then looked up in their block table (gtable) instead
    const uint32_t *fresh;       // tile_ctl[kCtlFresh + parity]: != 0 when the radix passes ran this step (the table is
                                 // of NOW: nobody is a straggler, the lists and the rosters are not used)
    const uint2 *table;
    uint32_t entries;
    int32_t blocks_x, blocks_y;  // table index of block (bx, by) = (by - by0) * blocks_x + (bx - bx0)
    int32_t bx0, by0;            // first block of the block box (0, 0 unless sharded)
    const uint32_t *counts;      // sharded runs: [0] = owned particles, kept on the device; else NULL
    float cell_size;
    float stiffness;
    int32_t gx, gy;              // cell box
    int32_t tiles_x, tiles_y;    // tile grid of the dense launch
    uint32_t band_tiles;         // ... dealt to the XCDs in bands of this many consecutive tiles (dense_launch_tile)
    // sharded runs that exchange beside the step: the FRAME of the tile grid (frame_l / _r columns, frame_b / _t rows: the
    // tiles whose particles can come to lie outside the pack's safe box) is resolved first, by k_collide_border
    int32_t frame_l, frame_r, frame_b, frame_t;
    int32_t tile_x0, tile_y0;    // its first tile (sharded runs cut the grid to the rank's active box)
    const uint32_t *order_keys;  // sharded runs: in-cell order by order_keys[local index]; else NULL
    uint32_t *tile_ctl;          // kCtl* words
    uint32_t *overflow1;         // packed (ty << 16 | tx) of over-capacity 32x32 tiles
    uint32_t overflow1_cap;
    // While tiles run over the direct-slot form (the host's lagged statistic) a launch between the dense and the
    // over-capacity one redoes each as two 32x16 HALVES in the same direct-slot form (k_collide_halves: half the cells,
    // so 1.6 x the particles per cell fit, at the dense launch's cost per particle); the halves it cannot take either
    // are listed in overflow2 (packed ty16 << 16 | tx32) and the over-capacity launch works through that list.
    uint32_t *overflow2;
    uint32_t quarters_of_halves; // list 1 was taken by the half-tile launch: the over-capacity launch only takes overflow2 (two quarters per half)
    // hints (kCtlHints): hints[parity * kHintMax + k] = ty << 16 | tx; front_wgs == 0: no hints this launch
    uint32_t *hints;
    uint32_t step_stamp, front_wgs, hint_parity;
    uint32_t hints_on;           // tiles that run over register themselves (front_wgs != 0: ... and this launch redoes the registered ones)
    // spill arena (global memory) for the particle arrays of such tiles
    float *arena_px, *arena_py, *arena_rad;
    uint32_t *arena_id, *arena_hm, *arena_mem;   // arena_mem holds 4 entries per particle
    uint8_t *arena_sblk;
    uint32_t arena_cap;
    // K12 fused into the write-back (particle_integration.wgsl:25-77) when fuse_verlet != 0
    float2 *prev;
    uint64_t n_owned;
    uint32_t fuse_verlet;
    VerletParams vp;
    // Tile rosters (direct-slot tiles of a run that keeps its block table): the particles a tile's lookup finds do not
    // change between two sorts, so the tile that looks them up right after a sort writes them down -- roster_ids[tile *
    // kRosterCap ..], roster_hdr[tile] = (count | 0xFFFFFFFF: more than the tile stages, stamp = sorts so far + 1, largest
    // 24x24-cell window population, -) -- and the steps until the next sort start from that list: one coalesced load
    // issued with the kernel's first instructions instead of table lookup -> scan -> slot map -> ids (three barriers and
    // a dependent global round trip).  A roster whose stamp is not the current one is ignored and rewritten.
    uint4 *roster_hdr;           // NULL: no rosters
    uint32_t *roster_ids;
    const uint32_t *sorts_seen;  // tile_ctl[kCtlSortsSeen]
    uint32_t roster_write;       // this run keeps its table: write rosters down
    unsigned long long *stamps;  // diagnostic builds only (-DGPE_TILE_STAMPS): cycles per phase, thread 0
    // sharded runs: the tiles along the rank's border pack their own particles for the neighbours as they write them
    // back (gpe_internal.h, pack_particle); pack.on == 0 otherwise
    PackArgs pack;
};
// The pack as the tiles do it: a particle whose new position lies inside the `safe` box (the rank's rectangle shrunk by
// one block and a cell on every side that has a neighbour) sits in a block this rank owns and no other rank borders --
// four compares say so, no table lookup; only the others go through pack_particle (a wave none of whose lanes holds one
// skips it: every tile away from the rank's border, whatever the speed of its particles).
__device__ __forceinline__ void pack_if_near_border(const PackArgs &P, const bool mine, const uint32_t id, const float2 o,
                                                    const float2 c, const float rad, const uint32_t key, const float cell_size)
{
    const bool near = mine && (o.x < P.safe_x0 || o.x >= P.safe_x1 || o.y < P.safe_y0 || o.y >= P.safe_y1);
    if (ballot64(near) == 0) return;
    if (P.on == 2u) {
        // an interior tile, resolved while the segments are already on their way: its particles lie more than a block
        // (+ a cell) inside the safe box at the start of the step, so only a particle that moved further than the
        // exchange's protocol allows (one block per step, k_shard.hip) gets here -- the run is in error, loudly
        if (near) atomicOr(P.err, kShardErrNoSlot);
        return;
    }
    pack_particle(P, near, id, o, c, rad, key, cell_size);
}
constexpr int kRosterCap = GPE_QMAX_MAIN_VALUE * 512;   // == TileDirect<32, .., 512>::RAWCAP

#ifdef GPE_TILE_CYCLES
// diagnostic builds only (scripts/tile_cycles.py): what every tile of the dense launch and every quarter of the
// over-capacity launch cost -- g_tile_cycles[tile of the tile box] = (cycles of the dense launch's workgroup, outcome:
// 0 done / 1 handed on, cycles of its quarters in the over-capacity launch, particles it looked up)
__device__ uint4 *g_tile_cycles;
__device__ uint32_t g_bail_reasons[8];         // why direct-slot tiles were handed on, counted since the last reset
#define GPE_BAIL(code) atomicAdd(&g_bail_reasons[code], threadIdx.x == 0 ? 1u : 0u)
#else
#define GPE_BAIL(code) ((void)0)
#endif
#ifdef GPE_DBG_RT
// diagnostic builds only (scripts/overflow_phases.py): parts of the counting-sort windows' colour passes switched off at
// run time (1 lane groups, 2 one-lane cells, 4 whole-wave cells, 8 everything behind P1) -- what a part costs in a scene
// that only exists after a thousand steps of the product's kernels.  Results are wrong while a bit is set.
__device__ uint32_t g_dbg_skip_rt;
#define GPE_RT_SKIP(bit) ((g_dbg_skip_rt & (bit)) != 0u)
#else
#define GPE_RT_SKIP(bit) false
#endif
#ifdef GPE_TILE_STAMPS
#define GPE_STAMP_BEGIN() long long _t_prev = clock64()
#define GPE_STAMP(i)                                                                  \
    do {                                                                              \
        if (A.stamps && threadIdx.x == 0 && (blockIdx.x & 127u) == 5u) {                  \
            const long long _t = clock64();                                           \
            atomicAdd(&A.stamps[i], (unsigned long long)(_t - _t_prev));              \
            atomicAdd(&A.stamps[16 + (i)], 1ull);                                      \
            _t_prev = _t;                                                             \
        }                                                                             \
    } while (0)
#else
#define GPE_STAMP_BEGIN() do {} while (0)
#define GPE_STAMP(i) do {} while (0)
#endif

// LID: a sharded run's windows also keep every particle's local index (4 bytes per slot): `id` then holds the order key
// the members are sorted by, and the write-back needs the local index again.
template <int T, int CAP, bool LID = false>
struct TileLds {
    static constexpr int TILE = T;
    static constexpr bool kGlobal = false;
    static constexpr bool kLid = LID;
    static constexpr int kSlots = CAP;                 // particles the window keeps
    uint32_t lid[LID ? CAP : 1];
    // The blocks of tile +- 8 cells are looked up (block granule), but only particles whose home cell lies in the
    // tile's dependency cone are kept -- 5 / 4 cells left / right of the tile, 3 / 2 below / above (see kConeLeft ...):
    // a third fewer particles in LDS at 32x32, and the window overflows that much later.
    static constexpr int HXL = kConeLeft + 1, HXR = kConeRight + 1, HYL = kConeDown + 1, HYR = kConeUp + 1;
    static constexpr int RWX = T + HXL + HXR, RWY = T + HYL + HYR;
    // The cell array carries a ring of one cell around the window: a kept particle's phantom cells then always have
    // a slot (home + a constant offset, no bounds tests).  Ring cells lie outside every colour's zone: never walked.
    static constexpr int PX = RWX + 2, PY = RWY + 2;
    static constexpr int NCELL = PX * PY;
    static constexpr int NB = (T + 2 * kHalo) / 8;
    static constexpr int NBLK = NB * NB;
    static constexpr int PER = (NCELL + kNatThreads - 1) / kNatThreads;   // cells per thread in the scan
    // Looked-up particles a window takes (P0 hands the tile on when its 3x3 / 4x4 / 6x6 blocks hold more, before any
    // gather): what the kept window holds at uniform density -- it keeps 66 % of what a 32x32 tile looks up, 51 % at
    // 16x16, 38 % at 8x8 -- plus a margin; a tile that passes this test and still keeps more than CAP is handed on
    // after the gather.  (Until the order-key windows kept their local indices this was bound to 1536 by an 11-bit
    // slot field: an 8x8 tile went to the spill arena at 590 kept particles.)  The 32x32 tile stays at 1536: in a
    // compressed scene its dense tiles are resolved faster as four quarters -- one crowded cell then holds up one
    // quarter's colour pass, not 512 threads' (100 M soak at step 1500: 22.9 ms with 1536, 22.2 with 1280, 24.3 /
    // 25.2 with 1700 / 1800 or 2048; profiles/r02/soak_1500_main_lookup_capacity.txt).
#ifndef GPE_QMAX_MAIN
#define GPE_QMAX_MAIN GPE_QMAX_MAIN_VALUE
#endif
    static constexpr int QMAX = T >= 32 ? GPE_QMAX_MAIN : (T >= 16 ? GPE_QMAX_MID : 8);
    static constexpr int RAWCAP = QMAX * kNatThreads;
    static_assert(NCELL < 2048, "hm packs home (11 bit) | overlap mask (8) | own (1)");
    static constexpr int QZ = (T + 8) * (T + 4) / 4;                      // cells of one colour inside its zone
    float px[CAP], py[CAP], rad[CAP];
    uint32_t id[CAP];
    uint32_t hm[CAP];          // bits 0-10 index of the home cell in the padded cell array; bits 11-18 overlap mask of
                               // the 8 neighbour cells (k_native_hash); bit 19 the home cell lies in the tile (the
                               // particle is the tile's own)
    // cell[lc + 1]: members of cell lc (P1) -> first slot of its list (P2) -> one past its last slot (P3);
    // cell[0] = 0, so from P3 on the list of cell lc is mem[cell[lc] .. cell[lc + 1]).  Values stay below
    // 4 * CAP < 65536: two cells share a word (LDS atomics are 32 bit, so cell_inc adds 1 or 1 << 16 --
    // no carry can leave the low half), which buys 160 more particles per window than 32-bit cells.
    static_assert(4 * CAP < 65536, "cell offsets are 16 bit");
    union {
        uint32_t cellw[(NCELL + 2) / 2];
        uint16_t cellh[NCELL + 2];
    };
    __device__ __forceinline__ void cell_clear(int tid)
    {
        for (int i = tid; i < (NCELL + 2) / 2; i += kNatThreads) cellw[i] = 0;
    }
    __device__ __forceinline__ uint32_t cell_get(int i) const { return cellh[i]; }
    __device__ __forceinline__ void cell_set(int i, uint32_t v) { cellh[i] = (uint16_t)v; }
    __device__ __forceinline__ uint32_t cell_inc(int i)                // returns the value before the add
    {
        const uint32_t sh = (uint32_t)(i & 1) * 16u;
        return (atomicAdd(&cellw[i >> 1], 1u << sh) >> sh) & 0xFFFFu;
    }
    union {
        uint16_t mem[4 * CAP]; // member lists (local particle slots)
        uint8_t sblk[RAWCAP];  // P0-P1 only: region block a looked-up slot came from
        static_assert(RAWCAP <= 8 * CAP, "sblk fits under mem");
    };
    uint16_t list[4 * QZ];     // active cells, one segment per colour
    // cells of 9..64 members (piles): resolved by a whole wave each; WC per colour, the rest falls back to one lane
    static constexpr int WC = T >= 32 ? 16 : 64;
    uint16_t wlist[4 * WC];
    // cells of 9..16 members of the sub-tile windows: by the sixteen lanes of a DPP row (resolve_row), four per wave
    static constexpr bool kRows = T < 32;
    uint16_t rlist[kRows ? 4 * WC : 2];
    uint32_t lcnt[20];         // per colour: [c] cells resolved by one lane, [4 + c] by a lane group, [8 + c] by a wave, [12 + c] by a
                               // row; [16 + c] the ticket the waves draw whole-wave cells and quadruples of rows with
    // looked-up blocks; an order-key (sharded) window looks every block up twice: among the owned particles (the kept
    // table) and among the ghosts (their own table, rebuilt every step): virtual blocks [NBLK, 2 NBLK)
    static constexpr int VBMAX = (kLid || kGlobal) ? 2 * NBLK : NBLK;
    uint32_t bstart[VBMAX];
    uint32_t bcnt[VBMAX];
    uint32_t boff[VBMAX + 1];
    uint32_t s_w[16];
    uint32_t misc[4];
};

// Same window, but the per-particle arrays live in a slice of the global spill arena: for 8x8 tiles whose
// 24x24-cell region holds more particles than any LDS window stages (thousands of particles piled into a few
// cells).  Slow and rare; it exists so that the native path is exact for every input, not only for sparse ones.
template <int T>
struct TileGlobal {
    static constexpr int TILE = T;
    static constexpr bool kGlobal = true;
    static constexpr bool kLid = false;
    static constexpr int kSlots = 1;                   // (no per-lane own arrays: the write-back loops over the slots)
    // keeps every looked-up particle: slot == looked-up slot
    static constexpr int HXL = kHalo, HXR = kHalo, HYL = kHalo, HYR = kHalo;
    static constexpr int RWX = T + 2 * kHalo, RWY = T + 2 * kHalo;
    static constexpr int PX = RWX + 2, PY = RWY + 2;               // padded as in TileLds
    static constexpr int NCELL = PX * PY;
    static constexpr int NB = RWX / 8;
    static constexpr int NBLK = NB * NB;
    static constexpr int PER = (NCELL + kNatThreads - 1) / kNatThreads;
    static constexpr int QMAX = 1;
    static constexpr int QZ = (T + 8) * (T + 4) / 4;
    float *px, *py, *rad;
    uint32_t *id, *hm, *mem;
    uint8_t *sblk;
    uint32_t cell[NCELL + 1];      // 32-bit cells: the member count is not bounded by an LDS capacity here
    __device__ __forceinline__ void cell_clear(int tid)
    {
        for (int i = tid; i <= NCELL; i += kNatThreads) cell[i] = 0;
    }
    __device__ __forceinline__ uint32_t cell_get(int i) const { return cell[i]; }
    __device__ __forceinline__ void cell_set(int i, uint32_t v) { cell[i] = v; }
    __device__ __forceinline__ uint32_t cell_inc(int i) { return atomicAdd(&cell[i], 1u); }
    uint16_t list[4 * QZ];
    static constexpr int WC = 64;
    uint16_t wlist[4 * WC];
    static constexpr bool kRows = true;
    uint16_t rlist[4 * WC];
    uint32_t lcnt[20];         // per colour: [c] cells resolved by one lane, [4 + c] by a lane group, [8 + c] by a wave, [12 + c] by a
                               // row; [16 + c] the ticket the waves draw whole-wave cells and quadruples of rows with
    // looked-up blocks; an order-key (sharded) window looks every block up twice: among the owned particles (the kept
    // table) and among the ghosts (their own table, rebuilt every step): virtual blocks [NBLK, 2 NBLK)
    static constexpr int VBMAX = (kLid || kGlobal) ? 2 * NBLK : NBLK;
    uint32_t bstart[VBMAX];
    uint32_t bcnt[VBMAX];
    uint32_t boff[VBMAX + 1];
    uint32_t s_w[16];
    uint32_t misc[4];
};

// Members of one cell into ascending object index (the order the reference's stable sort gives them).
template <class L>
__device__ __forceinline__ void sort_members(L &S, const uint32_t b, const uint32_t e)
{
    const uint32_t n = e - b;
    if (n == 2) {
        const uint32_t m0 = S.mem[b], m1 = S.mem[b + 1];
        if (S.id[m0] > S.id[m1]) { S.mem[b] = m1; S.mem[b + 1] = m0; }
        return;
    }
    if (n == 3) {
        uint32_t m0 = S.mem[b], m1 = S.mem[b + 1], m2 = S.mem[b + 2];
        uint32_t i0 = S.id[m0], i1 = S.id[m1], i2 = S.id[m2];
        bool ch = false;
        if (i0 > i1) { uint32_t t = m0; m0 = m1; m1 = t; uint32_t u = i0; i0 = i1; i1 = u; ch = true; }
        if (i1 > i2) { uint32_t t = m1; m1 = m2; m2 = t; uint32_t u = i1; i1 = i2; i2 = u; ch = true; }
        if (i0 > i1) { uint32_t t = m0; m0 = m1; m1 = t; ch = true; }
        if (ch) { S.mem[b] = m0; S.mem[b + 1] = m1; S.mem[b + 2] = m2; }
        return;
    }
    if (n <= 8) {
        // 4..8 members: rank count (ids are distinct) -- two LDS round trips whatever the order, where an
        // insertion sort walks a dependent LDS chain per element; the colour pass waits for its slowest cell
        constexpr int M = 8;
        uint32_t slot[M], id[M];
#pragma unroll
        for (int k = 0; k < M; ++k) slot[k] = ((uint32_t)k < n) ? (uint32_t)S.mem[b + k] : 0u;
#pragma unroll
        for (int k = 0; k < M; ++k) id[k] = ((uint32_t)k < n) ? S.id[slot[k]] : 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < M; ++k) {
            uint32_t rank = 0;
#pragma unroll
            for (int j = 0; j < M; ++j)
                if (j != k) rank += (id[j] < id[k]) ? 1u : 0u;
            if ((uint32_t)k < n && rank != (uint32_t)k) S.mem[b + rank] = slot[k];
        }
        return;
    }
    for (uint32_t i = b + 1; i < e; ++i) {                             // insertion sort
        const uint32_t x = S.mem[i];
        const uint32_t kx = S.id[x];
        uint32_t j = i;
        while (j > b && S.id[S.mem[j - 1]] > kx) { S.mem[j] = S.mem[j - 1]; --j; }
        S.mem[j] = x;
    }
}

// One pair of the reference's response (collision_solver.wgsl:91-111), shared by the one-lane-per-cell and the
// lane-group resolution.  One wave-uniform early-out (no lane of the wave can collide), then straight-line code
// whose result a lane keeps or not by select.  Bit-exact shortcuts:
//  * r1 == r2 (and 1/r finite, non-zero): inv1 == inv2 and inv1 + inv1 == 2 inv1 exactly, so both weights
//    (:107-108) are exactly 0.5 -- the three divisions are skipped, not approximated.
//  * q = vx*vx + vy*vy > 1.000001 rs^2 implies rs^2 <= distance^2 (distance = sqrt(q) correctly rounded, so
//    distance^2 >= q (1 - 2^-22)): no collision (:95); q < 9.9e-9 implies distance < 0.0001 (:95; 0.0001f squared
//    is 9.99999995e-9): no collision either.  Neither needs the square root.
//  * the correctly rounded square root and quotients without the steps hipcc's sequences spend on operands that
//    cannot occur here (below).
// Instruction budget (round 4; the tiles are bound by VALU issue at 100 M particles):
//  * x and y travel as one 64-bit register pair (f32x2): the subtraction, the squares, the two quotients' refinement
//    chains, the scalings and the final additions are the same operation on both components, and gfx950 issues
//    v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 -- IEEE binary32 per component, the same rounding as the scalar
//    forms -- in the slot of one scalar instruction.  Left to itself hipcc packed a quarter of them.
//  * the predicates are LANE MASKS (ballot64 / lanes_of, gpe_internal.h): every comparison is voted on its own and the masks are combined
//    by scalar ANDs.  A vote on `a && b` costs two VALU instructions (hipcc materialises the combined predicate as
//    0 / 1 and compares it again); the response votes three times per pair.
// `active`: the lanes that have a pair; `plain`: the lanes whose r1 is an ordinary number (1e-30 .. 1e30).  Returns the
// lanes that collided; (p1, p2) are updated in place for them.
#ifdef GPE_COUNT_PAIRS
// diagnostic builds only (scripts/soak_pairs.py): pairs the colour passes walk / resolve, over all tiles (the halo cells a
// tile recomputes for its neighbours included) -- what "ms per 10^9 pairs" in BASELINE.md is measured with
// (4096 counters each, by workgroup, 64 bytes apart: two counters for the whole device took 300 ms per step at 100 M; and
// only while g_pairs_on is set, so that a run reaches the step of interest at nearly the product's speed)
constexpr int kPairCounters = 4096;
__device__ unsigned long long g_pairs_walked[kPairCounters * 8], g_pairs_hit[kPairCounters * 8];
__device__ uint32_t g_pairs_on;
__device__ __forceinline__ void count_pairs(unsigned long long *ctr, const uint64_t m)
{
    if (m != 0 && g_pairs_on != 0u && lane_id() == (int)__builtin_ctzll(m))
        atomicAdd(&ctr[(blockIdx.x & (kPairCounters - 1)) * 8], (unsigned long long)__popcll(m));
}
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(const float v) { return (f32x2){v, v}; }
__device__ __forceinline__ f32x2 fma2(const f32x2 a, const f32x2 b, const f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 select2(const bool c, const f32x2 a, const f32x2 b) { return (f32x2){c ? a.x : b.x, c ? a.y : b.y}; }
__device__ __forceinline__ uint64_t plain_radius_lanes(const float r) { return ballot64(r >= 1e-30f) & ballot64(r <= 1e30f); }

// BOTH = false: only p1 is updated (the lane groups, where each lane of a pair computes its own half).
template <bool BOTH = true>
__device__ __forceinline__ uint64_t pair_response(const uint64_t active, f32x2 &p1, std::conditional_t<BOTH, f32x2 &, const f32x2 &> p2,
                                                  const float r1, const float r2, const uint64_t plain,
                                                  const float stiffness, const uint64_t counted = ~0ull)
{
    (void)counted;                                                    // (diagnostic builds: the lanes whose pair counts)
#ifdef GPE_COUNT_PAIRS
    count_pairs(g_pairs_walked, active & counted);
#endif
    const f32x2 v = p1 - p2;                                          // :91 (live positions, :86)
    const f32x2 vv = v * v;
    const float q = vv.x + vv.y;
    const float radius_sum = r1 + r2;                                 // :61
    const float rs2 = radius_sum * radius_sum;
    const uint64_t cand = active & ballot64(q <= rs2 * 1.000001f) & ballot64(q >= 9.9e-9f);
    if (cand == 0) return 0;                                          // wave-uniform
    // The correctly rounded square root and quotients WITHOUT the steps hipcc's sequences spend on operands that
    // cannot occur here: a candidate has q in [9.9e-9, 1.000001 rs^2] and the quotients' numerators are differences
    // of positions (0 or >= one ulp of a position), so nothing is denormal, zero-divided, infinite or NaN; the lanes
    // that are not candidates compute garbage that the selects below discard.
    //   sqrt: v_sqrt_f32 is within 1 ulp; try the neighbours with an exact residual (one fma each).
    //   x / d: r = 1/d refined once (shared by the two quotients); q0 = x r; q1 = q0 + (x - d q0) r; result =
    //   q1 + (x - d q1) r -- the core of v_div_scale / v_div_fmas / v_div_fixup, which only add scaling and specials.
    float distance;
    {
        const float s0 = __builtin_amdgcn_sqrtf(q);
        const float s_dn = __int_as_float(__float_as_int(s0) - 1), s_up = __int_as_float(__float_as_int(s0) + 1);
        const float r_dn = __builtin_fmaf(-s_dn, s0, q);
        float sres = r_dn <= 0.0f ? s_dn : s0;
        const float r_up = __builtin_fmaf(-s_up, s0, q);
        sres = r_up > 0.0f ? s_up : sres;
        distance = sres;                                              // :93
    }
    const uint64_t hit = cand & ballot64(rs2 > distance * distance) & ballot64(distance > 0.0001f);   // :95
    const float depth = radius_sum - distance;                        // :97
    f32x2 u;
    {
        const float r0 = __builtin_amdgcn_rcpf(distance);
        const float e0 = __builtin_fmaf(-distance, r0, 1.0f);
        const f32x2 rq = splat2(__builtin_fmaf(e0, r0, r0)), nd = splat2(-distance);
        const f32x2 q0 = v * rq;
        const f32x2 q1 = fma2(fma2(nd, q0, v), rq, q0);
        u = fma2(fma2(nd, q1, v), rq, q1);
    }
    const f32x2 c = (u * splat2(depth)) * splat2(stiffness);          // :98,101
    float w1 = 0.5f, w2 = 0.5f;                                       // == inv1 / (inv1 + inv1), exactly
    const uint64_t general = hit & ~(ballot64(r1 == r2) & plain);     // unequal (or odd) radii somewhere
    if (general != 0) {                                               // wave-uniform
        if (lanes_of(general)) {
            const float inv1 = 1.0f / r1, inv2 = 1.0f / r2;           // :103,104
            w1 = inv1 / (inv1 + inv2);                                // :107
            w2 = inv2 / (inv1 + inv2);                                // :108
        }
    }
    const bool mine = lanes_of(hit);
    p1 = select2(mine, p1 + c * splat2(w1), p1);                      // :110
    if constexpr (BOTH) p2 = select2(mine, p2 - c * splat2(w2), p2);  // :111
#ifdef GPE_COUNT_PAIRS
    count_pairs(g_pairs_hit, hit & counted);
#endif
    return hit;
}

// The reference's pair resolution (collision_solver.wgsl:66-118), on LDS-resident positions: one lane walks the
// pairs (a, b), a < b, of its cell.  Bit-exact restatements that shorten the dependent chain:
//  * the next partner's position is fetched while the current pair is computed: within one `a` loop every pair
//    touches a different partner, so that position cannot change in between;
//  * see pair_response for the arithmetic.
// (Called under divergent control flow -- the lanes walk cells of different sizes: the votes cover the lanes that are
// in the same iteration.)
template <class L>
__device__ __forceinline__ void resolve_cell(L &S, const uint32_t b, const uint32_t e, const float stiffness)
{
    for (uint32_t ia = b; ia + 1 < e; ++ia) {                         // :68
        const uint32_t a = S.mem[ia];
        uint32_t nb = S.mem[ia + 1];
        f32x2 p1 = {S.px[a], S.py[a]};
        const float r1 = S.rad[a];
        f32x2 nxt = {S.px[nb], S.py[nb]};
        float nr = S.rad[nb];
        const uint64_t plain = plain_radius_lanes(r1);
        bool dirty = false;
        for (uint32_t ib = ia + 1; ib < e; ++ib) {                    // :77
            const uint32_t bb = nb;
            f32x2 p2 = nxt;                                           // :86 live position
            const float r2 = nr;
            if (ib + 1 < e) { nb = S.mem[ib + 1]; nxt = (f32x2){S.px[nb], S.py[nb]}; nr = S.rad[nb]; }
            const uint64_t hit = pair_response(ballot64(true), p1, p2, r1, r2, plain, stiffness);
            if (hit != 0) {
                if (lanes_of(hit)) { S.px[bb] = p2.x; S.py[bb] = p2.y; dirty = true; }
            }
        }
        if (dirty) { S.px[a] = p1.x; S.py[a] = p1.y; }
    }
}

// The one-lane cells of a wave -- 2 or 3 members, rarely a pile the wave list had no room for -- as straight-line
// code on registers: slots in one LDS round trip, ids / positions / radii in a second, a three-element sorting
// network (ascending object index, collision_solver.wgsl:66-118; a cell of two needs none: the response is symmetric
// under exchanging the particles -- v -> -v negates the correction exactly, sums and products commute), the pairs
// (0,1), (0,2), (1,2) back to back with pair_response's wave-uniform early-outs, the moved positions stored at the
// end.  The general loop (resolve_cell) walked the same pairs through two nested loops with an LDS round trip per
// partner: ~60 wave instructions and five dependent round trips more per pass.
// `on`, `n <= 3` ... must reach this function as comparisons (they are voted on one by one, see pair_response).
template <class L>
__device__ __forceinline__ void resolve_small_cells(L &S, const bool on, const uint32_t b, const uint32_t n,
                                                    const float stiffness)
{
    const uint64_t on_m = ballot64(on);
    const uint64_t small_m = on_m & ballot64(n <= 3u), three_m = on_m & ballot64(n == 3u);
    const bool small = lanes_of(small_m), three = lanes_of(three_m);
    uint32_t m0 = 0, m1 = 0, m2 = 0;
    if (small) { m0 = S.mem[b]; m1 = S.mem[b + 1]; m2 = three ? (uint32_t)S.mem[b + 2] : m1; }
    uint32_t i0 = 0, i1 = 1, i2 = 2;
    f32x2 p0 = {0.f, 0.f}, p1 = {3.f, 3.f}, p2 = {6.f, 6.f};
    float r0 = 1.f, r1 = 1.f, r2 = 1.f;
    const bool any3 = three_m != 0;                                    // wave-uniform
    // The three-element network on (object index, slot) only; positions and radii are loaded by the ordered slots
    // afterwards: 10 selects instead of the 28 that ordered positions and radii as well, for one more LDS round trip in a
    // wave that holds a cell of three (round 4: -27 VALU lane-slots per particle, -1.5 % at every size).
    if (any3) {
        if (three) { i0 = S.id[m0]; i1 = S.id[m1]; i2 = S.id[m2]; }
#define GPE_CSWAP(A, B)                                                                                       \
        {                                                                                                     \
            const bool sw = lanes_of(three_m & ballot64(i##A > i##B));                                        \
            const uint32_t ta = sw ? i##B : i##A, tb = sw ? i##A : i##B, ma = sw ? m##B : m##A, mb = sw ? m##A : m##B; \
            i##A = ta; i##B = tb; m##A = ma; m##B = mb;                                                       \
        }
        GPE_CSWAP(0, 1) GPE_CSWAP(1, 2) GPE_CSWAP(0, 1)
#undef GPE_CSWAP
    }
    if (small) {
        p0 = (f32x2){S.px[m0], S.py[m0]}; r0 = S.rad[m0];
        p1 = (f32x2){S.px[m1], S.py[m1]}; r1 = S.rad[m1];
    }
    if (any3) {
        if (three) { p2 = (f32x2){S.px[m2], S.py[m2]}; r2 = S.rad[m2]; }
    }
    const uint64_t plain0 = plain_radius_lanes(r0);
    uint64_t h01 = 0, h02 = 0, h12 = 0;
    h01 = pair_response(small_m, p0, p1, r0, r1, plain0, stiffness);
    if (any3) {
        h02 = pair_response(three_m, p0, p2, r0, r2, plain0, stiffness);
        h12 = pair_response(three_m, p1, p2, r1, r2, plain_radius_lanes(r1), stiffness);
    }
    if ((h01 | h02 | h12) != 0) {                                      // wave-uniform
        if (lanes_of(h01 | h02)) { S.px[m0] = p0.x; S.py[m0] = p0.y; }
        if (lanes_of(h01 | h12)) { S.px[m1] = p1.x; S.py[m1] = p1.y; }
        if (lanes_of(h02 | h12)) { S.px[m2] = p2.x; S.py[m2] = p2.y; }
    }
    if (on && n > 3u) {                                                // a pile in the one-lane list: the general walk
        sort_members(S, b, b + n);
        resolve_cell(S, b, b + n, stiffness);
    }
}

// A cell of 4..8 members resolved by kGroupLanes consecutive lanes of one wave.  The reference's pair sequence
// (a, b), a < b in ascending object index (:68-118) only orders pairs that share a particle; pair (a, b) can
// run as soon as (a, b-1) and (a-1, b) are done, i.e. at step s = a + b of a wavefront schedule (s = 1 .. 2n - 3)
// -- 2n - 3 steps instead of n (n-1) / 2, every particle still seeing its updates in the reference's order.
// Lane a of the group owns particle a (registers) for the whole walk.  At step s its partner is particle s - a: the
// two lanes of a pair each fetch the other's live position and radius through two DPP moves -- the reflection
// i <-> s - i of an 8-lane group is row_half_mirror (i <-> 7 - i) followed by a row shift by |7 - s|, both fixed at
// compile time because the steps are unrolled -- and EACH COMPUTES ITS OWN HALF of the response as "particle 1".
// That is exact: seen from the other side v is negated, and negation commutes with every operation of the chain
// (squares and the distance are the same; products, the fused refinement steps and the final addition are odd in v;
// r1 + r2 and inv1 + inv2 commute), so lane b's p_b + (-c) w_b is the reference's p_b - c w_b bit for bit -- the
// symmetry the two-member cells already rely on.  (Rounds 1-3 moved the visitors through the lanes instead: a pipe up,
// a feed pipe down and their bookkeeping -- seven DPP moves and as many selects per step, and both halves of every
// response in one lane.  The steps mostly stop at the candidate test, so their fixed cost is what a group costs.)
// No LDS access inside the walk.  A colour pass lasts as long as its slowest cell: this is what shortens it.
constexpr uint32_t kGroupLanes = 8, kGroupMin = 4;
template <int CTRL>
__device__ __forceinline__ float dpp_mov(const float v)
{
    // (bound_ctrl: a source lane outside the row, or switched off, reads as 0)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
constexpr int kDppHalfMirror = 0x141;
// the value lane (S - i) of the 8-lane group holds, given the group's half-mirrored values (lane i holds lane 7 - i's)
template <int S>
__device__ __forceinline__ float dpp_reflect_from_mirrored(const float m)
{
    if constexpr (S < 7) return dpp_mov<0x100 + (7 - S)>(m);          // row_shl: lane i reads lane i + (7 - S)
    else if constexpr (S > 7) return dpp_mov<0x110 + (S - 7)>(m);     // row_shr: lane i reads lane i - (S - 7)
    else return m;
}
// lanes whose partner at step S can exist at all: 0 <= S - a, S - a != a  (a = lane mod 8)
template <int S, bool LOWER = false>
constexpr uint64_t group_step_lanes()
{
    uint64_t m = 0;
    for (int lane = 0; lane < 64; ++lane) {
        const int a = lane & 7;
        if (a <= S && 2 * a != S && (!LOWER || 2 * a < S)) m |= 1ull << lane;
    }
    return m;
}
template <int S, int SMAX, class F>
__device__ __forceinline__ void for_each_group_step(F &&f)
{
    f(std::integral_constant<int, S>{});
    if constexpr (S < SMAX) for_each_group_step<S + 1, SMAX>(f);
}
// NMAX: the most members a group cell of this window form can have
template <int NMAX, class L>
__device__ __forceinline__ void resolve_group(L &S, const uint32_t b, const uint32_t n, const int a,
                                              const float stiffness)
{
    static_assert(NMAX >= (int)kGroupMin && NMAX <= (int)kGroupLanes, "group cells: 4 .. 8 members");
    const int group_base = lane_id() & ~((int)kGroupLanes - 1);
    // order the members: rank = members with a smaller object index (they are distinct)
    const uint64_t has_m = ballot64((uint32_t)a < n);
    const bool has = lanes_of(has_m);
    const uint32_t my_slot = has ? (uint32_t)S.mem[b + a] : 0u;
    const uint32_t my_id = has ? S.id[my_slot] : 0xFFFFFFFFu;
    uint32_t rank = 0;
#pragma unroll
    for (int i = 0; i < NMAX; ++i) rank += ((uint32_t)__shfl((int)my_id, i, kGroupLanes) < my_id) ? 1u : 0u;
    // forward permute: lane r receives the slot of the member of rank r
    const uint32_t a_slot = (uint32_t)__builtin_amdgcn_ds_permute((group_base + (int)(has ? rank : (uint32_t)a)) << 2,
                                                                  (int)my_slot);
    f32x2 o = {0.f, 0.f};                                             // step-start state of the lane's particle
    float r1 = 1.f;
    if (has) { o = (f32x2){S.px[a_slot], S.py[a_slot]}; r1 = S.rad[a_slot]; }
    f32x2 p1 = o;
    const uint64_t plain = plain_radius_lanes(r1);
    const float r_mirrored = dpp_mov<kDppHalfMirror>(r1);
    const uint32_t n_plus_a = n + (uint32_t)a;                        // partner s - a exists while s - a < n
    // how long the longest schedule of the wave is (the cells of a wave differ): steps beyond 2 * 4 - 3 are skipped
    // by a scalar test unless some cell has the members for them
    const uint64_t any5 = has_m & ballot64(n >= 5u), any6 = has_m & ballot64(n >= 6u), any7 = has_m & ballot64(n >= 7u),
                   any8 = has_m & ballot64(n >= 8u);
    for_each_group_step<1, 2 * NMAX - 3>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        constexpr int need = (s + 4) / 2;                              // smallest n with 2n - 3 >= s
        if constexpr (need >= 5) {
            const uint64_t any = need == 5 ? any5 : (need == 6 ? any6 : (need == 7 ? any7 : any8));
            if (any == 0) return;                                      // wave-uniform
        }
        const f32x2 ip = {dpp_reflect_from_mirrored<s>(dpp_mov<kDppHalfMirror>(p1.x)),
                          dpp_reflect_from_mirrored<s>(dpp_mov<kDppHalfMirror>(p1.y))};
        const float ir = dpp_reflect_from_mirrored<s>(r_mirrored);
        constexpr uint64_t kStepLanes = group_step_lanes<s>();
        const uint64_t pairing = has_m & kStepLanes & ballot64((uint32_t)s < n_plus_a);
        constexpr uint64_t kLower = group_step_lanes<s, true>();      // (the lane with the smaller index of each pair)
        (void)pair_response<false>(pairing, p1, ip, r1, ir, plain, stiffness, kLower);
    });
    if (has && (__float_as_uint(p1.x) != __float_as_uint(o.x) || __float_as_uint(p1.y) != __float_as_uint(o.y))) {
        S.px[a_slot] = p1.x;
        S.py[a_slot] = p1.y;
    }
}

// A cell of 9..16 members resolved by the sixteen lanes of one DPP row: the symmetric walk of resolve_group at twice the
// width -- the reflection i <-> s - i of a row is row_mirror (i <-> 15 - i) followed by a row shift by |15 - s| -- four
// cells per wave side by side, 2n - 3 steps of one half-response each.  (Until round 4 these cells went to resolve_wave,
// one cell per wave, 2n - 2 steps of a whole response and seven wave-wide shifts each: ~15 k cycles for eleven members,
// and a colour pass of a compressed region's window waits for its slowest wave; phase stamps in profiles/r04.)
constexpr uint32_t kRowLanes = 16;
constexpr int kDppRowMirror = 0x140;
template <int S>
__device__ __forceinline__ float dpp_reflect_from_row_mirrored(const float m)
{
    if constexpr (S < 15) return dpp_mov<0x100 + (15 - S)>(m);        // row_shl: lane i reads lane i + (15 - S)
    else if constexpr (S > 15) return dpp_mov<0x110 + (S - 15)>(m);   // row_shr: lane i reads lane i - (S - 15)
    else return m;
}
template <int S, bool LOWER = false>
constexpr uint64_t row_step_lanes()
{
    uint64_t m = 0;
    for (int lane = 0; lane < 64; ++lane) {
        const int a = lane & 15;
        if (a <= S && 2 * a != S && (!LOWER || 2 * a < S)) m |= 1ull << lane;
    }
    return m;
}
template <int K>
__device__ __forceinline__ uint32_t row_rotated(const uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 + K /* row_ror:K */, 0xF, 0xF, false);
}
template <int K, int KMAX>
__device__ __forceinline__ uint32_t row_rank(const uint32_t id)
{
    const uint32_t r = row_rotated<K>(id) < id ? 1u : 0u;
    if constexpr (K < KMAX) return r + row_rank<K + 1, KMAX>(id); else return r;
}
// every lane of the wave calls (n == 0: no cell in this row); a = lane & 15
template <class L>
__device__ __forceinline__ void resolve_row(L &S, const uint32_t b, const uint32_t n, const int a, const float stiffness)
{
    const int row_base = lane_id() & ~((int)kRowLanes - 1);
    const uint64_t has_m = ballot64((uint32_t)a < n);
    const bool has = lanes_of(has_m);
    const uint32_t my_slot = has ? (uint32_t)S.mem[b + a] : 0u;
    const uint32_t my_id = has ? S.id[my_slot] : 0xFFFFFFFFu;
    // rank = members of the row with a smaller object index (distinct; lanes without a member hold the largest value)
    const uint32_t rank = row_rank<1, (int)kRowLanes - 1>(my_id);
    const uint32_t a_slot = (uint32_t)__builtin_amdgcn_ds_permute((row_base + (int)(has ? rank : (uint32_t)a)) << 2, (int)my_slot);
    f32x2 o = {0.f, 0.f};
    float r1 = 1.f;
    if (has) { o = (f32x2){S.px[a_slot], S.py[a_slot]}; r1 = S.rad[a_slot]; }
    f32x2 p1 = o;
    const uint64_t plain = plain_radius_lanes(r1);
    const float r_mirrored = dpp_mov<kDppRowMirror>(r1);
    const uint32_t n_plus_a = n + (uint32_t)a;                        // partner s - a exists while s - a < n
    for_each_group_step<1, 2 * (int)kRowLanes - 3>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        constexpr int need = (s + 4) / 2;                              // smallest n with 2n - 3 >= s
        if constexpr (need >= 10) {
            if ((has_m & ballot64(n >= (uint32_t)need)) == 0) return;  // wave-uniform: no row of the wave is that long
        }
        const f32x2 ip = {dpp_reflect_from_row_mirrored<s>(dpp_mov<kDppRowMirror>(p1.x)),
                          dpp_reflect_from_row_mirrored<s>(dpp_mov<kDppRowMirror>(p1.y))};
        const float ir = dpp_reflect_from_row_mirrored<s>(r_mirrored);
        constexpr uint64_t kStepLanes = row_step_lanes<s>();
        const uint64_t pairing = has_m & kStepLanes & ballot64((uint32_t)s < n_plus_a);
        constexpr uint64_t kLower = row_step_lanes<s, true>();
        (void)pair_response<false>(pairing, p1, ip, r1, ir, plain, stiffness, kLower);
    });
    if (has && (__float_as_uint(p1.x) != __float_as_uint(o.x) || __float_as_uint(p1.y) != __float_as_uint(o.y))) {
        S.px[a_slot] = p1.x;
        S.py[a_slot] = p1.y;
    }
}

// A cell of 9..64 members (a pile: particles pressed into one cell) resolved by a whole wave: the systolic array
// of resolve_group over 64 lanes (wave-wide DPP shifts).  One lane would walk n (n - 1) / 2 pairs one after the
// other -- 1225 for 50 members, ~0.25 ms, which every colour pass of every tile near the pile would wait for; the
// wavefront schedule takes 2 n - 3 steps.  Same operations per particle in the same order, so the same bits.
__device__ __forceinline__ float wave_from_lane_below(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /* wave_shr:1 */, 0xF, 0xF, true));
}
__device__ __forceinline__ int wave_from_lane_below(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, true);
}
__device__ __forceinline__ float wave_from_lane_above(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130 /* wave_shl:1 */, 0xF, 0xF, true));
}
__device__ __forceinline__ f32x2 wave_from_lane_below(f32x2 v) { return (f32x2){wave_from_lane_below(v.x), wave_from_lane_below(v.y)}; }
__device__ __forceinline__ f32x2 wave_from_lane_above(f32x2 v) { return (f32x2){wave_from_lane_above(v.x), wave_from_lane_above(v.y)}; }
template <class L>
__device__ __forceinline__ void resolve_wave(L &S, const uint32_t b, const uint32_t n, const float stiffness)
{
    const int a = lane_id();
    const uint64_t has_m = ballot64((uint32_t)a < n);
    const bool has = lanes_of(has_m);
    const uint32_t my_slot = has ? (uint32_t)S.mem[b + a] : 0u;
    const uint32_t my_id = has ? S.id[my_slot] : 0xFFFFFFFFu;
    uint32_t rank = 0;
    for (int i = 0; i < (int)n; ++i)                                    // n is wave-uniform: v_readlane
        rank += ((uint32_t)__builtin_amdgcn_readlane((int)my_id, i) < my_id) ? 1u : 0u;
    const uint32_t a_slot = (uint32_t)__builtin_amdgcn_ds_permute((int)(has ? rank : (uint32_t)a) << 2, (int)my_slot);
    f32x2 o = {0.f, 0.f};
    float r1 = 1.f;
    if (has) { o = (f32x2){S.px[a_slot], S.py[a_slot]}; r1 = S.rad[a_slot]; }
    f32x2 p1 = o;
    const uint64_t plain = plain_radius_lanes(r1);
    f32x2 qp = {0.f, 0.f};
    float qr = 1.f;
    int qb = -1;
    const int last = 2 * (int)n - 3;
    const bool first = a == 0;
    f32x2 fp = wave_from_lane_above(o);
    float fr = wave_from_lane_above(r1);
    for (int t = 0; t <= last; ++t) {
        f32x2 ip = wave_from_lane_below(qp);
        float ir = wave_from_lane_below(qr);
        int ib = wave_from_lane_below(qb);
        if (first) { ip = fp; ir = fr; ib = (t + 1 < (int)n) ? t + 1 : -1; }
        fp = wave_from_lane_above(fp); fr = wave_from_lane_above(fr);
        if (ib == a) p1 = ip;                                          // the lane's own particle has arrived
        const uint64_t pairing_m = has_m & ballot64(ib > a);
        (void)pair_response(pairing_m, p1, ip, r1, ir, plain, stiffness);
        // (what a lane without a pair passes on is marked "none" and never looked at)
        qp = ip; qr = ir; qb = lanes_of(pairing_m) ? ib : -1;
    }
    if (has && (__float_as_uint(p1.x) != __float_as_uint(o.x) || __float_as_uint(p1.y) != __float_as_uint(o.y))) {
        S.px[a_slot] = p1.x;
        S.py[a_slot] = p1.y;
    }
}

// Cells of 65..256 members (a crushed pile), sub-tile and spill windows only: one wave walks the pair matrix in
// 64 x 64 blocks, block row by block row.  The reference order -- pairs (a, b), a < b, ascending -- only constrains
// pairs that share a particle: (a, b) needs (a, b-1) and (a-1, b).  Diagonal block I: resolve_wave on chunk I.
// Block (I, J), J > I: the 64 owners of chunk I sit in the lanes (registers, for the whole block row), the members
// of chunk J enter at lane 0 one per step, meet one owner per step on their way up the lanes, and are stored back
// when they leave the last owner's lane: 64 + |J| - 1 steps.  Row-major block order gives every pair its two
// predecessors, so every particle sees its updates in the reference's order: same bits, ~n^2/64 steps instead
// of n (n - 1) / 2 pairs in one lane.
// (Up to 1024 members since round 4: by step 2000 of the 100 M gravity-on scene the two floor corners hold cells of 330,
// by step 2450 of 600 members; beyond this limit a cell falls to ONE lane -- 180 k pairs one after the other, each with
// the spill window's global round trips in its chain: one such tile took 107 M cycles, 2.7 x the over-capacity launch's
// whole balanced duration; profiles/r04/tile_cycles_tail.txt.)
constexpr uint32_t kWaveCellMax = 1024;
template <class L>
__device__ __forceinline__ void resolve_wave_blocked(L &S, const uint32_t b, const uint32_t n, const float stiffness)
{
    const int a = lane_id();
    constexpr int K = (int)kWaveCellMax / 64;
    const uint32_t chunks = (n + 63u) / 64u;
    // members into ascending object index: rank count over the whole cell, up to sixteen members per lane; the rank
    // rides in the slot word's upper bits (slots stay below 2^22: P4 sends no cell of a larger window here)
    {
        constexpr uint32_t kRankOne = 1u << 22;
        uint32_t pk[K], id[K];
        // (branch-free: a lane without a member in chunk k re-reads the cell's first member and discards it -- behind
        // per-chunk branches hipcc kept sixteen copies of the two arrays alive and spilled them)
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t m = (uint32_t)a + 64u * k;
            pk[k] = (uint32_t)S.mem[b + (m < n ? m : 0u)];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t m = (uint32_t)a + 64u * k;
            const uint32_t v = S.id[pk[k]];
            id[k] = m < n ? v : 0xFFFFFFFFu;
        }
#pragma unroll 1
        for (uint32_t kk = 0; kk < chunks; ++kk) {
            // (one copy of the loops: the chunk's ids by a chain of selects on the wave-uniform kk, not sixteen unrolled
            // bodies -- that form doubled the kernel's code and left its register arrays in scratch)
            uint32_t cur = id[0];
#pragma unroll
            for (int k = 1; k < K; ++k) cur = kk == (uint32_t)k ? id[k] : cur;
            const int lim = (int)min(64u, n - 64u * kk);                  // wave-uniform
#pragma unroll 1
            for (int i = 0; i < lim; ++i) {
                const uint32_t other = (uint32_t)__builtin_amdgcn_readlane((int)cur, i);
#pragma unroll
                for (int k = 0; k < K; ++k) pk[k] += other < id[k] ? kRankOne : 0u;
            }
        }
        wave_lds_order();                                              // every read of mem above precedes the writes
#pragma unroll
        for (int k = 0; k < K; ++k)
            if ((uint32_t)a + 64u * k < n) S.mem[b + (pk[k] >> 22)] = pk[k] & (kRankOne - 1u);
        wave_lds_order();
    }
    for (uint32_t I = 0; I < chunks; ++I) {
        const uint32_t bI = b + 64u * I, cI = min(64u, n - 64u * I);
        resolve_wave(S, bI, cI, stiffness);                            // pairs inside chunk I (it re-ranks: already sorted)
        if (I + 1 == chunks) break;
        wave_lds_order();
        // the owners of this block row
        const uint64_t has_m = ballot64((uint32_t)a < cI);
        const bool has = lanes_of(has_m);
        const uint32_t o_slot = has ? (uint32_t)S.mem[bI + a] : 0u;
        f32x2 p1 = {0.f, 0.f};
        float r1 = 1.f;
        if (has) { p1 = (f32x2){S.px[o_slot], S.py[o_slot]}; r1 = S.rad[o_slot]; }
        const f32x2 o = p1;
        const uint64_t plain = plain_radius_lanes(r1);
        const bool first = a == 0;
        for (uint32_t J = I + 1; J < chunks; ++J) {
            const uint32_t bJ = b + 64u * J, cJ = min(64u, n - 64u * J);
            // the visitors of chunk J: fetched by the 64 lanes at once, handed to lane 0 one per step with v_readlane
            // (a spill window keeps its arrays in global memory: fetched by lane 0 one step ahead, as until round 4,
            // every step waited for two dependent global round trips)
            int vs = 0;
            f32x2 vp = {0.f, 0.f};
            float vr = 1.f;
            if ((uint32_t)a < cJ) { vs = (int)S.mem[bJ + a]; vp = (f32x2){S.px[vs], S.py[vs]}; vr = S.rad[vs]; }
            f32x2 qp = {0.f, 0.f};
            float qr = 1.f;
            int qb = -1, qs = 0;
            const int steps = (int)(cI + cJ) - 1;
            for (int t = 0; t < steps; ++t) {
                f32x2 ip = wave_from_lane_below(qp);
                float ir = wave_from_lane_below(qr);
                int ib = wave_from_lane_below(qb), is = wave_from_lane_below(qs);
                const int tv = t & 63;                                 // (t >= cJ: no visitor enters, the values are unused)
                const f32x2 np = {__int_as_float(__builtin_amdgcn_readlane(__float_as_int(vp.x), tv)),
                                  __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vp.y), tv))};
                const float nr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vr), tv));
                const int ns = __builtin_amdgcn_readlane(vs, tv);
                if (first) {
                    ip = np; ir = nr; is = ns;
                    ib = t < (int)cJ ? t : -1;
                }
                const uint64_t pairing_m = has_m & ballot64(ib >= 0);
                (void)pair_response(pairing_m, p1, ip, r1, ir, plain, stiffness);
                const bool pairing = lanes_of(pairing_m);
                // the visitor leaves the block behind the last owner: store it (it may have been moved by lower
                // lanes; an unchanged store is harmless)
                if (pairing && (uint32_t)a + 1u == cI) { S.px[is] = ip.x; S.py[is] = ip.y; }
                qp = ip; qr = ir; qb = pairing ? ib : -1; qs = is;
            }
            wave_lds_order();
        }
        if (has && (__float_as_uint(p1.x) != __float_as_uint(o.x) || __float_as_uint(p1.y) != __float_as_uint(o.y))) {
            S.px[o_slot] = p1.x;
            S.py[o_slot] = p1.y;
        }
        wave_lds_order();
    }
}

// Offset of neighbour k (grid.wgsl:68-90 scan order: y outer, x inner, centre skipped) in a cell array of row
// stride PX, eight signed bytes packed into one constant.
template <int PX>
__device__ __forceinline__ int neighbour_offset(const int k)
{
    static_assert(PX + 1 < 128, "offsets fit a signed byte");
    constexpr unsigned long long packed =
        ((unsigned long long)(uint8_t)(int8_t)(-PX - 1)) | ((unsigned long long)(uint8_t)(int8_t)(-PX) << 8) |
        ((unsigned long long)(uint8_t)(int8_t)(-PX + 1) << 16) | ((unsigned long long)(uint8_t)(int8_t)(-1) << 24) |
        ((unsigned long long)(uint8_t)(int8_t)(1) << 32) | ((unsigned long long)(uint8_t)(int8_t)(PX - 1) << 40) |
        ((unsigned long long)(uint8_t)(int8_t)(PX) << 48) | ((unsigned long long)(uint8_t)(int8_t)(PX + 1) << 56);
    return (int)(int8_t)(uint8_t)(packed >> (8 * k));
}

// One tile: returns false when the region exceeds the window's capacity (nothing written).
// ORD: a sharded run -- the members of a cell are ordered by A.order_keys[local index] (the particle's index in the
// unsharded system) instead of by the local index.  A template parameter, not a run-time test: the instantiation
// of ordinary runs then contains no load of the order keys, and hipcc has no reason to wait for one
// (s_waitcnt vmcnt(0) in front of every prefetch of a previous position -- the rounds' loads ran one after the other).
template <bool ORD, class L>
__device__ __forceinline__ bool process_tile(L &S, const CollideArgs &A, const int tx, const int ty)
{
    constexpr int T = L::TILE;
    constexpr int RWX = L::RWX, RWY = L::RWY, PX = L::PX, NCELL = L::NCELL, NB = L::NB, NBLK = L::NBLK, PER = L::PER, QMAX = L::QMAX;
    // virtual blocks: an order-key (sharded) window looks every block up among the owned particles AND among the ghosts
    constexpr int VB = ORD ? 2 * NBLK : NBLK;
    static_assert(VB <= 255 && VB <= L::VBMAX, "sblk is 8 bit; the lookup arrays hold the virtual blocks");
    int tid_ = (int)threadIdx.x;
    // (the windows of the over-capacity launch run in a loop over tickets: everything that depends on the thread index
    // alone -- P4's cell coordinates and zone tests, the scans' offsets -- is invariant in that loop, and hipcc hoisted
    // 37 registers of it in front of the loop, spilled them there and reloaded them inside, s_waitcnt vmcnt(0) behind
    // each reload; recomputing them per window costs a few VALU instructions)
    if constexpr (T < 32) asm volatile("" : "+v"(tid_));
    const int tid = tid_;
    const int lane = tid & 63;
    constexpr int HX = L::HXL, HY = L::HYL;                            // cells kept left of / below the tile
    constexpr bool kTrim = !L::kGlobal;
    const int ox = tx * T - HX, oy = ty * T - HY;                      // origin of the cell window
    const int box = (tx * T - kHalo) >> 3, boy = (ty * T - kHalo) >> 3;   // first looked-up block
    GPE_STAMP_BEGIN();
    // (issued here, consumed behind P0: a crowded scene runs tens of thousands of short-lived tiles and sub-tiles, and a
    // global round trip exposed in each of them cost the dense launch a quarter of its time again: 5.95 -> 7.5 ms at
    // step 1000 of the 100 M soak)
    const uint32_t fresh_word = *A.fresh;
    const uint32_t owned_word = A.counts ? A.counts[0] : (uint32_t)(A.n_owned < 0xFFFFFFFFull ? A.n_owned : 0xFFFFFFFFull);
    uint32_t exc_word = 0;
    const int ptx = (tx * T) >> 5, pty = (ty * T) >> 5;                // the 32x32 parent tile (lists are kept per parent)
    const bool in_tb = A.tb.holds(ptx, pty);
    const uint32_t pt = in_tb ? A.tb.index(ptx, pty) : 0u;
    if (A.exc_count && in_tb) exc_word = A.exc_count[pt];
    uint32_t gho_word = 0, gsort_word = 0;
    if constexpr (ORD) {
        if (A.ghost_sort) gsort_word = *A.ghost_sort;
        if (A.gho_count && in_tb) gho_word = A.gho_count[pt];
    }

    // ---- P0: clear, look the region's blocks up, slot -> block map ---------------------------------
    S.cell_clear(tid);
    if (tid < 20) S.lcnt[tid] = 0;
    if (tid < VB) {
        const int rb = tid % NBLK;                                     // the block; tid >= NBLK: among the ghosts
        const int bi = rb % NB, bj = rb / NB;
        const int bx = box + bi, by = boy + bj;
        uint32_t start = 0, count = 0;
        const int lbx = bx - A.bx0, lby = by - A.by0;
        // (the ghosts: through their block table only when a ghost list ran over this step, or there are no lists)
        const bool ghosts_by_table = ORD && (A.gho_count == nullptr || __builtin_amdgcn_readfirstlane((int)gsort_word) != 0);
        const uint2 *tab = (ORD && tid >= NBLK) ? (ghosts_by_table ? A.gtable : nullptr) : A.table;
        if (tab && lbx >= 0 && lby >= 0 && lbx < A.blocks_x && lby < A.blocks_y) {
            const uint32_t mb = (uint32_t)(lby * A.blocks_x + lbx);
            if (mb < A.entries) {
                const uint2 se = tab[mb];                            // empty blocks hold (0xFFFFFFFF, 0)
                if (se.y > se.x) { start = se.x; count = se.y - se.x; }
            }
        }
        S.bstart[tid] = start;
        S.bcnt[tid] = count;
    }
    __syncthreads();
    if (tid < 64) {
        // exclusive scan of the block populations by one wave; particles of the tile's own blocks
        uint32_t carry = 0, own = 0;
        for (int base = 0; base < VB; base += 64) {
            const int b = base + lane;
            const uint32_t cb = (b < VB) ? S.bcnt[b] : 0u;
            const uint32_t inc = wave_inclusive_scan(cb);
            if (b < VB) {
                S.boff[b] = carry + inc - cb;
                const int bi = (b % NBLK) % NB, bj = (b % NBLK) / NB;
                if (bi >= 1 && bi < NB - 1 && bj >= 1 && bj < NB - 1) own += cb;
            }
            carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        }
        own = wave_sum(own);
        if (lane == 0) {
            S.boff[VB] = carry; S.misc[0] = carry; S.misc[1] = own; S.misc[3] = 0;
            if (!L::kGlobal) S.misc[2] = 0;                            // main tile: "a cell of more than 64 members"
        }
    } else if (tid < 64 + (NB - 2) * (NB - 2)) {
        // population of each 3x3-block window of this tile (what an 8x8-cell sub-tile would stage):
        // the host reads the step's maximum (lagged) to leave the native path before windows overfill
        const int wi = (tid - 64) % (NB - 2), wj = (tid - 64) / (NB - 2);
        uint32_t w = 0;
#pragma unroll
        for (int dj = 0; dj < 3; ++dj)
#pragma unroll
            for (int di = 0; di < 3; ++di) w += S.bcnt[(wj + dj) * NB + wi + di];
        if (w > kWindowReport) atomicMax(&A.tile_ctl[kCtlWindowMax], w);
    }
    __syncthreads();
    const uint32_t P = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.misc[0]);   // the same in every lane: keep it scalar
    const bool stale = __builtin_amdgcn_readfirstlane((int)fresh_word) == 0;
    const uint32_t n_owned_now = (uint32_t)__builtin_amdgcn_readfirstlane((int)owned_word);
    const uint32_t straggler_bit = stale ? kCodeStraggler : 0u;       // (listed under a block it is out of reach of: skipped)
    // stragglers handed to this tile's 32x32 parent by the hash kernel (P1 files them behind the looked-up particles)
    const uint32_t n_exc = stale ? min((uint32_t)__builtin_amdgcn_readfirstlane((int)exc_word), kExcSlots) : 0u;
    // ghosts listed for the tile's parent (sharded runs; none when the ghosts come through their block table this step)
    uint32_t n_gho = 0;
    if constexpr (ORD) {
        if (A.gho_count != nullptr && __builtin_amdgcn_readfirstlane((int)gsort_word) == 0)
            n_gho = min((uint32_t)__builtin_amdgcn_readfirstlane((int)gho_word), kGhostSlots);
    }
    // Nothing of its own to write?  With the table of this step: no particle in the tile's own blocks.  With a kept
    // table a particle may have drifted up to kDrift* cells out of the block that lists it, so only an empty lookup
    // region (own blocks + the ring around them) AND an empty straggler list tell -- a straggler that flew into empty
    // space exists for its tile through the list alone (test_stragglers_flying_into_empty_space_are_not_lost).
    if (stale ? (P == 0 && n_exc == 0) : S.misc[1] == 0) return true;
    if constexpr (L::kGlobal) {
        // take a slice of the global spill arena for this tile's particle arrays
        if (tid == 0) {
            const uint32_t extra = kExcSlots + (ORD ? kGhostSlots : 0u);                 // (+ the stragglers' and the listed ghosts' slots)
            const uint32_t base = atomicAdd(&A.tile_ctl[kCtlArena], P + extra);
            const bool ok = (uint64_t)base + P + extra <= (uint64_t)A.arena_cap;
            S.misc[2] = ok ? 1u : 0u;
            S.px = A.arena_px + base; S.py = A.arena_py + base; S.rad = A.arena_rad + base;
            S.id = A.arena_id + base; S.hm = A.arena_hm + base; S.sblk = A.arena_sblk + base;
            S.mem = A.arena_mem + 4ull * base;
        }
        __syncthreads();
        if (S.misc[2] == 0) return false;
    } else {
        if (P > (uint32_t)L::RAWCAP) return false;                     // more looked-up particles than slots
    }
    {
        // slot -> block map: kNatThreads / NBLK threads share each block's slots
        constexpr int SHARE = (kNatThreads / VB) > 0 ? (kNatThreads / VB) : 1;
        for (int b = tid % VB, sub = tid / VB; sub < SHARE && b < VB; b += kNatThreads) {
            const uint32_t lo = S.boff[b], hi = S.boff[b + 1];
            for (uint32_t i = lo + sub; i < hi; i += SHARE) S.sblk[i] = (uint8_t)b;
        }
    }
    __syncthreads();
    GPE_STAMP(0);

    // ---- P1: gather the region's particles (all loads of a thread in flight together), count the
    //          cell memberships -----------------------------------------------------------------------
    // Two rounds of 512 looked-up particles per pass of this loop: the window takes three (QMAX), but the mean tile
    // looks up 1.7 x 512, so the third round's instructions -- executed by every wave whether or not a lane has a
    // particle -- are left to a second pass that a scalar branch skips unless P > 1024.
    constexpr int QP = QMAX >= 2 ? 2 : 1;
    for (uint32_t s0 = 0; s0 < P; s0 += (uint32_t)QP * kNatThreads) {
        uint32_t pid[QP], blk[QP], cc[QP];
        float2 pp[QP];
        float pr[QP];
#pragma unroll
        for (int q = 0; q < QP; ++q) {
            // Branch-free on purpose: behind `if (s < P)` hipcc merges the loaded value with the default through
            // register copies that wait for the load (s_waitcnt vmcnt(0) right behind it), so the rounds' loads ran one
            // after the other -- three dependent global round trips.  Slots beyond P re-read slot P - 1 (a cache hit)
            // and are discarded by `keep` below.  (P >= 1: see P0.)
            const uint32_t s = min(s0 + (uint32_t)tid + (uint32_t)q * kNatThreads, P - 1u);
            blk[q] = S.sblk[s];
            const uint32_t *ids = (ORD && blk[q] >= (uint32_t)NBLK) ? A.gsorted_ids : A.sorted_ids;
            pid[q] = ids[S.bstart[blk[q]] + (s - S.boff[blk[q]])];
        }
#ifdef GPE_TILE_STAMPS
        { uint32_t acc = 0; for (int q = 0; q < QP; ++q) acc += pid[q]; asm volatile("" :: "v"(acc)); }
        GPE_STAMP(7);
#endif
#pragma unroll
        for (int q = 0; q < QP; ++q) {
            pp[q] = A.pos_in[pid[q]];
            pr[q] = A.radius[pid[q]];
            cc[q] = A.codes[pid[q]];
        }
        // sharded run: the member order is the particle's index in the unsharded system; the local index is looked
        // up again at write-back (P6)
        uint32_t lidq[QP];
#pragma unroll
        for (int q = 0; q < QP; ++q) lidq[q] = pid[q];
        if constexpr (ORD) {
#pragma unroll
            for (int q = 0; q < QP; ++q) pid[q] = A.order_keys[pid[q]];
        }
#ifdef GPE_TILE_STAMPS
        { float acc = 0; for (int q = 0; q < QP; ++q) acc += pp[q].x + pr[q]; asm volatile("" :: "v"(acc)); }
        GPE_STAMP(8);
#endif
        // home cell: the slot's block among the looked-up ones + the cell inside the block (k_native_hash),
        // relative to the cell window; particles whose home lies outside the window are dropped here
        int lxq[QP], lyq[QP];
        bool keep[QP];
        uint32_t slot[QP];
        uint64_t mq[QP];
#pragma unroll
        for (int q = 0; q < QP; ++q) {
            const uint32_t s = s0 + (uint32_t)tid + (uint32_t)q * kNatThreads;
            lxq[q] = code_window_x(cc[q], ox);
            lyq[q] = code_window_y(cc[q], oy);
            // (voted comparison by comparison, the masks combined by scalar ANDs: see pair_response)
            mq[q] = ballot64(s < P) & ballot64(lxq[q] < RWX) & ballot64(lyq[q] < RWY) & ballot64((cc[q] & straggler_bit) == 0u);
            // (an order-key window: the kept table may still list indices the owned range has shrunk below -- those
            // particles are ghosts now, or gone: they come through the ghosts' table, or not at all)
            if constexpr (ORD) {
                if (A.gtable != nullptr) mq[q] &= ballot64(blk[q] >= (uint32_t)NBLK) | ballot64(lidq[q] < n_owned_now);
            }
            keep[q] = lanes_of(mq[q]);
            slot[q] = s;
        }
        if constexpr (kTrim) {
            // kept particles get consecutive slots: one LDS atomic per wave (the order of the slots is free)
            uint32_t cnt = 0;
#pragma unroll
            for (int q = 0; q < QP; ++q) cnt += (uint32_t)__popcll(mq[q]);
            uint32_t base = 0;
            if (lane == 0 && cnt) base = atomicAdd(&S.misc[3], cnt);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
            for (int q = 0; q < QP; ++q) {
                slot[q] = base + popc_below_lane(mq[q]);
                base += (uint32_t)__popcll(mq[q]);
                keep[q] = keep[q] && slot[q] < (uint32_t)(sizeof(S.px) / sizeof(float));   // over capacity: see below
            }
        }
#pragma unroll
        for (int q = 0; q < QP; ++q) {
            if constexpr (!kTrim) {
                // The spill window's slots are the looked-up slots: a particle that has drifted out of the looked-up
                // region (it is listed under its old block) still owns one.  File it under ring cell 0 -- outside every
                // colour's zone, never walked, never written back -- so that the loops over the slots find a valid entry.
                if (!keep[q] && s0 + (uint32_t)tid + (uint32_t)q * kNatThreads < P) {
                    const uint32_t s = slot[q];
                    S.px[s] = pp[q].x; S.py[s] = pp[q].y; S.rad[s] = pr[q]; S.id[s] = pid[q];
                    S.cell_inc(0 + 1);
                    S.hm[s] = 0u;
                }
            }
            if (!keep[q]) continue;
            const uint32_t s = slot[q];
            const int lx = lxq[q], ly = lyq[q];
            S.px[s] = pp[q].x; S.py[s] = pp[q].y; S.rad[s] = pr[q]; S.id[s] = pid[q];
            if constexpr (L::kLid) S.lid[s] = lidq[q];
            const int home = (ly + 1) * PX + lx + 1;                  // index in the padded cell array
            S.cell_inc(home + 1);
            // phantom cells: the first three set bits of the overlap mask (grid.wgsl:68-90 keeps at most three)
            uint32_t over = (cc[q] >> kCodeOverlapShift) & 0xFFu;
            const uint32_t own = (lx >= HX && lx < HX + T && ly >= HY && ly < HY + T) ? (1u << 19) : 0u;
            S.hm[s] = (uint32_t)home | (over << 11) | own;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (over == 0) break;
                const int k = __ffs((int)over) - 1;
                over &= over - 1u;
                S.cell_inc(home + neighbour_offset<PX>(k) + 1);       // (always inside the padded array)
            }
        }
    }
    if (n_exc != 0 && tid < 64) {                                      // (scalar condition; one wave files them)
        // Stragglers: particles out of reach of the block the table lists them under, handed over with their cell.
        const bool have = (uint32_t)tid < n_exc;
        const uint2 en = A.exc_entry[(uint64_t)pt * kExcSlots + (have ? (uint32_t)tid : 0u)];
        uint32_t pid = en.x;
        const float2 pp = A.pos_in[pid];
        const float pr = A.radius[pid];
        const uint32_t cc = A.codes[pid];
        const uint32_t lidq = pid;
        if constexpr (ORD) pid = A.order_keys[pid];
        const int lx = (int)(en.y & 0xFFFFu) - ox, ly = (int)(en.y >> 16) - oy;
        bool keep = have && lx >= 0 && lx < RWX && ly >= 0 && ly < RWY;
        uint32_t sl = P + (uint32_t)tid;                               // the spill window: slots behind the looked-up ones
        if constexpr (kTrim) {
            const uint64_t mk = ballot64(keep);
            uint32_t base = 0;
            if (lane == 0 && mk) base = atomicAdd(&S.misc[3], (uint32_t)__popcll(mk));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            sl = base + popc_below_lane(mk);
            keep = keep && sl < (uint32_t)(sizeof(S.px) / sizeof(float));
        } else if (have && !keep) {
            S.px[sl] = pp.x; S.py[sl] = pp.y; S.rad[sl] = pr; S.id[sl] = pid;
            S.cell_inc(0 + 1);
            S.hm[sl] = 0u;
        }
        if (keep) {
            S.px[sl] = pp.x; S.py[sl] = pp.y; S.rad[sl] = pr; S.id[sl] = pid;
            if constexpr (L::kLid) S.lid[sl] = lidq;
            const int home = (ly + 1) * PX + lx + 1;
            S.cell_inc(home + 1);
            uint32_t over = (cc >> kCodeOverlapShift) & 0xFFu;
            const uint32_t own = (lx >= HX && lx < HX + T && ly >= HY && ly < HY + T) ? (1u << 19) : 0u;
            S.hm[sl] = (uint32_t)home | (over << 11) | own;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (over == 0) break;
                const int k = __ffs((int)over) - 1;
                over &= over - 1u;
                S.cell_inc(home + neighbour_offset<PX>(k) + 1);
            }
        }
    }
    if constexpr (ORD) {
        // Ghosts listed for the tile's parent by the hash kernel: filed like the stragglers, placed by the cell in their
        // code word (a listed ghost lies in the parent's window, so its cell mod 128 names one cell of this window or none).
        for (uint32_t g0 = 0; g0 < n_gho; g0 += kNatThreads) {         // (scalar bounds: whole waves, the slots use ballots)
            const uint32_t gi = g0 + (uint32_t)tid;
            const bool have = gi < n_gho;
            const uint32_t lidq = A.gho_entry[(uint64_t)pt * kGhostSlots + (have ? gi : 0u)];
            const float2 pp = A.pos_in[lidq];
            const float pr = A.radius[lidq];
            const uint32_t cc = A.codes[lidq];
            const uint32_t pid = A.order_keys[lidq];
            const int lx = code_window_x(cc, ox), ly = code_window_y(cc, oy);
            bool keep = have && lx < RWX && ly < RWY;
            uint32_t sl = P + n_exc + gi;                               // the spill window: behind the stragglers
            if constexpr (kTrim) {
                const uint64_t mk = ballot64(keep);
                uint32_t base = 0;
                if (lane == 0 && mk) base = atomicAdd(&S.misc[3], (uint32_t)__popcll(mk));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                sl = base + popc_below_lane(mk);
                keep = keep && sl < (uint32_t)(sizeof(S.px) / sizeof(float));
            } else if (have && !keep) {
                S.px[sl] = pp.x; S.py[sl] = pp.y; S.rad[sl] = pr; S.id[sl] = pid;
                S.cell_inc(0 + 1);
                S.hm[sl] = 0u;
            }
            if (keep) {
                S.px[sl] = pp.x; S.py[sl] = pp.y; S.rad[sl] = pr; S.id[sl] = pid;
                if constexpr (L::kLid) S.lid[sl] = lidq;
                const int home = (ly + 1) * PX + lx + 1;
                S.cell_inc(home + 1);
                uint32_t over = (cc >> kCodeOverlapShift) & 0xFFu;
                const uint32_t own = (lx >= HX && lx < HX + T && ly >= HY && ly < HY + T) ? (1u << 19) : 0u;
                S.hm[sl] = (uint32_t)home | (over << 11) | own;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (over == 0) break;
                    const int k = __ffs((int)over) - 1;
                    over &= over - 1u;
                    S.cell_inc(home + neighbour_offset<PX>(k) + 1);
                }
            }
        }
    }
    __syncthreads();
    GPE_STAMP(1);
    // particles in the window from here on: the kept ones (the global window keeps every looked-up particle: its
    // cell window is the looked-up blocks; the stragglers and the listed ghosts sit behind them)
    uint32_t PS = P + (kTrim ? 0u : n_exc + n_gho);
    if constexpr (kTrim) {
        PS = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.misc[3]);
        if (PS > (uint32_t)(sizeof(S.px) / sizeof(float))) return false;    // more kept particles than the window stages
        if (PS == 0) return true;                                      // (looked up, none inside the window)
    }

    // ---- P2: exclusive scan of the per-cell counts -> list starts ----------------------------------
    {
        const int c0 = tid * PER;
        uint32_t cn[PER];
        uint32_t sum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) { cn[k] = (c0 + k < NCELL) ? S.cell_get(c0 + k + 1) : 0u; sum += cn[k]; }
        uint32_t run = nat_block_exclusive_scan(sum, S.s_w, nullptr);
#pragma unroll
        for (int k = 0; k < PER; ++k)
            if (c0 + k < NCELL) { S.cell_set(c0 + k + 1, run); run += cn[k]; }
    }
    __syncthreads();
    GPE_STAMP(2);

    // ---- P3: fill the member lists (order fixed later by the per-cell sort) --------------------------
#ifdef GPE_DBG_SKIP
    if (!(GPE_DBG_SKIP & 4))
#endif
    for (uint32_t s = tid; s < PS; s += kNatThreads) {
        const uint32_t hm = S.hm[s];
        const int home = (int)(hm & 0x7FFu);
        uint32_t k = S.cell_inc(home + 1);
        S.mem[k] = s;
        uint32_t over = (hm >> 11) & 0xFFu;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (over == 0) break;
            const int kb = __ffs((int)over) - 1;
            over &= over - 1u;
            k = S.cell_inc(home + neighbour_offset<PX>(kb) + 1);
            S.mem[k] = s;
        }
    }
    __syncthreads();
    GPE_STAMP(3);

    // The tile's own particles and, when K12 is fused into the write-back, their previous positions: fetched
    // here so that the global round trip runs under the colour passes instead of at the end of the tile.
    // (a scalar, so that comparing against it waits for no load: the count of a sharded run is read here, once)
    const uint32_t n_owned = n_owned_now;
    constexpr int QOWN = (L::kSlots + kNatThreads - 1) / kNatThreads;  // ceil(kept capacity / threads)
    // (The windows of the over-capacity launch fetch at P6 instead: their kernel holds three window forms and the
    // blocked whole-wave walk, and twelve registers kept across the colour passes were spilled there -- each previous
    // position right behind its load, s_waitcnt vmcnt(0) in between: P4 took 18.6 k cycles per window instead of 4 k.)
    constexpr bool kFetchEarly = T >= 32;
    uint32_t own_id[QOWN];
    float2 own_prev[QOWN];
    if constexpr (kTrim && kFetchEarly) {
        // Branch-free (see P1): every lane loads -- a lane without a particle of the tile reads element 0 -- and
        // the loads of all rounds are issued before anything uses one of them, so they are in flight together and
        // nothing waits for them before P6.
        uint32_t fetch[QOWN];
#pragma unroll
        for (int q = 0; q < QOWN; ++q) {
            const uint32_t s = (uint32_t)tid + (uint32_t)q * kNatThreads;
            own_id[q] = 0xFFFFFFFFu;
            fetch[q] = 0u;
            if (q >= 2 && PS <= (uint32_t)q * kNatThreads) continue;   // (scalar: the third round is nearly always empty)
            const uint32_t sc = min(s, PS - 1u);                       // PS >= 1 (checked behind P1)
            const uint32_t hm = S.hm[sc];
            const bool own = s < PS && (hm & (1u << 19)) != 0;
            uint32_t id = S.id[sc];
            asm volatile("" : "+v"(id));                             // keep this an LDS read (no pointer select -> flat load)
            static_assert(!ORD || L::kLid, "order-key windows keep the local indices");
            if constexpr (ORD) id = S.lid[sc];                         // S.id holds the order key; the local index was kept
            own_id[q] = own ? id : 0xFFFFFFFFu;
            fetch[q] = (own && id < n_owned) ? id : 0u;
        }
#pragma unroll
        for (int q = 0; q < QOWN; ++q) {
            own_prev[q] = make_float2(0.f, 0.f);
            if (q >= 2 && PS <= (uint32_t)q * kNatThreads) continue;
            if (A.fuse_verlet) own_prev[q] = A.prev[fetch[q]];
        }
    }

#ifdef GPE_DBG_SKIP
    if (!(GPE_DBG_SKIP & 2))
#endif
    // ---- P4: active cells per colour.  Colour-major walk: every wave round looks at 64 cells of EACH colour
    //          (the four colours' reads are in flight together), so one ballot and one LDS atomic per colour,
    //          class and round compact them -------------------------------------------------------------
    {
        // only cells inside the widest exactness zone (colour 1: [x0-4, x1+3] x [y0-2, y1+1]) can be active:
        // (T + 8) (T + 4) / 4 per colour
        constexpr int ZW = (T + 8) / 2, ZH = (T + 4) / 2, QC = ZW * ZH, QZ = L::QZ;
        static_assert(QC == QZ, "one list slot per zone cell of a colour");
        // the walk starts at the even cell (x0 - 4, y0 - 2) and visits 2 x 2 groups: colour = position in the group
        static_assert(kConeLeft % 2 == 0 && kConeDown % 2 == 0 && T % 2 == 0, "groups start at even cells");
        static_assert(HX > kConeLeft && HY > kConeDown && L::HXR > kConeRight && L::HYR > kConeUp, "zones inside the window");
        for (int base = 0; base < QC; base += kNatThreads) {
            const int i = base + tid;
            const int hx = 2 * (i % ZW) + (HX - kConeLeft), hy = 2 * (i / ZW) + (HY - kConeDown);
            int lc[4];
            uint32_t cnt[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                // colour - 1 = (gx & 1) + 2 * (gy & 1) (collision_solver.wgsl:55-58); ox + hx, oy + hy are even
                lc[c] = (hy + (c >> 1) + 1) * PX + hx + (c & 1) + 1;
                cnt[c] = 0;
                if (i < QC) cnt[c] = S.cell_get(lc[c] + 1) - S.cell_get(lc[c]);
            }
            // (the classes as lane masks, voted comparison by comparison: see pair_response)
            uint64_t ms[4], mg[4];
            bool single[4], group[4];
            constexpr int WC = L::WC;
            const uint64_t in_m = ballot64(i < QC);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int lx = hx + (c & 1), ly = hy + (c >> 1);
                const int gxx = ox + lx, gyy = oy + ly;
                // cells beyond the tile's edges on each side; the colour's zone (kCone*): colour c + 1 reaches
                // 4 - c cells left, 3 - c right, 2 - (c >> 1) down, 1 - (c >> 1) up
                const int exl = HX - lx, exr = lx - (HX + T - 1), eyl = HY - ly, eyr = ly - (HY + T - 1);
                const uint64_t zone_m = ballot64(exl <= kConeLeft - c) & ballot64(exr <= kConeRight - c) &
                                        ballot64(eyl <= kConeDown - (c >> 1)) & ballot64(eyr <= kConeUp - (c >> 1));
                // morton(-1,-1) == 0xFFFFFFFF == UNUSED_CELL_ID: never a collision cell
                // (collision_cell_builder.wgsl:56); cells outside the colour's exactness zone are skipped
                const uint64_t alias_m = ballot64((gxx & 0xFFFF) == 0xFFFF) & ballot64((gyy & 0xFFFF) == 0xFFFF);
                const uint64_t act_m = in_m & ballot64(cnt[c] >= 2u) & zone_m & ~alias_m;
                // cells of 4..8 members go to the BACK of the colour's segment: they are resolved by a group
                // of 8 lanes (resolve_group); the others fill the segment from the front (one lane each)
                mg[c] = act_m & ballot64(cnt[c] >= kGroupMin) & ballot64(cnt[c] <= kGroupLanes);
                // cells of 9..64 members go to a whole wave each (resolve_wave), as far as the colour's list takes
                // (65..256 members: blocked, in the sub-tile and spill windows only; a main tile that meets such a
                // cell hands itself over to them)
                // (the blocked form packs a member's rank above its 22-bit slot: a spill window of 4 M particles and more
                // -- the whole system in 24 x 24 cells -- leaves its piles to one lane)
                const uint32_t kWaveMax = (T >= 32 || PS >= (1u << 22)) ? 64u : kWaveCellMax;
                const uint64_t big_m = act_m & ballot64(cnt[c] > kGroupLanes);
                uint64_t wave_m = big_m & ballot64(cnt[c] <= kWaveMax);
                if (big_m != 0) {                                      // (scalar: no such cell in most rounds)
                    if (T >= 32 && lanes_of(big_m) && cnt[c] > 64u) S.misc[2] = 1u;
                    bool wavec = lanes_of(wave_m);
                    bool rowc = false;
                    if constexpr (L::kRows) {
                        rowc = wavec && cnt[c] <= kRowLanes;
                        if (rowc) {
                            const uint32_t k = atomicAdd(&S.lcnt[12 + c], 1u);
                            if (k < (uint32_t)WC) { S.rlist[c * WC + k] = (uint16_t)lc[c]; wavec = false; } else rowc = false;
                        }
                    }
                    if (wavec) {
                        const uint32_t k = atomicAdd(&S.lcnt[8 + c], 1u);
                        if (k < (uint32_t)WC) S.wlist[c * WC + k] = (uint16_t)lc[c]; else wavec = false;
                    }
                    wave_m = ballot64(wavec || rowc);                  // (cells the lists had no room for: one lane)
                }
                ms[c] = act_m & ~mg[c] & ~wave_m;
                group[c] = lanes_of(mg[c]);
                single[c] = lanes_of(ms[c]);
            }
            // the eight list counters (colour x class) are bumped by eight lanes at once: one LDS round trip for
            // the wave instead of eight dependent ones
            uint32_t mine = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                mine = (lane == c) ? (uint32_t)__popcll(ms[c]) : mine;
                mine = (lane == 4 + c) ? (uint32_t)__popcll(mg[c]) : mine;
            }
            uint32_t mybase = 0;
            if (lane < 8 && mine) mybase = atomicAdd(&S.lcnt[lane], mine);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t bs = (uint32_t)__builtin_amdgcn_readlane((int)mybase, c);
                const uint32_t bg = (uint32_t)__builtin_amdgcn_readlane((int)mybase, 4 + c);
                if (single[c]) S.list[c * QZ + bs + popc_below_lane(ms[c])] = (uint16_t)lc[c];
                if (group[c]) S.list[c * QZ + (QZ - 1) - (bg + popc_below_lane(mg[c]))] = (uint16_t)lc[c];
            }
        }
    }
    __syncthreads();
    GPE_STAMP(4);
    if constexpr (T >= 32) {
        // a cell of more than 64 members: the sub-tile windows resolve those with a wave per cell
        if (S.misc[2]) return false;
    }

    // ---- P5: the four colour passes (collision_solver.rs:224), one lane per collision cell ----------
    // The colour passes are one long dependent chain per cell (sqrt, divisions, LDS round trips): give these
    // waves issue priority over the other tiles' throughput phases that share the SIMD.
#if GPE_P5_PRIO
    __builtin_amdgcn_s_setprio(GPE_P5_PRIO);
#endif
#ifdef GPE_DBG_SKIP
    if (!(GPE_DBG_SKIP & 1))                                           // diagnostic builds: phase cost by omission
#endif
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
#ifdef GPE_TILE_STAMPS
        const long long _tw0 = clock64();
#endif
        const uint32_t ns = S.lcnt[k], group_lanes = S.lcnt[4 + k] * kGroupLanes;
        const uint32_t nw = min(S.lcnt[8 + k], (uint32_t)L::WC);
        const uint32_t single_base = (group_lanes + 63u) & ~63u;      // waves are all-group or all-single
        const uint32_t work = single_base + ns;
        for (uint32_t i0 = 0; i0 < work; i0 += kNatThreads) {
            const uint32_t i = i0 + (uint32_t)tid;
            if (i < group_lanes) {
                const int lc = S.list[k * L::QZ + (L::QZ - 1) - (i / kGroupLanes)];
                const uint32_t b = S.cell_get(lc), e = S.cell_get(lc + 1);
#ifdef GPE_DBG_SKIP
                if (!(GPE_DBG_SKIP & 8))
#endif
                if (!GPE_RT_SKIP(1u))
                resolve_group<(int)kGroupLanes>(S, b, e - b, (int)(i % kGroupLanes), A.stiffness);
            } else if (i >= single_base && (i & ~63u) < work) {         // (whole waves: the walk uses ballots)
                const bool on = i < work;
                uint32_t b = 0, e = 0;
                if (on) {
                    const int lc = S.list[k * L::QZ + (i - single_base)];
                    b = S.cell_get(lc); e = S.cell_get(lc + 1);
                }
#ifdef GPE_DBG_SKIP
                if (!(GPE_DBG_SKIP & 16))
#endif
                if (!GPE_RT_SKIP(2u)) {
                    resolve_small_cells(S, on, b, e - b, A.stiffness);
                }
            }
        }
#ifdef GPE_TILE_STAMPS
        const long long _tw_cells = clock64();
#endif
        // Whole-wave cells and rows: drawn with a ticket, the whole-wave cells (the longest items) first -- the waves that
        // hold the colour's lane groups or one-lane cells come for them when they are through, the idle ones at once.
        // (Fixed shares were measured: with the items dealt out from the first wave up the group waves carried them on
        // top of their groups; from the last wave down the one-lane waves did, and step 2000 of the 100 M soak lost 3 %.)
        if constexpr (L::kRows) {
            const uint32_t nr = min(S.lcnt[12 + k], (uint32_t)L::WC);
            const uint32_t items = nw + (nr + 3u) / 4u;
            while (items != 0u) {                                         // (wave-uniform)
                uint32_t t = 0;
                if (lane == 0) t = atomicAdd(&S.lcnt[16 + k], 1u);
                t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
                if (t >= items || GPE_RT_SKIP(4u)) break;
                if (t < nw) {
                    const int lc = S.wlist[k * L::WC + t];
                    const uint32_t b = S.cell_get(lc), e = S.cell_get(lc + 1);
                    if (e - b <= 64u) resolve_wave(S, b, e - b, A.stiffness);
                    else resolve_wave_blocked(S, b, e - b, A.stiffness);
                } else {
                    const uint32_t ci = 4u * (t - nw) + (uint32_t)(lane >> 4);
                    uint32_t b = 0, e = 0;
                    if (ci < nr) {
                        const int lc = S.rlist[k * L::WC + ci];
                        b = S.cell_get(lc); e = S.cell_get(lc + 1);
                    }
                    resolve_row(S, b, e - b, lane & 15, A.stiffness);
                }
            }
        } else
        for (uint32_t i = (uint32_t)(tid >> 6); i < nw; i += kNatWaves) {     // wave-uniform
            if (GPE_RT_SKIP(4u)) break;
            const int lc = S.wlist[k * L::WC + i];
            const uint32_t b = S.cell_get(lc), e = S.cell_get(lc + 1);
            if (e - b <= 64u) resolve_wave(S, b, e - b, A.stiffness);
        }
#ifdef GPE_TILE_STAMPS
        GPE_STAMP(9 + k);
        {
            const long long _tw1 = clock64();
            __syncthreads();
            const long long _tw2 = clock64();
            if (A.stamps && lane == 0 && (blockIdx.x & 127u) == 5u) {
                const int w = tid >> 6;
                const int cls = ((uint32_t)(w * 64) < group_lanes) ? 0 : ((uint32_t)(w * 64) < work ? 1 : 2);   // groups / singles / idle
                atomicAdd(&A.stamps[32 + cls], (unsigned long long)(_tw1 - _tw0));
                atomicAdd(&A.stamps[36 + cls], (unsigned long long)(_tw2 - _tw1));
                atomicAdd(&A.stamps[40 + cls], 1ull);
                // the colour pass by class of cell: cells, and this wave's cycles in the whole-wave cells
                atomicAdd(&A.stamps[48], (unsigned long long)(_tw1 - _tw_cells));
                if (w == 0) {
                    atomicAdd(&A.stamps[44], (unsigned long long)ns);
                    atomicAdd(&A.stamps[45], (unsigned long long)(group_lanes / kGroupLanes));
                    atomicAdd(&A.stamps[46], (unsigned long long)nw);
                    atomicAdd(&A.stamps[47], 1ull);
                    uint32_t members = 0, largest = 0;
                    for (uint32_t i = 0; i < nw; ++i) {
                        const int lc = S.wlist[k * L::WC + i];
                        const uint32_t m = S.cell_get(lc + 1) - S.cell_get(lc);
                        members += m; largest = max(largest, m);
                        atomicAdd(&A.stamps[52 + (m <= 16u ? 0 : m <= 32u ? 1 : m <= 64u ? 2 : 3)], 1ull);
                    }
                    atomicAdd(&A.stamps[51], (unsigned long long)min(S.lcnt[12 + k], (uint32_t)L::WC));
                    atomicAdd(&A.stamps[49], (unsigned long long)members);
                    atomicAdd(&A.stamps[50], (unsigned long long)largest);
                }
            }
        }
#else
        __syncthreads();
#endif
    }
#if GPE_P5_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    GPE_STAMP(5);

    // ---- P6: write the tile's own particles back ------------------------------------------------------
    const bool packs = ORD && A.pack.on != 0u;                         // (scalar) a sharded step: the tiles pack
    if constexpr (kTrim && !kFetchEarly) {
        uint32_t fetch[QOWN];
#pragma unroll
        for (int q = 0; q < QOWN; ++q) {
            const uint32_t s = (uint32_t)tid + (uint32_t)q * kNatThreads;
            own_id[q] = 0xFFFFFFFFu;
            fetch[q] = 0u;
            if (q >= 1 && PS <= (uint32_t)q * kNatThreads) continue;   // (scalar)
            const uint32_t sc = min(s, PS - 1u);
            const uint32_t hm = S.hm[sc];
            const bool own = s < PS && (hm & (1u << 19)) != 0;
            uint32_t id = S.id[sc];
            asm volatile("" : "+v"(id));                             // keep this an LDS read (no pointer select -> flat load)
            if constexpr (ORD) id = S.lid[sc];
            own_id[q] = own ? id : 0xFFFFFFFFu;
            fetch[q] = (own && id < n_owned) ? id : 0u;
        }
#pragma unroll
        for (int q = 0; q < QOWN; ++q) {
            own_prev[q] = make_float2(0.f, 0.f);
            if (q >= 1 && PS <= (uint32_t)q * kNatThreads) continue;
            if (A.fuse_verlet) own_prev[q] = A.prev[fetch[q]];
        }
    }
    if constexpr (kTrim) {
#pragma unroll
        for (int q = 0; q < QOWN; ++q) {
            if (q >= (kFetchEarly ? 2 : 1) && PS <= (uint32_t)q * kNatThreads) continue;   // (scalar, as where own_id was filled)
            const uint32_t id = own_id[q];
            const bool have = id != 0xFFFFFFFFu;
            const uint32_t s = min((uint32_t)tid + (uint32_t)q * kNatThreads, (uint32_t)L::kSlots - 1u);
            const float2 c = make_float2(S.px[s], S.py[s]);
            const float rr = S.rad[s];
            float2 o = c;
            const bool mine = have && A.fuse_verlet && id < n_owned;
            if (mine) {
                // K12 on the resolved position: the integrated position becomes the live one, the resolved
                // position the previous one (particle_integration.wgsl:64,76)
                verlet_one(c.x, c.y, own_prev[q].x, own_prev[q].y, rr, A.vp, o.x, o.y);
                A.prev[id] = c;
                A.pos_out[id] = o;
            } else if (have) {
                A.pos_out[id] = c;
            }
            if constexpr (ORD) {
                if (packs) pack_if_near_border(A.pack, mine, id, o, c, rr, S.id[s], A.cell_size);
            }
        }
    } else
    for (uint32_t s0 = 0; s0 < PS; s0 += kNatThreads) {                // (whole waves: the pack uses ballots)
        const uint32_t s = s0 + (uint32_t)tid;
        const bool own = s < PS && (S.hm[min(s, PS - 1u)] & (1u << 19)) != 0;
        bool mine = false;
        uint32_t pk_id = 0, pk_key = 0;
        float2 pk_o = make_float2(0.f, 0.f), pk_c = pk_o;
        float pk_r = 0.f;
        if (own) {                                                     // the tile's own particle
            uint32_t id = S.id[s];
            pk_key = id;
            asm volatile("" : "+v"(id));                             // keep this an LDS read (no pointer select -> flat load)
            if constexpr (ORD) {                                       // S.id holds the order key: find the block of
                const uint32_t raw = s;                                // the looked-up slot, re-read the local index
                if (raw >= P + n_exc) {
                    id = A.gho_entry[(uint64_t)pt * kGhostSlots + (raw - P - n_exc)];   // a listed ghost
                } else if (raw >= P) {
                    // a straggler (filed behind the looked-up slots by P1): no block lists it -- its local index is
                    // the entry of the tile's straggler list it came from
                    id = A.exc_entry[(uint64_t)pt * kExcSlots + (raw - P)].x;
                } else {
                    int lo = 0, hi = VB;
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (S.boff[mid] <= raw) lo = mid; else hi = mid;
                    }
                    id = (lo >= NBLK ? A.gsorted_ids : A.sorted_ids)[S.bstart[lo] + (raw - S.boff[lo])];
                }
            }
            const float2 c = make_float2(S.px[s], S.py[s]);
            if (A.fuse_verlet && id < n_owned) {
                const float2 q = A.prev[id];
                float2 o;
                verlet_one(c.x, c.y, q.x, q.y, S.rad[s], A.vp, o.x, o.y);
                A.prev[id] = c;
                A.pos_out[id] = o;
                mine = true; pk_id = id; pk_o = o; pk_c = c; pk_r = S.rad[s];
            } else {
                A.pos_out[id] = c;
            }
        }
        if constexpr (ORD) {
            if (packs) pack_if_near_border(A.pack, mine, pk_id, pk_o, pk_c, pk_r, pk_key, A.cell_size);
        }
    }
    __syncthreads();
    GPE_STAMP(6);
    return true;
}


// One 16x16 quarter (tx, ty in 16-cell units) of an over-capacity 32x32 tile: as a 16x16 tile, else as four 8x8
// tiles, and an 8x8 tile whose 24x24-cell window exceeds even that LDS capacity gets its particle arrays from the
// global spill arena.
template <bool ORD>
struct OverflowLds {
    using Mid = TileLds<kTileMid, ORD ? kCapOrd : kCapMid, ORD>;
    using Small = TileLds<kTileSmall, ORD ? kCapOrd : kCapSmall, ORD>;
    using Spill = TileGlobal<kTileSmall>;
    union { Mid mid; Small small; Spill spill; };
};
template <bool ORD>
__device__ __forceinline__ void resolve_quarter(OverflowLds<ORD> &u, const CollideArgs &A, const int tx, const int ty)
{
    using Spill = typename OverflowLds<ORD>::Spill;
    const bool done = process_tile<ORD>(u.mid, A, tx, ty);
    __syncthreads();                                                   // the union's views alias each other
    if (done) return;
    if (threadIdx.x == 0) atomicAdd(&A.tile_ctl[kCtlSubTiles], 1u);
    for (int sub = 0; sub < 4; ++sub) {
        const int sx = tx * 2 + (sub & 1), sy = ty * 2 + (sub >> 1);
        bool ok = process_tile<ORD>(u.small, A, sx, sy);
        __syncthreads();
        if (ok) continue;
        if (threadIdx.x == 0) atomicAdd(&A.tile_ctl[kCtlSpills], 1u);
        ok = process_tile<ORD>(u.spill, A, sx, sy);
        __syncthreads();
        if (ok) continue;
        // The spill arena is exhausted: flag it (sticky; gpe_sync / gpe_download report the error and the state of
        // this step is NOT a result) and pass the tile's own particles through unresolved and unintegrated, so that
        // what a host reads back stays finite.  "Own" as P1 decides it -- by the home cell in the code word, not by
        // block membership: with a kept table a block still lists particles that have drifted into a neighbouring
        // tile (that tile writes them) and misses those that drifted in; stragglers come through the tile's list.
        if (threadIdx.x == 0) atomicOr(&A.tile_ctl[kCtlError], kErrTileOverflow);
        {
            constexpr int VBS = ORD ? 2 * Spill::NBLK : Spill::NBLK;
            const int wx = sx * kTileSmall - Spill::HXL, wy = sy * kTileSmall - Spill::HYL;
            const bool stale = *A.fresh == 0u;
            const uint32_t owned_now = A.counts ? A.counts[0] : (uint32_t)(A.n_owned < 0xFFFFFFFFull ? A.n_owned : 0xFFFFFFFFull);
            for (int b = threadIdx.x; b < VBS; b += kNatThreads) {
                const uint32_t *ids = (ORD && b >= Spill::NBLK) ? A.gsorted_ids : A.sorted_ids;
                for (uint32_t q = 0; q < u.spill.bcnt[b]; ++q) {
                    const uint32_t id = ids[u.spill.bstart[b] + q];
                    const uint32_t code = A.codes[id];
                    const int lx = code_window_x(code, wx) - Spill::HXL, ly = code_window_y(code, wy) - Spill::HYL;
                    bool own = lx >= 0 && lx < kTileSmall && ly >= 0 && ly < kTileSmall && !(stale && (code & kCodeStraggler));
                    if (ORD) own = own && (A.gtable == nullptr || b >= Spill::NBLK || id < owned_now);
                    if (own) A.pos_out[id] = A.pos_in[id];
                }
            }
            const int qtx = (sx * kTileSmall) >> 5, qty = (sy * kTileSmall) >> 5;
            if (stale && A.exc_count && A.tb.holds(qtx, qty)) {
                const uint32_t pt = A.tb.index(qtx, qty);
                const uint32_t ne = min(A.exc_count[pt], kExcSlots);
                for (uint32_t e = threadIdx.x; e < ne; e += kNatThreads) {
                    const uint2 en = A.exc_entry[(uint64_t)pt * kExcSlots + e];
                    const int lx = (int)(en.y & 0xFFFFu) - sx * kTileSmall, ly = (int)(en.y >> 16) - sy * kTileSmall;
                    if (lx >= 0 && lx < kTileSmall && ly >= 0 && ly < kTileSmall) A.pos_out[en.x] = A.pos_in[en.x];
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// The 32x32 tile of the dense launch, second form: DIRECT cell slots.
//
// process_tile builds the member lists of a window by a counting sort -- count the memberships per cell (P1), scan the
// counts (P2), fill the lists (P3): three barrier-separated phases whose LDS round trips every wave of the tile waits
// for, and the tile is bound by exactly that, the latency of its chain of phases (profiles/r03/
// ab_p5_candidate_cull_rejected.txt).  At the densities the dense launch is for, cells are small (25 % packing: 0.2 % of
// the cells hold more than six members), so here every cell of the tile's zone owns six member slots outright: the
// atomic that counts a membership hands out the slot, and P2 / P3 are gone.  The colour passes read a cell's members
// with one LDS access instead of list -> cell offsets -> members.  A seventh member goes to a short side list; cells of
// 7..64 members are gathered from it by the wave that resolves them.  A window with more than kBigCap such entries (a
// compressed scene) is left to process_tile's windows in the over-capacity launch, as is anything else this form has
// no room for.  Same operations per particle pair in the same order: same bits.
// ---------------------------------------------------------------------------------------------------
// The phantom cells of a particle, decoded once: entry[overlap mask] = count (2 bits) | for each of the first three set
// bits k of the mask (grid.wgsl:68-90 keeps at most three, in scan order: y outer, x inner, centre skipped) dx + 1 (2 bits)
// and dy + 1 (2 bits).  512 bytes that stay in the vector L1: one load per staged particle instead of find-first-set,
// clear, the centre skip and a division by three per membership (~7 VALU instructions x 3 memberships x 2 rounds per wave).
struct PhantomTable { uint16_t v[256]; };
constexpr PhantomTable make_phantom_table()
{
    PhantomTable t{};
    for (int m = 0; m < 256; ++m) {
        int cnt = 0, e = 0;
        for (int k = 0; k < 8 && cnt < 3; ++k) {
            if (!((m >> k) & 1)) continue;
            const int kk = k + (k >= 4 ? 1 : 0);
            e |= ((kk % 3) | ((kk / 3) << 2)) << (2 + 4 * cnt);
            ++cnt;
        }
        t.v[m] = (uint16_t)(e | cnt);
    }
    return t;
}
__device__ const PhantomTable kPhantomTable = make_phantom_table();
#ifndef GPE_VAR_PHANTOM_TABLE
#define GPE_VAR_PHANTOM_TABLE 1
#endif
constexpr int kDirectSlots = 6;                // member slots a zone cell owns
constexpr int kBigCap = 96;                    // memberships beyond that, per tile
template <int TX_, int TY_, int CAP, bool LID, int NT_>
struct TileDirect {
    static constexpr int TX = TX_, TY = TY_, NT = NT_, NW = NT_ / 64;   // tile (cells), threads, waves
    static constexpr bool kGlobal = false;
    static constexpr bool kLid = LID;
    static constexpr int kSlots = CAP;
    static constexpr int HXL = kConeLeft + 1, HXR = kConeRight + 1, HYL = kConeDown + 1, HYR = kConeUp + 1;
    static constexpr int RWX = TX + HXL + HXR, RWY = TY + HYL + HYR;   // cell window: where a kept particle's home may lie
    // zone cells: the 2 x 2 colour groups from the even cell (x0 - 4, y0 - 2) on, (T + 8) x (T + 4) cells; the colour
    // zones (kCone*) lie inside.  Only memberships of zone cells are filed.
    static constexpr int ZX = TX + 8, ZY = TY + 4, NZ = ZX * ZY, QZ = NZ / 4;
    static constexpr int ZOX = HXL - kConeLeft, ZOY = HYL - kConeDown;   // window coordinate of zone cell (0, 0)
    static constexpr int NBX = (TX + 2 * kHalo) / 8, NBY = (TY + 2 * kHalo) / 8;
    static constexpr int NBLK = NBX * NBY;
    // (a 32x16 half looks up 6 x 4 blocks, 3/4 of what a 32x32 tile does, for half the cells: it exists for tiles of up to
    // ~4 x the benchmark density, so it stages more looked-up particles, side-list entries and whole-wave cells)
    static constexpr int QMAX = TY >= 32 ? GPE_QMAX_MAIN : 7;
    static constexpr int RAWCAP = QMAX * NT;
    static constexpr int WC = TY >= 32 ? 16 : 32;
    static constexpr int kBig = TY >= 32 ? kBigCap : 2 * kBigCap;
    uint32_t lid[LID ? CAP : 1];
    float px[CAP], py[CAP], rad[CAP];
    uint32_t id[CAP];
    uint8_t own[CAP];          // the particle's home cell lies in the tile
    // member slots a zone cell owns: six; an order-key (sharded) window keeps a local index per particle as well and
    // pays for it with the sixth slot (its cells of six members go to the side list and a wave, like cells of seven)
    static constexpr int kMemSlots = LID ? kDirectSlots - 1 : kDirectSlots;
    uint32_t cntw[(NZ + 1) / 2];   // members per zone cell, two 16-bit counters per word (LDS atomics are 32 bit)
    __device__ __forceinline__ uint32_t cnt_inc(int i)                // returns the value before the add
    {
        const uint32_t sh = (uint32_t)(i & 1) * 16u;
        return (atomicAdd(&cntw[i >> 1], 1u << sh) >> sh) & 0xFFFFu;
    }
    __device__ __forceinline__ uint32_t cnt_get(int i) const { return (cntw[i >> 1] >> ((uint32_t)(i & 1) * 16u)) & 0xFFFFu; }
    // members: kDirectSlots per zone cell, then 64 per wave for the cells gathered from the side list.  (Named like
    // TileLds' member array: the resolvers index S.mem[b + k].)
    uint16_t mem[kMemSlots * NZ + NW * 64];
    union {
        uint16_t list[4 * QZ]; // active cells, one segment per colour (P4 on)
        uint8_t sblk[RAWCAP];  // P0-P1 only: region block a looked-up slot came from
    };
    uint16_t wlist[4 * WC];    // cells of more than kDirectSlots members, per colour
    uint32_t big[kBig];        // memberships that found their cell's slots taken: zone cell << 16 | particle slot
    uint32_t lcnt[12];
    static constexpr int VBMAX = NBLK;                    // (an order-key window gets its ghosts from the tile's ghost list)
    uint32_t bstart[VBMAX];
    uint32_t bcnt[VBMAX];
    uint32_t boff[VBMAX + 1];
    uint32_t s_w[16];
    uint32_t misc[6];          // [0] looked up, [2] hand the tile on, [3] kept, [4] side-list entries
};

template <bool ORD, class L, bool HINTS = false>
__device__ __forceinline__ bool process_tile_direct(L &S, const CollideArgs &A, const int tx, const int ty)
{
    constexpr int TX = L::TX, TY = L::TY, NT = L::NT, NW = L::NW;
    constexpr int RWX = L::RWX, RWY = L::RWY, NBX = L::NBX, NBY = L::NBY, NBLK = L::NBLK, QMAX = L::QMAX;
    constexpr int VB = NBLK;                                           // (the ghosts of an order-key window: from the ghost list)
    static_assert(VB <= 255 && VB <= L::VBMAX, "sblk is 8 bit; the lookup arrays hold the blocks");
    constexpr int ZX = L::ZX, ZY = L::ZY, NZ = L::NZ, QZ = L::QZ;
    constexpr int MS = L::kMemSlots;
    constexpr int HX = L::HXL, HY = L::HYL;
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int ox = tx * TX - HX, oy = ty * TY - HY;                    // origin of the cell window
    const int box = (tx * TX - kHalo) >> 3, boy = (ty * TY - kHalo) >> 3;   // first looked-up block
    GPE_STAMP_BEGIN();
    const uint32_t fresh_word = *A.fresh;                              // (issued here, consumed behind P0: see process_tile)
    const uint32_t owned_word = A.counts ? A.counts[0] : (uint32_t)(A.n_owned < 0xFFFFFFFFull ? A.n_owned : 0xFFFFFFFFull);
    // stragglers handed to this tile by the hash kernel (the lists are kept per 32x32 tile: a half reads its parent's)
    static_assert(TX == 32 && (TY == 32 || TY == 16), "straggler lists, ghost lists and rosters are kept per 32x32 tile");
    const int ptx = tx, pty = (ty * TY) >> 5;
    const bool in_tb = A.tb.holds(ptx, pty);                           // (always, for the tiles of the dense launch)
    const uint32_t pt = in_tb ? A.tb.index(ptx, pty) : 0u;
    const uint32_t exc_word = (A.exc_count && in_tb) ? A.exc_count[pt] : 0u;
    uint32_t gho_word = 0, gsort_word = 0;
    if constexpr (ORD) {
        if (A.ghost_sort) gsort_word = *A.ghost_sort;
        if (A.gho_count && in_tb) gho_word = A.gho_count[pt];
    }
    // the tile's roster (CollideArgs): header, and the first ids on the chance that it is valid
    constexpr bool kRoster = TX == 32 && TY == 32 && NT == 512;
    constexpr int QP = QMAX >= 2 ? 2 : 1;
    static_assert(!kRoster || L::RAWCAP == kRosterCap, "roster stride");
    const bool rosters = kRoster && A.roster_hdr != nullptr && in_tb;
    const uint64_t roster_base = (uint64_t)pt * (uint64_t)kRosterCap;
    uint4 hdr = make_uint4(0u, 0u, 0u, 0u);
    uint32_t sorts_word = 0;
    uint32_t first_ids[QP];
#pragma unroll
    for (int q = 0; q < QP; ++q) first_ids[q] = 0u;
    if (rosters) {
        hdr = A.roster_hdr[pt];
        sorts_word = *A.sorts_seen;
#pragma unroll
        for (int q = 0; q < QP; ++q) first_ids[q] = A.roster_ids[roster_base + (uint32_t)tid + (uint32_t)q * NT];
    }

    // ---- P0: clear the counters, look the region's blocks up, slot -> block map (as process_tile) ------------------
    for (int i = tid; i < (NZ + 1) / 2; i += NT) S.cntw[i] = 0;
    if (tid < 12) S.lcnt[tid] = 0;
    const bool stale = __builtin_amdgcn_readfirstlane((int)fresh_word) == 0;
    if constexpr (kRoster && HINTS) {
        // A hinted tile (kCtlHints): two of the launch's first workgroups redo it as halves.  (In front of every other way
        // out: a sharded tile must not be taken twice -- it would pack its particles twice.)
        const uint32_t hinted_for = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr.w);
        if (rosters && A.front_wgs != 0u && (hinted_for == A.step_stamp || hinted_for == A.step_stamp + 1u)) return true;
    }
    if constexpr (ORD) {
        // a ghost list ran over this step: the ghosts come through their block table, which only the counting-sort
        // windows of the over-capacity launch look up -- hand the tile on (a crowded border; the host's `crowded` policy
        // moves such scenes to those windows altogether)
        if (A.gho_count == nullptr || __builtin_amdgcn_readfirstlane((int)gsort_word) != 0) return false;
    }
    const uint32_t stamp_now = (uint32_t)__builtin_amdgcn_readfirstlane((int)sorts_word) + 1u;
    // (scalar) the roster is of the table in use: no lookup
    const bool listed = rosters && stale && (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr.y) == stamp_now;
    const bool record = rosters && !listed && A.roster_write != 0u;
    if (listed) {
        const uint32_t count = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr.x);
        if (count == 0xFFFFFFFFu) { GPE_BAIL(1); return false; }       // more looked-up particles than the tile stages
        const uint32_t wmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr.z);
        if (tid == 0) {
            S.misc[0] = count; S.misc[2] = 0; S.misc[3] = 0; S.misc[4] = 0;
            if (wmax > kWindowReport) atomicMax(&A.tile_ctl[kCtlWindowMax], wmax);
        }
    } else {
    if (tid == 0) S.misc[5] = 0;
    if (tid < VB) {
        const int rb = tid % NBLK;
        const int bi = rb % NBX, bj = rb / NBX;
        const int lbx = box + bi - A.bx0, lby = boy + bj - A.by0;
        uint32_t start = 0, count = 0;
        const uint2 *tab = A.table;
        if (tab && lbx >= 0 && lby >= 0 && lbx < A.blocks_x && lby < A.blocks_y) {
            const uint32_t mb = (uint32_t)(lby * A.blocks_x + lbx);
            if (mb < A.entries) {
                const uint2 se = tab[mb];                            // empty blocks hold (0xFFFFFFFF, 0)
                if (se.y > se.x) { start = se.x; count = se.y - se.x; }
            }
        }
        S.bstart[tid] = start;
        S.bcnt[tid] = count;
    }
    __syncthreads();
    if (tid < 64) {
        uint32_t carry = 0;
        for (int base = 0; base < VB; base += 64) {
            const int b = base + lane;
            const uint32_t cb = (b < VB) ? S.bcnt[b] : 0u;
            const uint32_t inc = wave_inclusive_scan(cb);
            if (b < VB) S.boff[b] = carry + inc - cb;
            carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        }
        if (lane == 0) { S.boff[VB] = carry; S.misc[0] = carry; S.misc[2] = 0; S.misc[3] = 0; S.misc[4] = 0; }
        {                                                              // particles in the tile's own blocks (misc[1])
            static_assert(NBLK <= 64, "one lane per block");
            const int bi = lane % NBX, bj = lane / NBX;
            const bool own_blk = lane < NBLK && bi >= 1 && bi < NBX - 1 && bj >= 1 && bj < NBY - 1;
            uint32_t v = 0;
            if (own_blk) v = S.bcnt[lane];
            const uint32_t own = wave_sum(v);                          // (all 64 lanes of wave 0 take part)
            if (lane == 0) S.misc[1] = own;
        }
    } else if (tid < 64 + (NBX - 2) * (NBY - 2)) {
        const int wi = (tid - 64) % (NBX - 2), wj = (tid - 64) / (NBX - 2);
        uint32_t w = 0;
#pragma unroll
        for (int dj = 0; dj < 3; ++dj)
#pragma unroll
            for (int di = 0; di < 3; ++di) w += S.bcnt[(wj + dj) * NBX + wi + di];
        if (w > kWindowReport) atomicMax(&A.tile_ctl[kCtlWindowMax], w);
        if (record && w > kWindowReport) atomicMax(&S.misc[5], w);
    }
    __syncthreads();
    }
    const uint32_t P = listed ? (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr.x)
                              : (uint32_t)__builtin_amdgcn_readfirstlane((int)S.misc[0]);
    const uint32_t n_owned = (uint32_t)__builtin_amdgcn_readfirstlane((int)owned_word);
    const uint32_t straggler_bit = stale ? kCodeStraggler : 0u;
    const uint32_t n_exc = stale ? min((uint32_t)__builtin_amdgcn_readfirstlane((int)exc_word), kExcSlots) : 0u;
    const uint32_t n_gho = ORD ? min((uint32_t)__builtin_amdgcn_readfirstlane((int)gho_word), kGhostSlots) : 0u;   // listed ghosts
    if (record && tid == 0) {
        // the header of the roster the gather below writes (an empty lookup is a valid, empty roster)
        const uint32_t count = P > (uint32_t)L::RAWCAP ? 0xFFFFFFFFu : P;
        A.roster_hdr[pt] = make_uint4(count, stamp_now, S.misc[5], 0u);   // (w: no hint)
    }
    {
        // nothing of its own to write?  (see process_tile: own blocks with the table of this step; the whole lookup
        // region and the straggler lists with a kept table)
        const uint32_t any_exc = n_exc;
        // (A tile that writes a roster goes through the gather for it even when nothing of this step's is its own.)
        if (stale ? (P == 0 && any_exc == 0) : (S.misc[1] == 0 && !(record && P != 0))) return true;
    }
    if (P > (uint32_t)L::RAWCAP) { GPE_BAIL(1); return false; }        // more looked-up particles than slots
    if (!listed) {
        constexpr int SHARE = (NT / VB) > 0 ? (NT / VB) : 1;
        for (int b = tid % VB, sub = tid / VB; sub < SHARE && b < VB; b += NT) {
            const uint32_t lo = S.boff[b], hi = S.boff[b + 1];
            for (uint32_t i = lo + sub; i < hi; i += SHARE) S.sblk[i] = (uint8_t)b;
        }
    }
    __syncthreads();
    GPE_STAMP(0);

    // ---- P1: gather, keep, and file every membership straight into its cell's slots ---------------------------------
    // one membership: the counter's old value is the slot; the seventh member of a cell goes to the side list
    auto file = [&](const int zx, const int zy, const uint32_t s) {
#ifdef GPE_DBG_SKIP
        if (GPE_DBG_SKIP & 4) return;
#endif
        if ((unsigned)zx < (unsigned)ZX && (unsigned)zy < (unsigned)ZY) {
            const int zc = zy * ZX + zx;
            const uint32_t k = S.cnt_inc(zc);
            if (k < (uint32_t)MS) S.mem[zc * MS + (int)k] = (uint16_t)s;
            else {
                const uint32_t e = atomicAdd(&S.misc[4], 1u);
                if (e < (uint32_t)L::kBig) S.big[e] = ((uint32_t)zc << 16) | s;
            }
        }
    };
    auto insert = [&](const uint32_t s, const float2 pp, const float pr, const uint32_t pid, const uint32_t lidv,
                      const int lx, const int ly, uint32_t over) {
        S.px[s] = pp.x; S.py[s] = pp.y; S.rad[s] = pr; S.id[s] = pid;
        if constexpr (L::kLid) S.lid[s] = lidv;
        S.own[s] = (lx >= HX && lx < HX + TX && ly >= HY && ly < HY + TY) ? 1 : 0;
        const int zx = lx - L::ZOX, zy = ly - L::ZOY;
        file(zx, zy, s);
        // phantom cells: the first three set bits of the overlap mask (grid.wgsl:68-90 keeps at most three); neighbour
        // k of the scan (y outer, x inner, centre skipped): dx = {-1,0,1,-1,1,-1,0,1}[k], dy = {-1,-1,-1,0,0,1,1,1}[k]
#if GPE_VAR_PHANTOM_TABLE
        // (`over` is the particle's entry of kPhantomTable here: the gather looked it up)
        const int cnt = (int)(over & 3u);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (j >= cnt) break;
            file(zx - 1 + (int)((over >> (2 + 4 * j)) & 3u), zy - 1 + (int)((over >> (4 + 4 * j)) & 3u), s);
        }
#else
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (over == 0) break;
            const int k = __ffs((int)over) - 1;
            over &= over - 1u;
            const int kk = k + (k >= 4 ? 1 : 0);                       // position in the 3 x 3 scan with the centre
            file(zx + kk % 3 - 1, zy + kk / 3 - 1, s);
        }
#endif
    };
    // what insert() takes as `over`: the mask itself, or its table entry
    auto phantoms_of = [&](const uint32_t code) -> uint32_t {
        const uint32_t mask = (code >> kCodeOverlapShift) & 0xFFu;
#if GPE_VAR_PHANTOM_TABLE
        return kPhantomTable.v[mask];
#else
        return mask;
#endif
    };
#ifdef GPE_DBG_SKIP
    if (!(GPE_DBG_SKIP & 64))                                          // diagnostic builds: phase cost by omission (results wrong)
#endif
    for (uint32_t s0 = 0; s0 < P; s0 += (uint32_t)QP * NT) {
        uint32_t pid[QP], blk[QP], cc[QP];
        float2 pp[QP];
        float pr[QP];
        if (listed) {                                                  // (scalar)
#pragma unroll
            for (int q = 0; q < QP; ++q) {
                const uint32_t sr = s0 + (uint32_t)tid + (uint32_t)q * NT;
                blk[q] = 0u;
                uint32_t v = first_ids[q];
                if (s0 != 0) v = A.roster_ids[roster_base + min(sr, P - 1u)];
                pid[q] = sr < P ? v : 0u;                              // (what lies behind the roster's end is not an id)
            }
        } else {
#pragma unroll
        for (int q = 0; q < QP; ++q) {                                 // branch-free, all loads in flight: see process_tile
            const uint32_t s = min(s0 + (uint32_t)tid + (uint32_t)q * NT, P - 1u);
            blk[q] = S.sblk[s];
            pid[q] = A.sorted_ids[S.bstart[blk[q]] + (s - S.boff[blk[q]])];
        }
        if (record) {
#pragma unroll
            for (int q = 0; q < QP; ++q) {
                const uint32_t sr = s0 + (uint32_t)tid + (uint32_t)q * NT;
                if (sr < P) A.roster_ids[roster_base + sr] = pid[q];
            }
        }
        }
#pragma unroll
        for (int q = 0; q < QP; ++q) {
            pp[q] = A.pos_in[pid[q]];
            pr[q] = A.radius[pid[q]];
            cc[q] = A.codes[pid[q]];
        }
        uint32_t lidq[QP];
#pragma unroll
        for (int q = 0; q < QP; ++q) lidq[q] = pid[q];
        if constexpr (ORD) {
#pragma unroll
            for (int q = 0; q < QP; ++q) pid[q] = A.order_keys[pid[q]];
        }
        uint32_t ph[QP];                                               // (issued here: in flight under the keep votes)
#pragma unroll
        for (int q = 0; q < QP; ++q) ph[q] = phantoms_of(cc[q]);
        int lxq[QP], lyq[QP];
        bool keep[QP];
        uint32_t slot[QP];
        uint64_t mq[QP];
        uint32_t cnt = 0;
#pragma unroll
        for (int q = 0; q < QP; ++q) {
            const uint32_t s = s0 + (uint32_t)tid + (uint32_t)q * NT;
            lxq[q] = code_window_x(cc[q], ox);
            lyq[q] = code_window_y(cc[q], oy);
            // (voted comparison by comparison: see pair_response)
            mq[q] = ballot64(s < P) & ballot64(lxq[q] < RWX) & ballot64(lyq[q] < RWY) & ballot64((cc[q] & straggler_bit) == 0u);
            // (an order-key window: the kept table -- or the roster written from it -- may still list indices the owned
            // range has shrunk below: those particles are ghosts now, or gone; they come through the ghost list, or not at all)
            if constexpr (ORD) mq[q] &= ballot64(lidq[q] < n_owned);
            keep[q] = lanes_of(mq[q]);
            cnt += (uint32_t)__popcll(mq[q]);
        }
        uint32_t base = 0;
        if (lane == 0 && cnt) base = atomicAdd(&S.misc[3], cnt);       // kept particles get consecutive slots
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
        for (int q = 0; q < QP; ++q) {
            slot[q] = base + popc_below_lane(mq[q]);
            base += (uint32_t)__popcll(mq[q]);
            keep[q] = keep[q] && slot[q] < (uint32_t)L::kSlots;          // over capacity: handed on below
        }
#pragma unroll
        for (int q = 0; q < QP; ++q)
            if (keep[q]) insert(slot[q], pp[q], pr[q], pid[q], lidq[q], lxq[q], lyq[q], ph[q]);
    }
    if (n_exc != 0 && tid < 64) {                                      // stragglers handed to the tile (see process_tile)
        const bool have = (uint32_t)tid < n_exc;
        const uint2 en = A.exc_entry[(uint64_t)pt * kExcSlots + (have ? (uint32_t)tid : 0u)];
        uint32_t pid = en.x;
        const float2 pp = A.pos_in[pid];
        const float pr = A.radius[pid];
        const uint32_t cc = A.codes[pid];
        const uint32_t lidq = pid;
        if constexpr (ORD) pid = A.order_keys[pid];
        const int lx = (int)(en.y & 0xFFFFu) - ox, ly = (int)(en.y >> 16) - oy;
        const uint64_t mk = ballot64(have) & ballot64((uint32_t)lx < (uint32_t)RWX) & ballot64((uint32_t)ly < (uint32_t)RWY);
        bool keep = lanes_of(mk);
        uint32_t base = 0;
        if (lane == 0 && mk) base = atomicAdd(&S.misc[3], (uint32_t)__popcll(mk));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        const uint32_t sl = base + popc_below_lane(mk);
        keep = keep && sl < (uint32_t)L::kSlots;
        if (keep) insert(sl, pp, pr, pid, lidq, lx, ly, phantoms_of(cc));
    }
    if constexpr (ORD) {
        // ghosts listed for the tile by the hash kernel (see process_tile)
        for (uint32_t g0 = 0; g0 < n_gho; g0 += (uint32_t)NT) {        // (scalar bounds: whole waves, the slots use ballots)
            const uint32_t gi = g0 + (uint32_t)tid;
            const bool have = gi < n_gho;
            const uint32_t lidq = A.gho_entry[(uint64_t)pt * kGhostSlots + (have ? gi : 0u)];
            const float2 pp = A.pos_in[lidq];
            const float pr = A.radius[lidq];
            const uint32_t cc = A.codes[lidq];
            const uint32_t pid = A.order_keys[lidq];
            const int lx = code_window_x(cc, ox), ly = code_window_y(cc, oy);
            const uint64_t mk = ballot64(have) & ballot64(lx < RWX) & ballot64(ly < RWY);
            bool keep = lanes_of(mk);
            uint32_t base = 0;
            if (lane == 0 && mk) base = atomicAdd(&S.misc[3], (uint32_t)__popcll(mk));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            const uint32_t sl = base + popc_below_lane(mk);
            keep = keep && sl < (uint32_t)L::kSlots;
            if (keep) insert(sl, pp, pr, pid, lidq, lx, ly, phantoms_of(cc));
        }
    }
    __syncthreads();
    GPE_STAMP(1);
    const uint32_t PS = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.misc[3]);
    if (PS > (uint32_t)L::kSlots) { GPE_BAIL(2); return false; }       // more kept particles than the window stages
    if (PS == 0) return true;
    const uint32_t n_big = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.misc[4]);
    if (n_big > (uint32_t)L::kBig) { GPE_BAIL(3); return false; }      // crowded cells: process_tile's windows take it

    // the tile's own particles and their previous positions: fetched here, used in P6 (as process_tile)
    constexpr int QOWN = (L::kSlots + NT - 1) / NT;
    uint32_t own_id[QOWN];
    float2 own_prev[QOWN];
    {
        uint32_t fetch[QOWN];
#pragma unroll
        for (int q = 0; q < QOWN; ++q) {
            const uint32_t s = (uint32_t)tid + (uint32_t)q * NT;
            own_id[q] = 0xFFFFFFFFu;
            fetch[q] = 0u;
            if (q >= 1 && PS <= (uint32_t)q * NT) continue;   // (scalar)
            const uint32_t sc = min(s, PS - 1u);
            const bool own = s < PS && S.own[sc] != 0;
            uint32_t id = S.id[sc];
            asm volatile("" : "+v"(id));                             // keep this an LDS read (no pointer select -> flat load)
            if constexpr (ORD) id = S.lid[sc];
            own_id[q] = own ? id : 0xFFFFFFFFu;
            fetch[q] = (own && id < n_owned) ? id : 0u;
        }
#pragma unroll
        for (int q = 0; q < QOWN; ++q) {
            own_prev[q] = make_float2(0.f, 0.f);
            if (q >= 1 && PS <= (uint32_t)q * NT) continue;
#ifdef GPE_DBG_SKIP
            if (GPE_DBG_SKIP & 32) continue;
#endif
            if (A.fuse_verlet) own_prev[q] = A.prev[fetch[q]];
        }
    }

    // ---- P4: active cells per colour (the walk of process_tile over the 2 x 2 colour groups; a cell's member count is
    //          its counter).  List entry: zone cell | class data << 12: cells of 2-3 members (one lane each; bit 12: three)
    //          from the front, cells of 4-6 (a lane group; members - 4) from the back; cells of 7-64 in the wave list.
#ifdef GPE_DBG_SKIP
    if (!(GPE_DBG_SKIP & 2))
#endif
    {
        constexpr int ZW = ZX / 2, ZH = ZY / 2, QC = ZW * ZH;
        static_assert(QC == QZ && NZ < 4096, "list entries: 12 bits of cell");
        for (int base = 0; base < QC; base += NT) {
            const int i = base + tid;
            const int gx2 = 2 * (i % ZW), gy2 = 2 * (i / ZW);          // zone coordinates of the group's first cell
            int zc[4];
            uint32_t cnt[4];
            static_assert(ZX % 2 == 0, "the two cells of a group's row share a counter word");
            {
                const int z0 = gy2 * ZX + gx2;                         // even
                const uint32_t w0 = (i < QC) ? S.cntw[z0 >> 1] : 0u, w1 = (i < QC) ? S.cntw[(z0 + ZX) >> 1] : 0u;
                zc[0] = z0; zc[1] = z0 + 1; zc[2] = z0 + ZX; zc[3] = z0 + ZX + 1;
                cnt[0] = w0 & 0xFFFFu; cnt[1] = w0 >> 16; cnt[2] = w1 & 0xFFFFu; cnt[3] = w1 >> 16;
            }
            // (the classes as lane masks, voted comparison by comparison: see pair_response)
            uint64_t ms[4], mg[4];
            bool single[4], group[4];
            const uint64_t in_m = ballot64(i < QC);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int lx = gx2 + (c & 1) + L::ZOX, ly = gy2 + (c >> 1) + L::ZOY;   // window coordinates
                const int gxx = ox + lx, gyy = oy + ly;
                const int exl = HX - lx, exr = lx - (HX + TX - 1), eyl = HY - ly, eyr = ly - (HY + TY - 1);
                const uint64_t zone_m = ballot64(exl <= kConeLeft - c) & ballot64(exr <= kConeRight - c) &
                                        ballot64(eyl <= kConeDown - (c >> 1)) & ballot64(eyr <= kConeUp - (c >> 1));
                // morton(-1,-1) == 0xFFFFFFFF == UNUSED_CELL_ID: never a collision cell (collision_cell_builder.wgsl:56).
                // Cell (-1, -1) lies in the zone of tile (0, 0) alone (the box has at most 65000 cells per axis, so no
                // other coordinate ends in 0xFFFF): a scalar test spares every other tile the comparisons.
                uint64_t alias_m = 0;
                if (tx == 0 && ty == 0) alias_m = ballot64((gxx & 0xFFFF) == 0xFFFF) & ballot64((gyy & 0xFFFF) == 0xFFFF);
                const uint64_t act_m = in_m & ballot64(cnt[c] >= 2u) & zone_m & ~alias_m;
                const uint64_t wave_m = act_m & ballot64(cnt[c] > (uint32_t)MS);
                mg[c] = act_m & ballot64(cnt[c] >= kGroupMin) & ~wave_m;
                ms[c] = act_m & ~mg[c] & ~wave_m;
                group[c] = lanes_of(mg[c]);
                single[c] = lanes_of(ms[c]);
                if (wave_m != 0) {                                     // (scalar: rare)
                    if (lanes_of(wave_m)) {
                        if (cnt[c] > 64u) { S.misc[2] = 1u; GPE_BAIL(4); }   // a pile: the sub-tile windows resolve those
                        const uint32_t k = atomicAdd(&S.lcnt[8 + c], 1u);
                        if (k < (uint32_t)L::WC) S.wlist[c * L::WC + k] = (uint16_t)zc[c]; else { S.misc[2] = 1u; GPE_BAIL(5); }
                    }
                }
            }
            // the eight list counters (colour x class) are bumped by eight lanes at once: the counts are scalars, each
            // written into its lane by one v_writelane
            int mine = 0;
            write_lane<0>(mine, (int)__popcll(ms[0])); write_lane<1>(mine, (int)__popcll(ms[1]));
            write_lane<2>(mine, (int)__popcll(ms[2])); write_lane<3>(mine, (int)__popcll(ms[3]));
            write_lane<4>(mine, (int)__popcll(mg[0])); write_lane<5>(mine, (int)__popcll(mg[1]));
            write_lane<6>(mine, (int)__popcll(mg[2])); write_lane<7>(mine, (int)__popcll(mg[3]));
            uint32_t mybase = 0;
            if (lane < 8 && mine) mybase = atomicAdd(&S.lcnt[lane], (uint32_t)mine);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t bs = (uint32_t)__builtin_amdgcn_readlane((int)mybase, c);
                const uint32_t bg = (uint32_t)__builtin_amdgcn_readlane((int)mybase, 4 + c);
                if (single[c]) S.list[c * QZ + bs + popc_below_lane(ms[c])] = (uint16_t)(zc[c] | ((cnt[c] == 3u) ? 0x1000 : 0));
                if (group[c]) S.list[c * QZ + (QZ - 1) - (bg + popc_below_lane(mg[c]))] = (uint16_t)(zc[c] | ((cnt[c] - kGroupMin) << 12));
            }
        }
    }
    __syncthreads();
    GPE_STAMP(4);
    if (S.misc[2]) return false;

    // ---- P5: the four colour passes (collision_solver.rs:224) ----------------------------------------------------
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
        const uint32_t ns = S.lcnt[k], group_lanes = S.lcnt[4 + k] * kGroupLanes;
        const uint32_t nw = min(S.lcnt[8 + k], (uint32_t)L::WC);
        const uint32_t single_base = (group_lanes + 63u) & ~63u;      // waves are all-group or all-single
        const uint32_t work = single_base + ns;
        for (uint32_t i0 = 0; i0 < work; i0 += NT) {
            const uint32_t i = i0 + (uint32_t)tid;
#ifdef GPE_DBG_SKIP
            if (GPE_DBG_SKIP & 1) continue;
            if ((GPE_DBG_SKIP & 8) && i < group_lanes) continue;
            if ((GPE_DBG_SKIP & 16) && i >= group_lanes) continue;
#endif
            if (i < group_lanes) {
                const uint32_t en = S.list[k * QZ + (QZ - 1) - (i / kGroupLanes)];
                resolve_group<MS>(S, (en & 0xFFFu) * MS, (en >> 12) + kGroupMin, (int)(i % kGroupLanes), A.stiffness);
            } else if (i >= single_base && (i & ~63u) < work) {         // (whole waves: the walk uses ballots)
                const bool on = i < work;
                uint32_t b = 0, n = 0;
                if (on) {
                    const uint32_t en = S.list[k * QZ + (i - single_base)];
                    b = (en & 0xFFFu) * MS; n = 2u + (en >> 12);
                }
                resolve_small_cells(S, on, b, n, A.stiffness);
            }
        }
        for (uint32_t i = (uint32_t)(NW - 1 - (tid >> 6)); i < nw; i += NW) {     // wave-uniform
            // a cell of 7..64 members: its first six from its own slots, the others from the side list, into the wave's
            // scratch run of the member array; then the whole-wave walk
            const uint32_t zc = S.wlist[k * L::WC + i];
            const uint32_t n = S.cnt_get((int)zc);
            const uint32_t wb = (uint32_t)(MS * NZ) + (uint32_t)(tid >> 6) * 64u;
            if (lane < MS) S.mem[wb + lane] = S.mem[zc * MS + lane];
            uint32_t filled = MS;
            for (uint32_t e0 = 0; e0 < n_big; e0 += 64u) {
                const uint32_t e = e0 + (uint32_t)lane;
                const uint32_t v = e < n_big ? S.big[e] : 0xFFFFFFFFu;
                const bool hit = e < n_big && (v >> 16) == zc;
                const uint64_t mh = ballot64(hit);
                if (hit) S.mem[wb + filled + popc_below_lane(mh)] = (uint16_t)(v & 0xFFFFu);
                filled += (uint32_t)__popcll(mh);
            }
            wave_lds_order();
            resolve_wave(S, wb, n, A.stiffness);                       // (filled == n: every membership is in one of the two)
            wave_lds_order();
        }
        __syncthreads();
        GPE_STAMP(9 + k);
    }

    // ---- P6: write the tile's own particles back, K12 applied (as process_tile) ---------------------------------------
#pragma unroll
    for (int q = 0; q < QOWN; ++q) {
#ifdef GPE_DBG_SKIP
        if (GPE_DBG_SKIP & 32) continue;
#endif
        if (q >= 1 && PS <= (uint32_t)q * NT) continue;                  // (scalar, as where own_id was filled)
        const uint32_t id = own_id[q];
        const bool have = id != 0xFFFFFFFFu;
        const uint32_t s = min((uint32_t)tid + (uint32_t)q * NT, (uint32_t)L::kSlots - 1u);
        const float2 c = make_float2(S.px[s], S.py[s]);
        const float rr = S.rad[s];
        float2 o = c;
        const bool mine = have && A.fuse_verlet && id < n_owned;
        if (mine) {
            verlet_one(c.x, c.y, own_prev[q].x, own_prev[q].y, rr, A.vp, o.x, o.y);
            A.prev[id] = c;
            A.pos_out[id] = o;
        } else if (have) {
            A.pos_out[id] = c;
        }
        if constexpr (ORD) {
            // a sharded step: the tiles pack their own particles for the neighbours (S.id holds the order key)
            if (A.pack.on) pack_if_near_border(A.pack, mine, id, o, c, rr, S.id[s], A.cell_size);
        }
    }
    __syncthreads();
    GPE_STAMP(6);
    return true;
}

#ifndef GPE_CAP_DIRECT
#define GPE_CAP_DIRECT 928
#endif
#ifndef GPE_CAP_DIRECT_ORD
#define GPE_CAP_DIRECT_ORD 880                 // order-key windows: 21 B per particle instead of 17, five member slots per cell
#endif
// 32 x 32-cell tiles on 512 threads, four workgroups per CU.  A tile the window has no room for is listed for
// k_collide_overflow.  (64 x 32-cell tiles on 1024 threads -- two per CU, the window 1.32 x the tile's own cells instead of
// 1.48 x -- were built, are exact and 5-6.5 % slower: profiles/r03/ab_wide_tiles_64x32_1024_threads_rejected.txt, the
// code in profiles/r04/wide_tiles_64x32_removed.patch.)
// Which tile does workgroup `wg` of the dense launch take?  Workgroups go round the eight XCDs (wg mod 8), and the tiles an
// XCD works through should be neighbours -- they share halo particles in that XCD's L2 -- so every XCD walks BANDS of
// A.band_tiles consecutive tiles (row-major: about two tile rows), band b belonging to XCD b mod 8, from the bottom of the
// box to its top.  (Rounds 1-3 gave each XCD one contiguous eighth of the rows.  A scene stratified in y -- anything
// under gravity: a crushed pile at the bottom, free fall above, nothing at the top -- then loads the XCDs so unevenly that
// the launch took twice the time its work amounts to: 6.4 ms for 3.3 ms of workgroup time at step 1250 of the 100 M
// soak, profiles/r04/tile_cycles_*.txt.)  Returns false when the workgroup has no tile.
__device__ __forceinline__ bool dense_launch_tile(const CollideArgs &A, const uint32_t wg, int *tx, int *ty)
{
    const uint32_t xcd = wg & 7u, j = wg >> 3;
    const uint32_t band = (j / A.band_tiles) * 8u + xcd;               // the j-th tile of this XCD lies in this band
    const uint32_t t = band * A.band_tiles + j % A.band_tiles;
    if (t >= (uint32_t)A.tiles_x * (uint32_t)A.tiles_y) return false;
    *tx = A.tile_x0 + (int)(t % (uint32_t)A.tiles_x);
    *ty = A.tile_y0 + (int)(t / (uint32_t)A.tiles_x);
    return true;
}
// The band size for a tile grid: bands a multiple of eight (every XCD the same number, so the busiest XCD has at most a
// band's remainder more than its share: 30 tile rows of 87 in whole rows give six XCDs 348 tiles and two 261 -- +6.7 % on
// the busiest against the 326 of an even deal; bands of 82 tiles give 328), about two tile rows each, at least four per XCD.
static uint32_t dense_launch_band(uint32_t tiles_x, uint32_t tiles_y, bool eighths)
{
    const uint64_t total = (uint64_t)tiles_x * tiles_y;
    const uint64_t per_xcd = eighths ? 1u : std::max<uint64_t>(4u, (total + 8u * tiles_x) / (16ull * tiles_x));
    return (uint32_t)std::max<uint64_t>(1u, (total + 8u * per_xcd - 1u) / (8u * per_xcd));
}
// ... and the grid that covers every tile
static uint32_t dense_launch_grid(uint32_t tiles_x, uint32_t tiles_y, uint32_t band_tiles)
{
    const uint64_t total = (uint64_t)tiles_x * tiles_y;
    const uint64_t bands = (total + band_tiles - 1) / band_tiles;
    return (uint32_t)(((bands + 7u) / 8u) * band_tiles * 8u);
}

// (three workgroups per CU: 80 VGPRs and 53 KB of LDS each -- room for 2000 particles, what a half of a tile at ~6 x the
// benchmark density keeps; an order-key window pays 4 more bytes per particle)
#ifndef GPE_CAP_HALF
#define GPE_CAP_HALF 2000
#endif
#ifndef GPE_CAP_HALF_ORD
#define GPE_CAP_HALF_ORD 1680
#endif
#ifndef GPE_CAP_HALF_FRONT
#define GPE_CAP_HALF_FRONT 1392                 // a half redone by the dense launch's front workgroups (kCtlHints): the tile's LDS
#endif
#ifndef GPE_CAP_HALF_FRONT_ORD
#define GPE_CAP_HALF_FRONT_ORD 1192             // ... of an order-key (sharded) run: 21 bytes per particle in 40 748
#endif
// Registers a tile that ran over for the front workgroups of the next step's dense launch (kCtlHints).
__device__ __forceinline__ void hint_tile(const CollideArgs &A, const int tx, const int ty, const uint32_t age = 0u)
{
    if (A.hints_on == 0u || !A.tb.holds(tx, ty) || (uint32_t)tx >= 2048u || (uint32_t)ty >= 2048u) return;
    const uint32_t next = A.hint_parity ^ 1u;
    const uint32_t k = atomicAdd(&A.tile_ctl[kCtlHints + next], 1u);
    if (k >= kHintMax) return;
    A.hints[next * kHintMax + k] = (age << 22) | ((uint32_t)ty << 11) | (uint32_t)tx;
    A.roster_hdr[A.tb.index(tx, ty)].w = A.step_stamp + 1u;
}

// HINTS: the launch carries front workgroups for the registered tiles (kCtlHints) -- a kernel of its own, launched only
// while such tiles exist: the plain kernel is 3 % faster without the code (48.0 against 49.5 us at 1 M).
template <int TX, int CAP, bool ORD, int NT, bool HINTS = false>
__global__ __launch_bounds__(NT, 8) void k_collide_direct(CollideArgs A)
{
    // (The half-tile form of the hinted tiles lives in the tile's own LDS: 40 928 bytes, four workgroups per CU, hold a
    // half of 1392 particles -- 59 % of a tile's window at 1.5 x its capacity; the half-tile launch's 2000 would make it
    // 51 KB and three per CU: 54.2 instead of 47.8 us at 1 M, profiles/r04/ab_hints_bisect.txt.)
    struct NoHalf { char unused; };
    using Half = typename std::conditional<HINTS, TileDirect<32, 16, ORD ? GPE_CAP_HALF_FRONT_ORD : GPE_CAP_HALF_FRONT, ORD, 512>, NoHalf>::type;
    __shared__ TileDirect<TX, 32, CAP, ORD, NT> S;
    static_assert(sizeof(Half) <= sizeof(S) && alignof(Half) <= alignof(TileDirect<TX, 32, CAP, ORD, NT>), "the half form fits the tile's LDS");
    uint32_t wg = blockIdx.x;
    if constexpr (HINTS) {
        if (wg < A.front_wgs) {
            // half (wg & 1) of hinted tile wg >> 1
            const uint32_t count = min(A.tile_ctl[kCtlHints + A.hint_parity], kHintMax);
            if ((wg >> 1) >= count) return;
            const uint32_t tile = A.hints[A.hint_parity * kHintMax + (wg >> 1)];
            const int htx = (int)(tile & 0x7FFu), hty = (int)((tile >> 11) & 0x7FFu);
            const uint32_t age = tile >> 22;
            if (!A.tb.holds(htx, hty)) return;
            const uint32_t hinted_for = A.roster_hdr[A.tb.index(htx, hty)].w;   // (not registered for this launch: its own workgroup takes it)
            if (hinted_for != A.step_stamp && hinted_for != A.step_stamp + 1u) return;
            const int hy = hty * 2 + (int)(wg & 1u);
            const bool done = process_tile_direct<ORD>(*reinterpret_cast<Half *>(&S), A, htx, hy);
            if (threadIdx.x == 0) {
                if (!done) {
                    const uint32_t slot = atomicAdd(&A.tile_ctl[kCtlOverflow2], 1u);
                    if (slot < 2u * A.overflow1_cap) A.overflow2[slot] = ((uint32_t)hy << 16) | (uint32_t)htx;
                    else atomicOr(&A.tile_ctl[kCtlError], kErrTileOverflow);
                }
                // again in the next launch, kHintAge launches long
                if ((wg & 1u) == 0u && age + 1u < kHintAge) hint_tile(A, htx, hty, age + 1u);
            }
            return;
        }
        wg -= A.front_wgs;                                             // (a multiple of 8: the XCD of a tile stays)
    }
    int tx, ty;
    if (!dense_launch_tile(A, wg, &tx, &ty)) return;
#ifdef GPE_TILE_CYCLES
    const long long tc0 = clock64();
#endif
    const bool done = process_tile_direct<ORD, TileDirect<TX, 32, CAP, ORD, NT>, HINTS>(S, A, tx, ty);
#ifdef GPE_TILE_CYCLES
    if (g_tile_cycles && threadIdx.x == 0 && A.tb.holds(tx, ty)) {
        uint4 *e = &g_tile_cycles[A.tb.index(tx, ty)];
        e->x = (uint32_t)(clock64() - tc0); e->y = done ? 0u : 1u; e->z = 0u; e->w = S.misc[0];
    }
#endif
    if (!done) {
        // (the launch that takes the tile off list 1 -- half tiles or over-capacity windows -- registers it: kCtlHints)
        if (threadIdx.x == 0) {
            const uint32_t slot = atomicAdd(&A.tile_ctl[kCtlOverflow1], 1u);
            if (slot < A.overflow1_cap) A.overflow1[slot] = ((uint32_t)ty << 16) | (uint32_t)tx;
            else atomicOr(&A.tile_ctl[kCtlError], kErrTileOverflow);
        }
    }
}

// The frame of a sharded rank's tile grid (CollideArgs::frame_*), one workgroup per tile: the tiles whose particles may
// have to be packed for the neighbours.  Launched BEFORE the interior tiles so that the exchange can start while those
// are resolved; everything such a tile needs happens in this launch -- a tile the direct-slot form has no room for is
// redone on the spot as four 16x16 quarters (the over-capacity launch would pack its particles after the segments left).
template <bool ORD>
__global__ __launch_bounds__(512, 6) void k_collide_border(CollideArgs A)
{
    __shared__ union { TileDirect<32, 32, ORD ? GPE_CAP_DIRECT_ORD : GPE_CAP_DIRECT, ORD, 512> tile; OverflowLds<ORD> windows; } u;
    // frame tile t: the bottom rows, the top rows, then the left and right columns of the rows between
    const uint32_t nx = (uint32_t)A.tiles_x, ny = (uint32_t)A.tiles_y;
    const uint32_t fb = (uint32_t)A.frame_b, ft = (uint32_t)A.frame_t, fl = (uint32_t)A.frame_l, fr = (uint32_t)A.frame_r;
    uint32_t t = blockIdx.x, x, y;
    if (t < fb * nx) { x = t % nx; y = t / nx; }
    else if (t < (fb + ft) * nx) { t -= fb * nx; x = t % nx; y = ny - ft + t / nx; }
    else {
        t -= (fb + ft) * nx;
        const uint32_t w = fl + fr;
        if (w == 0 || t / w >= ny - fb - ft) return;
        const uint32_t col = t % w;
        x = col < fl ? col : nx - fr + (col - fl);
        y = fb + t / w;
    }
    const int tx = A.tile_x0 + (int)x, ty = A.tile_y0 + (int)y;
    if (process_tile_direct<ORD>(u.tile, A, tx, ty)) return;
    __syncthreads();
    for (int q = 0; q < 4; ++q) resolve_quarter<ORD>(u.windows, A, tx * 2 + (q & 1), ty * 2 + (q >> 1));
}

// Level 0: one workgroup per 32x32 tile, tiles dealt so that each XCD (blockIdx % 8) works through a
// contiguous run of tile rows (neighbouring tiles share halo particles in that XCD's L2).
template <int T, int CAP, bool ORD>
__global__ __launch_bounds__(kNatThreads, 2048 / kNatThreads * 2) void k_collide_dense(CollideArgs A)
{
    __shared__ TileLds<T, CAP, ORD> S;
    int tx, ty;
    if (!dense_launch_tile(A, blockIdx.x, &tx, &ty)) return;
    // A tile whose window exceeds the capacity returns early and is listed for k_collide_overflow.  (Redoing it
    // here, quarter by quarter, on steps whose statistics let the host skip that launch: the extra code costs this
    // kernel 3 % at 1 M and 5 % at 100 M, more than the 4.5 us launch: profiles/r02/ab_inline_fallback_rejected.txt.)
    if (!process_tile<ORD>(S, A, tx, ty)) {
        if (threadIdx.x == 0) {
            const uint32_t slot = atomicAdd(&A.tile_ctl[kCtlOverflow1], 1u);
            if (slot < A.overflow1_cap) A.overflow1[slot] = ((uint32_t)ty << 16) | (uint32_t)tx;
            else atomicOr(&A.tile_ctl[kCtlError], kErrTileOverflow);
        }
    }
}

// The 32x32 tiles the dense launch handed on, each redone as two 32x16 halves in the direct-slot form (CollideArgs::
// overflow2): a ticketed grid like the over-capacity launch's, only launched while the host's lagged statistic reports
// such tiles.  A half that does not fit either is listed for the over-capacity launch's 16x16 / 8x8 windows.
// (Why: one tile a little over 928 particles used to cost the step a whole over-capacity launch -- a counting-sort 16x16
// window takes ~30 us whatever it holds, behind the dense launch: +40 % on the 1 M step once the undamped benchmark cloud
// has clumped, from step ~1000 on.  A half costs what a dense-launch tile costs, ~10 us.)
template <bool ORD>
__global__ __launch_bounds__(512, 6) void k_collide_halves(CollideArgs A)
{
    __shared__ TileDirect<32, 16, ORD ? GPE_CAP_HALF_ORD : GPE_CAP_HALF, ORD, 512> S;
    __shared__ uint32_t s_item;
    uint32_t count = A.tile_ctl[kCtlOverflow1];
    if (count > A.overflow1_cap) count = A.overflow1_cap;
    const uint32_t work = count * 2u;
    for (;;) {
        if (threadIdx.x == 0) s_item = atomicAdd(&A.tile_ctl[kCtlHalfTicket], 1u);
        __syncthreads();
        const uint32_t i = s_item;
        __syncthreads();
        if (i >= work) break;
        const uint32_t parent = A.overflow1[i >> 1];
        const int tx = (int)(parent & 0xFFFFu), ty = (int)((parent >> 16) * 2u + (i & 1u));
        if ((i & 1u) == 0u && threadIdx.x == 0) hint_tile(A, tx, (int)(parent >> 16));   // (for the next dense launch: kCtlHints)
        const bool done = process_tile_direct<ORD>(S, A, tx, ty);
        __syncthreads();
        if (!done && threadIdx.x == 0) {
            const uint32_t slot = atomicAdd(&A.tile_ctl[kCtlOverflow2], 1u);
            if (slot < 2u * A.overflow1_cap) A.overflow2[slot] = ((uint32_t)ty << 16) | (uint32_t)tx;
            else atomicOr(&A.tile_ctl[kCtlError], kErrTileOverflow);
        }
    }
}

// Over-capacity 32x32 tiles are redone by a fixed grid that takes work items off the device-side list (HIP has no
// indirect dispatch): one work item per 16x16 quarter.  One launch, nothing to wait for.
// (Eight waves per SIMD = 64 VGPRs: the kernel spills -- the item loop keeps the arguments of three inlined tile
// walks live -- and is still faster than with 85 or 128 VGPRs and three or two workgroups per CU: 12.1 against 13.1 /
// 15.2 ms in the compressed 100 M scene, profiles/r02/soak_1500_overflow_kernel_register_budget.txt.)
template <bool ORD>
__global__ __launch_bounds__(kNatThreads, GPE_OVF_WAVES) void k_collide_overflow(CollideArgs A)
{
    __shared__ OverflowLds<ORD> u;
    __shared__ uint32_t s_item;
    // work items: the four quarters of every tile of list 1 (unless the half-tile launch took that list), then the two
    // of every half of list 2 (CollideArgs)
    const uint32_t count1 = A.quarters_of_halves != 0u ? 0u : min(A.tile_ctl[kCtlOverflow1], A.overflow1_cap);
    const uint32_t count2 = min(A.tile_ctl[kCtlOverflow2], 2u * A.overflow1_cap);
    const uint32_t work1 = count1 * 4u, work = work1 + count2 * 2u;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // the hints this collide launch has used: counted for the statistics, the list free for the launch after the next
        const uint32_t used = min(A.tile_ctl[kCtlHints + A.hint_parity], kHintMax);
        A.tile_ctl[kCtlHintsSeen] = A.front_wgs != 0u ? used : 0u;
        A.tile_ctl[kCtlHints + A.hint_parity] = 0u;
    }
    if (work == 0) return;
    // Work items are taken from a ticket counter: their durations differ by orders of magnitude (in a compressed scene
    // the lower quarters of a tile hold several times the particles of the upper ones), and a fixed stride of 1024
    // gives a workgroup the same quarter of every tile it meets.
    for (;;) {
        if (threadIdx.x == 0) s_item = atomicAdd(&A.tile_ctl[kCtlOverflowTicket], 1u);
        __syncthreads();
        const uint32_t i = s_item;
        __syncthreads();
        if (i >= work) break;
        // the quarter, in 16-cell units, and its 32x32 tile
        uint32_t qx, qy;
        if (i >= work1) { const uint32_t h = A.overflow2[(i - work1) >> 1]; qx = (h & 0xFFFFu) * 2u + (i & 1u); qy = h >> 16; }
        else {
            const uint32_t parent = A.overflow1[i >> 2]; qx = (parent & 0xFFFFu) * 2u + (i & 1u); qy = (parent >> 16) * 2u + ((i >> 1) & 1u);
            if ((i & 3u) == 0u && threadIdx.x == 0) hint_tile(A, (int)(parent & 0xFFFFu), (int)(parent >> 16));   // (kCtlHints)
        }
#ifdef GPE_TILE_CYCLES
        const long long tq0 = clock64();
#endif
        resolve_quarter<ORD>(u, A, (int)qx, (int)qy);
#ifdef GPE_TILE_CYCLES
        if (g_tile_cycles && threadIdx.x == 0 && A.tb.holds((int)(qx >> 1), (int)(qy >> 1)))
            atomicAdd(&g_tile_cycles[A.tb.index((int)(qx >> 1), (int)(qy >> 1))].z, (uint32_t)(clock64() - tq0));
#endif
    }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
static uint32_t host_split(uint32_t n)
{
    uint32_t x = n & 0x0000FFFFu;
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}

void native_release(gpe_ctx *c)
{
    NativeState &N = c->native;
    if (N.block_table) (void)hipFree(N.block_table);
    if (N.keys) (void)hipFree(N.keys);
    if (N.codes) (void)hipFree(N.codes);
    if (N.sorted_key) (void)hipFree(N.sorted_key);
    if (N.exc_count) (void)hipFree(N.exc_count);
    if (N.gho_count) (void)hipFree(N.gho_count);
    if (N.roster_hdr) (void)hipFree(N.roster_hdr);
    if (N.roster_ids) (void)hipFree(N.roster_ids);
    if (N.gkeys) (void)hipFree(N.gkeys);
    if (N.gids) (void)hipFree(N.gids);
    if (N.gkeys_b) (void)hipFree(N.gkeys_b);
    if (N.gids_b) (void)hipFree(N.gids_b);
    if (N.gtable) (void)hipFree(N.gtable);
    if (N.ghist) (void)hipFree(N.ghist);
    if (N.ids) (void)hipFree(N.ids);
    if (N.keys_b) (void)hipFree(N.keys_b);
    if (N.ids_b) (void)hipFree(N.ids_b);
    if (N.tile_ctl) (void)hipFree(N.tile_ctl);
    if (N.overflow1) (void)hipFree(N.overflow1);
    if (N.arena) (void)hipFree(N.arena);
    if (N.host_stat) (void)hipHostFree(N.host_stat);
    N = NativeState();
}

static gpe_status arena_reserve(gpe_ctx *c, uint64_t want);

// hash -> [sort -> block table].  *sorted_ids receives the particle ids grouped by 8x8-cell block.
// The radix passes are enqueued every step but run only when the hash kernel finds a particle that has left the
// reach of the grouping they last produced (kDrift* cells beyond its block): the sorted ids and the block table are
// kept across steps, the per-particle codes carry the particle's cell (mod 128) and whether it is still within reach of
// its old block, and the tiles look up more blocks than they keep particles from.  Decided on the device, step by step; the host waits for nothing.
// In the benchmark cloud (gravity off) a sort is needed every few dozen steps, in free fall every 5-25 steps.
// always_sort: this call must not rely on the kept grouping (configuration-time probes).
static gpe_status native_prepare_step(gpe_ctx *c, uint32_t **sorted_ids, bool always_sort = false)
{
    NativeState &N = c->native;
    const uint64_t n = c->n;
    constexpr size_t kHistSet = (size_t)kHistCopies * 4 * 256;          // words of one set of digit histograms
    if (!c->os_ws.hist_clean) {                                        // another sort used the histograms since
        GPE_HIP(c, hipMemsetAsync(c->os_ws.hist4, 0, 2 * kHistSet * sizeof(uint32_t), c->stream));
        c->os_ws.hist_clean = true;
        c->os_ws.hist_set = 0;
    }
    uint32_t *hist_now = c->os_ws.hist4 + (size_t)c->os_ws.hist_set * kHistSet;
    uint32_t *hist_next = c->os_ws.hist4 + (size_t)(c->os_ws.hist_set ^ 1u) * kHistSet;
    c->os_ws.hist_set ^= 1u;
    const uint32_t parity = (N.step_seq++) & 1u;
    // The kept grouping can be used when it belongs to these particles and this box.  Sharded runs (ghosts come and go
    // every step, padding keys) and one-pass sorts (the table would be reset and filled by the same launch) always sort.
    // A sharded run with its counts on the device (k_shard.hip) keeps the grouping of its OWNED particles the same way:
    // their indices are stable (a hole left by a migrant is filled from the tail, an arrival is appended: both reach
    // the tiles as stragglers until the next sort), while the ghosts -- new every step -- are grouped by a small sort
    // of their own, every step, into a second block table.  Other sharded set-ups (host-side counts) always sort.
    // A scene in which nearly every step sorts anyway (a crushed pile: a third of the particles move further than the
    // kept table reaches, every step) gains nothing from the kept table and pays for it in the hash kernel (the old
    // keys, the drift test: 1.2 instead of 0.6 ms at 100 M).  The passes' own counter says so (lagged): when three
    // quarters of the last 64 steps sorted, the next 256 steps sort unconditionally; then the table gets another try.
    // (While the hold lasts every step sorts by decree, which says nothing about the scene: when it ends the count starts
    // afresh -- one window for the lagged counter to settle, one to judge -- so a scene that has calmed down keeps its table.)
    if (N.host_stat && !always_sort) {
        if (N.sort_hold > 0) {
            if (--N.sort_hold == 0) { N.watch_steps = 0; N.watch_valid = false; }
        } else if (++N.watch_steps >= 64) {
            const uint32_t sorts = N.host_stat[kStatSorts];
            if (N.watch_valid && sorts - N.watch_sorts >= 48) N.sort_hold = 256;
            N.watch_sorts = sorts; N.watch_steps = 0; N.watch_valid = true;
        }
    }
    const bool sharded = c->shard.on || c->use_order_keys || c->has_active_box;
    const bool kept_sharded = c->shard.on && c->shard.active && c->use_order_keys && N.gkeys != nullptr && !N.always_sort;
    const bool gated = N.passes >= 2 && (!sharded || kept_sharded);
    const bool reuse = gated && !always_sort && N.sort_state_valid && (kept_sharded || N.sorted_n == n) && !N.always_sort &&
                       N.sort_hold == 0 && N.exc_count != nullptr;     // (no room for the straggler lists: sort every step)
    const uint64_t pairs = ((uint64_t)N.table_entries + 1) / 2;        // the table is allocated in 16-byte units
    // Who counts the radix digits of a step that keeps its table (k_native_hash, fuse_hist): the hash kernel, fused, while
    // at least a quarter of the last 16 steps sorted (the passes' own counter, lagged: a falling or crushed cloud before
    // sort_hold takes over -- the separate launch reads all keys again, 0.2-0.4 ms at 100 M when it runs, against the
    // 0.075 ms per step the fused count costs the hash);
    // otherwise the gated launch, which returns at once on the steps that do not sort.
    if (N.host_stat && ++N.hist_watch_steps >= 16) {
        const uint32_t sorts = N.host_stat[kStatSorts];
        N.hist_fused = (sorts - N.hist_watch_sorts) * 4u >= N.hist_watch_steps;
        N.hist_watch_sorts = sorts; N.hist_watch_steps = 0;
    }
    const bool fuse_hist = N.hist_fused || (c->cfg.flags & GPE_FLAG_FUSED_HISTOGRAMS) != 0;
    HashGhosts hg;
    hg.sorted_count = N.tile_ctl + kCtlSortedCount;
    uint64_t g_bound = 0;
    if (kept_sharded) {
        // upper bound of the ghost count from the pinned mirror (lags by the steps in flight), as for the total
        const ShardState &SH = c->shard;
        g_bound = c->cap;
        const uint32_t epoch = __atomic_load_n(&SH.host_counts[kShardEpoch], __ATOMIC_ACQUIRE);
        if ((int32_t)(epoch - SH.begin_epoch) > 0) {
            const uint64_t gh = SH.host_counts[kShardTotal] >= SH.host_counts[kShardOwned]
                                    ? SH.host_counts[kShardTotal] - SH.host_counts[kShardOwned] : 0;
            g_bound = std::min<uint64_t>(c->cap, gh + std::max<uint64_t>(16384, gh / 4));
        }
        g_bound = std::min<uint64_t>(g_bound, n);
        hg.owned = SH.counts_now() + kShardOwned;
        hg.gkeys = N.gkeys; hg.gids = N.gids; hg.g_bound = g_bound;
        hg.gtable2 = (uint4 *)N.gtable; hg.gtable_pairs = pairs;
        hg.ghist_now = N.ghist + (size_t)N.ghist_set * kHistSet;       // (two sets, alternating over the steps that use them:
        hg.ghist_next = N.ghist + (size_t)(N.ghist_set ^ 1u) * kHistSet;   //  this step's hash zeroes the next one's)
        N.ghist_set ^= 1u;
        hg.ghost_sort = N.tile_ctl + kCtlGhostSort + parity;
        hg.ghost_sort_next = N.tile_ctl + kCtlGhostSort + (parity ^ 1u);
        if (N.gho_count && N.gho_cap >= N.exc_tiles) {                 // ghost lists: two sets by step parity, like the stragglers'
            hg.gl_count = N.gho_count + (size_t)parity * N.exc_tiles;
            hg.gl_count_next = N.gho_count + (size_t)(parity ^ 1u) * N.exc_tiles;
            hg.gl_entry = N.gho_count + 2 * N.exc_tiles + (size_t)parity * N.exc_tiles * kGhostSlots;
        }
    }
    {
        Scope s(c, "native/hash");
        // at least 4 keys per lane (measured: profiles/r01/tune_hash.txt)
        // (two particles per thread until the grid is full: 14.0 against 14.3 us at 1 M with four, 15.2 with one --
        // the kernel is launch and latency there; from 4 M particles on the grid is kHashGridMax either way)
        const int grid = (int)std::min<uint64_t>(kHashGridMax, std::max<uint64_t>(1, n / (2ull * kHashBlock)));
        const uint32_t *n_valid = (c->shard.on && c->shard.active) ? c->shard.counts_now() + kShardTotal : nullptr;
        const uint64_t div_magic = ((1ull << 40) + (uint64_t)N.blocks_x - 1) / (uint64_t)N.blocks_x;
        const auto hash_kernel = kept_sharded ? k_native_hash<true> : k_native_hash<false>;
        hipLaunchKernelGGL(hash_kernel, dim3(grid), dim3(kHashBlock), 0, c->stream, c->pos, c->radius, n, n_valid,
                           c->cell_size, N.gx, N.gy, N.bx0, N.by0, N.blocks_x, N.blocks_y, N.table_entries, N.keys,
                           N.codes, N.passes, hist_now, hist_next, c->os_ws.ctl, N.tile_ctl,
                           (uint4 *)N.block_table, gated ? 0ull : pairs,    // gated: the first radix pass resets the table
                           N.host_stat, reuse ? N.sorted_key : nullptr, parity, div_magic,
                           N.exc_count ? N.exc_count + (size_t)parity * N.exc_tiles : nullptr,
                           N.exc_count ? N.exc_entry + (size_t)parity * N.exc_tiles * kExcSlots : nullptr,
                           N.exc_count ? N.exc_count + (size_t)(parity ^ 1u) * N.exc_tiles : nullptr, N.tb,
                           (uint32_t)std::max<uint64_t>(64, n >> 11),       // more stragglers than 0.05 % of the particles: sort
                           fuse_hist ? 1u : 0u, hg);
        GPE_HIP(c, hipGetLastError());
    }
    uint32_t *sk = nullptr, *sv = nullptr;
    {
        // the last radix pass also fills the block table (first / one-past-last position of every block, by
        // atomic min / max at the ends of each tile's key runs); every pass derives its digit bases from hist_now
        Scope s(c, "native/sort");
        OnesweepGate g;
        g.need = N.tile_ctl + kCtlNeedSort + parity;
        if (reuse && !fuse_hist) {
            // (the hash kernel counted nothing: see fuse_hist there)
            const int hgrid = (int)std::min<uint64_t>(kHistGatedGridMax, std::max<uint64_t>(1, n / (4ull * kHistGatedBlock)));
            hipLaunchKernelGGL(k_native_hist_gated, dim3(hgrid), dim3(kHistGatedBlock), 0, c->stream, N.keys, n, N.passes, hist_now, g.need);
            GPE_HIP(c, hipGetLastError());
        }
        g.fresh = N.tile_ctl + kCtlFresh + parity;
        g.sorts = N.tile_ctl + kCtlSorts;
        g.sorts_seen = N.tile_ctl + kCtlSortsSeen;
        if (gated) {
            g.key_copy = N.sorted_key; g.table_reset = (uint4 *)N.block_table; g.table_pairs = pairs;
            g.key_blocks_x = (uint32_t)N.blocks_x;
            g.key_div_magic = ((1ull << 40) + (uint64_t)N.blocks_x - 1) / (uint64_t)N.blocks_x;
            g.count_now = hg.owned; g.sorted_count = N.tile_ctl + kCtlSortedCount;
        }
        GPE_TRY(onesweep_sort(c, N.keys, N.ids, N.keys_b, N.ids_b, n, N.passes, true, true, &sk, &sv, true,
                              N.block_table, N.table_entries, hist_now, &g));
    }
    (void)sk;
    N.gsorted_ids_now = nullptr;
    if (kept_sharded && g_bound > 0) {
        // the ghosts' own grouping: (block key, particle index) pairs written by the hash, sorted every step; the last
        // pass fills the ghosts' block table
        Scope s(c, "shard/ghost-sort");
        uint32_t *gk = nullptr, *gv = nullptr;
        OnesweepGate gg;                                               // (its own tile tickets; runs when a ghost list ran over,
        gg.ticket_base = 8;                                            //  or always when there are no lists)
        gg.need = hg.ghost_sort;
        GPE_TRY(onesweep_sort(c, N.gkeys, N.gids, N.gkeys_b, N.gids_b, g_bound, N.passes, true, false, &gk, &gv, true,
                              N.gtable, N.table_entries, hg.ghist_now, &gg));
        (void)gk;
        N.gsorted_ids_now = gv;
    }
    N.sort_state_valid = gated;           // (the passes of this call ran, or the kept state was and stays valid)
    N.sorted_n = n;
    N.fresh_word = N.tile_ctl + kCtlFresh + parity;
    N.gho_count_now = hg.gl_count;
    N.gho_entry_now = hg.gl_entry;
    N.ghost_sort_now = hg.ghost_sort;
    N.exc_count_now = reuse ? N.exc_count + (size_t)parity * N.exc_tiles : nullptr;
    N.exc_entry_now = reuse ? N.exc_entry + (size_t)parity * N.exc_tiles * kExcSlots : nullptr;
    *sorted_ids = sv;
    return GPE_OK;
}

// (Re)derive the cell box from the world and the cell size, size the workspaces, and check on the
// device that (a) every particle lies inside the box and (b) no 24x24-cell window holds more particles
// than the smallest LDS cell window stages.  Called from the configuration entry points (set/add
// particles, set world, set max radius, set mode) -- never on the step path; synchronises.
gpe_status native_configure(gpe_ctx *c)
{
    NativeState &N = c->native;
    N.eligible = false;
    N.in_box = false;
    N.dense_hold = false;
    N.steps_since_check = 0;
    N.sort_state_valid = false;          // particles, box or keys changed: the kept grouping is of something else
    N.quiet_steps = 0;
    N.crowded = false;
    N.hist_fused = false; N.hist_watch_steps = 0; N.hist_watch_sorts = N.host_stat ? N.host_stat[kStatSorts] : 0u;
    N.sort_hold = 0; N.watch_steps = 0; N.watch_valid = false;
    N.always_sort = (c->cfg.flags & GPE_FLAG_SORT_EVERY_STEP) != 0;
    N.reason = GPE_REASON_NO_PARTICLES;
    if (c->n == 0 || !(c->cell_size > 0.0f)) return GPE_OK;
    // largest home coordinate a clamped particle can take: floor(world / cell_size)
    // (K12 clamps to [r, world - r], particle_integration.wgsl:70-71)
    const float fx = floorf(c->cfg.world_width / c->cell_size), fy = floorf(c->cfg.world_height / c->cell_size);
    N.reason = GPE_REASON_GRID_TOO_WIDE;
    if (!(fx >= 0.0f) || !(fy >= 0.0f) || fx > 65000.0f || fy > 65000.0f) return GPE_OK;   // 16-bit cell coords
    N.gx = (int32_t)fx + 1;
    N.gy = (int32_t)fy + 1;
    // The sort key is the particle's 8x8-cell BLOCK, row-major over the box: the tiles look particles up per
    // block and order the members of a cell themselves, so the order inside a block is free.  Against the
    // Morton id of the home cell (what the reference sorts by) that is 6 bits less plus the padding Morton
    // interleaving adds to a non-square box: one radix pass less at 1 M (2 instead of 3) and at 100 M (3 / 4).
    N.bx0 = N.by0 = 0;
    N.blocks_x = (N.gx + 7) >> 3;
    N.blocks_y = (N.gy + 7) >> 3;
    if (c->has_active_box) {
        // sharded: the block box is this rank's active box (own blocks + ghost ring), so the keys stay as short
        // as a single-device run of the same size has them
        const int32_t b0x = std::max(0, c->active_box[0] >> 3), b0y = std::max(0, c->active_box[1] >> 3);
        const int32_t b1x = std::min(N.blocks_x - 1, c->active_box[2] >> 3), b1y = std::min(N.blocks_y - 1, c->active_box[3] >> 3);
        if (b1x >= b0x && b1y >= b0y) {
            N.bx0 = b0x; N.by0 = b0y;
            N.blocks_x = b1x - b0x + 1; N.blocks_y = b1y - b0y + 1;
        }
    }
    N.table_entries = (uint32_t)N.blocks_x * (uint32_t)N.blocks_y;
    int bits = 0;
    // key == table_entries is the padding key of a sharded run (k_native_hash): it needs its bits too
    const uint32_t max_key = c->shard.on ? N.table_entries : N.table_entries - 1;
    while (bits < 32 && (max_key >> bits) != 0) ++bits;
    N.passes = (bits + 7) / 8;
    if (N.passes < 1) N.passes = 1;
    N.reason = GPE_REASON_TABLE_TOO_LARGE;
    if (N.table_entries > (1u << 27)) return GPE_OK;                   // > 1 GiB of table: stay on compat
    if (N.table_cap < N.table_entries) {
        if (N.block_table) GPE_HIP(c, hipFree(N.block_table));
        N.block_table = nullptr; N.table_cap = 0;
        GPE_HIP(c, hipMalloc((void **)&N.block_table, ((size_t)N.table_entries + 2) * sizeof(uint2)));
        N.table_cap = N.table_entries;
    }
    if (N.cap < c->cap) {
        uint32_t **bufs[4] = {&N.keys, &N.ids, &N.keys_b, &N.ids_b};
        for (uint32_t **b : bufs) {
            if (*b) GPE_HIP(c, hipFree(*b));
            *b = nullptr;
            GPE_HIP(c, hipMalloc((void **)b, (c->cap + 16) * sizeof(uint32_t)));
        }
        if (N.codes) GPE_HIP(c, hipFree(N.codes));
        N.codes = nullptr;
        GPE_HIP(c, hipMalloc((void **)&N.codes, (c->cap + 16) * sizeof(uint32_t)));
        if (N.sorted_key) GPE_HIP(c, hipFree(N.sorted_key));
        N.sorted_key = nullptr;
        GPE_HIP(c, hipMalloc((void **)&N.sorted_key, (c->cap + 16) * sizeof(uint32_t)));
        N.cap = c->cap;
        N.gcap = 0;                                                    // (the ghost buffers follow below)
    }
    if (c->shard.on && (N.gcap < c->cap || N.gtable_cap < N.table_entries)) {
        // sharded runs: the ghosts' sort buffers and block table
        uint32_t **gb[4] = {&N.gkeys, &N.gids, &N.gkeys_b, &N.gids_b};
        for (uint32_t **b : gb) {
            if (*b) GPE_HIP(c, hipFree(*b));
            *b = nullptr;
            GPE_HIP(c, hipMalloc((void **)b, (c->cap + 16) * sizeof(uint32_t)));
        }
        N.gcap = c->cap;
        if (N.gtable) GPE_HIP(c, hipFree(N.gtable));
        N.gtable = nullptr;
        GPE_HIP(c, hipMalloc((void **)&N.gtable, ((size_t)N.table_entries + 2) * sizeof(uint2)));
        N.gtable_cap = N.table_entries;
        if (!N.ghist) {
            GPE_HIP(c, hipMalloc((void **)&N.ghist, 2 * (size_t)kHistCopies * 4 * 256 * sizeof(uint32_t)));
            GPE_HIP(c, hipMemsetAsync(N.ghist, 0, 2 * (size_t)kHistCopies * 4 * 256 * sizeof(uint32_t), c->stream));
        }
    }
    const uint64_t tiles = (uint64_t)((N.gx + kTileMain - 1) / kTileMain) * ((N.gy + kTileMain - 1) / kTileMain);
    {
        // straggler lists: per 32x32 tile of the cell box a count and kExcSlots entries, two sets (step parity)
        // the tile box: the whole cell box, or a sharded rank's active box
        N.tb.x0 = 0; N.tb.y0 = 0; N.tb.nx = (N.gx + 31) / 32; N.tb.ny = (N.gy + 31) / 32;
        if (c->has_active_box) {
            const int32_t cx0 = std::max(0, c->active_box[0]), cy0 = std::max(0, c->active_box[1]);
            const int32_t cx1 = std::min(N.gx - 1, c->active_box[2]), cy1 = std::min(N.gy - 1, c->active_box[3]);
            if (cx1 >= cx0 && cy1 >= cy0) {
                N.tb.x0 = cx0 / 32; N.tb.y0 = cy0 / 32;
                N.tb.nx = cx1 / 32 - N.tb.x0 + 1; N.tb.ny = cy1 / 32 - N.tb.y0 + 1;
            }
        }
        N.exc_tiles = (uint64_t)N.tb.nx * (uint64_t)N.tb.ny;
        if (N.exc_cap < N.exc_tiles) {
            if (N.exc_count) GPE_HIP(c, hipFree(N.exc_count));
            N.exc_count = nullptr; N.exc_entry = nullptr; N.exc_cap = 0;
            // (264 B per tile and set: a sparse scene in a huge world -- up to 8 M tiles -- may not get them; the run
            // then sorts every step, native_prepare_step, instead of failing to configure)
            const size_t bytes = 2 * N.exc_tiles * sizeof(uint32_t) + 16 + 2 * N.exc_tiles * kExcSlots * sizeof(uint2);
            if (hipMalloc((void **)&N.exc_count, bytes) == hipSuccess) N.exc_cap = N.exc_tiles;
            else { (void)hipGetLastError(); N.exc_count = nullptr; }
        }
        if (N.exc_count) {
            // (entries behind the counts of both sets, 8-byte aligned)
            N.exc_entry = (uint2 *)(N.exc_count + ((2 * N.exc_tiles + 1) & ~1ull));
            GPE_HIP(c, hipMemsetAsync(N.exc_count, 0, 2 * N.exc_tiles * sizeof(uint32_t), c->stream));
        }
    }
    // Rosters scale with the world's tile count, not with n (6160 B per tile): at the benchmark density that is 16 B per
    // particle; a sparse scene in a large world would pay gigabytes for lists of a few ids each.  Beyond 16 roster
    // slots per particle (4 x the benchmark's ratio) the run does without them.
    const bool rosters_pay = N.exc_tiles * (uint64_t)kRosterCap <= 16ull * std::max<uint64_t>(c->n, 1u << 16);
    if (!rosters_pay && N.roster_hdr) {
        GPE_HIP(c, hipFree(N.roster_hdr)); GPE_HIP(c, hipFree(N.roster_ids));
        N.roster_hdr = nullptr; N.roster_ids = nullptr; N.roster_cap = 0;
    }
    if (c->shard.on && c->has_active_box && N.gho_cap < N.exc_tiles) {
        // ghost lists (sharded runs): a count and kGhostSlots ids per tile, two sets.  Optional: without them the ghosts
        // are sorted into their block table every step
        if (N.gho_count) GPE_HIP(c, hipFree(N.gho_count));
        N.gho_count = nullptr; N.gho_cap = 0;
        if (hipMalloc((void **)&N.gho_count, 2 * N.exc_tiles * (1 + (size_t)kGhostSlots) * sizeof(uint32_t) + 64) == hipSuccess)
            N.gho_cap = N.exc_tiles;
        else { (void)hipGetLastError(); N.gho_count = nullptr; }
    }
    if (N.gho_count) GPE_HIP(c, hipMemsetAsync(N.gho_count, 0, 2 * N.exc_tiles * sizeof(uint32_t), c->stream));
    if (rosters_pay && N.exc_count &&
        (c->cfg.flags & (GPE_FLAG_SORT_EVERY_STEP | GPE_FLAG_COUNTING_SORT_TILES)) == 0) {
        // tile rosters (CollideArgs): 16 + 4 kRosterCap bytes per 32x32 tile.  Optional: a device that has no room for
        // them runs without (every step then looks its blocks up)
        if (N.roster_cap < N.exc_tiles) {
            if (N.roster_hdr) GPE_HIP(c, hipFree(N.roster_hdr));
            if (N.roster_ids) GPE_HIP(c, hipFree(N.roster_ids));
            N.roster_hdr = nullptr; N.roster_ids = nullptr; N.roster_cap = 0;
            hipError_t e1 = hipMalloc((void **)&N.roster_hdr, N.exc_tiles * sizeof(uint4));
            hipError_t e2 = e1 == hipSuccess ? hipMalloc((void **)&N.roster_ids, N.exc_tiles * (size_t)kRosterCap * sizeof(uint32_t)) : e1;
            if (e1 != hipSuccess || e2 != hipSuccess) {
                (void)hipGetLastError();
                if (N.roster_hdr) (void)hipFree(N.roster_hdr);
                N.roster_hdr = nullptr; N.roster_ids = nullptr;
            } else N.roster_cap = N.exc_tiles;
        }
        // (stamp 0 is never current: the tiles compare with sorts + 1)
        if (N.roster_hdr) GPE_HIP(c, hipMemsetAsync(N.roster_hdr, 0, N.exc_tiles * sizeof(uint4), c->stream));
    }
    if (N.overflow_cap < tiles) {
        if (N.overflow1) GPE_HIP(c, hipFree(N.overflow1));
        N.overflow1 = nullptr; N.overflow_cap = 0;
        GPE_HIP(c, hipMalloc((void **)&N.overflow1, (3 * tiles + 32 + 2 * kHintMax) * sizeof(uint32_t)));   // (the tiles, then their halves: CollideArgs::overflow2, then the hints)
        N.overflow_cap = tiles;
    }
    {
        // spill arena: every particle can be staged by the 9 windows around it, but a scene that dense has
        // left the native path long before (native_should_run); one slot per particle, 1 M .. 32 M slots
        const uint64_t want = std::min<uint64_t>(std::max<uint64_t>(c->cap, 1ull << 20), 32ull << 20);
        GPE_TRY(arena_reserve(c, std::max<uint64_t>(want, N.arena_cap)));
    }
    if (!N.tile_ctl) {
        GPE_HIP(c, hipMalloc((void **)&N.tile_ctl, kCtlWords * sizeof(uint32_t)));
        GPE_HIP(c, hipMemsetAsync(N.tile_ctl, 0, kCtlWords * sizeof(uint32_t), c->stream));
    }
    if (!N.host_stat) GPE_HIP(c, hipHostMalloc((void **)&N.host_stat, 64, hipHostMallocDefault));
    memset(N.host_stat, 0, 64);
    GPE_TRY(onesweep_reserve(c, c->cap));
    GPE_HIP(c, hipMemsetAsync(N.tile_ctl, 0, kCtlSorts * sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(k_native_check_box, dim3(stream_grid(c->n)), dim3(kStreamBlock), 0, c->stream, c->pos, c->n,
                       (c->shard.on && c->shard.active) ? c->shard.counts_now() + kShardTotal : nullptr, c->cell_size, N.gx, N.gy,
                       N.tile_ctl + kCtlError);
    GPE_HIP(c, hipGetLastError());
    uint32_t flag = 1;
    GPE_HIP(c, hipMemcpyAsync(&flag, N.tile_ctl + kCtlError, sizeof(flag), hipMemcpyDeviceToHost, c->stream));
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    GPE_HIP(c, hipMemsetAsync(N.tile_ctl, 0, kCtlSorts * sizeof(uint32_t), c->stream));         // the check's verdict is not a step error
    N.reason = GPE_REASON_OUT_OF_BOX;
    if (flag != 0) return GPE_OK;                                      // a particle outside the box: compat kernels
    N.in_box = true;
    // window population of the current state
    const bool prof = c->profiling;
    c->profiling = false;
    uint32_t *ids = nullptr;
    gpe_status st = native_prepare_step(c, &ids, true);
    c->profiling = prof;
    GPE_TRY(st);
    hipLaunchKernelGGL(k_native_window_max, dim3(stream_grid(N.table_entries)), dim3(kStreamBlock), 0, c->stream,
                       N.block_table, N.table_entries, N.blocks_x, N.blocks_y, N.tile_ctl + kCtlWindowMax);
    GPE_HIP(c, hipGetLastError());
    uint32_t wmax = 0xffffffffu;
    GPE_HIP(c, hipMemcpyAsync(&wmax, N.tile_ctl + kCtlWindowMax, sizeof(wmax), hipMemcpyDeviceToHost, c->stream));
    GPE_HIP(c, hipStreamSynchronize(c->stream));
    GPE_HIP(c, hipMemsetAsync(N.tile_ctl, 0, kCtlSorts * sizeof(uint32_t), c->stream));
    N.window_max = wmax;
    N.eligible = wmax <= kWindowEligible;
    N.reason = N.eligible ? GPE_REASON_NONE : GPE_REASON_DENSE_WINDOWS;
    // gpe_config.flags: keep over-dense scenes on the native kernels (their windows then go through the spill arena);
    // print the step statistics every 128 steps; sort every step
    N.force = (c->cfg.flags & GPE_FLAG_NATIVE_FORCE) != 0;
    N.print_stats = (c->cfg.flags & GPE_FLAG_NATIVE_STATS) != 0;
    if (N.force) { N.eligible = true; N.reason = GPE_REASON_NONE; }
    return GPE_OK;
}

// The spill arena: every particle of a dense region can be staged by the 9 windows around it.  One slot per
// particle to start with (1 M .. 32 M slots); native_should_run doubles it when a step used more than half.
static gpe_status arena_reserve(gpe_ctx *c, uint64_t want)
{
    NativeState &N = c->native;
    if (N.arena_cap >= want) return GPE_OK;
    // the new arena first: on failure the old one stays in place (a run that must stay on the native kernels keeps
    // working with it) and the error is the caller's to report
    void *fresh = nullptr;
    hipError_t e = hipMalloc(&fresh, want * kArenaBytesPerSlot + 256);
    if (e != hipSuccess) (void)hipGetLastError();
    if (e == hipErrorOutOfMemory) return fail(c, GPE_ERR_OOM, "native collide: out of device memory for the spill arena");
    if (e != hipSuccess) return fail(c, GPE_ERR_HIP, std::string("hipMalloc (spill arena): ") + hipGetErrorName(e));
    GPE_HIP(c, hipStreamSynchronize(c->stream));                       // (kernels in flight may still use the old one)
    if (N.arena) GPE_HIP(c, hipFree(N.arena));
    N.arena = fresh;
    N.arena_cap = want;
    return GPE_OK;
}

// While a dense scene is held on the compat kernels: measure the window population of the current state WITHOUT a
// host synchronisation -- hash + sort + window maximum are enqueued, the answer lands in pinned memory and is read
// by a later call.  (Round 1 re-ran native_configure here: two stream synchronisations and possibly a reallocation
// inside gpe_run every 256 steps.)
static gpe_status native_probe_async(gpe_ctx *c)
{
    NativeState &N = c->native;
    const bool prof = c->profiling;
    c->profiling = false;
    uint32_t *ids = nullptr;
    const gpe_status st = native_prepare_step(c, &ids, true);
    c->profiling = prof;
    GPE_TRY(st);
    hipLaunchKernelGGL(k_native_window_max, dim3(stream_grid(N.table_entries)), dim3(kStreamBlock), 0, c->stream,
                       N.block_table, N.table_entries, N.blocks_x, N.blocks_y, N.tile_ctl + kCtlWindowMax);
    hipLaunchKernelGGL(k_native_publish_probe, dim3(1), dim3(64), 0, c->stream, N.tile_ctl, N.host_stat);
    GPE_HIP(c, hipGetLastError());
    return GPE_OK;
}

// Should this step take the native kernels?  The tiles report the step's largest 24x24-cell window population and
// the spill-arena slots they used to pinned host memory (asynchronously, so the values lag by the steps still in
// flight; gpe_run bounds that).  Windows above the handover population (one-lane O(n^2) cells that overlapping
// windows would repeat) send the context to the compat kernels -- unless the run needs the native ones (order keys
// of a sharded run: there the dense windows keep going through the spill arena).  A held context probes the state
// every 256 steps without synchronising and returns when the windows have thinned out.  Nothing here frees or
// allocates on ordinary steps; the arena grows (one synchronisation) when a step has used more than half of it.
bool native_should_run(gpe_ctx *c)
{
    NativeState &N = c->native;
    if (c->cfg.mode != GPE_MODE_NATIVE) return false;
    const bool must_stay = N.force || c->use_order_keys;
    if (N.print_stats && N.host_stat && (++N.stat_calls & 127u) == 0)
        fprintf(stderr, "[gpe native] call %u: window max %u, arena slots used %u of %llu, 32x32 tiles over capacity %u, "
                        "quarters redone as 8x8 tiles %u, 8x8 tiles through the arena %u\n", N.stat_calls,
                N.host_stat[kStatWindowMax], N.host_stat[kStatArena], (unsigned long long)N.arena_cap,
                N.host_stat[kStatOverflow], N.host_stat[kStatSubTiles], N.host_stat[kStatSpills]);
    if (!N.eligible && must_stay && N.in_box) { N.eligible = true; N.dense_hold = false; N.reason = GPE_REASON_NONE; }   // density alone never stops such a run
    if (N.eligible) {
        if (N.host_stat && (uint64_t)N.host_stat[kStatArena] * 2 > N.arena_cap && N.arena_cap < kArenaMaxSlots) {
            // (on failure the old arena stays and the step goes on with it: a window it cannot hold raises
            // kErrTileOverflow, which gpe_sync reports -- never a silent hand-over of a run that must stay native)
            if (arena_reserve(c, std::min<uint64_t>(N.arena_cap * 2, kArenaMaxSlots)) != GPE_OK && !must_stay) return false;
            N.host_stat[kStatArena] = 0;
        }
        if (!must_stay && N.host_stat && N.host_stat[kStatWindowMax] > kWindowHandover) {
            N.eligible = false;
            N.dense_hold = true;
            N.reason = GPE_REASON_DENSE_WINDOWS;
            N.steps_since_check = 0;
            N.host_stat[kStatProbe] = 0;
        }
        return N.eligible;
    }
    if (N.dense_hold) {
        const uint32_t probe = N.host_stat ? N.host_stat[kStatProbe] : 0u;
        if (probe != 0 && probe - 1u <= kWindowHandover * 3 / 4) {    // the last probe found the windows thin again
            N.dense_hold = false;
            N.eligible = true;
            N.reason = GPE_REASON_NONE;
            N.host_stat[kStatWindowMax] = probe - 1u;
            N.host_stat[kStatProbe] = 0;
            return true;
        }
        if (++N.steps_since_check >= 256) {
            N.steps_since_check = 0;
            N.host_stat[kStatProbe] = 0;
            (void)native_probe_async(c);
        }
    }
    return false;
}

// Hinted tiles (kCtlHints): the dense launch's first workgroups redo them as halves.  With rosters only (the hint travels
// in the roster header the tile loads anyway), and only while tiles have run over lately (lagged statistic, hinted tiles
// included): the kernel that carries the front workgroups is 3 % slower than the plain one.  Until it is launched a
// registered tile simply tries itself again.  (Up to 8 M particles, and while the front workgroups can take at least half
// of the tiles that run over: the 3 % are 1.5 us of the 1 M launch, against ~20 us of half-tile launch behind it, but
// 0.1 ms at 100 M.)
static void native_hint_policy(gpe_ctx *c, CollideArgs *A)
{
    NativeState &N = c->native;
    if (A->roster_hdr == nullptr || (c->cfg.flags & GPE_FLAG_NO_HALF_TILES) != 0 || c->n > (8ull << 20) || !N.host_stat) return;
    A->hints_on = 1u;
    const uint32_t over = N.host_stat[kStatOverflow];
    if (over != 0 && over <= 2u * kHintMax) N.hint_quiet = 0; else if (N.hint_quiet < 0xFFFFFFFFu) ++N.hint_quiet;
    if (N.hint_quiet < 32u && over <= 2u * kHintMax) A->front_wgs = 2u * kHintMax;
}

// pos_in (step-start positions) -> pos_out (after the four colour passes), every particle written.
gpe_status native_collide(gpe_ctx *c, const float2 *pos_in, float2 *pos_out, const VerletParams *verlet)
{
    NativeState &N = c->native;
    uint32_t *sorted_ids = nullptr;
    GPE_TRY(native_prepare_step(c, &sorted_ids));
    CollideArgs A;
    A.pos_in = pos_in;
    A.radius = c->radius;
    A.pos_out = pos_out;
    A.sorted_ids = sorted_ids;
    A.codes = N.codes;
    A.fresh = N.fresh_word;
    A.gtable = N.gsorted_ids_now ? N.gtable : nullptr;
    A.gsorted_ids = N.gsorted_ids_now;
    A.exc_count = N.exc_count_now;
    A.exc_entry = N.exc_entry_now;
    A.tb = N.tb;
    A.gho_count = N.gsorted_ids_now ? N.gho_count_now : nullptr;     // (kept sharded run with ghosts this step)
    A.gho_entry = N.gho_entry_now;
    A.ghost_sort = N.ghost_sort_now;
    A.roster_hdr = (N.roster_cap >= N.exc_tiles) ? N.roster_hdr : nullptr;
    A.roster_ids = N.roster_ids;
    A.sorts_seen = N.tile_ctl + kCtlSortsSeen;
    A.roster_write = N.exc_count_now != nullptr ? 1u : 0u;            // (this step could do without a sort: the table is kept)
    A.table = N.block_table;
    A.entries = N.table_entries;
    A.blocks_x = N.blocks_x;
    A.blocks_y = N.blocks_y;
    A.bx0 = N.bx0;
    A.by0 = N.by0;
    A.counts = (c->shard.on && c->shard.active) ? c->shard.counts_now() + kShardOwned : nullptr;
    A.cell_size = c->cell_size;
    A.stiffness = c->cfg.stiffness;
    A.gx = N.gx;
    A.gy = N.gy;
    A.tile_ctl = N.tile_ctl;
    A.overflow1 = N.overflow1;
    A.overflow1_cap = (uint32_t)N.overflow_cap;
    A.overflow2 = N.overflow1 + N.overflow_cap + 16;
    A.quarters_of_halves = 0u;
    A.hints = N.overflow1 + 3 * N.overflow_cap + 32;                   // 2 x kHintMax words behind the two lists
    A.hint_parity = N.collide_seq & 1u;
    A.step_stamp = N.collide_seq + 16u;                                // (never the 0 of a cleared roster header)
    ++N.collide_seq;
    A.front_wgs = 0u;
    A.hints_on = 0u;
    {
        // arena layout: px | py | rad | id | hm | mem (4 per slot) | sblk
        float *f = (float *)N.arena;
        const uint64_t m = N.arena_cap;
        A.arena_px = f; A.arena_py = f + m; A.arena_rad = f + 2 * m;
        A.arena_id = (uint32_t *)(f + 3 * m); A.arena_hm = (uint32_t *)(f + 4 * m);
        A.arena_mem = (uint32_t *)(f + 5 * m);
        A.arena_sblk = (uint8_t *)(f + 9 * m);
        A.arena_cap = (uint32_t)m;
    }
    A.order_keys = c->use_order_keys ? c->order_keys : nullptr;
    A.tile_x0 = A.tile_y0 = 0;
    A.prev = c->prev;
    A.n_owned = c->n_owned;
    A.fuse_verlet = verlet ? 1u : 0u;
    if (verlet) A.vp = *verlet; else memset(&A.vp, 0, sizeof(A.vp));
    A.stamps = nullptr;
    A.frame_l = A.frame_r = A.frame_b = A.frame_t = 0;
    A.pack = PackArgs();
    if (c->shard.on && c->shard.active && c->shard.have_rect && verlet && A.order_keys) shard_pack_args(c, &A.pack);
#ifdef GPE_TILE_STAMPS
    static unsigned long long *g_stamps = nullptr;
    // ([0, 64): the dense launch's tiles; [64, 128): the windows of the over-capacity launch)
    if (!g_stamps) { (void)hipMalloc((void **)&g_stamps, 128 * 8); (void)hipMemset(g_stamps, 0, 128 * 8); }
    A.stamps = g_stamps;
    static int g_calls = 0;
    if (++g_calls % 20 == 0) for (int part = 0; part < 2; ++part) {
        unsigned long long h[64];
        (void)hipStreamSynchronize(c->stream);
        (void)hipMemcpy(h, g_stamps + 64 * part, sizeof(h), hipMemcpyDeviceToHost);
        fprintf(stderr, part == 0 ? "[dense launch]\n" : "[over-capacity launch]\n");
        for (int cls = 0; cls < 3; ++cls)
            fprintf(stderr, "[P5 waves] %s: busy %.0f  barrier wait %.0f cycles per colour pass (%llu wave-passes)\n",
                    cls == 0 ? "group waves" : cls == 1 ? "single waves" : "idle waves", h[40 + cls] ? (double)h[32 + cls] / h[40 + cls] : 0.0,
                    h[40 + cls] ? (double)h[36 + cls] / h[40 + cls] : 0.0, h[40 + cls]);
        fprintf(stderr, "[tile stamps] n=%llu", (unsigned long long)c->n);
        if (h[47])
            fprintf(stderr, "[P5 cells] per colour pass: %.1f one-lane cells, %.1f lane-group cells, %.2f whole-wave cells of %.1f members (largest %.1f);"
                    " a wave spends %.0f cycles in them and the rows; %.2f row cells; whole-wave cells by members <=16 / <=32 / <=64 / more: %.2f %.2f %.2f %.2f\n",
                    (double)h[44] / h[47], (double)h[45] / h[47], (double)h[46] / h[47],
                    h[46] ? (double)h[49] / h[46] : 0.0, (double)h[50] / h[47], (double)h[48] / (double)(h[40] + h[41] + h[42]),
                    (double)h[51] / h[47], (double)h[52] / h[47], (double)h[53] / h[47], (double)h[54] / h[47], (double)h[55] / h[47]);
        double all = 0;
        for (int i = 0; i < 14; ++i) all += (double)h[i];
        for (int i = 0; i < 14; ++i)
            fprintf(stderr, "  P%d %.0f (%.1f%%)", i, h[16 + i] ? (double)h[i] / (double)h[16 + i] : 0.0, all > 0 ? 100.0 * (double)h[i] / all : 0.0);
        fprintf(stderr, "  (tiles %llu of %llu started)\n", h[16 + 6], h[16 + 0]);
        if (part == 1) (void)hipMemset(g_stamps, 0, 128 * 8);
    }
#endif
    int32_t cx0 = 0, cy0 = 0, cx1 = N.gx - 1, cy1 = N.gy - 1;
    if (c->has_active_box) {                                           // sharded: only this rank's cells
        cx0 = std::max(cx0, c->active_box[0]); cy0 = std::max(cy0, c->active_box[1]);
        cx1 = std::min(cx1, c->active_box[2]); cy1 = std::min(cy1, c->active_box[3]);
        if (cx1 < cx0 || cy1 < cy0) { cx1 = cx0; cy1 = cy0; }
    }
    A.tile_x0 = cx0 / kTileMain;
    A.tile_y0 = cy0 / kTileMain;
    A.tiles_x = cx1 / kTileMain - A.tile_x0 + 1;
    A.tiles_y = cy1 / kTileMain - A.tile_y0 + 1;
    const uint32_t total = (uint32_t)A.tiles_x * (uint32_t)A.tiles_y;
    bool direct_form = false;                                          // the dense launch runs direct-slot tiles
    {
        Scope s(c, verlet ? "native/collide+verlet" : "native/collide");
        A.band_tiles = dense_launch_band((uint32_t)A.tiles_x, (uint32_t)A.tiles_y, (c->cfg.flags & GPE_FLAG_XCD_EIGHTHS) != 0);
        const uint32_t grid = dense_launch_grid((uint32_t)A.tiles_x, (uint32_t)A.tiles_y, A.band_tiles);
        // Which form of the tile?  The direct-slot form is the faster one while tiles fit it; it holds 928 particles and
        // hands a tile on when its cells crowd (more than 96 memberships beyond a cell's sixth, a cell of more than 64).
        // In a compressed scene (the 100 M cloud after a few hundred steps of gravity) most tiles would take that
        // detour through the over-capacity launch, while the counting-sort form holds 1192 particles and resolves
        // crowded cells in place.  So the choice follows the tiles' own report (lagged by the steps in flight; either
        // form is exact): counting-sort tiles once more than 2 % of the direct-slot tiles ran over; back to direct
        // slots when no 24x24-cell window has held more than 512 particles (2.3 x the mean of the benchmark density; a
        // direct-slot window overflows around 350) and no tile has run over for 64 steps.
        // (order-key windows carry four more bytes per particle: in the direct form that leaves 728 slots, 1.27 x the
        // mean tile's particles, and too many tiles run over; the counting-sort form holds 1024)
        if (N.host_stat) {
            const uint32_t over = N.host_stat[kStatOverflow];
            if (!N.crowded) {
                if (over > total / 50u + 4u) { N.crowded = true; N.calm_steps = 0; }
            } else {
                N.calm_steps = (over == 0 && N.host_stat[kStatWindowMax] == 0) ? N.calm_steps + 1 : 0;
                if (N.calm_steps >= 64) N.crowded = false;
            }
        }
        // (order-key windows: the direct form needs the ghost lists of a kept sharded run; every other sharded set-up
        // looks its ghosts up in the block tables, which only the counting-sort form does)
        const bool legacy = (c->cfg.flags & GPE_FLAG_COUNTING_SORT_TILES) != 0 || N.crowded ||
                            (A.order_keys != nullptr && A.gho_count == nullptr);
        direct_form = !legacy;
        if (legacy) {
            if (A.order_keys)
                hipLaunchKernelGGL((k_collide_dense<kTileMain, kCapOrd, true>), dim3(grid), dim3(kNatThreads), 0, c->stream, A);
            else
                hipLaunchKernelGGL((k_collide_dense<kTileMain, kCapMain, false>), dim3(grid), dim3(kNatThreads), 0, c->stream, A);
        } else if (A.order_keys) {
            // A sharded step whose tiles pack (A.pack.on) and whose exchange runs beside it (ShardState::overlap): the frame
            // of the tile grid first -- the tiles whose particles can come to lie outside the pack's safe box: their cells
            // reach within a block and a cell (the most a particle may move per step, + the box's margin) of it -- then
            // the event the exchange waits for, then the interior tiles, which must not have anything to pack.
            ShardState &SH = c->shard;
            bool split = false;
            if (A.pack.on == 1u && SH.overlap && SH.ev_packed) {
                const int reach = 8 + 1 + 8 + 1;                       // cells: from a tile's edge to the safe box's edge
                const int rx0 = SH.rect[0] * 8, ry0 = SH.rect[1] * 8, rx1 = SH.rect[2] * 8, ry1 = SH.rect[3] * 8;
                const bool nb_l = SH.rect[0] > 0, nb_r = SH.rect[2] < SH.blocks_x, nb_d = SH.rect[1] > 0, nb_u = SH.rect[3] < SH.blocks_y;
                int fl = 0, fr = 0, fb = 0, ft = 0;
                for (int t = 0; t < A.tiles_x; ++t) {
                    const int c0 = (A.tile_x0 + t) * kTileMain, c1 = c0 + kTileMain - 1;
                    if (nb_l && c0 < rx0 + reach) fl = t + 1;
                    if (nb_r && c1 >= rx1 - reach && fr == 0) fr = A.tiles_x - t;
                }
                for (int t = 0; t < A.tiles_y; ++t) {
                    const int c0 = (A.tile_y0 + t) * kTileMain, c1 = c0 + kTileMain - 1;
                    if (nb_d && c0 < ry0 + reach) fb = t + 1;
                    if (nb_u && c1 >= ry1 - reach && ft == 0) ft = A.tiles_y - t;
                }
                if (fl + fr < A.tiles_x && fb + ft < A.tiles_y) {
                    split = true;
                    A.frame_l = fl; A.frame_r = fr; A.frame_b = fb; A.frame_t = ft;
                    const uint32_t frame = (uint32_t)((fb + ft) * A.tiles_x + (A.tiles_y - fb - ft) * (fl + fr));
                    if (frame) {
                        hipLaunchKernelGGL(k_collide_border<true>, dim3(frame), dim3(512), 0, c->stream, A);
                        GPE_HIP(c, hipGetLastError());
                    }
                    GPE_HIP(c, hipEventRecord(SH.ev_packed, c->stream));
                    SH.packed_recorded = true;
                    // the interior: a tile box of its own (bands as above), nothing to pack
                    A.tile_x0 += fl; A.tile_y0 += fb; A.tiles_x -= fl + fr; A.tiles_y -= fb + ft;
                    A.pack.on = 2u;
                    A.band_tiles = dense_launch_band((uint32_t)A.tiles_x, (uint32_t)A.tiles_y, (c->cfg.flags & GPE_FLAG_XCD_EIGHTHS) != 0);
                    const uint32_t igrid = dense_launch_grid((uint32_t)A.tiles_x, (uint32_t)A.tiles_y, A.band_tiles);
                    hipLaunchKernelGGL((k_collide_direct<32, GPE_CAP_DIRECT_ORD, true, 512>), dim3(igrid), dim3(512), 0, c->stream, A);
                }
            }
            if (!split) {
                native_hint_policy(c, &A);
                if (A.front_wgs)
                    hipLaunchKernelGGL((k_collide_direct<32, GPE_CAP_DIRECT_ORD, true, 512, true>), dim3(grid + A.front_wgs), dim3(512), 0, c->stream, A);
                else
                    hipLaunchKernelGGL((k_collide_direct<32, GPE_CAP_DIRECT_ORD, true, 512>), dim3(grid), dim3(512), 0, c->stream, A);
            }
        } else {
            native_hint_policy(c, &A);
            if (A.front_wgs)
                hipLaunchKernelGGL((k_collide_direct<32, GPE_CAP_DIRECT, false, 512, true>), dim3(grid + A.front_wgs), dim3(512), 0, c->stream, A);
            else
                hipLaunchKernelGGL((k_collide_direct<32, GPE_CAP_DIRECT, false, 512>), dim3(grid), dim3(512), 0, c->stream, A);
        }
        GPE_HIP(c, hipGetLastError());
    }
    {
        // tiles whose window exceeded the LDS capacity: 16x16 tiles, 8x8 tiles, spill arena.  (Measured and dropped:
        // running this launch on a second stream beside the dense one -- the stream fork/join costs ~8 us per step,
        // more than the normally empty launch it hides; it only pays in clustered scenes.)
        // The launch takes its work items from a ticket counter, so ANY grid is correct; an empty launch of 1024
        // workgroups (8 waves and 36 KB of LDS each) costs ~6 us, 8 % of the 1 M step.  While the tiles have reported no
        // over-capacity tile for a while (the statistic lags by the steps in flight) the grid is 128 workgroups; the
        // first reported tile brings the full grid back.  A surprise only makes that one step's launch slower.
        Scope s(c, "native/collide-dense-regions");
        // (quiet_steps: since the dense launch last handed a tile on ITSELF -- hinted tiles do not count, they never reach
        // list 1; dense_quiet: since anything reached list 1 or list 2)
        if (N.host_stat && N.host_stat[A.front_wgs ? kStatOverflowNew : kStatOverflow] != 0) N.quiet_steps = 0;
        else if (N.quiet_steps < 0xFFFFFFFFu) ++N.quiet_steps;
        if (N.host_stat && (N.host_stat[kStatOverflowNew] != 0 || N.host_stat[kStatHalvesOver] != 0 || (!A.front_wgs && N.host_stat[kStatOverflow] != 0))) N.dense_quiet = 0;
        else if (N.dense_quiet < 0xFFFFFFFFu) ++N.dense_quiet;
        // The half-tile launch: while the direct-slot launch has handed tiles on lately (lagged; either way is exact --
        // without it the over-capacity launch takes the tiles of list 1).  Not behind counting-sort tiles: what does not
        // fit their 1192 particles is dense enough for the windows.
        // (A scene in which more than 2 % of the tiles run over has its dense launch on counting-sort tiles by then --
        // `crowded` above: this launch is for the few tiles of a clumped cloud, not for piles; keeping the direct-slot form
        // with halves behind it up to 50 % of the tiles was measured: step 2000 of the 100 M soak 35.5 instead of 31.0 ms.)
        // (With front workgroups in the dense launch list 1 only holds tiles that ran over for the FIRST time -- one every
        // ~60 steps in the clumped 1 M cloud, which no lagged statistic foresees: the over-capacity launch takes those as
        // quarters, and the half-tile launch comes back when list 1 stays occupied, i.e. the hints are full.)
        if (N.host_stat[kStatOverflowNew] != 0) { if (N.new_streak < 0xFFFFFFFFu) ++N.new_streak; } else N.new_streak = 0;
        const bool halves_wanted = A.front_wgs ? N.new_streak >= 4u : N.quiet_steps < 32u;
        if (direct_form && halves_wanted && (c->cfg.flags & GPE_FLAG_NO_HALF_TILES) == 0) {
            A.quarters_of_halves = 1u;
            const uint32_t hgrid = (uint32_t)std::min<uint64_t>(1024, std::max<uint64_t>(64, 2ull * N.host_stat[A.front_wgs ? kStatOverflowNew : kStatOverflow] + 32));
            if (A.order_keys)
                hipLaunchKernelGGL(k_collide_halves<true>, dim3(hgrid), dim3(512), 0, c->stream, A);
            else
                hipLaunchKernelGGL(k_collide_halves<false>, dim3(hgrid), dim3(512), 0, c->stream, A);
            GPE_HIP(c, hipGetLastError());
        }
        // (Only where the empty launch matters: from a few million particles on its 6 us are noise, and a surprise -- the
        // statistic lags by up to 64 steps -- would cost those steps milliseconds each.)
        // ... or, with front workgroups, while the lists have held few work items lately (a first-time tile is four)
        const uint64_t items = 4ull * N.host_stat[kStatOverflowNew] + 2ull * N.host_stat[kStatHalvesOver];
        const bool small_grid = c->n <= (4ull << 20) && (N.dense_quiet > 96 || (A.front_wgs != 0u && items <= 128u));
        const uint32_t ogrid = small_grid ? 128u : 1024u;
#ifdef GPE_TILE_STAMPS
        A.stamps += 64;
#endif
        if (A.order_keys)
            hipLaunchKernelGGL(k_collide_overflow<true>, dim3(ogrid), dim3(kNatThreads), 0, c->stream, A);
        else
            hipLaunchKernelGGL(k_collide_overflow<false>, dim3(ogrid), dim3(kNatThreads), 0, c->stream, A);
        GPE_HIP(c, hipGetLastError());
    }
    // (a sharded step that did not split its tiles: everything has packed now)
    if (A.pack.on == 1u && c->shard.overlap && c->shard.ev_packed) {
        GPE_HIP(c, hipEventRecord(c->shard.ev_packed, c->stream));
        c->shard.packed_recorded = true;
    }
    return GPE_OK;
}

// ---------------------------------------------------------------------------------------------------
// sharded runs: which owned particles must travel (gpe_shard_classify)
// ---------------------------------------------------------------------------------------------------
// One streaming pass over the owned particles: R pos 8 B + one table byte/word per particle.  The few
// that sit in a block owned by another rank (migrants) or bordering other ranks (ghost candidates) are
// appended with one global atomic per wave.
__global__ __launch_bounds__(kStreamBlock) void k_shard_classify(const float2 *__restrict__ pos, uint64_t n_owned,
                                                                  float cell_size,
                                                                  const uint8_t *__restrict__ owner_of_block,
                                                                  const uint32_t *__restrict__ dest_mask_of_block,
                                                                  int32_t blocks_x, int32_t blocks_y, uint32_t my_rank,
                                                                  uint32_t *__restrict__ out_index,
                                                                  uint32_t *__restrict__ out_info,
                                                                  uint32_t *__restrict__ out_count, uint64_t out_capacity)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t rounds = (n_owned + stride - 1) / stride;
    for (uint64_t r = 0; r < rounds; ++r) {
        const uint64_t i = r * stride + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        uint32_t info = 0;
        if (i < n_owned) {
            const float2 p = pos[i];
            int bx = cell_coord(p.x, cell_size) >> 3, by = cell_coord(p.y, cell_size) >> 3;
            bx = min(max(bx, 0), blocks_x - 1);
            by = min(max(by, 0), blocks_y - 1);
            const uint32_t b = (uint32_t)by * (uint32_t)blocks_x + (uint32_t)bx;
            const uint32_t owner = owner_of_block[b];
            // bits 0-25: ranks bordering the block the particle sits in NOW (they need it as a ghost);
            // bits 26-30: 1 + owner of that block when it is not this rank (the particle migrates)
            info = (dest_mask_of_block[b] & 0x03FFFFFFu) | ((owner != my_rank) ? ((owner + 1u) << 26) : 0u);
        }
        const uint64_t m = ballot64(info != 0);
        if (m == 0) continue;
        const int leader = (int)__builtin_ctzll(m);
        uint32_t base = 0;
        if (lane_id() == leader) base = atomicAdd(out_count, (uint32_t)__popcll(m));
        base = __shfl(base, leader, 64);
        if (info != 0) {
            const uint64_t slot = (uint64_t)base + popc_below_lane(m);
            if (slot < out_capacity) { out_index[slot] = (uint32_t)i; out_info[slot] = info; }
        }
    }
}

gpe_status launch_shard_classify(gpe_ctx *c, const uint8_t *owner_of_block, const uint32_t *dest_mask_of_block,
                                 int32_t blocks_x, int32_t blocks_y, uint32_t my_rank, uint32_t *out_index,
                                 uint32_t *out_info, uint32_t *out_count, uint64_t out_capacity)
{
    if (c->n_owned == 0) return GPE_OK;
    Scope s(c, "shard/classify");
    hipLaunchKernelGGL(k_shard_classify, dim3(stream_grid(c->n_owned)), dim3(kStreamBlock), 0, c->stream, c->pos,
                       c->n_owned, c->cell_size, owner_of_block, dest_mask_of_block, blocks_x, blocks_y, my_rank,
                       out_index, out_info, out_count, out_capacity);
    GPE_HIP(c, hipGetLastError());
    return GPE_OK;
}

}  // namespace gpe

#ifdef GPE_TILE_CYCLES
// diagnostic builds only: start recording (out == NULL: a buffer of one uint4 per tile of the tile box is attached) or
// download what the last step recorded (out: tiles x 4 words; *tiles_x / *tiles_y the tile box)
extern "C" gpe_status gpe_debug_tile_cycles(gpe_ctx *c, uint32_t *out, uint64_t words, uint32_t *tiles_x, uint32_t *tiles_y)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    gpe::NativeState &N = c->native;
    static uint4 *buf = nullptr;
    static uint64_t cap = 0;
    (void)hipStreamSynchronize(c->stream);
    const uint64_t tiles = N.exc_tiles;
    if (tiles_x) *tiles_x = (uint32_t)N.tb.nx;
    if (tiles_y) *tiles_y = (uint32_t)N.tb.ny;
    if (!out) {
        if (cap < tiles) { if (buf) (void)hipFree(buf); if (hipMalloc((void **)&buf, tiles * sizeof(uint4)) != hipSuccess) return GPE_ERR_HIP; cap = tiles; }
        (void)hipMemset(buf, 0, tiles * sizeof(uint4));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(gpe::g_tile_cycles), &buf, sizeof(buf));
        return GPE_OK;
    }
    if (!buf || words < tiles * 4) return GPE_ERR_INVALID_ARG;
    (void)hipMemcpy(out, buf, tiles * sizeof(uint4), hipMemcpyDeviceToHost);
    return GPE_OK;
}
#endif

#ifdef GPE_TILE_CYCLES
// diagnostic builds only: how often each reason handed a direct-slot tile on since the last call (1 looked up > 1536,
// 2 kept > capacity, 3 more than kBigCap memberships beyond a cell's slots, 4 a cell of more than 64 (lanes counted),
// 5 more than 16 cells of 7+ members in a colour (lanes counted))
extern "C" gpe_status gpe_debug_bail_reasons(gpe_ctx *c, uint32_t *out8)
{
    if (!c || !out8) return GPE_ERR_INVALID_ARG;
    (void)hipStreamSynchronize(c->stream);
    (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(gpe::g_bail_reasons), 8 * sizeof(uint32_t));
    uint32_t z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(gpe::g_bail_reasons), z, sizeof(z));
    return GPE_OK;
}
#endif

#ifdef GPE_DBG_RT
extern "C" gpe_status gpe_debug_skip(gpe_ctx *c, uint32_t mask)
{
    if (!c) return GPE_ERR_INVALID_ARG;
    (void)hipStreamSynchronize(c->stream);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(gpe::g_dbg_skip_rt), &mask, sizeof(mask));
    return GPE_OK;
}
#endif

#ifdef GPE_COUNT_PAIRS
// diagnostic builds only: (pairs walked, pairs resolved) since the last reset
extern "C" gpe_status gpe_debug_pair_counts(gpe_ctx *c, uint64_t *out2, int32_t reset)
{
    // out2 = (pairs walked, pairs resolved) since the last reset; reset != 0 also switches the counting on, reset < 0 off
    if (!c || !out2) return GPE_ERR_INVALID_ARG;
    (void)hipStreamSynchronize(c->stream);
    static unsigned long long w[gpe::kPairCounters * 8], h[gpe::kPairCounters * 8];
    (void)hipMemcpyFromSymbol(w, HIP_SYMBOL(gpe::g_pairs_walked), sizeof(w));
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(gpe::g_pairs_hit), sizeof(h));
    out2[0] = out2[1] = 0;
    for (int i = 0; i < gpe::kPairCounters; ++i) { out2[0] += w[i * 8]; out2[1] += h[i * 8]; }
    if (reset) {
        memset(w, 0, sizeof(w));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(gpe::g_pairs_walked), w, sizeof(w));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(gpe::g_pairs_hit), w, sizeof(w));
        const uint32_t on = reset > 0 ? 1u : 0u;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(gpe::g_pairs_on), &on, sizeof(on));
    }
    return GPE_OK;
}
#endif
