// k_onesweep.hip -- onesweep-style LSD radix sort of (u32 key, u32 payload) pairs for gfx950.
//
//   1. ONE upfront pass builds the 256-bin histogram of every 8-bit digit (LDS histograms, flushed
//      with one global atomic per bin per workgroup).  The native step fuses this into the cell-hash
//      kernel (k_native.hip), so the keys are not even re-read.
//   2. k_os_prepare turns the histograms into exclusive digit bases and resets the tile tickets.
//   3. One kernel per digit: a workgroup takes a tile ticket, ranks its 8192 keys (lanes of a digit meet in an LDS match table)
//      (stable), publishes its per-digit tile counts, resolves the counts of all preceding tiles by
//      decoupled look-back over 8-byte {epoch, flag, value} status words, reorders the tile in LDS and
//      writes each digit's run as one coalesced store.  Per pass: R 8 B + W 8 B per pair.
//
// Inter-workgroup protocol (cdna_hip_programming.md Guideline 16, form R2 "the data is the flag"):
// a status word is ONE naturally aligned 8-byte granule written by one relaxed agent-scope atomic
// store (sc1) and read by relaxed agent-scope atomic loads; no other memory is handed over inside a
// launch, so no fences are needed.  Tickets come from an atomic counter, so a tile only ever waits
// for tiles that are already resident => forward progress whatever the dispatch order.  Status
// words carry an epoch (sort call x pass), so the array is never cleared between passes; spins are
// bounded and report through an error word instead of hanging the GPU.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "gpe_internal.h"

#ifdef GPE_OS_STAMPS
#define OS_STAMP(i)                                                                   \
    do {                                                                              \
        if (threadIdx.x == 0 && (s_tile & 63u) == 7u) {                               \
            const long long _t = clock64();                                           \
            atomicAdd(&ctl[16 + (i)], (uint32_t)(_t - _t_prev));                      \
            _t_prev = _t;                                                             \
        }                                                                             \
    } while (0)
#else
#define OS_STAMP(i) do {} while (0)
#endif

namespace gpe {

#ifndef GPE_OS_MINWAVES
#define GPE_OS_MINWAVES 4
#endif
constexpr int kOsBlock = 512;
constexpr int kOsWaves = kOsBlock / 64;
// 8192-key tiles: measured best from 1 M to 100 M keys (2048-key tiles and a 32-word look-back
// window were each 10-50% slower per pass at 1 M, 4 M and 16 M keys -- profiles/r01/tune_onesweep.txt)
constexpr int kOsItems = 16;                     // keys per thread: 8192-key tiles ...
constexpr int kOsItemsSmall = 8;                 // ... and 4096-key tiles for small sorts: at 1 M keys 8192-key tiles
                                                 // leave half the CUs without a workgroup (one pass 21.9 -> 20.9 us;
                                                 // at 16 M and 100 M keys the small tiles are 20-33 % slower:
                                                 // profiles/r01/tune_onesweep_items.txt)
constexpr uint64_t kOsSmallSort = 3u << 20;      // sorts up to this many keys take the small tiles
constexpr int kWin = 4;                          // predecessors per look-back round trip: 4 beats 8 by 0.6 us per pass
                                                 // at 1 M keys and by 15 us (3 %) at 100 M, 16 loses 1.6 us at 1 M
                                                 // (profiles/r02/ab_lookback_window.txt)

constexpr uint64_t kFlagAggregate = 1ull;        // value = this tile's count of the digit
constexpr uint64_t kFlagPrefix = 2ull;           // value = count of the digit in tiles 0..this
constexpr uint32_t kSpinLimit = 1u << 24;

typedef unsigned long long u64;

// 1. all four digit histograms in one read of the keys: hist4[p*256 + d]
__global__ __launch_bounds__(kStreamBlock) void k_os_hist4(const uint32_t *__restrict__ keys, uint64_t n,
                                                        uint32_t *__restrict__ hist4)
{
    __shared__ uint32_t s_hist[4 * 256];
#pragma unroll
    for (int p = 0; p < 4; ++p) s_hist[p * 256 + threadIdx.x] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t rounds = (n + stride - 1) / stride;
    for (uint64_t r = 0; r < rounds; ++r) {
        const uint64_t idx = r * stride + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        const bool valid = idx < n;
        const uint32_t key = valid ? keys[idx] : 0u;
#pragma unroll
        for (int p = 0; p < 4; ++p) hist_add(s_hist + p * 256, (key >> (8 * p)) & 255u, valid);
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint32_t v = s_hist[p * 256 + threadIdx.x];
        if (v) atomicAdd(&hist4[p * 256 + threadIdx.x], v);
    }
}

// 2. exclusive digit bases per pass; reset the four tile tickets and the error word.
//    ctl: [0..3] tickets, [4] error
__global__ __launch_bounds__(256) void k_os_prepare(const uint32_t *__restrict__ hist4,
                                                     uint32_t *__restrict__ bases4, uint32_t *__restrict__ ctl)
{
    __shared__ uint32_t s_w[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint32_t v = hist4[p * 256 + threadIdx.x];
        bases4[p * 256 + threadIdx.x] = block256_exclusive_scan(v, s_w, nullptr);
    }
    if (threadIdx.x < 8) ctl[threadIdx.x] = 0;
}

__device__ __forceinline__ u64 status_pack(uint32_t epoch, uint64_t flag, uint32_t value)
{
    return ((u64)epoch << 34) | (flag << 32) | (u64)value;
}

// spin (bounded) until the status word carries this epoch and a flag
// (returns 0 -- no flag -- when the bound is hit)
__device__ __noinline__ u64 os_wait_status(const u64 *p, uint32_t epoch)
{
    u64 sv;
    uint32_t spins = 0;
    do {
        if (++spins > kSpinLimit) return 0;
        __builtin_amdgcn_s_sleep(1);
        sv = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } while ((uint32_t)(sv >> 34) != epoch || ((sv >> 32) & 3ull) == 0ull);
    return sv;
}

// exclusive scan of one value per thread over the kOsBlock threads of the pass kernel
__device__ __forceinline__ uint32_t os_block_exclusive_scan(uint32_t v, uint32_t *s_w)
{
    const int lane = lane_id();
    const int w = (int)(threadIdx.x >> 6);
    const uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int i = 0; i < kOsWaves; ++i)
        if (i < w) base += s_w[i];
    __syncthreads();
    return base + inc - v;
}

// 3. one digit pass.  IOTA: the payload of input element i is i itself (first pass of a sort whose
//    payload is the identity), saving the payload read.
//    512 threads x 16 keys = 8192 keys per tile: the look-back reads 2 KB of status per predecessor and
//    tile, so larger tiles halve that traffic per key; keys and payloads are reordered through ONE
//    32 KB LDS buffer, one after the other (42 KB of LDS per workgroup).
template <bool IOTA, int ITEMS>
__global__ __launch_bounds__(kOsBlock, GPE_OS_MINWAVES) void k_os_pass(const uint32_t *__restrict__ keys_in,
                                                       const uint32_t *__restrict__ vals_in,
                                                       uint32_t *__restrict__ keys_out,
                                                       uint32_t *__restrict__ vals_out, uint64_t n,
                                                       uint32_t shift, uint32_t pass,
                                                       const uint32_t *__restrict__ bases4, u64 *status,
                                                       uint32_t *ctl, uint32_t epoch, uint2 *table,
                                                       uint32_t table_entries, const uint32_t *__restrict__ hist_src,
                                                       OnesweepGate G)
{
    // Gated sort (the native step): the producer of the keys decides on the device whether this sort is needed at all
    // (k_native_hash: has any particle left the reach of the old block table?).  The passes are enqueued every step
    // and return here when the word is 0 -- the host never waits for the decision.
    if (G.need && *G.need == 0u) return;
    constexpr int kOsItems = ITEMS, kOsTile = kOsBlock * ITEMS, kOsWaveSpan = 64 * ITEMS;   // shadow the defaults
    __shared__ uint32_t s_stage[kOsTile];
    __shared__ uint32_t s_whist[kOsWaves][256];
    __shared__ u64 s_match[kOsWaves][256];                         // per-wave digit -> lane mask, zero between rounds
    __shared__ uint32_t s_excl[256];
    __shared__ uint32_t s_delta[256];
    __shared__ uint32_t s_w[kOsWaves];
    __shared__ uint32_t s_tile;

#ifdef GPE_OS_STAMPS
    long long _t_prev = clock64();
#endif
    if (threadIdx.x == 0) s_tile = atomicAdd(&ctl[G.ticket_base + pass], 1u);     // ticket: tiles start in ticket order
    for (int i = threadIdx.x; i < kOsWaves * 256; i += kOsBlock) { (&s_whist[0][0])[i] = 0; (&s_match[0][0])[i] = 0ull; }
    __syncthreads();
    const uint32_t tile = s_tile;

    const int lane = lane_id();
    const int w = (int)(threadIdx.x >> 6);
    const uint64_t tile_base = (uint64_t)tile * kOsTile;
    const uint64_t wave_base = tile_base + (uint64_t)w * kOsWaveSpan;
    const uint32_t tile_n = (uint32_t)((n - tile_base < (uint64_t)kOsTile) ? (n - tile_base) : kOsTile);

    uint32_t key[kOsItems], val[kOsItems];
    uint16_t slot[kOsItems];                                       // rank in the wave, then slot in the tile
#pragma unroll
    for (int k = 0; k < kOsItems; ++k) {
        const uint64_t idx = wave_base + (uint64_t)k * 64 + lane;
        const bool valid = idx < n;
        key[k] = valid ? keys_in[idx] : 0xffffffffu;
        if (IOTA) val[k] = (uint32_t)idx;
        else val[k] = valid ? vals_in[idx] : 0u;
    }
    if (G.key_copy) {
        // first pass of a gated sort: the keys in input order are what the NEXT steps compare against (the block every
        // particle is being sorted into), and the block table the last pass fills is reset here, a launch earlier
#pragma unroll
        for (int k = 0; k < kOsItems; ++k) {
            const uint64_t idx = wave_base + (uint64_t)k * 64 + lane;
            if (idx < n) {
                // (the block's coordinates in the block box, x | y << 16: what the hash kernel's drift test needs --
                // one division per sort here instead of one per particle and step there)
                const uint32_t by = (uint32_t)(((uint64_t)key[k] * G.key_div_magic) >> 40);
                G.key_copy[idx] = (key[k] - by * G.key_blocks_x) | (by << 16);
            }
        }
        for (uint64_t i = (uint64_t)tile * kOsBlock + threadIdx.x; i < G.table_pairs; i += (uint64_t)gridDim.x * kOsBlock)
            G.table_reset[i] = make_uint4(0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u);   // (first, one past last) = (max, 0): empty
        // how many particles this grouping covers: indices from there on are not in it (sharded runs: the owned count
        // lives on the device and moves; ghosts and arrivals behind it reach the tiles by other routes)
        if (tile == 0 && threadIdx.x == 0) *G.sorted_count = G.count_now ? *G.count_now : (uint32_t)n;
    }
    if (G.fresh && tile == 0 && threadIdx.x == 0) {                  // last pass: the table is of this step
        *G.fresh = 1u;
        const uint32_t before = atomicAdd(G.sorts, 1u);
        if (G.sorts_seen) *G.sorts_seen = before + 1u;
    }
    // hist_src (the native step): the producer of the keys left the digit histograms (kHistCopies copies) and no
    // bases -- every tile sums and scans its pass's 256 bins itself, the loads in flight beside the keys'.  That takes
    // a serial "last workgroup scans" tail (~5 us) out of the producer.
    uint32_t digit_base = 0;
    if (hist_src) {
        uint32_t v = 0;
        if (threadIdx.x < 256) {
#pragma unroll
            for (int k = 0; k < kHistCopies; ++k) v += hist_src[k * 1024 + pass * 256 + threadIdx.x];
        }
        digit_base = os_block_exclusive_scan(v, s_w);              // threads >= 256 contribute 0, come last
    }
#ifdef GPE_OS_STAMPS
    { uint32_t acc = 0; for (int k = 0; k < kOsItems; ++k) acc += key[k] + val[k]; asm volatile("" :: "v"(acc)); }
    OS_STAMP(0);
#endif
    // Rank: (round k, lane) is the input order inside the wave's span, so
    // rank = keys of this digit in earlier rounds + lower lanes of this round  => stable.
    // Rank: (round k, lane) is the input order inside the wave's span, so
    // rank = keys of this digit in earlier rounds + lower lanes of this round  => stable.
    // The lanes of a round that share a digit find each other through LDS: every lane ORs its lane bit into
    // its digit's 64-bit word of the wave's match table, reads the word back (the LDS runs a wave's
    // instructions in order, so the read sees the whole round), and clears it for the next round.  That is 3
    // LDS operations and ~10 VALU instructions per key where eight ballot-and-select steps took ~70 VALU
    // instructions -- the pass was VALU-bound on those.
    uint32_t *wh = s_whist[w];
    u64 *wm = s_match[w];
    const u64 my_bit = 1ull << lane;
    constexpr int kBatch = 8;                                      // rounds whose match words are in flight together
    constexpr int kPeel = 4;                                       // digit values matched by ballot before the table
#pragma unroll
    for (int k0 = 0; k0 < kOsItems; k0 += kBatch) {
        // OR / read / clear of a round do not wait for each other's results: the LDS keeps a wave's
        // instructions in order, so a batch is issued back to back and its reads return together
        u64 m[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const int k = k0 + j;
            const bool valid = wave_base + (uint64_t)k * 64 + lane < n;
            const uint32_t d = (key[k] >> shift) & 255u;
            // Clustered digits (keys of neighbouring particles: a wave holds a handful of values) are peeled off
            // with ballots, one value per step -- many lanes ORing into ONE LDS word serialise; what is left
            // after kPeel values (scattered digits) meets through the LDS match table.
            u64 rem = ballot64(valid);
            u64 mine = 0ull;
#pragma unroll
            for (int it = 0; it < kPeel; ++it) {
                if (rem == 0) break;                                   // wave-uniform
                const int first = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(rem));
                const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, first);
                const u64 same = ballot64(valid && d == d0) & rem;
                if (d == d0) mine = same;
                rem &= ~same;
            }
            const bool left = (rem >> lane) & 1ull;
            if (rem != 0) {                                            // wave-uniform
                if (left) __hip_atomic_fetch_or(&wm[d], my_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                wave_lds_order();
                if (left) mine = wave_lds_load(&wm[d]);
                wave_lds_order();
                if (left) wave_lds_store(&wm[d], 0ull);
                wave_lds_order();
            }
            m[j] = valid ? mine : 0ull;
        }
        // the per-wave digit counters: one dependent LDS read -> write per round
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const int k = k0 + j;
            const bool valid = wave_base + (uint64_t)k * 64 + lane < n;
            const uint32_t d = (key[k] >> shift) & 255u;
            const uint32_t below = (uint32_t)__popcll(m[j] & (my_bit - 1ull));
            const uint32_t pre = wave_lds_load(&wh[d]);
            wave_lds_order();
            if (valid && below == 0) wave_lds_store(&wh[d], pre + (uint32_t)__popcll(m[j]));
            wave_lds_order();
            slot[k] = (uint16_t)(pre + below);
        }
    }
    __syncthreads();
    OS_STAMP(1);

    // one thread per digit: tile count, wave offsets, publish the tile's aggregate
    const uint32_t d = threadIdx.x & 255u;
    const bool digit_thread = threadIdx.x < 256;
    u64 *mine = status + (uint64_t)tile * 256 + d;
    uint32_t count = 0;
    if (digit_thread) {
        uint32_t run = 0;
#pragma unroll
        for (int i = 0; i < kOsWaves; ++i) { const uint32_t ci = s_whist[i][d]; s_whist[i][d] = run; run += ci; }
        count = run;
        if (tile > 0)
            __hip_atomic_store(mine, status_pack(epoch, kFlagAggregate, count), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
    {
        const uint32_t ex = os_block_exclusive_scan(count, s_w);   // threads >= 256 contribute 0, come last
        if (digit_thread) s_excl[d] = ex;
    }
    __syncthreads();

    // reorder the keys in LDS by digit (stable) BEFORE looking back: it needs only tile-local offsets, and
    // meanwhile the predecessors get on with publishing their inclusive prefixes (shorter walk, no spinning)
#pragma unroll
    for (int k = 0; k < kOsItems; ++k) {
        const uint64_t idx = wave_base + (uint64_t)k * 64 + lane;
        if (idx < n) {
            const uint32_t dk = (key[k] >> shift) & 255u;
            slot[k] = (uint16_t)(s_excl[dk] + s_whist[w][dk] + slot[k]);
            s_stage[slot[k]] = key[k];
        }
    }
    OS_STAMP(2);

    // decoupled look-back, a WINDOW of predecessors per memory round trip
    if (digit_thread) {
        uint32_t before = 0;                                   // digit d in tiles [0, tile)
        bool failed = false;
        uint32_t hops = 0, total_spins = 0;
        bool done = (tile == 0);
        for (int64_t t = (int64_t)tile - 1; !done && !failed; t -= kWin) {
            u64 sw[kWin];
#pragma unroll
            for (int i = 0; i < kWin; ++i)
                sw[i] = (t - i >= 0) ? __hip_atomic_load(status + (uint64_t)(t - i) * 256 + d, __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT)
                                     : status_pack(epoch, kFlagPrefix, 0u);       // before tile 0: nothing
#pragma unroll
            for (int i = 0; i < kWin; ++i) {
                if (!done && !failed) {
                    u64 sv = sw[i];
                    if ((uint32_t)(sv >> 34) != epoch || ((sv >> 32) & 3ull) == 0ull) {      // not published yet
                        sv = os_wait_status(status + (uint64_t)(t - i) * 256 + d, epoch);
                        failed = (sv == 0);
                        ++total_spins;
                    }
                    if (!failed) {
                        ++hops;
                        before += (uint32_t)sv;
                        if (((sv >> 32) & 3ull) == kFlagPrefix) done = true;
                    }
                }
            }
        }
#ifdef GPE_OS_STAMPS
        if ((s_tile & 63u) == 7u && d == 0) { atomicAdd(&ctl[24], hops); atomicAdd(&ctl[25], total_spins); atomicAdd(&ctl[26], 1u); }
#else
        (void)hops; (void)total_spins;
#endif
        if (failed) atomicOr(&ctl[4], 1u);
        __hip_atomic_store(mine, status_pack(epoch, kFlagPrefix, before + count), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        s_delta[d] = (hist_src ? digit_base : bases4[pass * 256 + d]) + before - s_excl[d];
    }
    __syncthreads();
    OS_STAMP(3);

    // write-out: slot j of digit d goes to s_delta[d] + j -- each digit's run is one contiguous store
    uint32_t dst[kOsItems];
#pragma unroll
    for (int q = 0; q < kOsItems; ++q) {
        const uint32_t j = threadIdx.x + (uint32_t)q * kOsBlock;
        dst[q] = 0;
        if (j < tile_n) {
            const uint32_t kk = s_stage[j];
            dst[q] = s_delta[(kk >> shift) & 255u] + j;
            keys_out[dst[q]] = kk;
            if (table && kk < table_entries) {
                // Last pass of the native step's sort: the keys are block indices and land in their final
                // places, equal keys next to each other (in the tile's staging order too, which is grouped by
                // this digit and otherwise keeps the input order = sorted by the lower digits).  The first and
                // the last key of a run bound the block's range from this tile; min / max over the tiles is
                // the block table entry (initialised to (0xFFFFFFFF, 0) by the hash kernel) -- no table launch.
                const uint32_t before = j > 0 ? s_stage[j - 1] : ~kk;
                const uint32_t after = j + 1 < tile_n ? s_stage[j + 1] : ~kk;
                if (before != kk) atomicMin(&table[kk].x, dst[q]);
                if (after != kk) atomicMax(&table[kk].y, dst[q] + 1u);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kOsItems; ++k) {
        const uint64_t idx = wave_base + (uint64_t)k * 64 + lane;
        if (idx < n) s_stage[slot[k]] = val[k];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kOsItems; ++q) {
        const uint32_t j = threadIdx.x + (uint32_t)q * kOsBlock;
        if (j < tile_n) vals_out[dst[q]] = s_stage[j];
    }
    OS_STAMP(4);
}

static int os_items(uint64_t n) { return n <= kOsSmallSort ? kOsItemsSmall : kOsItems; }
static uint64_t os_tiles(uint64_t n)
{
    const uint64_t tile = (uint64_t)kOsBlock * os_items(n);
    return (n + tile - 1) / tile;
}

gpe_status onesweep_reserve(gpe_ctx *c, uint64_t n)
{
    OnesweepWorkspace &ws = c->os_ws;
    // sorts of up to n keys: the small-tile sorts among them may have more tiles than the largest one
    const uint64_t small_n = n < kOsSmallSort ? n : kOsSmallSort;
    const uint64_t need = std::max(os_tiles(n), os_tiles(small_n)) * 256 + 256;
    if (ws.status_cap < need) {
        if (ws.status) GPE_HIP(c, hipFree(ws.status));
        ws.status = nullptr; ws.status_cap = 0;
        GPE_HIP(c, hipMalloc((void **)&ws.status, need * sizeof(uint64_t)));
        GPE_HIP(c, hipMemsetAsync(ws.status, 0, need * sizeof(uint64_t), c->stream));   // epoch 0 = never
        ws.status_cap = need;
    }
    if (!ws.hist4) {
        // kHistCopies histograms (the fused hash kernel spreads its flush atomics over them; the generic
        // path uses copy 0), the digit bases, the control words
        // (two sets of copies: the native step alternates between them)
        const size_t words = 2 * (size_t)kHistCopies * 4 * 256 + 4 * 256 + 64 + 4 * 256;
        GPE_HIP(c, hipMalloc((void **)&ws.hist4, words * sizeof(uint32_t)));
        ws.bases4 = ws.hist4 + 2 * (size_t)kHistCopies * 4 * 256;
        ws.ctl = ws.bases4 + 4 * 256;
        ws.hist_plain = ws.ctl + 64;     // the histograms of sorts that bring none (the native step's sets stay untouched)
        GPE_HIP(c, hipMemsetAsync(ws.hist4, 0, words * sizeof(uint32_t), c->stream));
    }
    return GPE_OK;
}

void onesweep_release(gpe_ctx *c)
{
    OnesweepWorkspace &ws = c->os_ws;
    if (ws.status) (void)hipFree(ws.status);
    if (ws.hist4) (void)hipFree(ws.hist4);
    ws = OnesweepWorkspace();
}

gpe_status onesweep_zero_hist(gpe_ctx *c)
{
    GPE_HIP(c, hipMemsetAsync(c->os_ws.hist_plain, 0, 4 * 256 * sizeof(uint32_t), c->stream));
    return GPE_OK;
}

// Passes over digits [0, passes): (keys, vals) ping-pong with (keys_b, vals_b).  Returns through
// *out_keys/*out_vals the buffers holding the result (the caller's when `passes` is even).
// hist_ready: hist4 already holds the four digit histograms of `keys` (fused producer);
// bases_ready: that producer has also reset the tickets, and the passes take their digit bases from hist_src
// (kHistCopies histogram copies, summed and scanned by every tile).
// iota_vals: the payload is the identity permutation and `vals` need not be initialised.
gpe_status onesweep_sort(gpe_ctx *c, uint32_t *keys, uint32_t *vals, uint32_t *keys_b, uint32_t *vals_b,
                         uint64_t n, int passes, bool hist_ready, bool iota_vals, uint32_t **out_keys,
                         uint32_t **out_vals, bool bases_ready, uint2 *table, uint32_t table_entries,
                         const uint32_t *hist_src, const OnesweepGate *gate)
{
    if (out_keys) *out_keys = keys;
    if (out_vals) *out_vals = vals;
    if (n == 0) return GPE_OK;
    if (n > 0xffffffffull) return fail(c, GPE_ERR_INVALID_ARG, "onesweep: n must be < 2^32");
    if (passes < 1 || passes > 4) return fail(c, GPE_ERR_INVALID_ARG, "onesweep: passes must be 1..4");
    OnesweepWorkspace &ws = c->os_ws;
    const uint64_t tiles = os_tiles(n);
    if (!hist_ready) {
        Scope s(c, "sort/hist");
        GPE_TRY(onesweep_zero_hist(c));
        hipLaunchKernelGGL(k_os_hist4, dim3(stream_grid(n, kStreamBlock)), dim3(kStreamBlock), 0, c->stream, keys, n,
                           ws.hist_plain);
        GPE_HIP(c, hipGetLastError());
    }
    if (!(hist_ready && bases_ready)) {
        Scope s(c, "sort/prepare");
        hipLaunchKernelGGL(k_os_prepare, dim3(1), dim3(256), 0, c->stream, hist_ready ? ws.hist4 : ws.hist_plain, ws.bases4,
                           ws.ctl);
        GPE_HIP(c, hipGetLastError());
    }
    uint32_t *ka = keys, *va = vals, *kb = keys_b, *vb = vals_b;
    for (int p = 0; p < passes; ++p) {
        ws.epoch += 1;
        if (ws.epoch >= (1u << 30)) {             // epoch field is 30 bits: restart from a clean array
            GPE_HIP(c, hipMemsetAsync(ws.status, 0, ws.status_cap * sizeof(uint64_t), c->stream));
            ws.epoch = 1;
        }
        Scope s(c, "sort/onesweep");
        const bool iota = (p == 0 && iota_vals);
        const bool small = os_items(n) == kOsItemsSmall;
        const auto kern = small ? (iota ? k_os_pass<true, kOsItemsSmall> : k_os_pass<false, kOsItemsSmall>)
                                : (iota ? k_os_pass<true, kOsItems> : k_os_pass<false, kOsItems>);
        const bool last = p == passes - 1;
        OnesweepGate g;                                            // (all NULL: an ordinary sort)
        if (gate) {
            g.need = gate->need;
            g.ticket_base = gate->ticket_base;
            if (p == 0) {
                g.key_copy = gate->key_copy; g.table_reset = gate->table_reset; g.table_pairs = gate->table_pairs;
                g.key_blocks_x = gate->key_blocks_x; g.key_div_magic = gate->key_div_magic;
                g.count_now = gate->count_now; g.sorted_count = gate->sorted_count;
            }
            if (last) { g.fresh = gate->fresh; g.sorts = gate->sorts; g.sorts_seen = gate->sorts_seen; }
        }
        hipLaunchKernelGGL(kern, dim3((uint32_t)tiles), dim3(kOsBlock), 0, c->stream, ka, va, kb, vb, n,
                           (uint32_t)(8 * p), (uint32_t)p, ws.bases4, (u64 *)ws.status, ws.ctl, ws.epoch,
                           last ? table : nullptr, table_entries, (hist_ready && bases_ready) ? hist_src : nullptr, g);
        GPE_HIP(c, hipGetLastError());
        uint32_t *t = ka; ka = kb; kb = t;
        t = va; va = vb; vb = t;
    }
    if (out_keys) *out_keys = ka;
    if (out_vals) *out_vals = va;
#ifdef GPE_OS_STAMPS
    {
        static int calls = 0;
        if (++calls % 10 == 0) {
            uint32_t h[64];
            (void)hipStreamSynchronize(c->stream);
            (void)hipMemcpy(h, ws.ctl, sizeof(h), hipMemcpyDeviceToHost);
            const double nt = h[26] ? (double)h[26] : 1.0;
            fprintf(stderr, "[os stamps] n=%llu passes=%d sampled=%u  load %.0f  rank %.0f  reorder %.0f  lookback %.0f  write %.0f cyc;"
                            " hops/tile %.1f spins/tile %.1f\n", (unsigned long long)n, passes, h[26], h[16] / nt, h[17] / nt,
                    h[18] / nt, h[19] / nt, h[20] / nt, h[24] / nt, h[25] / nt);
        }
        (void)hipMemsetAsync(ws.ctl + 16, 0, 16 * sizeof(uint32_t), c->stream);
    }
#endif
    return GPE_OK;
}

}  // namespace gpe
