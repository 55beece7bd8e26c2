// k_radix_sort.hip -- stable LSD radix sort of (u32 key, u32 payload) pairs, 8 bits x 4 passes,
// the replacement for the reference's GPUSorter (utils/radix_sort/radix_sort.rs:199-217,
// radix_sort.wgsl:23-186).
//
// Per pass: (a) per-tile digit counts, (b) device scan of the [digit][tile] count matrix,
// (c) scatter.  The scatter ranks keys with wave64 ballots (match-any on the 8 digit bits, rank =
// popcount of lower lanes) so the order inside a tile is the input order => the sort is stable,
// like the reference's per-bucket ballot masks (radix_sort.wgsl:160-176).  A tile is reordered in
// LDS so that each digit's run leaves the CU as one contiguous, coalesced store.
// The reference's scatter re-reads every workgroup's histogram in every workgroup
// (radix_sort.wgsl:99-112, O(workgroups^2)); here the global bases come from one scan.
#include "gpe_internal.h"

namespace gpe {

constexpr int kSortBlock = 256;                       // 4 waves
constexpr int kSortWaves = kSortBlock / 64;
constexpr int kSortItems = 16;                        // keys per lane
constexpr int kSortTile = kSortBlock * kSortItems;    // 4096 keys per workgroup
constexpr int kWaveSpan = 64 * kSortItems;            // contiguous keys owned by one wave

// Lanes holding the same 8-bit digit (among `valid` lanes).  8 ballots, one per digit bit.
__device__ __forceinline__ uint64_t match_digit(uint32_t d, bool valid)
{
    uint64_t m = ballot64(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t bal = ballot64(bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}

// (a) per-tile digit counts, stored digit-major: counts[d * tiles + tile]
__global__ __launch_bounds__(kSortBlock) void k_sort_count(const uint32_t *__restrict__ keys, uint64_t n,
                                                            uint32_t shift, uint32_t tiles,
                                                            uint32_t *__restrict__ counts)
{
    __shared__ uint32_t s_hist[256];
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    const int lane = lane_id();
    const int w = (int)(threadIdx.x >> 6);
    const uint64_t wave_base = (uint64_t)blockIdx.x * kSortTile + (uint64_t)w * kWaveSpan;
#pragma unroll 4
    for (int k = 0; k < kSortItems; ++k) {
        const uint64_t idx = wave_base + (uint64_t)k * 64 + lane;
        const bool valid = idx < n;
        const uint32_t key = valid ? keys[idx] : 0u;
        const uint32_t d = (key >> shift) & 255u;
        const uint64_t m = match_digit(d, valid);
        // one LDS atomic per distinct digit per wave-round (nearly sorted keys put all 64 lanes
        // in one bin; per-lane atomics would serialise)
        if (valid && popc_below_lane(m) == 0) atomicAdd(&s_hist[d], (uint32_t)__popcll(m));
    }
    __syncthreads();
    counts[(uint64_t)threadIdx.x * tiles + blockIdx.x] = s_hist[threadIdx.x];
}

// (c) scatter one tile.  excl_counts = INCLUSIVE scan of the count matrix; the exclusive base of
// entry e is scan[e-1].
__global__ __launch_bounds__(kSortBlock) void k_sort_scatter(const uint32_t *__restrict__ keys_in,
                                                              const uint32_t *__restrict__ vals_in,
                                                              uint32_t *__restrict__ keys_out,
                                                              uint32_t *__restrict__ vals_out, uint64_t n,
                                                              uint32_t shift, uint32_t tiles,
                                                              const uint32_t *__restrict__ scanned)
{
    __shared__ uint32_t s_keys[kSortTile];
    __shared__ uint32_t s_vals[kSortTile];
    __shared__ uint32_t s_whist[kSortWaves][256];   // per-wave digit counters, then wave offsets
    __shared__ uint32_t s_excl[256];                // first slot of each digit inside the tile
    __shared__ uint32_t s_delta[256];               // global base - s_excl (mod 2^32)
    __shared__ uint32_t s_w[4];

    const int lane = lane_id();
    const int w = (int)(threadIdx.x >> 6);
    const uint64_t tile_base = (uint64_t)blockIdx.x * kSortTile;
    const uint64_t wave_base = tile_base + (uint64_t)w * kWaveSpan;
    const uint32_t tile_n = (uint32_t)((n - tile_base < (uint64_t)kSortTile) ? (n - tile_base) : kSortTile);

#pragma unroll
    for (int i = 0; i < kSortWaves; ++i) s_whist[i][threadIdx.x] = 0;
    __syncthreads();

    uint32_t key[kSortItems], val[kSortItems];
    uint16_t rank[kSortItems];
    uint32_t *wh = s_whist[w];
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const uint64_t idx = wave_base + (uint64_t)k * 64 + lane;
        const bool valid = idx < n;
        key[k] = valid ? keys_in[idx] : 0xffffffffu;
        val[k] = valid ? vals_in[idx] : 0u;
    }
    // Rank: (round k, lane) is the input order inside the wave's span, so
    // rank = keys of this digit in earlier rounds + lower lanes of this round.
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const uint64_t idx = wave_base + (uint64_t)k * 64 + lane;
        const bool valid = idx < n;
        const uint32_t d = (key[k] >> shift) & 255u;
        const uint64_t m = match_digit(d, valid);
        const uint32_t below = popc_below_lane(m);
        const uint32_t pre = wave_lds_load(&wh[d]);       // all lanes of the group read the same word
        wave_lds_order();
        if (valid && below == 0) wave_lds_store(&wh[d], pre + (uint32_t)__popcll(m));   // group leader publishes
        wave_lds_order();
        rank[k] = (uint16_t)(pre + below);
    }
    __syncthreads();

    // Per digit (one thread each): offsets of the four waves inside the digit's run, tile total.
    {
        const uint32_t d = threadIdx.x;
        uint32_t c0 = s_whist[0][d], c1 = s_whist[1][d], c2 = s_whist[2][d], c3 = s_whist[3][d];
        s_whist[0][d] = 0; s_whist[1][d] = c0; s_whist[2][d] = c0 + c1; s_whist[3][d] = c0 + c1 + c2;
        const uint32_t total = c0 + c1 + c2 + c3;
        const uint32_t excl = block256_exclusive_scan(total, s_w, nullptr);
        s_excl[d] = excl;
        const uint64_t e = (uint64_t)d * tiles + blockIdx.x;
        const uint32_t gbase = (e > 0) ? scanned[e - 1] : 0u;
        s_delta[d] = gbase - excl;
    }
    __syncthreads();

    // Reorder the tile in LDS by digit (stable).
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const uint64_t idx = wave_base + (uint64_t)k * 64 + lane;
        if (idx < n) {
            const uint32_t d = (key[k] >> shift) & 255u;
            const uint32_t slot = s_excl[d] + s_whist[w][d] + rank[k];
            s_keys[slot] = key[k];
            s_vals[slot] = val[k];
        }
    }
    __syncthreads();

    // Coalesced write-out: consecutive slots of one digit are consecutive global addresses.
    for (uint32_t j = threadIdx.x; j < tile_n; j += kSortBlock) {
        const uint32_t kk = s_keys[j];
        const uint32_t d = (kk >> shift) & 255u;
        const uint32_t dst = s_delta[d] + j;
        keys_out[dst] = kk;
        vals_out[dst] = s_vals[j];
    }
}

// 256-bin histogram of one digit over all keys (GPUSorter::build_histogram stand-in)
__global__ __launch_bounds__(kSortBlock) void k_sort_global_hist(const uint32_t *__restrict__ keys, uint64_t n,
                                                                  uint32_t shift, uint32_t *__restrict__ hist)
{
    __shared__ uint32_t s_hist[256];
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t rounds = (n + stride - 1) / stride;
    for (uint64_t r = 0; r < rounds; ++r) {
        const uint64_t idx = r * stride + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        const bool valid = idx < n;
        const uint32_t d = valid ? ((keys[idx] >> shift) & 255u) : 0u;
        const uint64_t m = match_digit(d, valid);
        if (valid && popc_below_lane(m) == 0) atomicAdd(&s_hist[d], (uint32_t)__popcll(m));
    }
    __syncthreads();
    if (s_hist[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_hist[threadIdx.x]);
}

static uint64_t sort_tiles(uint64_t n) { return (n + kSortTile - 1) / kSortTile; }

gpe_status sort_reserve(gpe_ctx *c, uint64_t n)
{
    SortWorkspace &ws = c->sort_ws;
    if (ws.cap < n) {
        if (ws.keys_b) GPE_HIP(c, hipFree(ws.keys_b));
        if (ws.vals_b) GPE_HIP(c, hipFree(ws.vals_b));
        ws.keys_b = ws.vals_b = nullptr; ws.cap = 0;
        GPE_HIP(c, hipMalloc((void **)&ws.keys_b, (n + 16) * sizeof(uint32_t)));
        GPE_HIP(c, hipMalloc((void **)&ws.vals_b, (n + 16) * sizeof(uint32_t)));
        ws.cap = n;
    }
    const uint64_t need = 256ull * sort_tiles(n) + 16;
    if (ws.counts_cap < need) {
        if (ws.counts) GPE_HIP(c, hipFree(ws.counts));
        ws.counts = nullptr; ws.counts_cap = 0;
        GPE_HIP(c, hipMalloc((void **)&ws.counts, need * sizeof(uint32_t)));
        ws.counts_cap = need;
    }
    if (!ws.hist4) GPE_HIP(c, hipMalloc((void **)&ws.hist4, 4 * 256 * sizeof(uint32_t)));
    GPE_TRY(scan_reserve(c, need));
    GPE_TRY(onesweep_reserve(c, n));
    return GPE_OK;
}

void sort_release(gpe_ctx *c)
{
    SortWorkspace &ws = c->sort_ws;
    if (ws.keys_b) (void)hipFree(ws.keys_b);
    if (ws.vals_b) (void)hipFree(ws.vals_b);
    if (ws.counts) (void)hipFree(ws.counts);
    if (ws.hist4) (void)hipFree(ws.hist4);
    ws = SortWorkspace();
}

// One stable pass (ka,va) -> (kb,vb) on the digit at `shift`.  Workspace must be reserved.
gpe_status sort_scatter_pass(gpe_ctx *c, const uint32_t *ka, const uint32_t *va, uint32_t *kb,
                             uint32_t *vb, uint64_t n, uint32_t shift)
{
    if (n == 0) return GPE_OK;
    if (n > 0xffffffffull) return fail(c, GPE_ERR_INVALID_ARG, "sort: n must be < 2^32");
    const uint64_t tiles = sort_tiles(n);
    uint32_t *counts = c->sort_ws.counts;
    {
        Scope s(c, "sort/count");
        hipLaunchKernelGGL(k_sort_count, dim3((uint32_t)tiles), dim3(kSortBlock), 0, c->stream, ka, n, shift,
                           (uint32_t)tiles, counts);
        GPE_HIP(c, hipGetLastError());
    }
    {
        Scope s(c, "sort/scan");
        GPE_TRY(inclusive_scan(c, counts, 256ull * tiles));
    }
    {
        Scope s(c, "sort/scatter");
        hipLaunchKernelGGL(k_sort_scatter, dim3((uint32_t)tiles), dim3(kSortBlock), 0, c->stream, ka, va, kb,
                           vb, n, shift, (uint32_t)tiles, (const uint32_t *)counts);
        GPE_HIP(c, hipGetLastError());
    }
    return GPE_OK;
}

// radix_sort.rs:199-217: four ping-pong passes, result back in (keys, vals).
gpe_status sort_pairs(gpe_ctx *c, uint32_t *keys, uint32_t *vals, uint64_t n)
{
    if (n == 0) return GPE_OK;
    if (c->use_onesweep)   // four passes: the result lands back in (keys, vals)
        return onesweep_sort(c, keys, vals, c->sort_ws.keys_b, c->sort_ws.vals_b, n, 4, false, false, nullptr,
                             nullptr);
    uint32_t *ka = keys, *va = vals, *kb = c->sort_ws.keys_b, *vb = c->sort_ws.vals_b;
    for (uint32_t pass = 0; pass < 4; ++pass) {
        GPE_TRY(sort_scatter_pass(c, ka, va, kb, vb, n, pass * 8u));
        uint32_t *t = ka; ka = kb; kb = t;
        t = va; va = vb; vb = t;
    }
    return GPE_OK;
}

gpe_status sort_histogram(gpe_ctx *c, const uint32_t *keys, uint64_t n, uint32_t shift, uint32_t *hist256)
{
    GPE_HIP(c, hipMemsetAsync(hist256, 0, 256 * sizeof(uint32_t), c->stream));
    if (n == 0) return GPE_OK;
    hipLaunchKernelGGL(k_sort_global_hist, dim3(stream_grid(n, kSortBlock)), dim3(kSortBlock), 0, c->stream,
                       keys, n, shift, hist256);
    GPE_HIP(c, hipGetLastError());
    return GPE_OK;
}

}  // namespace gpe
