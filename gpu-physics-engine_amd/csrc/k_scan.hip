// k_scan.hip -- device-wide in-place inclusive u32 scan (wrap-around add), the replacement for
// the reference's PrefixSum (utils/prefix_sum/prefix_sum.rs:143-160, prefix_sum.wgsl:14-147).
//
// Reduce-then-scan over tiles of 4096 items: (1) per-tile sums, (2) recursive scan of the tile
// sums, (3) per-tile scan seeded with the preceding tiles' total.  Wave-level scans use wave64
// shuffles; no inter-workgroup communication inside a launch, so no forward-progress assumption.
// Traffic: 12 B/item (R 4 + R 4 + W 4); the reference's three passes move ~16 B/item.
#include "gpe_internal.h"

namespace gpe {

constexpr int kScanBlock = 256;
constexpr int kScanItems = 16;                       // 4 x uint4 per thread
constexpr int kScanTile = kScanBlock * kScanItems;   // 4096

// (1) tile sums
__global__ __launch_bounds__(kScanBlock) void k_scan_reduce(const uint32_t *__restrict__ data, uint64_t n,
                                                             uint32_t *__restrict__ tile_sums)
{
    __shared__ uint32_t s_w[4];
    const uint64_t base = (uint64_t)blockIdx.x * kScanTile;
    uint32_t sum = 0;
    const uint64_t t0 = base + (uint64_t)threadIdx.x * kScanItems;
    if (t0 + kScanItems <= n) {
        const uint4 *p = reinterpret_cast<const uint4 *>(data + t0);
#pragma unroll
        for (int k = 0; k < 4; ++k) { uint4 v = p[k]; sum += v.x + v.y + v.z + v.w; }
    } else {
        for (int k = 0; k < kScanItems; ++k) if (t0 + k < n) sum += data[t0 + k];
    }
    uint32_t inc = wave_inclusive_scan(sum);
    if (lane_id() == 63) s_w[threadIdx.x >> 6] = inc;
    __syncthreads();
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// (3) per-tile inclusive scan; tile b starts from scanned_sums[b-1] (inclusive scan of tile sums)
__global__ __launch_bounds__(kScanBlock) void k_scan_apply(uint32_t *__restrict__ data, uint64_t n,
                                                            const uint32_t *__restrict__ scanned_sums)
{
    __shared__ uint32_t s_w[4];
    const uint64_t base = (uint64_t)blockIdx.x * kScanTile;
    const uint32_t carry = (scanned_sums != nullptr && blockIdx.x > 0) ? scanned_sums[blockIdx.x - 1] : 0u;
    const uint64_t t0 = base + (uint64_t)threadIdx.x * kScanItems;
    uint32_t v[kScanItems];
    const bool full = (t0 + kScanItems <= n);
    if (full) {
        const uint4 *p = reinterpret_cast<const uint4 *>(data + t0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint4 q = p[k];
            v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) v[k] = (t0 + k < n) ? data[t0 + k] : 0u;
    }
    uint32_t run = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) { run += v[k]; v[k] = run; }
    const uint32_t excl = block256_exclusive_scan(run, s_w, nullptr) + carry;
    if (full) {
        uint4 *p = reinterpret_cast<uint4 *>(data + t0);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            p[k] = make_uint4(v[4 * k] + excl, v[4 * k + 1] + excl, v[4 * k + 2] + excl, v[4 * k + 3] + excl);
    } else {
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) if (t0 + k < n) data[t0 + k] = v[k] + excl;
    }
}

static uint64_t tiles_of(uint64_t n) { return (n + kScanTile - 1) / kScanTile; }

gpe_status scan_reserve(gpe_ctx *c, uint64_t n)
{
    ScanWorkspace &ws = c->scan_ws;
    size_t lvl = 0;
    for (uint64_t m = tiles_of(n); n > (uint64_t)kScanTile; ++lvl) {
        if (ws.level.size() <= lvl) { ws.level.push_back(nullptr); ws.cap.push_back(0); }
        if (ws.cap[lvl] < m) {
            if (ws.level[lvl]) GPE_HIP(c, hipFree(ws.level[lvl]));
            ws.level[lvl] = nullptr; ws.cap[lvl] = 0;
            uint64_t want = m + m / 2 + 16;
            GPE_HIP(c, hipMalloc((void **)&ws.level[lvl], want * sizeof(uint32_t)));
            ws.cap[lvl] = want;
        }
        n = m;
        m = tiles_of(n);
    }
    return GPE_OK;
}

void scan_release(gpe_ctx *c)
{
    for (uint32_t *p : c->scan_ws.level) if (p) (void)hipFree(p);
    c->scan_ws.level.clear();
    c->scan_ws.cap.clear();
}

static gpe_status scan_level(gpe_ctx *c, uint32_t *data, uint64_t n, size_t lvl)
{
    if (n == 0) return GPE_OK;
    const uint64_t tiles = tiles_of(n);
    if (tiles == 1) {
        hipLaunchKernelGGL(k_scan_apply, dim3(1), dim3(kScanBlock), 0, c->stream, data, n,
                           (const uint32_t *)nullptr);
        GPE_HIP(c, hipGetLastError());
        return GPE_OK;
    }
    if (tiles > 0x7fffffffull) return fail(c, GPE_ERR_INVALID_ARG, "scan: too many items");
    uint32_t *sums = c->scan_ws.level[lvl];
    hipLaunchKernelGGL(k_scan_reduce, dim3((uint32_t)tiles), dim3(kScanBlock), 0, c->stream, data, n, sums);
    GPE_HIP(c, hipGetLastError());
    GPE_TRY(scan_level(c, sums, tiles, lvl + 1));
    hipLaunchKernelGGL(k_scan_apply, dim3((uint32_t)tiles), dim3(kScanBlock), 0, c->stream, data, n,
                       (const uint32_t *)sums);
    GPE_HIP(c, hipGetLastError());
    return GPE_OK;
}

// Caller must have called scan_reserve(c, >= n) outside any no-allocation region.
gpe_status inclusive_scan(gpe_ctx *c, uint32_t *data, uint64_t n)
{
    return scan_level(c, data, n, 0);
}

}  // namespace gpe
