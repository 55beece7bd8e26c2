// k_grid.hip -- K5 build_cell_ids_array (grid/grid.wgsl:39-97), the reference's (cell, object)
// pair list: slot 0 = home cell, then the phantom cells in y-major neighbour order, rest UNUSED.
#include "gpe_internal.h"

namespace gpe {

// R pos 8 + radius 4 + old object ids 16, W cell ids 16 + object ids 16 per particle; every
// access is one 16-B (or 8/4-B) coalesced vector access per lane.
__global__ __launch_bounds__(kStreamBlock) void k_build_cell_ids(const float2 *__restrict__ pos,
                                                                  const float *__restrict__ radius,
                                                                  uint64_t n, float cell_size,
                                                                  uint4 *__restrict__ cell_ids4,
                                                                  uint4 *__restrict__ object_ids4)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float2 p = pos[i];
        const float r = radius[i];
        const float sq_radius = r * r;                                  // :50
        const int32_t hx = cell_coord(p.x, cell_size);                  // :53
        const int32_t hy = cell_coord(p.y, cell_size);
        const uint32_t obj = (uint32_t)i;

        uint32_t cid[4];
        uint32_t oid[4];
        // Unused object-id slots keep whatever the buffer held (:84-85 write used slots only;
        // tests/grid.rs:43,54 see the initial zeros there).
        const uint4 old = object_ids4[i];
        oid[0] = old.x; oid[1] = old.y; oid[2] = old.z; oid[3] = old.w;

        cid[0] = morton_encode(hx, hy);                                 // :62-64
        oid[0] = obj;
        uint32_t p_cell_count = 0;
#pragma unroll
        for (int y = -1; y <= 1; ++y) {                                 // :68
#pragma unroll
            for (int x = -1; x <= 1; ++x) {                             // :69
                if (x == 0 && y == 0) continue;                         // :70-74
                const int32_t nx = (int32_t)((uint32_t)hx + (uint32_t)x);
                const int32_t ny = (int32_t)((uint32_t)hy + (uint32_t)y);
                if (is_obj_in_cell(p.x, p.y, sq_radius, nx, ny, cell_size)) {   // :79
                    p_cell_count++;                                     // :82
                    // 2r < cell_size => at most 3 phantom cells; beyond that the reference
                    // overruns the next particle's slots, which is refused here (as in the oracle).
                    if (p_cell_count < GPE_MAX_CELLS_PER_OBJECT) {
                        const uint32_t h = morton_encode(nx, ny);
                        // static slot selection keeps cid/oid in registers
                        if (p_cell_count == 1) { cid[1] = h; oid[1] = obj; }
                        else if (p_cell_count == 2) { cid[2] = h; oid[2] = obj; }
                        else { cid[3] = h; oid[3] = obj; }
                    }
                }
            }
        }
        if (p_cell_count < 1) cid[1] = kUnused;                         // :92-94
        if (p_cell_count < 2) cid[2] = kUnused;
        if (p_cell_count < 3) cid[3] = kUnused;

        cell_ids4[i] = make_uint4(cid[0], cid[1], cid[2], cid[3]);
        object_ids4[i] = make_uint4(oid[0], oid[1], oid[2], oid[3]);
    }
}

gpe_status launch_build_cell_ids(gpe_ctx *c, const float2 *pos, const float *radius, uint64_t n,
                                 float cell_size, uint32_t *cell_ids, uint32_t *object_ids)
{
    if (n == 0) return GPE_OK;
    Scope s(c, "Build cell ids");   // grid.rs:324
    hipLaunchKernelGGL(k_build_cell_ids, dim3(stream_grid(n)), dim3(kStreamBlock), 0, c->stream, pos,
                       radius, n, cell_size, reinterpret_cast<uint4 *>(cell_ids),
                       reinterpret_cast<uint4 *>(object_ids));
    GPE_HIP(c, hipGetLastError());
    return GPE_OK;
}

}  // namespace gpe
