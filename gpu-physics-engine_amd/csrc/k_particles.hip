// k_particles.hip -- K1 home cell ids, K4 rearrange, K12 Verlet integration (gfx950).
// Streaming kernels: HBM-bound, grid-stride over a capped grid, 8-16 B per lane per access.
#include "gpe_internal.h"

namespace gpe {

// K1  particles/home_cell_ids.wgsl:16-34.  R 8 B, W 8 B per particle.
__global__ __launch_bounds__(kStreamBlock) void k_home_cell_ids(const float2 *__restrict__ pos,
                                                                 uint64_t n, float cell_size,
                                                                 uint32_t *__restrict__ home,
                                                                 uint32_t *__restrict__ ids)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float2 p = pos[i];
        home[i] = morton_encode(cell_coord(p.x, cell_size), cell_coord(p.y, cell_size));
        ids[i] = (uint32_t)i;
    }
}

// K4  particles/rearrange.wgsl:19-35.  Gather through the sorted particle ids into the copy set.
__global__ __launch_bounds__(kStreamBlock) void k_rearrange(const float2 *__restrict__ pos,
                                                             const float2 *__restrict__ prev,
                                                             const float *__restrict__ radius,
                                                             const uint32_t *__restrict__ ids,
                                                             uint64_t n, float2 *__restrict__ pos_out,
                                                             float2 *__restrict__ prev_out,
                                                             float *__restrict__ radius_out)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint32_t r = ids[i];
        float2 p = pos[r];
        float rad = radius[r];
        float2 q = prev[r];
        pos_out[i] = p;
        radius_out[i] = rad;
        prev_out[i] = q;
    }
}

// K12 particles/particle_integration.wgsl:25-77.  R 20 B, W 16 B per particle; two particles per
// lane per iteration so every access is 16 B/lane (pos, prev) or 8 B/lane (radius).
__global__ __launch_bounds__(kStreamBlock) void k_verlet(float2 *__restrict__ pos,
                                                          float2 *__restrict__ prev,
                                                          const float *__restrict__ radius,
                                                          uint64_t n, VerletParams P)
{
    const uint64_t pairs = n >> 1;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    float4 *pos4 = reinterpret_cast<float4 *>(pos);
    float4 *prev4 = reinterpret_cast<float4 *>(prev);
    const float2 *rad2 = reinterpret_cast<const float2 *>(radius);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += stride) {
        float4 c = pos4[i];
        float4 q = prev4[i];
        float2 r = rad2[i];
        float4 o;
        verlet_one(c.x, c.y, q.x, q.y, r.x, P, o.x, o.y);
        verlet_one(c.z, c.w, q.z, q.w, r.y, P, o.z, o.w);
        prev4[i] = c;                                        // :64 previous = current
        pos4[i] = o;                                         // :76
    }
    if ((n & 1ull) && blockIdx.x == 0 && threadIdx.x == 0) {
        uint64_t i = n - 1;
        float2 c = pos[i], q = prev[i];
        float2 o;
        verlet_one(c.x, c.y, q.x, q.y, radius[i], P, o.x, o.y);
        prev[i] = c;
        pos[i] = o;
    }
}

gpe_status launch_home_cell_ids(gpe_ctx *c, const float2 *pos, uint64_t n, float cell_size,
                                uint32_t *home, uint32_t *ids)
{
    if (n == 0) return GPE_OK;
    Scope s(c, "Particle home cells");   // particle_home_cell_ids_kernel.rs:134
    hipLaunchKernelGGL(k_home_cell_ids, dim3(stream_grid(n)), dim3(kStreamBlock), 0, c->stream, pos, n,
                       cell_size, home, ids);
    GPE_HIP(c, hipGetLastError());
    return GPE_OK;
}

gpe_status launch_rearrange(gpe_ctx *c, const float2 *pos, const float2 *prev, const float *radius,
                            const uint32_t *ids, uint64_t n, float2 *pos_out, float2 *prev_out,
                            float *radius_out)
{
    if (n == 0) return GPE_OK;
    Scope s(c, "Particle rearranging");  // particle_rearrange.rs:194
    hipLaunchKernelGGL(k_rearrange, dim3(stream_grid(n)), dim3(kStreamBlock), 0, c->stream, pos, prev,
                       radius, ids, n, pos_out, prev_out, radius_out);
    GPE_HIP(c, hipGetLastError());
    return GPE_OK;
}

VerletParams verlet_params(const gpe_ctx *c, float dt)
{
    VerletParams P;
    P.dt_squared = dt * dt;                   // particle_integration.wgsl:58
    P.world_w = c->cfg.world_width;
    P.world_h = c->cfg.world_height;
    P.acc_x = c->cfg.gravity_x;
    P.acc_y = c->cfg.gravity_y;
    P.mouse_pressed = (uint32_t)c->mouse_pressed;
    P.mouse_x = c->mouse_x;
    P.mouse_y = c->mouse_y;
    P.mouse_strength = c->cfg.mouse_strength;
    return P;
}

gpe_status launch_verlet(gpe_ctx *c, float2 *pos, float2 *prev, const float *radius, uint64_t n,
                         float dt)
{
    if (n == 0) return GPE_OK;
    Scope s(c, "Particle integration pass");  // particle_integration.rs:81
    const VerletParams P = verlet_params(c, dt);
    uint64_t pairs = (n + 1) >> 1;
    hipLaunchKernelGGL(k_verlet, dim3(stream_grid(pairs)), dim3(kStreamBlock), 0, c->stream, pos, prev,
                       radius, n, P);
    GPE_HIP(c, hipGetLastError());
    return GPE_OK;
}

}  // namespace gpe
