// k_collision.hip -- the reference's collision-cell list (K6 count_objects_for_each_chunk, K10
// build_collision_cells_array; physics/collision_cell_builder.wgsl:27-189) and its four-colour
// Gauss-Seidel solver (K11 solve_collisions; physics/collision_solver.wgsl:26-118) on the sorted
// (cell, object) pair list.  This is the COMPAT pipeline: it materialises exactly the buffers the
// reference materialises.
#include "gpe_internal.h"

namespace gpe {

constexpr uint32_t kChunk = GPE_COUNTING_CHUNK_SIZE;   // 4 sorted entries per thread

// A "collision cell" is a maximal run of equal keys, key != UNUSED, length >= 2; it belongs to the
// chunk its first entry lies in.  The WGSL walks each chunk with a small state machine
// (collision_cell_builder.wgsl:50-80) whose effect is exactly this predicate (the oracle follows
// the state machine line by line; tests/test_oracle_properties.py checks the equivalence).
__device__ __forceinline__ bool starts_run(uint32_t prev, uint32_t cur, uint32_t next, bool has_next)
{
    return cur != kUnused && prev != cur && has_next && next == cur;
}

// Loads the 4 keys of a chunk plus the key before and after it (guarded: the WGSL relies on
// wgpu's robust buffer access for cell_ids[first_idx-1], collision_cell_builder.wgsl:40).
__device__ __forceinline__ uint32_t chunk_flags(const uint32_t *__restrict__ cell_ids, uint64_t total,
                                                uint64_t chunk)
{
    const uint64_t first = chunk * kChunk;
    uint32_t k[6];
    k[0] = (first >= 1) ? cell_ids[first - 1] : kUnused;   // :40 select(UNUSED, ...)
    if (first + 4 <= total) {
        const uint4 q = *reinterpret_cast<const uint4 *>(cell_ids + first);
        k[1] = q.x; k[2] = q.y; k[3] = q.z; k[4] = q.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) k[1 + j] = (first + j < total) ? cell_ids[first + j] : kUnused;
    }
    const bool has5 = first + 4 < total;
    k[5] = has5 ? cell_ids[first + 4] : kUnused;
    uint32_t flags = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool in_range = first + j < total;
        const bool has_next = first + j + 1 < total;
        if (in_range && starts_run(k[j], k[j + 1], k[j + 2], has_next)) flags |= 1u << j;
    }
    return flags;
}

// K6: chunk_obj_count[chunk] = number of collision cells starting in the chunk.
__global__ __launch_bounds__(kStreamBlock) void k_count_chunks(const uint32_t *__restrict__ cell_ids,
                                                                uint64_t total, uint64_t num_chunks,
                                                                uint32_t *__restrict__ chunk_counts)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < num_chunks; c += stride)
        chunk_counts[c] = (uint32_t)__popc(chunk_flags(cell_ids, total, c));
}

// K10: collision_cells[scan[chunk-1] ...] = start index of each collision cell of the chunk;
// thread 0 also writes the indirect dispatch args (collision_cell_builder.wgsl:96-109).
__global__ __launch_bounds__(kStreamBlock) void k_build_collision_cells(
    const uint32_t *__restrict__ cell_ids, uint64_t total, const uint32_t *__restrict__ scanned,
    uint64_t num_chunks, uint32_t *__restrict__ collision_cells, uint32_t *__restrict__ indirect_args)
{
    const uint64_t gid0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid0 == 0) {
        const uint32_t total_items = scanned[num_chunks - 1];           // :100
        indirect_args[0] = (total_items + 64u - 1u) / 64u;              // :103
        indirect_args[1] = 1u;
        indirect_args[2] = 1u;
    }
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t c = gid0; c < num_chunks; c += stride) {
        const uint32_t start_index = (c >= 1) ? scanned[c - 1] : 0u;    // :128
        const uint32_t end_index = scanned[c];                          // :129
        if (end_index == start_index) continue;                         // :133-136
        const uint32_t flags = chunk_flags(cell_ids, total, c);
        uint32_t w = start_index;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (flags & (1u << j)) collision_cells[w++] = (uint32_t)(c * kChunk + j);   // :174
    }
}

// K11: one colour pass.  One lane per collision cell (grid-stride: HIP has no indirect dispatch,
// so K = scanned[num_chunks-1] is read on the device), sequential pair loop in run order with live
// positions (collision_solver.wgsl:66-118).  Cells of one colour never share a particle.
__global__ __launch_bounds__(kStreamBlock) void k_solve_color(
    const uint32_t *__restrict__ collision_cells, const uint32_t *__restrict__ scanned, uint64_t num_chunks,
    const uint32_t *__restrict__ cell_ids, const uint32_t *__restrict__ object_ids, uint64_t total,
    float2 *pos, const float *__restrict__ radius, float stiffness, uint32_t color)
{
    const uint32_t num_collision_cells = scanned[num_chunks - 1];       // :48-53
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; tid < num_collision_cells;
         tid += stride) {
        const uint32_t start = collision_cells[tid];                    // :38
        const uint32_t cell_hash = cell_ids[start];                     // :39
        if (cell_color(cell_hash) != color) continue;                   // :40-43
        for (uint64_t i = start; i < total; ++i) {                      // :68
            if (cell_ids[i] != cell_hash) break;                        // :69-71
            const uint32_t a = object_ids[i];                           // :72
            const float r1 = radius[a];
            // positions[a] is only written by this lane during this pass: keep it in registers
            // across the inner loop (the WGSL re-reads the same value, :85).
            float2 p1 = pos[a];
            bool dirty = false;
            for (uint64_t j = i + 1; j < total; ++j) {                  // :77
                if (cell_ids[j] != cell_hash) break;                    // :78-81
                const uint32_t b = object_ids[j];                       // :83
                const float2 p2 = pos[b];                               // :86
                const float r2 = radius[b];                             // :88
                const float vx = p1.x - p2.x, vy = p1.y - p2.y;         // :91
                const float distance = sqrtf(vx * vx + vy * vy);        // :93
                const float radius_sum = r1 + r2;                       // :61
                if (radius_sum * radius_sum > distance * distance && distance > 0.0001f) {   // :95
                    const float depth = radius_sum - distance;          // :97
                    const float cx = ((vx / distance) * depth) * stiffness;   // :98,101
                    const float cy = ((vy / distance) * depth) * stiffness;
                    const float inv1 = 1.0f / r1, inv2 = 1.0f / r2;     // :103-104
                    const float w1 = inv1 / (inv1 + inv2);              // :107
                    const float w2 = inv2 / (inv1 + inv2);              // :108
                    p1.x = p1.x + cx * w1;                              // :110
                    p1.y = p1.y + cy * w1;
                    float2 np2;
                    np2.x = p2.x - cx * w2;                             // :111
                    np2.y = p2.y - cy * w2;
                    pos[b] = np2;
                    dirty = true;
                }
            }
            if (dirty) pos[a] = p1;
        }
    }
}

gpe_status launch_count_chunks(gpe_ctx *c, const uint32_t *cell_ids, uint64_t total, uint32_t *chunk_counts)
{
    const uint64_t num_chunks = (total + kChunk - 1) / kChunk;
    if (num_chunks == 0) return GPE_OK;
    Scope s(c, "Collision cell count objects per chunk");   // collision_cell_builder.rs:216
    hipLaunchKernelGGL(k_count_chunks, dim3(stream_grid(num_chunks)), dim3(kStreamBlock), 0, c->stream,
                       cell_ids, total, num_chunks, chunk_counts);
    GPE_HIP(c, hipGetLastError());
    return GPE_OK;
}

gpe_status launch_build_collision_cells(gpe_ctx *c, const uint32_t *cell_ids, uint64_t total,
                                        const uint32_t *scanned, uint64_t num_chunks,
                                        uint32_t *collision_cells, uint32_t *indirect_args)
{
    if (num_chunks == 0) return GPE_OK;
    Scope s(c, "Build collision cells");   // collision_cell_builder.rs:233
    hipLaunchKernelGGL(k_build_collision_cells, dim3(stream_grid(num_chunks)), dim3(kStreamBlock), 0,
                       c->stream, cell_ids, total, scanned, num_chunks, collision_cells, indirect_args);
    GPE_HIP(c, hipGetLastError());
    return GPE_OK;
}

gpe_status launch_solve_color(gpe_ctx *c, const uint32_t *collision_cells, const uint32_t *scanned,
                              uint64_t num_chunks, const uint32_t *cell_ids, const uint32_t *object_ids,
                              uint64_t total, float2 *pos, const float *radius, float stiffness,
                              uint32_t color)
{
    if (num_chunks == 0) return GPE_OK;
    static const char *names[4] = {"Solve Collisions - Color 1", "Solve Collisions - Color 2",
                                   "Solve Collisions - Color 3", "Solve Collisions - Color 4"};
    Scope s(c, names[color - 1]);   // collision_solver.rs:226
    // at most total/2 collision cells; the grid is capped and strides
    hipLaunchKernelGGL(k_solve_color, dim3(stream_grid(total / 2 + 1)), dim3(kStreamBlock), 0, c->stream,
                       collision_cells, scanned, num_chunks, cell_ids, object_ids, total, pos, radius,
                       stiffness, color);
    GPE_HIP(c, hipGetLastError());
    return GPE_OK;
}

}  // namespace gpe
