/*
 * gpe_oracle.c -- CPU restatement of the reference particle step.  TEST INFRASTRUCTURE ONLY.
 * See gpe_oracle.h for scope, citation convention and parity status.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fPIC -shared (oracle/Makefile).
 * All float arithmetic is binary32, one rounding per WGSL operator, in the WGSL's
 * left-to-right evaluation order.  No fmaf anywhere.
 */
#include "gpe_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * scalar helpers
 * ---------------------------------------------------------------------------------------- */

/* WGSL vec2<i32>(f32): truncation toward zero, saturating, NaN -> 0 (WGSL spec 'i32(e)').
 * The argument is already floor()ed at every call site. */
/* Host threads for the loops whose iterations are independent (they are separate GPU threads / workgroups in the
 * WGSL): 1 = plain serial code (default; what the parity tests use).  More threads give the same bits -- every
 * parallel loop below writes disjoint elements -- and serve as the multi-core CPU baseline of bench.py. */
static int g_threads = 1;
void orc_set_threads(int threads) { g_threads = threads < 1 ? 1 : threads; }
int orc_get_threads(void) { return g_threads; }
#define ORC_PARALLEL_FOR _Pragma("omp parallel for schedule(static) num_threads(g_threads) if(g_threads > 1)")

static inline int32_t f32_to_i32_sat(float f)
{
    if (f != f) return 0;
    if (f >= 2147483648.0f) return INT32_MAX;
    if (f <= -2147483648.0f) return INT32_MIN;
    return (int32_t)f;
}

static inline float clampf(float x, float lo, float hi)
{
    /* WGSL clamp(e, low, high) = min(max(e, low), high) */
    float m = (x > lo) ? x : lo;       /* max(x, lo) */
    return (m < hi) ? m : hi;          /* min(m, hi) */
}

float orc_compute_cell_size(float max_radius)
{
    return max_radius * 2.2f;          /* grid.rs:20,159-161 */
}

void orc_params_default(orc_params *p, float world_w, float world_h, float max_radius)
{
    memset(p, 0, sizeof(*p));
    p->world_w = world_w;
    p->world_h = world_h;
    p->cell_size = orc_compute_cell_size(max_radius);
    p->gravity_x = 0.0f;               /* particle_integration.wgsl:21 */
    p->gravity_y = 0.0f;
    p->stiffness = 0.6f;               /* collision_solver.wgsl:2 */
    p->mouse_strength = 150.0f;        /* particle_integration.wgsl:22 */
    p->mouse_pressed = 0;
    p->mouse_x = 0.0f;                 /* particle_integration.rs:44 */
    p->mouse_y = 0.0f;
}

/* grid.wgsl:101-108 */
uint32_t orc_split_by_bits(uint32_t n)
{
    uint32_t x = n & 0x0000FFFFu;
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}

/* grid.wgsl:112-114 ; u32(i32) is a two's-complement bit cast */
uint32_t orc_morton_encode(int32_t x, int32_t y)
{
    return orc_split_by_bits((uint32_t)x) | (orc_split_by_bits((uint32_t)y) << 1);
}

/* collision_solver.wgsl:123-130 */
uint32_t orc_unsplit_by_bits(uint32_t n)
{
    uint32_t x = n & 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu;
    x = (x | (x >> 8)) & 0x0000FFFFu;
    return x;
}

/* collision_solver.wgsl:55-58 */
uint32_t orc_cell_color(uint32_t cell_hash)
{
    uint32_t cx = orc_unsplit_by_bits(cell_hash);
    uint32_t cy = orc_unsplit_by_bits(cell_hash >> 1);
    return 1u + (cx % 2u) + (cy % 2u) * 2u;
}

/* ------------------------------------------------------------------------------------------
 * K1  home_cell_ids.wgsl:16-34
 * ---------------------------------------------------------------------------------------- */
void orc_create_home_cell_ids(const float *pos_xy, uint32_t n, float cell_size,
                              uint32_t *home_cell_ids, uint32_t *particle_ids)
{
    ORC_PARALLEL_FOR
    for (uint32_t obj_id = 0; obj_id < n; ++obj_id) {
        float px = pos_xy[2 * obj_id], py = pos_xy[2 * obj_id + 1];
        /* :27  vec2<i32>(floor(pos / cell_size)) -- a division, not a reciprocal multiply */
        int32_t cx = f32_to_i32_sat(floorf(px / cell_size));
        int32_t cy = f32_to_i32_sat(floorf(py / cell_size));
        home_cell_ids[obj_id] = orc_morton_encode(cx, cy);   /* :28-31 */
        particle_ids[obj_id] = obj_id;                        /* :33    */
    }
}

/* ------------------------------------------------------------------------------------------
 * K4  rearrange.wgsl:19-35
 * ---------------------------------------------------------------------------------------- */
void orc_rearrange(const float *pos_xy, const float *prev_xy, const float *radius,
                   const uint32_t *particle_ids, uint32_t n,
                   float *pos_out, float *prev_out, float *radius_out)
{
    ORC_PARALLEL_FOR
    for (uint32_t obj_id = 0; obj_id < n; ++obj_id) {
        uint32_t r = particle_ids[obj_id];                    /* :27 */
        pos_out[2 * obj_id] = pos_xy[2 * r];                  /* :28,32 */
        pos_out[2 * obj_id + 1] = pos_xy[2 * r + 1];
        radius_out[obj_id] = radius[r];                       /* :29,33 */
        prev_out[2 * obj_id] = prev_xy[2 * r];                /* :30,34 */
        prev_out[2 * obj_id + 1] = prev_xy[2 * r + 1];
    }
}

/* ------------------------------------------------------------------------------------------
 * K5  grid.wgsl:39-97 (+ is_obj_in_cell :117-129)
 * ---------------------------------------------------------------------------------------- */
static int is_obj_in_cell(float px, float py, float sq_radius, int32_t cx, int32_t cy, float cs)
{
    float lo_x = (float)cx * cs, lo_y = (float)cy * cs;       /* :118 */
    float hi_x = lo_x + cs, hi_y = lo_y + cs;                 /* :119 */
    float qx = clampf(px, lo_x, hi_x);                        /* :122 */
    float qy = clampf(py, lo_y, hi_y);
    float dx = px - qx, dy = py - qy;                         /* :125 */
    float dist_sq = dx * dx + dy * dy;                        /* :126 dot() */
    return dist_sq < sq_radius;                               /* :128 strict < */
}

void orc_build_cell_ids(const float *pos_xy, const float *radius, uint32_t n, float cell_size,
                        uint32_t *cell_ids, uint32_t *object_ids)
{
    ORC_PARALLEL_FOR
    for (uint32_t obj_id = 0; obj_id < n; ++obj_id) {
        float px = pos_xy[2 * obj_id], py = pos_xy[2 * obj_id + 1];
        float r = radius[obj_id];
        float sq_radius = r * r;                                              /* :50 */
        int32_t hx = f32_to_i32_sat(floorf(px / cell_size));                  /* :53 */
        int32_t hy = f32_to_i32_sat(floorf(py / cell_size));
        uint32_t base = obj_id * ORC_MAX_CELLS_PER_OBJECT;                    /* :56 */

        cell_ids[base] = orc_morton_encode(hx, hy);                           /* :62-64 */
        object_ids[base] = obj_id;

        uint32_t p_cell_count = 0;
        for (int y = -1; y <= 1; ++y) {                                       /* :68 */
            for (int x = -1; x <= 1; ++x) {                                   /* :69 */
                if (x == 0 && y == 0) continue;                               /* :70-74 */
                /* i32 add wraps in WGSL; coordinates here are far from the i32 limits
                 * unless floor() saturated, so use unsigned wrap to stay defined in C. */
                int32_t nx = (int32_t)((uint32_t)hx + (uint32_t)x);           /* :76-77 */
                int32_t ny = (int32_t)((uint32_t)hy + (uint32_t)y);
                if (is_obj_in_cell(px, py, sq_radius, nx, ny, cell_size)) {   /* :79 */
                    p_cell_count++;                                           /* :82 */
                    /* The WGSL writes slot base+p_cell_count unconditionally (:83-85).  With
                     * 2*r < cell_size a particle spans <= 2 cells per axis, so p_cell_count <= 3.
                     * Outside that contract the reference overruns into the next particle's
                     * slots (a data race there); the restatement refuses to overrun. */
                    if (p_cell_count < ORC_MAX_CELLS_PER_OBJECT) {
                        cell_ids[base + p_cell_count] = orc_morton_encode(nx, ny);
                        object_ids[base + p_cell_count] = obj_id;
                    }
                }
            }
        }
        for (uint32_t s = p_cell_count + 1; s < ORC_MAX_CELLS_PER_OBJECT; ++s)  /* :92-94 */
            cell_ids[base + s] = ORC_UNUSED_CELL_ID;
    }
}

/* ------------------------------------------------------------------------------------------
 * K2/K3  radix_sort.wgsl ; host loop radix_sort.rs:199-217
 * ---------------------------------------------------------------------------------------- */
uint32_t orc_radix_num_wg(uint32_t n)
{
    /* radix_sort.rs:203-204 */
    uint32_t total_threads = (n + ORC_RADIX_BLOCKS_PER_WG - 1) / ORC_RADIX_BLOCKS_PER_WG;
    return (total_threads + ORC_RADIX_WG - 1) / ORC_RADIX_WG;
}

/* radix_sort.wgsl:23-59 */
void orc_radix_build_histogram(const uint32_t *keys, uint32_t n, uint32_t shift,
                               uint32_t num_wg, uint32_t blocks_per_wg, uint32_t *hist)
{
    ORC_PARALLEL_FOR
    for (uint32_t wg = 0; wg < num_wg; ++wg) {
        uint32_t shared_histogram[256];
        memset(shared_histogram, 0, sizeof(shared_histogram));              /* :35-38 */
        for (uint32_t i = 0; i < blocks_per_wg; ++i) {                       /* :45 */
            for (uint32_t local_idx = 0; local_idx < ORC_RADIX_WG; ++local_idx) {
                uint64_t index = (uint64_t)wg * blocks_per_wg * ORC_RADIX_WG
                               + (uint64_t)i * ORC_RADIX_WG + local_idx;     /* :46 */
                if (index < n) {
                    uint32_t bucket = (keys[index] >> shift) & 255u;         /* :49 */
                    shared_histogram[bucket] += 1;                           /* :51 */
                }
            }
        }
        for (uint32_t b = 0; b < 256; ++b) hist[256u * wg + b] = shared_histogram[b];  /* :56-58 */
    }
}

/* radix_sort.wgsl:77-186.  Workgroups, blocks and lanes are visited sequentially; the WGSL's
 * barriers make that order result-equivalent.  The per-bucket ballot masks + popcount of lower
 * bits (:160-176) give each element the number of lower-lane elements of the same bucket in its
 * 256-element block; a running per-bucket counter in lane order computes the same number. */
void orc_radix_scatter(const uint32_t *keys_a, const uint32_t *payload_a, uint32_t n,
                       uint32_t shift, uint32_t num_wg, uint32_t blocks_per_wg,
                       const uint32_t *hist, uint32_t *keys_b, uint32_t *payload_b)
{
    /* :96-131 -- bucket totals and the exclusive bucket prefix are the same for every WG */
    uint32_t bucket_counts[256], bucket_prefix[256];
    for (uint32_t b = 0; b < 256; ++b) {
        uint32_t accum = 0;
        for (uint32_t i = 0; i < num_wg; ++i) accum += hist[i * 256u + b];   /* :101-108 */
        bucket_counts[b] = accum;                                            /* :109-110 */
    }
    uint32_t acc = 0;
    for (uint32_t b = 0; b < 256; ++b) { bucket_prefix[b] = acc; acc += bucket_counts[b]; } /* :116-122 */

    /* per-bucket count of the earlier workgroups, for every workgroup (what the WGSL sums at :101-108) */
    uint32_t *wg_excl = (uint32_t *)calloc((size_t)(num_wg + 1) * 256, sizeof(uint32_t));
    for (uint32_t wg = 0; wg < num_wg; ++wg)
        for (uint32_t b = 0; b < 256; ++b)
            wg_excl[(size_t)(wg + 1) * 256 + b] = wg_excl[(size_t)wg * 256 + b] + hist[wg * 256u + b];
    ORC_PARALLEL_FOR
    for (uint32_t wg = 0; wg < num_wg; ++wg) {
        uint32_t shared_global_offsets[256];
        for (uint32_t b = 0; b < 256; ++b)
            shared_global_offsets[b] = bucket_prefix[b] + wg_excl[(size_t)wg * 256 + b];   /* :104-106,125-130 */

        for (uint32_t i = 0; i < blocks_per_wg; ++i) {                       /* :138 */
            uint32_t in_block[256];                                          /* per-bucket count  */
            memset(in_block, 0, sizeof(in_block));                           /* :143-149 flags=0  */
            uint64_t block_base = (uint64_t)wg * blocks_per_wg * ORC_RADIX_WG
                                + (uint64_t)i * ORC_RADIX_WG;                /* :140 */
            for (uint32_t local_id = 0; local_id < ORC_RADIX_WG; ++local_id) {
                uint64_t index = block_base + local_id;
                if (index >= n) break;                                       /* :155,165 */
                uint32_t element = keys_a[index];                            /* :156 */
                uint32_t payload = payload_a[index];                         /* :157 */
                uint32_t bucket = (element >> shift) & 255u;                 /* :158 */
                uint32_t bucket_offset = shared_global_offsets[bucket];      /* :159 (pre-update) */
                uint32_t prefix = in_block[bucket]++;                        /* :166-176 */
                keys_b[bucket_offset + prefix] = element;                    /* :177 */
                payload_b[bucket_offset + prefix] = payload;                 /* :178 */
            }
            for (uint32_t b = 0; b < 256; ++b)
                shared_global_offsets[b] += in_block[b];                     /* :179-181 */
        }
    }
    free(wg_excl);
}

/* radix_sort.rs:199-217 */
void orc_sort_pairs(uint32_t *keys, uint32_t *payload, uint32_t n,
                    uint32_t *tmp_k, uint32_t *tmp_v, uint32_t *hist)
{
    if (n == 0) return;
    uint32_t num_wg = orc_radix_num_wg(n);
    uint32_t *ka = keys, *va = payload, *kb = tmp_k, *vb = tmp_v;
    for (uint32_t pass = 0; pass < 4; ++pass) {                              /* :206 */
        uint32_t shift = pass * 8u;                                          /* :209 */
        orc_radix_build_histogram(ka, n, shift, num_wg, ORC_RADIX_BLOCKS_PER_WG, hist);
        orc_radix_scatter(ka, va, n, shift, num_wg, ORC_RADIX_BLOCKS_PER_WG, hist, kb, vb);
        uint32_t *t = ka; ka = kb; kb = t;                                   /* :215 ping-pong */
        t = va; va = vb; vb = t;
    }
    /* 4 passes: the result is back in the caller's buffers */
}

/* ------------------------------------------------------------------------------------------
 * K7-K9  prefix_sum.wgsl:14-147, prefix_sum.rs:143-160
 * The three passes (per-256 block inclusive scan; scan of the block sums, recursive at
 * >= 65,536 items; add block_sums[block-1]) compute exactly the running u32 sum with
 * wrap-around; stated directly.
 * ---------------------------------------------------------------------------------------- */
void orc_inclusive_scan(uint32_t *data, uint32_t n)
{
    uint32_t sum = 0;
    for (uint32_t i = 0; i < n; ++i) { sum += data[i]; data[i] = sum; }
}

/* ------------------------------------------------------------------------------------------
 * K6  collision_cell_builder.wgsl:27-85
 * ---------------------------------------------------------------------------------------- */
static inline uint32_t total_chunks_of(uint32_t total_cell_ids)
{
    return (total_cell_ids + ORC_CHUNK_SIZE - 1) / ORC_CHUNK_SIZE;           /* :88-90 */
}

void orc_count_objects_per_chunk(const uint32_t *cell_ids, uint32_t total_cell_ids,
                                 uint32_t *chunk_obj_count)
{
    uint32_t total_chunks = total_chunks_of(total_cell_ids);                 /* :31 */
    ORC_PARALLEL_FOR
    for (uint32_t chunk_id = 0; chunk_id < total_chunks; ++chunk_id) {
        uint32_t first_idx = chunk_id * ORC_CHUNK_SIZE;                      /* :37 */
        /* :40 select(UNUSED, cell_ids[first_idx-1], first_idx >= 1); the out-of-bounds read
         * for first_idx == 0 is clamped by wgpu and discarded by select -- guarded here. */
        uint32_t prev_cell_id = (first_idx >= 1) ? cell_ids[first_idx - 1] : ORC_UNUSED_CELL_ID;
        uint32_t obj_count = 0;                                              /* :43 */
        uint32_t currently_counting_cell = ORC_UNUSED_CELL_ID;               /* :44 */
        uint32_t current_count = 0;                                          /* :45 */
        uint32_t next_chunk_first_idx = first_idx + ORC_CHUNK_SIZE;          /* :47 */
        for (uint32_t i = first_idx; i < total_cell_ids; ++i) {              /* :50 */
            uint32_t cell_id = cell_ids[i];                                  /* :52 */
            int is_out_of_bounds = i >= next_chunk_first_idx;                /* :55 */
            int is_cell_unused = cell_id == ORC_UNUSED_CELL_ID;              /* :56 */
            int is_it_a_transition = cell_id != prev_cell_id;                /* :57 */
            if ((is_it_a_transition && is_out_of_bounds) || is_cell_unused ||
                (current_count == 0 && is_out_of_bounds && !is_it_a_transition)) break; /* :58 */
            int cell_was_seen_before = currently_counting_cell == cell_id;   /* :61 */
            if (cell_was_seen_before) {
                if (current_count == 1) obj_count += 1;                      /* :65-67 */
                current_count += 1;                                          /* :68 */
            }
            if (cell_id != prev_cell_id) {                                   /* :71 */
                current_count = 1;                                           /* :74 */
                currently_counting_cell = cell_id;                           /* :75 */
            }
            prev_cell_id = cell_id;                                          /* :79 */
        }
        chunk_obj_count[chunk_id] = obj_count;                               /* :83 */
    }
}

/* ------------------------------------------------------------------------------------------
 * K10 collision_cell_builder.wgsl:96-189
 * ---------------------------------------------------------------------------------------- */
uint32_t orc_build_collision_cells(const uint32_t *cell_ids, uint32_t total_cell_ids,
                                   const uint32_t *chunk_obj_count, uint32_t num_chunks,
                                   uint32_t *collision_cells, uint32_t *indirect_args)
{
    /* prepare_dispatch_buffer :96-109 (global thread 0) */
    uint32_t total_items = num_chunks ? chunk_obj_count[num_chunks - 1] : 0; /* :100 */
    if (indirect_args) {
        indirect_args[0] = (total_items + 64u - 1u) / 64u;                   /* :103 */
        indirect_args[1] = 1u;
        indirect_args[2] = 1u;
    }
    uint32_t total_chunks = total_chunks_of(total_cell_ids);                 /* :122 */
    ORC_PARALLEL_FOR
    for (uint32_t chunk_id = 0; chunk_id < total_chunks; ++chunk_id) {
        uint32_t start_index = (chunk_id >= 1) ? chunk_obj_count[chunk_id - 1] : 0u; /* :128, :93 */
        uint32_t end_index = chunk_obj_count[chunk_id];                      /* :129 */
        uint32_t num_objects_to_manage = end_index - start_index;            /* :130 */
        if (!(num_objects_to_manage > 0)) continue;                          /* :133-136 */
        uint32_t first_idx = chunk_id * ORC_CHUNK_SIZE;                      /* :141 */
        uint32_t prev_cell_id = (first_idx >= 1) ? cell_ids[first_idx - 1] : ORC_UNUSED_CELL_ID; /* :144 */
        uint32_t currently_counting_cell = ORC_UNUSED_CELL_ID;               /* :148 */
        uint32_t current_count = 0;                                          /* :149 */
        uint32_t next_chunk_first_idx = first_idx + ORC_CHUNK_SIZE;          /* :151 */
        uint32_t write_index = start_index;                                  /* :153 */
        for (uint32_t i = first_idx; i < total_cell_ids && write_index != end_index; ++i) { /* :156 */
            uint32_t cell_id = cell_ids[i];
            int is_out_of_bounds = i >= next_chunk_first_idx;                /* :161 */
            int is_cell_unused = cell_id == ORC_UNUSED_CELL_ID;
            int is_it_a_transition = cell_id != prev_cell_id;
            if ((is_it_a_transition && is_out_of_bounds) || is_cell_unused ||
                (current_count == 0 && is_out_of_bounds && !is_it_a_transition)) break; /* :164 */
            int cell_was_seen_before = currently_counting_cell == cell_id;   /* :167 */
            if (cell_was_seen_before) {
                if (current_count == 1) {                                    /* :171 */
                    collision_cells[write_index] = i - 1;                    /* :174 */
                    write_index++;
                }
                current_count += 1;                                          /* :177 */
            }
            if (cell_id != prev_cell_id) {                                   /* :180 */
                current_count = 1;
                currently_counting_cell = cell_id;
            }
            prev_cell_id = cell_id;                                          /* :188 */
        }
    }
    return total_items;
}

/* ------------------------------------------------------------------------------------------
 * K11 collision_solver.wgsl:26-118
 * ---------------------------------------------------------------------------------------- */
static void resolve_cell_collisions(uint32_t cell_hash, uint32_t start, const uint32_t *cell_ids,
                                    const uint32_t *object_ids, uint32_t total_cell_ids,
                                    float *pos, const float *radius, float stiffness)
{
    for (uint32_t i = start; i < total_cell_ids; ++i) {                      /* :68 */
        if (cell_ids[i] != cell_hash) break;                                 /* :69-71 */
        uint32_t object_id = object_ids[i];                                  /* :72 */
        for (uint32_t j = i + 1; j < total_cell_ids; ++j) {                  /* :77 */
            if (cell_ids[j] != cell_hash) break;                             /* :78-81 */
            uint32_t other_object_id = object_ids[j];                        /* :83 */
            /* :85-88 -- live positions, re-read for every pair */
            float p1x = pos[2 * object_id], p1y = pos[2 * object_id + 1];
            float p2x = pos[2 * other_object_id], p2y = pos[2 * other_object_id + 1];
            float r1 = radius[object_id], r2 = radius[other_object_id];
            float vx = p1x - p2x, vy = p1y - p2y;                            /* :91 */
            float distance = sqrtf(vx * vx + vy * vy);                       /* :93 length() */
            float radius_sum = r1 + r2;                                      /* :61 */
            float sq_radius_sum = radius_sum * radius_sum;                   /* :62 */
            if (sq_radius_sum > distance * distance && distance > 0.0001f) { /* :95 */
                float penetration_depth = (r1 + r2) - distance;             /* :97 */
                float dirx = vx / distance, diry = vy / distance;            /* :98 */
                /* :101  (dir * depth) * STIFFNESS, left to right */
                float cx = (dirx * penetration_depth) * stiffness;
                float cy = (diry * penetration_depth) * stiffness;
                float inv_mass_1 = 1.0f / r1;                                /* :103 */
                float inv_mass_2 = 1.0f / r2;                                /* :104 */
                float w1 = inv_mass_1 / (inv_mass_1 + inv_mass_2);           /* :107 */
                float w2 = inv_mass_2 / (inv_mass_1 + inv_mass_2);           /* :108 */
                float d1x = cx * w1, d1y = cy * w1;
                float d2x = cx * w2, d2y = cy * w2;
                pos[2 * object_id] = p1x + d1x;                              /* :110 */
                pos[2 * object_id + 1] = p1y + d1y;
                /* :111 re-reads positions[other] -- identical to p2 unless other == object,
                 * which cannot happen (each object appears once per cell). */
                pos[2 * other_object_id] = pos[2 * other_object_id] - d2x;
                pos[2 * other_object_id + 1] = pos[2 * other_object_id + 1] - d2y;
            }
        }
    }
}

void orc_solve_collisions_color(const uint32_t *collision_cells, uint32_t num_collision_cells,
                                const uint32_t *cell_ids, const uint32_t *object_ids,
                                uint32_t total_cell_ids, float *pos_xy, const float *radius,
                                float stiffness, uint32_t color)
{
    /* Threads of one colour pass touch disjoint particles, so visiting
     * them sequentially is result-equivalent. */
    ORC_PARALLEL_FOR
    for (uint32_t tid = 0; tid < num_collision_cells; ++tid) {               /* :33-36 */
        uint32_t start = collision_cells[tid];                               /* :38 */
        uint32_t cell_hash = cell_ids[start];                                /* :39 */
        if (orc_cell_color(cell_hash) == color)                              /* :40-43 */
            resolve_cell_collisions(cell_hash, start, cell_ids, object_ids, total_cell_ids,
                                    pos_xy, radius, stiffness);
    }
}

/* ------------------------------------------------------------------------------------------
 * K12 particle_integration.wgsl:25-77
 * ---------------------------------------------------------------------------------------- */
void orc_verlet_integration(float *pos_xy, float *prev_xy, const float *radius, uint32_t n,
                            const orc_params *p, float dt)
{
    ORC_PARALLEL_FOR
    for (uint32_t index = 0; index < n; ++index) {
        float cx = pos_xy[2 * index], cy = pos_xy[2 * index + 1];            /* :34 */
        float qx = prev_xy[2 * index], qy = prev_xy[2 * index + 1];          /* :35 */
        float vx = cx - qx, vy = cy - qy;                                    /* :40 */
        float ax = p->gravity_x, ay = p->gravity_y;                          /* :42 */
        if (p->mouse_pressed == 1u) {                                        /* :44 */
            float dx = p->mouse_x - cx, dy = p->mouse_y - cy;                /* :46 */
            float len = sqrtf(dx * dx + dy * dy);                            /* :50 normalize() */
            float nx = dx / len, ny = dy / len;
            ax = ax + nx * p->mouse_strength;                                /* :50,53 */
            ay = ay + ny * p->mouse_strength;
        }
        float dt_squared = dt * dt;                                          /* :58 */
        float nxp = (cx + vx) + ax * dt_squared;                             /* :59 */
        float nyp = (cy + vy) + ay * dt_squared;
        prev_xy[2 * index] = cx;                                             /* :64 */
        prev_xy[2 * index + 1] = cy;
        float r = radius[index];                                             /* :66 */
        nxp = clampf(nxp, r, p->world_w - r);                                /* :70 */
        nyp = clampf(nyp, r, p->world_h - r);                                /* :71 */
        pos_xy[2 * index] = nxp;                                             /* :76 */
        pos_xy[2 * index + 1] = nyp;
    }
}

/* ------------------------------------------------------------------------------------------
 * whole simulation
 * ---------------------------------------------------------------------------------------- */
static void *xmalloc(size_t bytes) { void *p = malloc(bytes ? bytes : 1); if (!p) abort(); return p; }

orc_sim *orc_sim_create(const float *pos_xy, const float *prev_xy, const float *radius,
                        uint32_t n, const orc_params *p)
{
    orc_sim *s = (orc_sim *)calloc(1, sizeof(orc_sim));
    if (!s) abort();
    s->n = n;
    s->params = *p;
    size_t v2 = (size_t)n * 2 * sizeof(float), v1 = (size_t)n * sizeof(float);
    size_t total = (size_t)n * ORC_MAX_CELLS_PER_OBJECT;
    s->pos = (float *)xmalloc(v2); s->prev = (float *)xmalloc(v2); s->radius = (float *)xmalloc(v1);
    s->pos_copy = (float *)xmalloc(v2); s->prev_copy = (float *)xmalloc(v2);
    s->radius_copy = (float *)xmalloc(v1);
    memcpy(s->pos, pos_xy, v2);
    memcpy(s->prev, prev_xy ? prev_xy : pos_xy, v2);      /* particle_system.rs:126-127: prev = cur */
    memcpy(s->radius, radius, v1);
    memcpy(s->pos_copy, s->pos, v2); memcpy(s->prev_copy, s->prev, v2);
    memcpy(s->radius_copy, s->radius, v1);
    s->home_cell_ids = (uint32_t *)xmalloc(n * sizeof(uint32_t));
    s->particle_ids = (uint32_t *)xmalloc(n * sizeof(uint32_t));
    for (uint32_t i = 0; i < n; ++i) {
        s->home_cell_ids[i] = ORC_UNUSED_CELL_ID;          /* particle_system.rs:130-133 */
        s->particle_ids[i] = i;                            /* particle_sort.rs:30        */
    }
    s->cell_ids = (uint32_t *)xmalloc(total * sizeof(uint32_t));
    s->object_ids = (uint32_t *)xmalloc(total * sizeof(uint32_t));
    s->collision_cells = (uint32_t *)xmalloc(total * sizeof(uint32_t));
    for (size_t i = 0; i < total; ++i) {
        s->cell_ids[i] = ORC_UNUSED_CELL_ID;               /* grid.rs:80-83 */
        s->object_ids[i] = 0;                              /* grid.rs:85-89 */
        s->collision_cells[i] = ORC_UNUSED_CELL_ID;        /* collision_cell_buffers.rs:23-27 */
    }
    uint32_t num_chunks = total_chunks_of((uint32_t)total);
    s->chunk_obj_count = (uint32_t *)calloc(num_chunks ? num_chunks : 1, sizeof(uint32_t));
    s->tmp_k = (uint32_t *)xmalloc(total * sizeof(uint32_t));
    s->tmp_v = (uint32_t *)xmalloc(total * sizeof(uint32_t));
    s->hist = (uint32_t *)xmalloc((size_t)256 * (orc_radix_num_wg((uint32_t)total) + 1) * sizeof(uint32_t));
    s->indirect_args[0] = s->indirect_args[1] = s->indirect_args[2] = 0;
    s->num_collision_cells = 0;
    return s;
}

void orc_sim_destroy(orc_sim *s)
{
    if (!s) return;
    free(s->pos); free(s->prev); free(s->radius);
    free(s->pos_copy); free(s->prev_copy); free(s->radius_copy);
    free(s->home_cell_ids); free(s->particle_ids);
    free(s->cell_ids); free(s->object_ids);
    free(s->chunk_obj_count); free(s->collision_cells);
    free(s->tmp_k); free(s->tmp_v); free(s->hist);
    free(s);
}

/* particle_sort.rs:58-69 then particle_rearrange.rs:205-238 (copy set -> live set) */
void orc_sim_morton_resort(orc_sim *s)
{
    orc_create_home_cell_ids(s->pos, s->n, s->params.cell_size, s->home_cell_ids, s->particle_ids);
    orc_sort_pairs(s->home_cell_ids, s->particle_ids, s->n, s->tmp_k, s->tmp_v, s->hist);
    orc_rearrange(s->pos, s->prev, s->radius, s->particle_ids, s->n,
                  s->pos_copy, s->prev_copy, s->radius_copy);
    memcpy(s->pos, s->pos_copy, (size_t)s->n * 2 * sizeof(float));
    memcpy(s->radius, s->radius_copy, (size_t)s->n * sizeof(float));
    memcpy(s->prev, s->prev_copy, (size_t)s->n * 2 * sizeof(float));
}

void orc_sim_grid_build(orc_sim *s)
{
    orc_build_cell_ids(s->pos, s->radius, s->n, s->params.cell_size, s->cell_ids, s->object_ids);
}

void orc_sim_grid_sort(orc_sim *s)
{
    orc_sort_pairs(s->cell_ids, s->object_ids, s->n * ORC_MAX_CELLS_PER_OBJECT,
                   s->tmp_k, s->tmp_v, s->hist);
}

void orc_sim_build_collision_cells(orc_sim *s)
{
    uint32_t total = s->n * ORC_MAX_CELLS_PER_OBJECT;
    uint32_t num_chunks = total_chunks_of(total);
    orc_count_objects_per_chunk(s->cell_ids, total, s->chunk_obj_count);
    orc_inclusive_scan(s->chunk_obj_count, num_chunks);
    s->num_collision_cells = orc_build_collision_cells(s->cell_ids, total, s->chunk_obj_count,
                                                       num_chunks, s->collision_cells,
                                                       s->indirect_args);
}

void orc_sim_solve_colors(orc_sim *s)
{
    uint32_t total = s->n * ORC_MAX_CELLS_PER_OBJECT;
    for (uint32_t color = 1; color <= 4; ++color)                            /* collision_solver.rs:224 */
        orc_solve_collisions_color(s->collision_cells, s->num_collision_cells, s->cell_ids,
                                   s->object_ids, total, s->pos, s->radius,
                                   s->params.stiffness, color);
}

void orc_sim_integrate(orc_sim *s, float dt)
{
    orc_verlet_integration(s->pos, s->prev, s->radius, s->n, &s->params, dt);
}

/* state.rs:115-131 */
void orc_sim_step(orc_sim *s, float dt, int resort)
{
    if (s->n == 0) return;
    if (resort) orc_sim_morton_resort(s);          /* :122-125 */
    orc_sim_grid_build(s);                         /* :126 grid.update */
    orc_sim_grid_sort(s);
    orc_sim_build_collision_cells(s);              /* :127 collision_system.solve_collisions */
    orc_sim_solve_colors(s);
    orc_sim_integrate(s, dt);                      /* :130 */
}

float *orc_sim_pos(orc_sim *s) { return s->pos; }
float *orc_sim_prev(orc_sim *s) { return s->prev; }
float *orc_sim_radius(orc_sim *s) { return s->radius; }
uint32_t *orc_sim_cell_ids(orc_sim *s) { return s->cell_ids; }
uint32_t *orc_sim_object_ids(orc_sim *s) { return s->object_ids; }
uint32_t *orc_sim_collision_cells(orc_sim *s) { return s->collision_cells; }
uint32_t *orc_sim_chunk_obj_count(orc_sim *s) { return s->chunk_obj_count; }
uint32_t *orc_sim_home_cell_ids(orc_sim *s) { return s->home_cell_ids; }
uint32_t *orc_sim_particle_ids(orc_sim *s) { return s->particle_ids; }
uint32_t orc_sim_num_collision_cells(orc_sim *s) { return s->num_collision_cells; }
