/*
 * gpe_oracle.h -- CPU restatement of the reference particle step.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle for the MI355X HIP path in gpu-physics-engine_amd/csrc/.  Nothing
 * in the product path may include, link, import or call anything in oracle/: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as the checker.
 *
 * Every function restates one compute entry point of the reference WGSL (cited per function,
 * paths relative to /root/reference/src) with IEEE-754 binary32 arithmetic, round-to-nearest,
 * no FMA contraction (compile with -ffp-contract=off, never -ffast-math).
 *
 * Parity status: the integer artefacts (Morton ids, (cell,object) pairs, sorted order,
 * collision-cell list, scan) are PINNED by the reference's own known-answer tests
 * (tests/golden/reference_vectors.json, transcribed from the .rs files under /root/reference/tests).
 * The float results of K11 (collision response) and K12 (Verlet integration) are
 * "oracle-defined": the reference holds no test or vector for them ("parity unpinned" for
 * those two kernels' float values; their control flow and operation order follow the WGSL text).
 */
#ifndef GPE_ORACLE_H
#define GPE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_UNUSED_CELL_ID 0xffffffffu
#define ORC_MAX_CELLS_PER_OBJECT 4u
#define ORC_CHUNK_SIZE 4u
#define ORC_RADIX_WG 256u
#define ORC_RADIX_BLOCKS_PER_WG 45u

typedef struct orc_params {
    float world_w, world_h;   /* particle_integration.rs:38-40 (SimParams)              */
    float cell_size;          /* grid.rs:159-161: max_radius * 2.2f                       */
    float gravity_x, gravity_y; /* particle_integration.wgsl:21 FORCE_OF_GRAVITY = (0,0) */
    float stiffness;          /* collision_solver.wgsl:2  STIFFNESS = 0.6                */
    float mouse_strength;     /* particle_integration.wgsl:22 = 150.0                    */
    uint32_t mouse_pressed;   /* SimParams.is_mouse_pressed                               */
    float mouse_x, mouse_y;   /* SimParams.mouse_pos                                      */
} orc_params;

void orc_params_default(orc_params *p, float world_w, float world_h, float max_radius);

/* grid.rs:159-161 */
/* host threads for the independent loops (1 = serial, the default; any count gives the same bits) */
void orc_set_threads(int threads);
int orc_get_threads(void);

float orc_compute_cell_size(float max_radius);

/* grid.wgsl:101-114 / home_cell_ids.wgsl:38-51 */
uint32_t orc_split_by_bits(uint32_t n);
uint32_t orc_morton_encode(int32_t x, int32_t y);
/* collision_solver.wgsl:123-136 */
uint32_t orc_unsplit_by_bits(uint32_t n);
/* collision_solver.wgsl:55-58 */
uint32_t orc_cell_color(uint32_t cell_hash);

/* K1  home_cell_ids.wgsl:16-34 */
void orc_create_home_cell_ids(const float *pos_xy, uint32_t n, float cell_size,
                              uint32_t *home_cell_ids, uint32_t *particle_ids);
/* K4  rearrange.wgsl:19-35 (gather into the copy set) */
void orc_rearrange(const float *pos_xy, const float *prev_xy, const float *radius,
                   const uint32_t *particle_ids, uint32_t n,
                   float *pos_out, float *prev_out, float *radius_out);
/* K5  grid.wgsl:39-97 ; object_ids of unused slots are left untouched */
void orc_build_cell_ids(const float *pos_xy, const float *radius, uint32_t n, float cell_size,
                        uint32_t *cell_ids, uint32_t *object_ids);

/* K2  radix_sort.wgsl:23-59 ; hist has 256*num_wg entries */
void orc_radix_build_histogram(const uint32_t *keys, uint32_t n, uint32_t shift,
                               uint32_t num_wg, uint32_t blocks_per_wg, uint32_t *hist);
/* K3  radix_sort.wgsl:77-186 ; one stable scatter pass */
void orc_radix_scatter(const uint32_t *keys_a, const uint32_t *payload_a, uint32_t n,
                       uint32_t shift, uint32_t num_wg, uint32_t blocks_per_wg,
                       const uint32_t *hist, uint32_t *keys_b, uint32_t *payload_b);
/* radix_sort.rs:199-217 : 4 x (K2,K3) ping-pong, result back in keys/payload.
 * tmp_k/tmp_v: n entries each; hist: 256*orc_radix_num_wg(n) entries. */
uint32_t orc_radix_num_wg(uint32_t n);
void orc_sort_pairs(uint32_t *keys, uint32_t *payload, uint32_t n,
                    uint32_t *tmp_k, uint32_t *tmp_v, uint32_t *hist);

/* K7-K9  prefix_sum.wgsl:14-147 + prefix_sum.rs:143-160 : in-place inclusive u32 scan */
void orc_inclusive_scan(uint32_t *data, uint32_t n);

/* K6  collision_cell_builder.wgsl:27-85 */
void orc_count_objects_per_chunk(const uint32_t *cell_ids, uint32_t total_cell_ids,
                                 uint32_t *chunk_obj_count);
/* K10 collision_cell_builder.wgsl:96-189 ; chunk_obj_count is the SCANNED buffer.
 * Writes collision_cells[0..K) and indirect_args[3]; returns K. */
uint32_t orc_build_collision_cells(const uint32_t *cell_ids, uint32_t total_cell_ids,
                                   const uint32_t *chunk_obj_count, uint32_t num_chunks,
                                   uint32_t *collision_cells, uint32_t *indirect_args);
/* K11 collision_solver.wgsl:26-118 ; one colour pass (colour in 1..4) */
void orc_solve_collisions_color(const uint32_t *collision_cells, uint32_t num_collision_cells,
                                const uint32_t *cell_ids, const uint32_t *object_ids,
                                uint32_t total_cell_ids, float *pos_xy, const float *radius,
                                float stiffness, uint32_t color);
/* K12 particle_integration.wgsl:25-77 */
void orc_verlet_integration(float *pos_xy, float *prev_xy, const float *radius, uint32_t n,
                            const orc_params *p, float dt);

/* ---- whole simulation: state.rs:115-131 ordering ---------------------------------------- */
typedef struct orc_sim {
    uint32_t n;
    orc_params params;
    float *pos, *prev, *radius;             /* live set (particle_buffers.rs:4-10)           */
    float *pos_copy, *prev_copy, *radius_copy; /* copy set (rearrange scratch)              */
    uint32_t *home_cell_ids, *particle_ids; /* particle_sort.rs:29-33                        */
    uint32_t *cell_ids, *object_ids;        /* grid.rs:80-89, 4N each                        */
    uint32_t *chunk_obj_count;              /* collision_cell_buffers.rs:17-21, N entries    */
    uint32_t *collision_cells;              /* collision_cell_buffers.rs:23-27, 4N, init U   */
    uint32_t indirect_args[3];
    uint32_t num_collision_cells;
    uint32_t *tmp_k, *tmp_v, *hist;         /* sorter scratch                                */
} orc_sim;

orc_sim *orc_sim_create(const float *pos_xy, const float *prev_xy /*NULL => = pos*/,
                        const float *radius, uint32_t n, const orc_params *p);
void orc_sim_destroy(orc_sim *s);
void orc_sim_morton_resort(orc_sim *s);     /* particle_sort.rs:58-69 + copies back          */
void orc_sim_grid_build(orc_sim *s);        /* grid.rs:296-306                               */
void orc_sim_grid_sort(orc_sim *s);         /* grid.rs:310-312                               */
void orc_sim_build_collision_cells(orc_sim *s); /* collision_cell_builder.rs:211-236         */
void orc_sim_solve_colors(orc_sim *s);      /* collision_solver.rs:219-244                   */
void orc_sim_integrate(orc_sim *s, float dt);
void orc_sim_step(orc_sim *s, float dt, int resort); /* state.rs:115-131                     */
/* accessors for ctypes */
float *orc_sim_pos(orc_sim *s);
float *orc_sim_prev(orc_sim *s);
float *orc_sim_radius(orc_sim *s);
uint32_t *orc_sim_cell_ids(orc_sim *s);
uint32_t *orc_sim_object_ids(orc_sim *s);
uint32_t *orc_sim_collision_cells(orc_sim *s);
uint32_t *orc_sim_chunk_obj_count(orc_sim *s);
uint32_t *orc_sim_home_cell_ids(orc_sim *s);
uint32_t *orc_sim_particle_ids(orc_sim *s);
uint32_t orc_sim_num_collision_cells(orc_sim *s);

#ifdef __cplusplus
}
#endif
#endif
