// reference_tests.cpp -- the reference's five integration test files (/root/reference/tests/*.rs) restated in C++ on top
// of the host mirror (gpu-physics-engine_amd/host/gpe_host.hpp) and the C-ABI.  Same test names, same inputs, same
// expected values (the data also lives in tests/golden/reference_vectors.json).  Runs on the GPU box:
//   g++ -std=c++17 tests/cpp/reference_tests.cpp -Lgpu-physics-engine_amd -lgpe -o tests/cpp/reference_tests
// Exit code 0 = all passed; `--list` prints the test names (used by the CPU-side compile check).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <functional>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#include "../../gpu-physics-engine_amd/host/gpe_host.hpp"

using namespace gpe;

static int g_failed = 0;
#define ASSERT_EQ(a, b)                                                                                   \
    do {                                                                                                  \
        if (!((a) == (b))) {                                                                              \
            std::printf("    ASSERT_EQ failed at %s:%d: %s == %s\n", __FILE__, __LINE__, #a, #b);          \
            throw std::runtime_error("assertion failed");                                                 \
        }                                                                                                 \
    } while (0)

// tests/grid.rs:76-90
static uint32_t split_by_bits(uint32_t n)
{
    uint32_t x = n & 0x0000FFFFu;
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}
static uint32_t morton_encode(uint32_t x, uint32_t y) { return split_by_bits(x) | (split_by_bits(y) << 1); }

// tests/common.rs:29-33
static ParticleSystem create_test_particle_system(const Context &ctx, const std::vector<Vec2> &positions,
                                                  const std::vector<float> &radius)
{
    return ParticleSystem::new_from_buffers(ctx, positions, radius);
}

// ---- tests/grid.rs --------------------------------------------------------------------------------------------
struct Case { std::vector<Vec2> positions; std::vector<float> radii; float max_radius; };
static Case case_1() { return {{{20.0f, 42.0f}, {77.0f, 77.0f}, {5.0f, 5.0f}}, {10.0f, 8.0f, 1.0f}, 10.0f}; }   // :107-133

static void expected_case_1(std::vector<uint32_t> &cell_ids, std::vector<uint32_t> &object_ids)
{
    cell_ids = {morton_encode(0, 1), morton_encode(1, 1), morton_encode(0, 2), morton_encode(1, 2),   // particle 0
                morton_encode(3, 3), UNUSED_CELL_ID, UNUSED_CELL_ID, UNUSED_CELL_ID,                  // particle 1
                0, UNUSED_CELL_ID, UNUSED_CELL_ID, UNUSED_CELL_ID};                                   // particle 2
    object_ids = {0, 0, 0, 0, 1, 0, 0, 0, 2, 0, 0, 0};
}

static void test_grid_build_cell_ids_with_multiple_particles()          // tests/grid.rs:12-71
{
    Context ctx;
    Case c = case_1();
    ParticleSystem ps = create_test_particle_system(ctx, c.positions, c.radii);
    Grid grid = Grid::new_without_camera(ctx, c.max_radius, ps);
    std::vector<uint32_t> expected_cell_ids, expected_object_ids;
    expected_case_1(expected_cell_ids, expected_object_ids);
    grid.build_cell_ids();
    ASSERT_EQ(grid.download_cell_ids(), expected_cell_ids);
    ASSERT_EQ(grid.download_object_ids(), expected_object_ids);
}

static void test_grid_build_cell_ids_and_sort()                         // tests/grid.rs:134-197
{
    Context ctx;
    Case c = case_1();
    ParticleSystem ps = create_test_particle_system(ctx, c.positions, c.radii);
    Grid grid = Grid::new_without_camera(ctx, c.max_radius, ps);
    grid.build_cell_ids();
    grid.sort_map();
    std::vector<uint32_t> ec, eo;
    expected_case_1(ec, eo);
    std::vector<std::pair<uint32_t, uint32_t>> expected;
    for (size_t i = 0; i < ec.size(); ++i) expected.push_back({ec[i], eo[i]});
    std::sort(expected.begin(), expected.end());
    auto gc = grid.download_cell_ids();
    auto go = grid.download_object_ids();
    std::vector<std::pair<uint32_t, uint32_t>> got;
    for (size_t i = 0; i < gc.size(); ++i) got.push_back({gc[i], go[i]});
    ASSERT_EQ(got, expected);
}

static void test_grid_build_cell_ids_sort_and_build_empty_collision_cells_list()   // tests/grid.rs:203-226
{
    Context ctx;
    Case c = case_1();
    ParticleSystem ps = create_test_particle_system(ctx, c.positions, c.radii);
    Grid grid = Grid::new_without_camera(ctx, c.max_radius, ps);
    CollisionSystem cs(ctx, 2, ps, grid);
    grid.build_cell_ids();
    grid.sort_map();
    cs.solve_collisions();
    std::vector<uint32_t> expected(grid.download_object_ids().size(), UNUSED_CELL_ID);
    ASSERT_EQ(cs.download_collision_cells(), expected);
}

static void test_grid_build_cell_ids_sort_and_build_collision_cells_list()         // tests/grid.rs:265-292
{
    Context ctx;
    const size_t n = 546;
    std::vector<Vec2> positions(n, Vec2{20.0f, 42.0f});
    std::vector<float> radii(n, 10.0f);
    ParticleSystem ps = create_test_particle_system(ctx, positions, radii);
    Grid grid = Grid::new_without_camera(ctx, 10.0f, ps);
    CollisionSystem cs(ctx, 2, ps, grid);
    grid.build_cell_ids();
    grid.sort_map();
    cs.solve_collisions();
    std::vector<uint32_t> expected;
    for (uint32_t i = 0; i < 4; ++i) expected.push_back(i * (uint32_t)n);
    expected.resize(grid.download_object_ids().size(), UNUSED_CELL_ID);
    ASSERT_EQ(cs.download_collision_cells(), expected);
}

// ---- tests/particle_sort.rs ------------------------------------------------------------------------------------
static void sort_particles_test()                                        // tests/particle_sort.rs:9-71
{
    Context ctx;
    Case c = case_1();
    ParticleSystem ps = create_test_particle_system(ctx, c.positions, c.radii);
    Grid::new_without_camera(ctx, c.max_radius, ps);
    ps.sort_by_cell_id(Grid::compute_cell_size(c.max_radius));
    ASSERT_EQ(ps.download_home_cell_ids(), (std::vector<uint32_t>{0, 2, 15}));
    ASSERT_EQ(ps.download_particle_ids(), (std::vector<uint32_t>{2, 0, 1}));
    ParticleBuffers b = ps.download_particle_buffers();
    const std::vector<Vec2> expected_positions = {{5.0f, 5.0f}, {20.0f, 42.0f}, {77.0f, 77.0f}};
    ASSERT_EQ(b.current_positions, expected_positions);
    ASSERT_EQ(b.previous_positions, expected_positions);
    ASSERT_EQ(b.radii, (std::vector<float>{1.0f, 10.0f, 8.0f}));
}

// ---- tests/radix_sort.rs ---------------------------------------------------------------------------------------
static void sort_test()                                                  // tests/radix_sort.rs:7-48
{
    Context ctx;
    const uint32_t n = 25006;
    std::vector<uint32_t> scrambled(n);
    for (uint32_t i = 0; i < n; ++i) scrambled[i] = n - 1 - i;
    GpuBuffer<uint32_t> keys(ctx, scrambled), payload(ctx, scrambled);
    GPUSorter sorter(ctx, n, keys, payload);
    sorter.sort(nullptr);
    std::vector<uint32_t> sorted(n);
    std::iota(sorted.begin(), sorted.end(), 0u);
    ASSERT_EQ(keys.download(), sorted);
    ASSERT_EQ(payload.download(), sorted);
}

static void sort_test_small_sized_array()                                // tests/radix_sort.rs:52-125
{
    Context ctx;
    std::vector<uint32_t> data = {357000000u, 90000u, 257u, 2u, 20000000u, 1u, 30000u, 65611u};
    const uint32_t n = (uint32_t)data.size();
    GpuBuffer<uint32_t> keys(ctx, data), payload(ctx, data);
    GPUSorter sorter(ctx, n, keys, payload);
    const uint32_t total_threads = (n + NUM_BLOCKS_PER_WORKGROUP - 1) / NUM_BLOCKS_PER_WORKGROUP;
    const uint32_t num_workgroups = (total_threads + WORKGROUP_SIZE - 1) / WORKGROUP_SIZE;
    PushConstants pc{n, 0, num_workgroups, NUM_BLOCKS_PER_WORKGROUP};
    sorter.build_histogram(pc, true);
    const std::vector<uint32_t> &histogram = sorter.get_histogram();
    ASSERT_EQ(std::accumulate(histogram.begin(), histogram.end(), 0u), n);
    ASSERT_EQ(histogram.size(), (size_t)256);
    std::vector<uint32_t> expected_histogram(256, 0);
    for (uint32_t e : data) expected_histogram[(e >> 0) & (RADIX_SORT_BUCKETS - 1)] += 1;
    ASSERT_EQ(histogram, expected_histogram);
    sorter.scatter(pc, true);
    ASSERT_EQ(sorter.get_keys_b(), (std::vector<uint32_t>{20000000u, 257u, 1u, 2u, 30000u, 357000000u, 65611u, 90000u}));
}

// ---- tests/prefix_sum.rs ---------------------------------------------------------------------------------------
static std::vector<uint32_t> host_scan(const std::vector<uint32_t> &v)
{
    std::vector<uint32_t> out(v.size());
    uint32_t sum = 0;
    for (size_t i = 0; i < v.size(); ++i) { sum += v[i]; out[i] = sum; }
    return out;
}
static void scan_case(const std::vector<uint32_t> &values)
{
    Context ctx;
    GpuBuffer<uint32_t> buffer(ctx, values);
    PrefixSum prefix_sum(ctx, buffer);
    prefix_sum.execute((uint32_t)values.size());
    ASSERT_EQ(buffer.download(), host_scan(values));
}
static void inclusive_prefix_sum_test()                                  // tests/prefix_sum.rs:8-46
{
    const uint32_t n = 81920;
    std::vector<uint32_t> v(n);
    for (uint32_t i = 0; i < n; ++i) v[i] = n - 1 - i;
    scan_case(v);
}
static void inclusive_prefix_sum_same_values_test() { scan_case(std::vector<uint32_t>(83090, 1u)); }   // :50-88
static void inclusive_prefix_sum_all_zero_test() { scan_case(std::vector<uint32_t>(81920, 0u)); }      // :91-129
static void inclusive_prefix_sum_random_test()                           // tests/prefix_sum.rs:133-168
{
    std::mt19937 rng(20251031);
    const uint32_t n = std::uniform_int_distribution<uint32_t>(10381920u, 14381920u)(rng);
    std::vector<uint32_t> v(n);
    std::uniform_int_distribution<uint32_t> d(0, 9);
    for (auto &x : v) x = d(rng);
    scan_case(v);
}
static void inclusive_prefix_sum_resize_test()                           // tests/prefix_sum.rs:171-243
{
    Context ctx;
    const size_t n = 83090;
    std::vector<uint32_t> original(n, 1u);
    {
        GpuBuffer<uint32_t> buffer(ctx, original);
        PrefixSum prefix_sum(ctx, buffer);
        prefix_sum.execute((uint32_t)n);
        ASSERT_EQ(buffer.download(), host_scan(original));
    }
    GpuBuffer<uint32_t> buffer(ctx, original);
    PrefixSum prefix_sum(ctx, buffer);
    const size_t new_len = n + 700000;
    original.resize(new_len, 1u);
    buffer.push_all(std::vector<uint32_t>(new_len - n, 1u));
    prefix_sum.update_buffers(buffer);
    prefix_sum.execute((uint32_t)new_len);
    ASSERT_EQ(buffer.download(), host_scan(original));
}

// ---- utils/gpu_buffer.rs:30-38,177-275: push / replace_elem / download_last (no reference test file covers them) --------
static void gpu_buffer_push_replace_download_last()
{
    Context ctx;
    GpuBuffer<uint32_t> buf(ctx, std::vector<uint32_t>{0, 1, 2, 3, 4});
    uint32_t last = 0;
    ASSERT_EQ(buf.capacity_bytes(), (size_t)20);
    ASSERT_EQ(buf.download_last(&last), true);
    ASSERT_EQ(last, 4u);
    {
        PrefixSum scan(ctx, buf);                                       // device: 0 1 3 6 10; the mirror keeps 0 1 2 3 4
        scan.execute(5);
    }
    buf.push(77u);                                                      // grows to 48 bytes, device contents kept
    ASSERT_EQ(buf.capacity_bytes(), (size_t)48);
    buf.push_all(std::vector<uint32_t>{5, 6, 7});
    ASSERT_EQ(buf.capacity_bytes(), (size_t)48);
    buf.replace_elem(1234u, 2);
    ASSERT_EQ(buf.download(), (std::vector<uint32_t>{0, 1, 1234, 6, 10, 77, 5, 6, 7}));
    ASSERT_EQ(buf.download_last(&last), true);
    ASSERT_EQ(last, 7u);
    bool threw = false;
    try { buf.replace_elem(1u, 9); } catch (const std::out_of_range &) { threw = true; }
    ASSERT_EQ(threw, true);
    GpuBuffer<uint32_t> empty(ctx, std::vector<uint32_t>{});
    ASSERT_EQ(empty.download_last(&last), false);
    empty.push(3u);
    ASSERT_EQ(empty.download(), (std::vector<uint32_t>{3}));
}

// ---- beyond the reference: the whole step through State, both pipelines must agree ------------------------------------
static void state_update_native_equals_compat()
{
    const size_t n = 50000;
    const Vec2 world{680.0f, 235.0f};
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> ux(0.0f, world.x), uy(0.0f, world.y);
    std::vector<Vec2> pos(n);
    for (auto &p : pos) p = {ux(rng), uy(rng)};
    std::vector<float> rad(n, 0.5f);
    State a(pos, rad, world, GPE_MODE_NATIVE), b(pos, rad, world, GPE_MODE_COMPAT);
    for (int s = 0; s < 8; ++s) {
        a.update(1.0f / 60.0f, s == 0);
        b.update(1.0f / 60.0f, s == 0);
    }
    ParticleBuffers pa = a.particles().download_particle_buffers(), pb = b.particles().download_particle_buffers();
    ASSERT_EQ(pa.current_positions, pb.current_positions);
    ASSERT_EQ(pa.previous_positions, pb.previous_positions);
}

int main(int argc, char **argv)
{
    const std::vector<std::pair<std::string, std::function<void()>>> tests = {
        {"gpu_buffer_push_replace_download_last", gpu_buffer_push_replace_download_last},
        {"test_grid_build_cell_ids_with_multiple_particles", test_grid_build_cell_ids_with_multiple_particles},
        {"test_grid_build_cell_ids_and_sort", test_grid_build_cell_ids_and_sort},
        {"test_grid_build_cell_ids_sort_and_build_empty_collision_cells_list",
         test_grid_build_cell_ids_sort_and_build_empty_collision_cells_list},
        {"test_grid_build_cell_ids_sort_and_build_collision_cells_list",
         test_grid_build_cell_ids_sort_and_build_collision_cells_list},
        {"sort_particles_test", sort_particles_test},
        {"sort_test", sort_test},
        {"sort_test_small_sized_array", sort_test_small_sized_array},
        {"inclusive_prefix_sum_test", inclusive_prefix_sum_test},
        {"inclusive_prefix_sum_same_values_test", inclusive_prefix_sum_same_values_test},
        {"inclusive_prefix_sum_all_zero_test", inclusive_prefix_sum_all_zero_test},
        {"inclusive_prefix_sum_random_test", inclusive_prefix_sum_random_test},
        {"inclusive_prefix_sum_resize_test", inclusive_prefix_sum_resize_test},
        {"state_update_native_equals_compat", state_update_native_equals_compat},
    };
    if (argc > 1 && std::strcmp(argv[1], "--list") == 0) {
        for (auto &t : tests) std::printf("%s\n", t.first.c_str());
        return 0;
    }
    for (auto &t : tests) {
        try {
            t.second();
            std::printf("test %s ... ok\n", t.first.c_str());
        } catch (const std::exception &e) {
            std::printf("test %s ... FAILED: %s\n", t.first.c_str(), e.what());
            ++g_failed;
        }
    }
    std::printf("test result: %s. %zu passed; %d failed\n", g_failed ? "FAILED" : "ok", tests.size() - g_failed, g_failed);
    return g_failed ? 1 : 0;
}
