"""GPU (-m gpu): the reference's five integration test files (tests/*.rs) restated against the HIP
path through the C-ABI, on the reference's own known-answer vectors (tests/golden/reference_vectors.json)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
U = 0xFFFFFFFF


@pytest.fixture()
def ctx(gpe):
    c = gpe.Context(world=(1920.0, 1080.0))      # tests/common.rs:21 ; particle_system.rs:86
    yield c
    c.close()


def _case1(gpe, ctx, golden):
    g = golden["grid_case_1"]
    ps = gpe.ParticleSystem.new_from_buffers(ctx, np.array(g["positions"], np.float32),
                                             np.array(g["radii"], np.float32))
    grid = gpe.Grid.new_without_camera(ctx, g["max_radius"], ps)
    return grid, ps, g


# ---- tests/grid.rs ------------------------------------------------------------------------------
def test_grid_build_cell_ids_with_multiple_particles(gpe, ctx, golden):
    grid, _, g = _case1(gpe, ctx, golden)                                   # tests/grid.rs:12-71
    assert grid.cell_size() == g["cell_size"]
    grid.build_cell_ids()
    assert grid.download_cell_ids().tolist() == g["expected_cell_ids"]
    assert grid.download_object_ids().tolist() == g["expected_object_ids"]


def test_grid_build_cell_ids_and_sort(gpe, ctx, golden):
    grid, _, _ = _case1(gpe, ctx, golden)                                   # tests/grid.rs:134-197
    grid.build_cell_ids()
    grid.sort_map()
    pairs = [list(t) for t in zip(grid.download_cell_ids().tolist(), grid.download_object_ids().tolist())]
    assert pairs == golden["grid_case_1_sorted"]["expected_pairs_sorted"]


def test_grid_build_cell_ids_sort_and_build_empty_collision_cells_list(gpe, ctx, golden):
    grid, ps, _ = _case1(gpe, ctx, golden)                                  # tests/grid.rs:203-226
    cs = gpe.CollisionSystem(ctx, 2, ps, grid)
    grid.build_cell_ids()
    grid.sort_map()
    cs.solve_collisions()
    assert cs.download_collision_cells().tolist() == golden["grid_case_1_collision_cells"]["expected_collision_cells"]
    assert cs.num_collision_cells() == 0


def test_grid_build_cell_ids_sort_and_build_collision_cells_list(gpe, ctx, golden):
    g = golden["grid_case_2"]                                               # tests/grid.rs:265-292
    n = g["num_particles"]
    pos = np.tile(np.array(g["position"], np.float32), (n, 1))
    ps = gpe.ParticleSystem.new_from_buffers(ctx, pos, np.full(n, g["radius"], np.float32))
    grid = gpe.Grid.new_without_camera(ctx, g["max_radius"], ps)
    cs = gpe.CollisionSystem(ctx, 2, ps, grid)
    grid.build_cell_ids()
    grid.sort_map()
    cs.solve_collisions()
    cc = cs.download_collision_cells()
    k = len(g["expected_collision_cells_prefix"])
    assert len(cc) == g["expected_collision_cells_len"]
    assert cc[:k].tolist() == g["expected_collision_cells_prefix"]
    assert (cc[k:] == U).all()
    assert cs.num_collision_cells() == k
    cur, _, _ = ps.download_particle_buffers()
    assert np.array_equal(cur, pos)      # coincident centres: d > 1e-4 guard (collision_solver.wgsl:95)


# ---- tests/particle_sort.rs ------------------------------------------------------------------------
def test_sort_particles(gpe, ctx, golden):
    g = golden["particle_sort"]                                             # tests/particle_sort.rs:9-71
    ps = gpe.ParticleSystem.new_from_buffers(ctx, np.array(g["positions"], np.float32),
                                             np.array(g["radii"], np.float32))
    gpe.Grid.new_without_camera(ctx, g["max_radius"], ps)
    ps.sort_by_cell_id(gpe.Grid.compute_cell_size(g["max_radius"]))
    assert ps.download_home_cell_ids().tolist() == g["expected_home_cell_ids"]
    assert ps.download_particle_ids().tolist() == g["expected_particle_ids"]
    cur, prev, rad = ps.download_particle_buffers()
    assert cur.tolist() == g["expected_positions"]
    assert prev.tolist() == g["expected_previous_positions"]
    assert rad.tolist() == g["expected_radii"]


# ---- tests/radix_sort.rs ---------------------------------------------------------------------------
def test_sort(gpe, ctx, golden):
    n = golden["radix_sort_reversed"]["n"]                                  # tests/radix_sort.rs:7-48
    data = np.arange(n, dtype=np.uint32)[::-1].copy()
    keys, payload = gpe.GpuBuffer(ctx, data), gpe.GpuBuffer(ctx, data)
    sorter = gpe.GPUSorter(ctx, n, keys, payload)
    sorter.sort(None)
    assert np.array_equal(keys.download(), np.arange(n, dtype=np.uint32))
    assert np.array_equal(payload.download(), np.arange(n, dtype=np.uint32))


def test_sort_small_sized_array(gpe, ctx, golden):
    g = golden["radix_sort_small"]                                          # tests/radix_sort.rs:52-125
    data = np.array(g["keys"], np.uint32)
    n = len(data)
    keys, payload = gpe.GpuBuffer(ctx, data), gpe.GpuBuffer(ctx, data)
    sorter = gpe.GPUSorter(ctx, n, keys, payload)
    sorter.build_histogram(n, g["shift"])
    hist = sorter.get_histogram()
    assert hist.sum() == n and len(hist) == g["histogram_len"]
    expected = np.zeros(256, np.uint32)
    for e in data:
        expected[(int(e) >> g["shift"]) & (gpe.RADIX_SORT_BUCKETS - 1)] += 1
    assert np.array_equal(hist, expected)
    sorter.scatter(n, g["shift"])
    assert sorter.get_keys_b().tolist() == g["expected_keys_b"]
    assert sorter.get_payload_b().tolist() == g["expected_keys_b"]


# ---- tests/prefix_sum.rs ---------------------------------------------------------------------------
def _scan_case(gpe, ctx, values):
    buf = gpe.GpuBuffer(ctx, values)
    ps = gpe.PrefixSum(ctx, buf)
    ps.execute(len(values))
    assert np.array_equal(buf.download(), np.cumsum(values, dtype=np.uint64).astype(np.uint32))
    return buf, ps


def test_inclusive_prefix_sum(gpe, ctx, golden):
    n = golden["prefix_sum"]["reversed_ramp_n"]                             # tests/prefix_sum.rs:8-46
    _scan_case(gpe, ctx, np.arange(n, dtype=np.uint32)[::-1].copy())


def test_inclusive_prefix_sum_same_values(gpe, ctx, golden):
    _scan_case(gpe, ctx, np.ones(golden["prefix_sum"]["ones_n"], np.uint32))   # :50-88


def test_inclusive_prefix_sum_all_zero(gpe, ctx, golden):
    _scan_case(gpe, ctx, np.zeros(golden["prefix_sum"]["zeros_n"], np.uint32))  # :91-129


def test_inclusive_prefix_sum_random(gpe, ctx, golden):
    g = golden["prefix_sum"]                                                # :133-168
    rng = np.random.default_rng(99)
    n = int(rng.integers(g["random_n_range"][0], g["random_n_range"][1] + 1))
    _scan_case(gpe, ctx, rng.integers(0, 10, n, dtype=np.uint32))


def test_inclusive_prefix_sum_resize(gpe, ctx, golden):
    g = golden["prefix_sum"]                                                # :171-243
    n = g["resize_from_n"]
    buf, ps = _scan_case(gpe, ctx, np.ones(n, np.uint32))
    buf2 = gpe.GpuBuffer(ctx, np.ones(n, np.uint32))
    buf2.push_all(np.ones(g["resize_add"], np.uint32))
    ps.update_buffers(buf2)
    ps.execute(buf2.len())
    assert np.array_equal(buf2.download(), np.arange(1, n + g["resize_add"] + 1, dtype=np.uint32))
