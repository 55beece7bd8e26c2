"""The CPU oracle against every known-answer vector the reference's own tests hold
(tests/golden/reference_vectors.json, transcribed from /root/reference/tests/*.rs).
CPU only -- this is what pins the oracle before the HIP path is compared with it."""
import numpy as np

U = 0xFFFFFFFF


def _case1(golden):
    g = golden["grid_case_1"]
    return (np.array(g["positions"], np.float32), np.array(g["radii"], np.float32),
            g["max_radius"], g)


def test_cell_size(oracle, golden):
    # grid.rs:159-161 ; tests/grid.rs:109 "implicitly sets cell_size to 22.0"
    assert oracle.compute_cell_size(golden["grid_case_1"]["max_radius"]) == golden["grid_case_1"]["cell_size"]
    assert oracle.compute_cell_size(0.5) == np.float32(0.5) * np.float32(2.2)


def test_morton_encode_examples(oracle, golden):
    # tests/grid.rs:24-54 expected ids are morton_encode(x, y) of the listed cell coordinates
    g = golden["grid_case_1"]
    flat = [c for per in g["expected_cell_coords"] for c in per]
    ids = [oracle.morton_encode(x, y) for x, y in flat]
    assert ids == [v for v in g["expected_cell_ids"] if v != U]
    assert oracle.morton_encode(3, 3) == 15          # grid.wgsl:111 doc example
    assert oracle.morton_encode(-1, 0) == 0x55555555  # 16-bit mask of a negative coordinate
    assert oracle.morton_encode(0, -1) == 0xAAAAAAAA


def test_build_cell_ids_case_1(oracle, golden):
    # tests/grid.rs:12-71
    pos, rad, max_r, g = _case1(golden)
    cell_ids, object_ids = oracle.build_cell_ids(pos, rad, oracle.compute_cell_size(max_r))
    assert cell_ids.tolist() == g["expected_cell_ids"]
    assert object_ids.tolist() == g["expected_object_ids"]


def test_build_cell_ids_and_sort_case_1(oracle, golden):
    # tests/grid.rs:134-197
    pos, rad, max_r, _ = _case1(golden)
    cell_ids, object_ids = oracle.build_cell_ids(pos, rad, oracle.compute_cell_size(max_r))
    k, v = oracle.sort_pairs(cell_ids, object_ids)
    assert [list(t) for t in zip(k.tolist(), v.tolist())] == golden["grid_case_1_sorted"]["expected_pairs_sorted"]


def test_empty_collision_cells_case_1(oracle, golden):
    # tests/grid.rs:203-226 (runs the whole solve too)
    pos, rad, max_r, _ = _case1(golden)
    p = oracle.default_params(1920.0, 1080.0, max_r)
    sim = oracle.Sim(pos, rad, p)
    sim.grid_build(); sim.grid_sort(); sim.build_collision_cells(); sim.solve_colors()
    assert sim.collision_cells.tolist() == golden["grid_case_1_collision_cells"]["expected_collision_cells"]
    assert sim.num_collision_cells == 0


def test_collision_cells_case_2(oracle, golden):
    # tests/grid.rs:265-292 : 546 coincident particles => 4 cells x 546 entries
    g = golden["grid_case_2"]
    n = g["num_particles"]
    pos = np.tile(np.array(g["position"], np.float32), (n, 1))
    rad = np.full(n, g["radius"], np.float32)
    p = oracle.default_params(1920.0, 1080.0, g["max_radius"])
    sim = oracle.Sim(pos, rad, p)
    sim.grid_build(); sim.grid_sort(); sim.build_collision_cells(); sim.solve_colors()
    cc = sim.collision_cells
    assert len(cc) == g["expected_collision_cells_len"]
    k = len(g["expected_collision_cells_prefix"])
    assert cc[:k].tolist() == g["expected_collision_cells_prefix"]
    assert (cc[k:] == U).all()
    assert sim.num_collision_cells == k
    # coincident particles: distance 0 fails the d > 1e-4 guard (collision_solver.wgsl:95)
    assert np.array_equal(sim.pos, pos)


def test_particle_sort(oracle, golden):
    # tests/particle_sort.rs:9-71
    g = golden["particle_sort"]
    pos = np.array(g["positions"], np.float32)
    rad = np.array(g["radii"], np.float32)
    p = oracle.default_params(1920.0, 1080.0, g["max_radius"])
    sim = oracle.Sim(pos, rad, p)
    sim.morton_resort()
    assert sim.home_cell_ids.tolist() == g["expected_home_cell_ids"]
    assert sim.particle_ids.tolist() == g["expected_particle_ids"]
    assert sim.pos.tolist() == g["expected_positions"]
    assert sim.prev.tolist() == g["expected_previous_positions"]
    assert sim.radius.tolist() == g["expected_radii"]


def test_radix_sort_reversed(oracle, golden):
    # tests/radix_sort.rs:7-48
    n = golden["radix_sort_reversed"]["n"]
    data = np.arange(n, dtype=np.uint32)[::-1].copy()
    k, v = oracle.sort_pairs(data, data)
    assert np.array_equal(k, np.arange(n, dtype=np.uint32))
    assert np.array_equal(v, np.arange(n, dtype=np.uint32))
    assert oracle.radix_num_wg(n) == 3


def test_radix_sort_small_histogram_and_scatter(oracle, golden):
    # tests/radix_sort.rs:52-125
    g = golden["radix_sort_small"]
    keys = np.array(g["keys"], np.uint32)
    hist = oracle.radix_build_histogram(keys, g["shift"])
    assert len(hist) == g["histogram_len"]
    assert hist.sum() == len(keys)
    expected = np.zeros(256, np.uint32)
    for e in keys:
        expected[(int(e) >> g["shift"]) & 255] += 1
    assert np.array_equal(hist, expected)
    kb, pb = oracle.radix_scatter(keys, keys, g["shift"], hist)
    assert kb.tolist() == g["expected_keys_b"]      # 257 before 1: stability
    assert pb.tolist() == g["expected_keys_b"]


def test_prefix_sum_vectors(oracle, golden):
    # tests/prefix_sum.rs:8-129
    g = golden["prefix_sum"]
    for data in (np.arange(g["reversed_ramp_n"], dtype=np.uint32)[::-1].copy(),
                 np.ones(g["ones_n"], np.uint32), np.zeros(g["zeros_n"], np.uint32)):
        assert np.array_equal(oracle.inclusive_scan(data), np.cumsum(data, dtype=np.uint32))


def test_prefix_sum_random_large(oracle, golden):
    # tests/prefix_sum.rs:133-168 (size drawn in the reference's range, values 0..=9)
    g = golden["prefix_sum"]
    rng = np.random.default_rng(1234)
    n = int(rng.integers(g["random_n_range"][0], g["random_n_range"][1] + 1))
    data = rng.integers(0, 10, n, dtype=np.uint32)
    assert np.array_equal(oracle.inclusive_scan(data), np.cumsum(data, dtype=np.uint32))


def test_scan_wraps_mod_2_32(oracle):
    data = np.full(10, 0x40000000, np.uint32)
    assert np.array_equal(oracle.inclusive_scan(data), np.cumsum(data, dtype=np.uint64).astype(np.uint32))


def test_threaded_oracle_gives_the_same_bits(oracle):
    """The multi-core CPU baseline of bench.py (orc_set_threads > 1, OpenMP over the loops that are separate GPU
    threads in the WGSL) must be the same computation as the serial oracle the parity tests use."""
    rng = np.random.default_rng(9)
    n, world = 30_000, (190.0, 160.0)
    pos = (rng.random((n, 2), dtype=np.float32) * np.array(world, np.float32)).astype(np.float32)
    rad = np.where(np.arange(n) % 3 == 0, np.float32(0.4), np.float32(0.5)).astype(np.float32)
    out = []
    try:
        for threads in (1, 4):
            oracle.set_threads(threads)
            assert oracle.get_threads() == threads
            p = oracle.default_params(world[0], world[1], 0.5, gravity=(1.0, -9.81))
            sim = oracle.Sim(pos, rad, p)
            for s in range(6):
                sim.step(1 / 60, resort=(s in (0, 3)))
            out.append((sim.pos.copy(), sim.prev.copy(), sim.cell_ids.copy(), sim.collision_cells.copy()))
            sim.close()
    finally:
        oracle.set_threads(1)
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)
