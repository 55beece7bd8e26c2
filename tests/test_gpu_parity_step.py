"""GPU (-m gpu): whole-step parity of the HIP path (through the C-ABI) against the CPU oracle on the
same seeded inputs.  Bar: bit-exact for every integer artefact (Morton ids, (cell,object) pairs, sorted
order, collision-cell list) and -- because both sides evaluate the same IEEE binary32 operations in the
same order with no FMA contraction -- bit-exact positions; the north-star tolerance for positions vs the
reference WGSL is 1e-5 relative (driver-defined sqrt/div rounding there), asserted as the fallback bound."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
U = 0xFFFFFFFF
REL_TOL = 1e-5


def _assert_positions(got, want, what):
    if np.array_equal(got, want):
        return
    # not bit-exact: report how far, then fail unless inside the north-star tolerance
    denom = np.maximum(np.abs(want), 1e-30)
    rel = np.max(np.abs(got - want) / denom)
    nbad = int((got != want).sum())
    assert rel <= REL_TOL, "%s: %d values differ, max rel %.3g" % (what, nbad, rel)
    pytest.fail("%s: within 1e-5 (max rel %.3g, %d values) but NOT bit-exact vs the oracle" % (what, rel, nbad))


def _scene(gpe, kind, n, seed):
    if kind == "reference_density":
        world = gpe.scenes.world_for(n)
        pos, rad = gpe.scenes.uniform_cloud(n, world, seed=seed)
    elif kind == "dense":
        world = gpe.scenes.world_for(n, density=1.0)
        pos, rad = gpe.scenes.uniform_cloud(n, world, seed=seed)
    elif kind == "mixed_radii":
        world = gpe.scenes.world_for(n, density=0.02)
        pos, rad = gpe.scenes.mixed_radius_cloud(n, world, seed=seed)
    else:
        raise ValueError(kind)
    return world, pos, rad


@pytest.mark.parametrize("kind,n,steps", [
    ("reference_density", 1, 3), ("reference_density", 2, 3), ("reference_density", 1000, 20),
    ("reference_density", 20_000, 12), ("dense", 20_000, 8), ("mixed_radii", 20_000, 8),
    ("reference_density", 200_000, 4),
])
def test_step_matches_oracle(gpe, oracle, kind, n, steps):
    world, pos, rad = _scene(gpe, kind, n, seed=1000 + n)
    max_r = float(np.abs(rad).max())
    st = gpe.State(pos, rad, world=world, mode=gpe.MODE_COMPAT)
    sim = oracle.Sim(pos, rad, oracle.default_params(world[0], world[1], max_r))
    dt = 1.0 / 60.0
    for s in range(steps):
        resort = (s == 0) or (s == 5)            # first frame always sorts (particle_system.rs:45)
        st.update(dt, resort=resort)
        sim.step(dt, resort=resort)
        if s == 0:
            # integer artefacts of the first step
            assert np.array_equal(st.particles.download_home_cell_ids(), sim.home_cell_ids)
            assert np.array_equal(st.particles.download_particle_ids(), sim.particle_ids)
            k = sim.num_collision_cells
            assert st.collision_system.num_collision_cells() == k
            cid, oid = st.grid.download_cell_ids(), st.grid.download_object_ids()
            assert np.array_equal(cid, sim.cell_ids)
            used = sim.cell_ids != U
            assert np.array_equal(oid[used], sim.object_ids[used])
            assert np.array_equal(st.collision_system.download_collision_cells()[:k], sim.collision_cells[:k])
    _assert_positions(st.positions(), sim.pos, "positions after %d steps" % steps)
    _assert_positions(st.previous_positions(), sim.prev, "previous positions")
    # later-step integer artefacts too (stale object ids behind UNUSED keys are not compared: SURVEY App. B.1)
    cid = st.grid.download_cell_ids()
    assert np.array_equal(cid, sim.cell_ids)
    used = cid != U
    assert np.array_equal(st.grid.download_object_ids()[used], sim.object_ids[used])
    k = sim.num_collision_cells
    assert st.collision_system.num_collision_cells() == k
    assert np.array_equal(st.collision_system.download_collision_cells()[:k], sim.collision_cells[:k])
    st.close(); sim.close()


def test_gravity_and_mouse_match_oracle(gpe, oracle):
    n = 5000
    world, pos, rad = _scene(gpe, "reference_density", n, seed=77)
    g = (0.0, -9.81)
    st = gpe.State(pos, rad, world=world, mode=gpe.MODE_COMPAT, gravity=g)
    p = oracle.default_params(world[0], world[1], 0.5, gravity=g)
    p.mouse_pressed, p.mouse_x, p.mouse_y = 1, world[0] * 0.4, world[1] * 0.6
    st.particles.mouse_click_callback(True, (p.mouse_x, p.mouse_y))
    sim = oracle.Sim(pos, rad, p)
    for s in range(15):
        st.update(1.0 / 60.0, resort=(s == 0))
        sim.step(1.0 / 60.0, resort=(s == 0))
    _assert_positions(st.positions(), sim.pos, "positions (gravity + mouse)")
    _assert_positions(st.previous_positions(), sim.prev, "previous positions (gravity + mouse)")
    st.close(); sim.close()


def test_negative_phantom_cells_and_unused_alias(gpe, oracle):
    """Particles closer than r to the x=0 / y=0 walls overlap cell -1, which aliases to coordinate 65535
    (grid.wgsl:102); the corner cell (-1,-1) hashes to 0xFFFFFFFF == UNUSED_CELL_ID and is never a
    collision cell (collision_cell_builder.wgsl:56).  SURVEY Appendix B.7."""
    pos = np.array([[0.2, 0.2], [0.3, 0.25], [0.1, 5.0], [0.15, 5.2], [7.0, 0.05], [7.3, 0.1],
                    [0.0, 0.0], [0.05, 0.02], [30.0, 30.0]], np.float32)
    rad = np.full(len(pos), 0.5, np.float32)
    world = (64.0, 64.0)
    st = gpe.State(pos, rad, world=world, mode=gpe.MODE_COMPAT)
    sim = oracle.Sim(pos, rad, oracle.default_params(world[0], world[1], 0.5))
    st.grid.build_cell_ids(); sim.grid_build()
    assert np.array_equal(st.grid.download_cell_ids(), sim.cell_ids)
    assert (sim.cell_ids.reshape(-1, 4)[:, 1:] == U).any() and (sim.cell_ids == 0x55555555).any()
    for s in range(6):
        st.update(1.0 / 60.0, resort=(s == 0)); sim.step(1.0 / 60.0, resort=(s == 0))
        if s == 0:
            k = sim.num_collision_cells
            assert st.collision_system.num_collision_cells() == k
            assert np.array_equal(st.collision_system.download_collision_cells()[:k], sim.collision_cells[:k])
    _assert_positions(st.positions(), sim.pos, "positions near the origin walls")
    st.close(); sim.close()


def test_module_calls_compose_like_step(gpe, oracle):
    """gpe_grid_build / gpe_grid_sort / gpe_solve_collisions / gpe_integrate one by one == gpe_step."""
    n = 10_000
    world, pos, rad = _scene(gpe, "dense", n, seed=5)
    a = gpe.State(pos, rad, world=world, mode=gpe.MODE_COMPAT)
    b = gpe.State(pos, rad, world=world, mode=gpe.MODE_COMPAT)
    for s in range(4):
        a.update(0.01, resort=(s == 0))
        if s == 0:
            b.particles.sort_by_cell_id(b.grid.cell_size())
        b.grid.build_cell_ids(); b.grid.sort_map(); b.collision_system.solve_collisions()
        b.particles.update_positions(0.01)
    assert np.array_equal(a.positions(), b.positions())
    assert np.array_equal(a.previous_positions(), b.previous_positions())
    a.close(); b.close()


def test_add_particles_matches_fresh_system(gpe, oracle):
    """State::add_particles (state.rs:187-200): grown buffers, new max radius => new cell size."""
    n = 3000
    world = (200.0, 120.0)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=21)
    extra_pos, extra_rad = gpe.scenes.mixed_radius_cloud(100, world, seed=22, radii=(1.0, 2.0, 3.0))
    st = gpe.State(pos, rad, world=world, mode=gpe.MODE_COMPAT)
    st.update(1 / 60, resort=True)
    cur, prev = st.positions(), st.previous_positions()
    perm = st.particles.download_particle_ids()
    st.add_particles(extra_pos, extra_rad)
    assert st.particles.len() == n + 100
    assert st.particles.get_max_radius() == 3.0
    assert st.grid.cell_size() == np.float32(3.0) * np.float32(2.2)
    all_pos = np.concatenate([cur, extra_pos]); all_prev = np.concatenate([prev, extra_pos])
    all_rad = np.concatenate([rad[perm], extra_rad])
    sim = oracle.Sim(all_pos, all_rad, oracle.default_params(world[0], world[1], 3.0), prev=all_prev)
    for s in range(5):
        st.update(1 / 60, resort=(s == 2)); sim.step(1 / 60, resort=(s == 2))
    _assert_positions(st.positions(), sim.pos, "positions after add_particles")
    st.close(); sim.close()


def test_run_equals_repeated_step(gpe):
    n = 20_000
    world, pos, rad = _scene(gpe, "reference_density", n, seed=9)
    a = gpe.State(pos, rad, world=world, mode=gpe.MODE_COMPAT)
    b = gpe.State(pos, rad, world=world, mode=gpe.MODE_COMPAT)
    a.run(1 / 60, 12, resort_every=5, resort_first=True)
    for s in range(12):
        b.update(1 / 60, resort=(s % 5 == 0))
    assert np.array_equal(a.positions(), b.positions())
    a.close(); b.close()


def test_full_size_1m_properties(gpe, oracle):
    """BASELINE config 2 (1M particles, reference world): size-independent properties + an oracle step."""
    n = 1_000_000
    world = gpe.scenes.REF_WORLD
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=0x5EED)
    st = gpe.State(pos, rad, world=world, mode=gpe.MODE_COMPAT)
    sim = oracle.Sim(pos, rad, oracle.default_params(world[0], world[1], 0.5))
    st.update(1 / 60, resort=True); sim.step(1 / 60, resort=True)
    ids = st.particles.download_particle_ids()
    assert np.array_equal(np.sort(ids), np.arange(n, dtype=np.uint32))           # a permutation
    home = st.particles.download_home_cell_ids()
    assert (np.diff(home.astype(np.int64)) >= 0).all()                           # Morton-sorted
    cid = st.grid.download_cell_ids()
    assert (np.diff(cid.astype(np.int64)) >= 0).all()
    assert np.array_equal(cid, sim.cell_ids)
    k = sim.num_collision_cells
    assert st.collision_system.num_collision_cells() == k
    assert np.array_equal(st.collision_system.download_collision_cells()[:k], sim.collision_cells[:k])
    _assert_positions(st.positions(), sim.pos, "1M positions after one step")
    p = st.positions()
    assert (p[:, 0] >= 0.5).all() and (p[:, 0] <= world[0] - 0.5).all()          # wall clamp
    st.close(); sim.close()


def test_failed_growth_leaves_the_context_intact(gpe, oracle):
    """gpe_reserve for more memory than the device has: GPE_ERR_OOM, and the context is exactly as before -- the new
    buffer set is allocated beside the old one and replaces it only when every allocation and copy has succeeded
    (round 1 nulled the pointers first: a failed growth lost the particles and the next step dereferenced NULL)."""
    n = 5000
    world = gpe.scenes.world_for(n)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=77)
    st = gpe.State(pos, rad, world=world, mode=gpe.MODE_NATIVE)
    sim = oracle.Sim(pos, rad, oracle.default_params(world[0], world[1], 0.5))
    st.update(1 / 60, resort=True); sim.step(1 / 60, resort=True)
    import ctypes as C
    blocker = C.c_void_p()
    st.ctx.call("gpe_buffer_alloc", 262 << 30, C.byref(blocker))       # leaves ~25 of the 288 GB free
    try:
        with pytest.raises(gpe.GpeError) as e:
            st.ctx.call("gpe_reserve", 400_000_000)                     # ~170 B per particle = 68 GB
        assert e.value.status == gpe._lib.GPE_ERR_OOM
    finally:
        st.ctx.call("gpe_buffer_free", blocker)
    for s in range(3):
        st.update(1 / 60, resort=(s == 1)); sim.step(1 / 60, resort=(s == 1))
    assert np.array_equal(st.positions(), sim.pos)
    st.ctx.sync()
    st.close(); sim.close()


def test_max_radius_takes_the_last_of_equal_magnitudes(gpe):
    """particle_system.rs:51 uses Iterator::max_by, which returns the LAST maximum: radii [2, -2] give max_radius -2
    (and a negative cell size), [-2, 2] give 2."""
    import ctypes as C
    pos = np.array([[5.0, 5.0], [9.0, 9.0]], np.float32)
    for radii, want in (([2.0, -2.0], -2.0), ([-2.0, 2.0], 2.0), ([1.0, 3.0, -3.0, 2.0], -3.0)):
        p = np.zeros((len(radii), 2), np.float32) + 5.0
        ctx = gpe.Context(world=(100.0, 100.0))
        r = np.array(radii, np.float32)
        ctx.call("gpe_set_particles", p.ctypes.data_as(C.c_void_p), None, r.ctypes.data_as(C.c_void_p), len(radii))
        out = C.c_float()
        ctx.call("gpe_max_radius", C.byref(out))
        assert out.value == want, (radii, out.value)
        ctx.close()


def test_native_context_allocates_grid_buffers_on_demand(gpe, oracle):
    """A NATIVE context whose steps run on the native kernels does not hold the reference's 4N grid /
    collision-cell arrays and 4N sort partners (84 B per particle); the first per-module call that needs them
    allocates them, and the module calls then give the reference's results."""
    import ctypes

    def free_bytes():
        # hipMemGetInfo of the runtime libgpe.so itself links (torch carries a second copy of the HIP runtime, which
        # refuses to initialise when this one already owns the device)
        hip = ctypes.CDLL("libamdhip64.so.7")
        free, total = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value

    n = 4_000_000
    world = gpe.scenes.world_for(n)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=77)
    warm = gpe.Context(); warm.close()                 # the runtime is up before the first reading
    free0 = free_bytes()
    st = gpe.State(pos, rad, world=world, mode=gpe.MODE_NATIVE)
    st.run(1 / 60, 3, resort_every=0, resort_first=True)
    st.ctx.sync()
    lean = free0 - free_bytes()
    st.grid.update()                                   # Grid::update -> the 4N arrays and the 4N sort partners
    st.ctx.sync()
    full = free0 - free_bytes()
    assert full - lean >= 72 * n, (lean, full)         # 52 B of 4N arrays + the sort partners growing from N to 4N
    assert lean <= 200 * n + (96 << 20), (lean, full)  # SoA x 2, index arrays, native sort buffers, spill arena
    # and the arrays hold what the oracle's grid holds for the same positions
    sim = oracle.Sim(pos, rad, oracle.default_params(world[0], world[1], 0.5))
    for s in range(3):
        sim.step(1 / 60, resort=(s == 0))
    sim.grid_build(); sim.grid_sort()
    cells = st.grid.download_cell_ids().ravel()
    assert np.array_equal(cells, sim.cell_ids.ravel())
    used = cells != 0xFFFFFFFF          # (the object ids of unused slots are whatever earlier steps left there:
    assert used.sum() >= n              #  grid.wgsl writes them for used slots only, and these arrays are fresh)
    assert np.array_equal(st.grid.download_object_ids().ravel()[used], sim.object_ids.ravel()[used])
    st.close(); sim.close()


def test_switching_pipelines_mid_run_keeps_the_trajectory(gpe, oracle):
    """gpe_set_mode (ABI only: the reference has one pipeline): NATIVE -> COMPAT -> NATIVE in the middle of a run.
    The first switch is what allocates the reference's 4N arrays in a context created NATIVE; positions stay
    bit-identical to the oracle throughout."""
    n = 50_000
    world = gpe.scenes.world_for(n)
    pos, rad = gpe.scenes.mixed_radius_cloud(n, gpe.scenes.world_for(n, density=0.05), seed=91)
    world = gpe.scenes.world_for(n, density=0.05)
    g = (0.0, -9.81)
    st = gpe.State(pos, rad, world=world, gravity=g, mode=gpe.MODE_NATIVE)
    st.ctx.set_profiling(True)
    sim = oracle.Sim(pos, rad, oracle.default_params(world[0], world[1], float(np.abs(rad).max()), gravity=g))
    for phase, mode in enumerate([None, gpe.MODE_COMPAT, gpe.MODE_NATIVE]):
        if mode is not None:
            st.ctx.call("gpe_set_mode", mode)
        for s in range(4):
            resort = (phase == 0 and s == 0) or (phase == 2 and s == 1)
            st.update(1 / 60, resort=resort); sim.step(1 / 60, resort=resort)
        assert np.array_equal(st.positions().view(np.uint32), sim.pos.view(np.uint32)), "phase %d" % phase
        assert np.array_equal(st.previous_positions().view(np.uint32), sim.prev.view(np.uint32)), "phase %d" % phase
    st.ctx.sync()
    tim = st.ctx.timings()
    assert tim["native/collide+verlet"][1] == 8                    # phases 0 and 2 ran on the native kernels
    assert tim["Sort map"][1] == 4                                 # phase 1 on the reference's pipeline (grid.rs:329)
    st.close(); sim.close()
