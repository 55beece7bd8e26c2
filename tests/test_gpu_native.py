"""GPU (-m gpu): the MI355X-native pipeline (GPE_MODE_NATIVE: N-key onesweep sort + LDS-staged cell windows,
all four colour passes fused) against the CPU oracle and against the compat pipeline, through the C-ABI.
Bar: positions bit-exact (same IEEE binary32 operation sequence per particle pair);
1e-5 relative is the north-star bound vs the reference WGSL and the reported fallback here."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REL_TOL = 1e-5


def _assert_positions(got, want, what):
    if np.array_equal(got, want):
        return
    denom = np.maximum(np.abs(want), 1e-30)
    rel = np.max(np.abs(got - want) / denom)
    nbad = int((got != want).any(axis=-1).sum()) if got.ndim == 2 else int((got != want).sum())
    assert rel <= REL_TOL, "%s: %d particles differ, max rel %.3g" % (what, nbad, rel)
    pytest.fail("%s: within 1e-5 (max rel %.3g, %d particles) but NOT bit-exact" % (what, rel, nbad))


def _native(gpe, pos, rad, world, **kw):
    st = gpe.State(pos, rad, world=world, mode=gpe.MODE_NATIVE, **kw)
    return st


@pytest.mark.parametrize("kind,n,steps,density", [
    ("uniform", 1, 3, None), ("uniform", 2, 3, None), ("uniform", 500, 10, None),
    ("uniform", 20_000, 12, None), ("uniform", 200_000, 6, None),
    ("uniform", 30_000, 6, 1.0),          # dense: most cells are collision cells
    ("uniform", 60_000, 4, 2.0),          # 32x32-cell regions over LDS capacity -> 8x8 sub-tiles
    ("mixed", 20_000, 8, 0.02),           # radii 0.5..3: phantom cells of big particles
])
def test_native_step_matches_oracle(gpe, oracle, kind, n, steps, density):
    if kind == "mixed":
        world = gpe.scenes.world_for(n, density=density)
        pos, rad = gpe.scenes.mixed_radius_cloud(n, world, seed=n)
    else:
        world = gpe.scenes.world_for(n) if density is None else gpe.scenes.world_for(n, density=density)
        pos, rad = gpe.scenes.uniform_cloud(n, world, seed=2000 + n)
    st = _native(gpe, pos, rad, world)
    sim = oracle.Sim(pos, rad, oracle.default_params(world[0], world[1], float(np.abs(rad).max())))
    for s in range(steps):
        resort = s in (0, 4)
        st.update(1 / 60, resort=resort)
        sim.step(1 / 60, resort=resort)
        if s == 0:
            _assert_positions(st.positions(), sim.pos, "positions after the first step")
    _assert_positions(st.positions(), sim.pos, "positions after %d steps" % steps)
    _assert_positions(st.previous_positions(), sim.prev, "previous positions")
    assert np.array_equal(st.particles.download_particle_ids(), sim.particle_ids)
    st.ctx.sync()
    st.close(); sim.close()


def test_native_equals_compat_1m(gpe):
    """BASELINE config 2 size: native and compat pipelines agree bit for bit over 12 steps."""
    n = 1_000_000
    world = gpe.scenes.REF_WORLD
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=0x5EED)
    a = _native(gpe, pos, rad, world)
    b = gpe.State(pos, rad, world=world, mode=gpe.MODE_COMPAT)
    a.run(1 / 60, 12, resort_every=5, resort_first=True)
    b.run(1 / 60, 12, resort_every=5, resort_first=True)
    assert np.array_equal(a.positions(), b.positions())
    assert np.array_equal(a.previous_positions(), b.previous_positions())
    a.ctx.sync()
    a.close(); b.close()


def test_native_equals_compat_100m(gpe):
    """BASELINE config 3 size (100 M particles, gravity on, 30480 x 10480): the native pipeline and the compat
    pipeline (the reference's algorithm buffer for buffer, itself checked against the oracle at sizes the oracle
    finishes) agree bit for bit over 3 steps; plus the size-independent properties of the step."""
    n = 100_000_000
    world = gpe.scenes.world_for(n)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=0x5EED)
    g = (0.0, -9.81)
    out = []
    for mode in (gpe.MODE_NATIVE, gpe.MODE_COMPAT):
        st = gpe.State(pos, rad, world=world, gravity=g, mode=mode)
        st.run(1 / 60, 3, resort_every=0, resort_first=True)
        out.append((st.positions(), st.previous_positions()))
        if mode == gpe.MODE_NATIVE:
            ids = st.particles.download_particle_ids()
            assert np.array_equal(np.sort(ids), np.arange(n, dtype=np.uint32))       # the re-sort is a permutation
            home = st.particles.download_home_cell_ids()
            assert (np.diff(home.astype(np.int64)) >= 0).all()                       # in Morton order
            del ids, home
        st.ctx.sync()
        st.close()
    del pos, rad
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1])
    p = out[0][0]
    assert np.isfinite(p).all()
    assert (p[:, 0] >= 0.5).all() and (p[:, 0] <= np.float32(world[0]) - 0.5).all()  # wall clamp
    assert (p[:, 1] >= 0.5).all() and (p[:, 1] <= np.float32(world[1]) - 0.5).all()


def test_config2_100m_gravity_on_matches_oracle_directly(gpe, oracle):
    """BASELINE.json configs[2] itself -- 100 M particles, gravity on, 30480 x 10480 -- NATIVE against the ORACLE (not
    against the compat kernels): 2 steps, the first one re-sorting, the oracle on the box's CPU share (same bits for any
    thread count, tests/test_oracle_golden.py).  Positions, previous positions and the re-sort permutation, bit for bit."""
    n = 100_000_000
    world = gpe.scenes.world_for(n)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=0x5EED)
    g = (0.0, -9.81)
    st = _native(gpe, pos, rad, world, gravity=g)
    oracle.set_threads(_threads())
    try:
        sim = oracle.Sim(pos, rad, oracle.default_params(world[0], world[1], 0.5, gravity=g))
        del pos, rad
        for s in range(2):
            st.update(1 / 60, resort=(s == 0)); sim.step(1 / 60, resort=(s == 0))
        assert np.array_equal(st.positions(), sim.pos), "positions (100 M, gravity on, 2 steps)"
        assert np.array_equal(st.previous_positions(), sim.prev), "previous positions (100 M, gravity on, 2 steps)"
        assert np.array_equal(st.particles.download_particle_ids(), sim.particle_ids)
        info = st.ctx.pipeline_info()
        assert info["pipeline"] == gpe._lib.PIPELINE_NATIVE and info["native_steps"] == 2 and info["sort_passes"] == 3, info
        st.ctx.sync()
        st.close(); sim.close()
    finally:
        oracle.set_threads(1)


def test_native_gravity_mouse_match_oracle(gpe, oracle):
    n = 8000
    world = gpe.scenes.world_for(n)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=31)
    g = (0.0, -9.81)
    st = _native(gpe, pos, rad, world, gravity=g)
    p = oracle.default_params(world[0], world[1], 0.5, gravity=g)
    p.mouse_pressed, p.mouse_x, p.mouse_y = 1, world[0] * 0.3, world[1] * 0.7
    st.particles.mouse_click_callback(True, (p.mouse_x, p.mouse_y))
    sim = oracle.Sim(pos, rad, p)
    for s in range(40):
        st.update(1 / 60, resort=(s == 0)); sim.step(1 / 60, resort=(s == 0))
    _assert_positions(st.positions(), sim.pos, "positions (gravity + mouse, 40 steps)")
    st.close(); sim.close()


def test_native_walls_and_unused_alias(gpe, oracle):
    """Phantom cells at coordinate -1 (x < r or y < r) and the (-1,-1) cell that aliases UNUSED_CELL_ID."""
    rng = np.random.default_rng(4)
    world = (40.0, 40.0)
    edge = np.concatenate([
        np.stack([rng.random(300, dtype=np.float32) * 0.6, rng.random(300, dtype=np.float32) * 40], 1),
        np.stack([rng.random(300, dtype=np.float32) * 40, rng.random(300, dtype=np.float32) * 0.6], 1),
        rng.random((200, 2), dtype=np.float32) * 0.8,
        np.stack([40 - rng.random(300, dtype=np.float32) * 0.6, rng.random(300, dtype=np.float32) * 40], 1),
        np.stack([rng.random(300, dtype=np.float32) * 40, 40 - rng.random(300, dtype=np.float32) * 0.6], 1),
        rng.random((600, 2), dtype=np.float32) * 40,
    ]).astype(np.float32)
    rad = np.full(len(edge), 0.5, np.float32)
    st = _native(gpe, edge, rad, world)
    sim = oracle.Sim(edge, rad, oracle.default_params(world[0], world[1], 0.5))
    for s in range(8):
        st.update(1 / 60, resort=(s == 0)); sim.step(1 / 60, resort=(s == 0))
        if s == 0:
            _assert_positions(st.positions(), sim.pos, "first step (before any wall clamp)")
    _assert_positions(st.positions(), sim.pos, "positions along the walls")
    st.close(); sim.close()


def test_native_coincident_particles(gpe, oracle, golden):
    """tests/grid.rs:229-263 scene (546 particles at one point, radius 10) through the native path."""
    g = golden["grid_case_2"]
    n = g["num_particles"]
    pos = np.tile(np.array(g["position"], np.float32), (n, 1))
    pos[::7] += np.float32(0.37)          # some distinct positions so that pairs do get resolved
    rad = np.full(n, g["radius"], np.float32)
    world = (1920.0, 1080.0)
    st = _native(gpe, pos, rad, world)
    sim = oracle.Sim(pos, rad, oracle.default_params(world[0], world[1], g["max_radius"]))
    for s in range(3):
        st.update(1 / 60, resort=(s == 0)); sim.step(1 / 60, resort=(s == 0))
    _assert_positions(st.positions(), sim.pos, "coincident pile")
    st.close(); sim.close()


def test_native_out_of_box_scene_uses_compat_kernels(gpe, oracle):
    """Particles outside [0,W]x[0,H] (negative cells wrap to 65535, grid.wgsl:102) are outside the native
    cell box: the context runs the compat kernels instead -- same bits, no error."""
    n = 2000
    world = (100.0, 80.0)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=3)
    pos[:50, 0] -= 30.0
    pos[50:100, 1] += 70.0
    st = _native(gpe, pos, rad, world)
    sim = oracle.Sim(pos, rad, oracle.default_params(world[0], world[1], 0.5))
    for s in range(4):
        st.update(1 / 60, resort=(s == 0)); sim.step(1 / 60, resort=(s == 0))
    _assert_positions(st.positions(), sim.pos, "out-of-box start")
    st.ctx.sync()
    info = st.ctx.pipeline_info()
    assert (info["pipeline"], info["reason"]) == (gpe._lib.PIPELINE_COMPAT, gpe._lib.REASON_OUT_OF_BOX), info
    st.close(); sim.close()


def test_native_overfull_windows_use_compat_kernels(gpe, oracle):
    """More particles in a 24x24-cell window than a context may ENTER the native kernels with (24576: here ~42 per
    unit^2): the configuration-time check sends the context to the compat kernels -- exact, no error, and
    gpe_get_pipeline_info says which kernels ran and why.  The same scene at a third of the density stays native
    (its windows go through the sub-tile and spill windows)."""
    L = gpe._lib
    for n, expect in ((30_000, (L.PIPELINE_COMPAT, L.REASON_DENSE_WINDOWS)), (12_000, (L.PIPELINE_NATIVE, L.REASON_NONE))):
        world = (26.0, 26.0) if n == 30_000 else (33.0, 33.0)
        pos, rad = gpe.scenes.uniform_cloud(n, world, seed=12)
        st = _native(gpe, pos, rad, world)
        sim = oracle.Sim(pos, rad, oracle.default_params(world[0], world[1], 0.5))
        oracle.set_threads(8)
        try:
            for s in range(3):
                st.update(1 / 60, resort=(s == 0)); sim.step(1 / 60, resort=(s == 0))
        finally:
            oracle.set_threads(1)
        _assert_positions(st.positions(), sim.pos, "over-dense scene, n = %d" % n)
        st.ctx.sync()
        info = st.ctx.pipeline_info()                      # the host can see which kernels ran, and why
        assert (info["pipeline"], info["reason"]) == expect, info
        assert (info["compat_steps"], info["native_steps"]) == ((3, 0) if expect[0] == L.PIPELINE_COMPAT else (0, 3)), info
        st.close(); sim.close()


def test_native_add_particles(gpe, oracle):
    n = 4000
    world = (220.0, 140.0)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=41)
    extra_pos, extra_rad = gpe.scenes.mixed_radius_cloud(100, world, seed=42, radii=(1.0, 2.0, 3.0))
    st = _native(gpe, pos, rad, world)
    st.update(1 / 60, resort=True)
    cur, prev, perm = st.positions(), st.previous_positions(), st.particles.download_particle_ids()
    st.add_particles(extra_pos, extra_rad)
    sim = oracle.Sim(np.concatenate([cur, extra_pos]), np.concatenate([rad[perm], extra_rad]),
                     oracle.default_params(world[0], world[1], 3.0), prev=np.concatenate([prev, extra_pos]))
    for s in range(6):
        st.update(1 / 60, resort=(s == 3)); sim.step(1 / 60, resort=(s == 3))
    _assert_positions(st.positions(), sim.pos, "positions after add_particles (native)")
    st.close(); sim.close()


def test_native_spill_arena_is_exact(gpe, oracle):
    """GPE_FLAG_NATIVE_FORCE keeps an over-dense scene on the native kernels: every 8x8 tile overflows the LDS
    windows and is resolved with its particle arrays in the global spill arena -- same bits as the oracle."""
    n = 12_000
    world = (33.0, 33.0)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=12)
    st = _native(gpe, pos, rad, world, flags=gpe._lib.FLAG_NATIVE_FORCE)
    st.ctx.set_profiling(True)
    sim = oracle.Sim(pos, rad, oracle.default_params(world[0], world[1], 0.5))
    for s in range(3):
        st.update(1 / 60, resort=(s == 0)); sim.step(1 / 60, resort=(s == 0))
    _assert_positions(st.positions(), sim.pos, "over-dense scene through the spill arena")
    st.ctx.sync()
    assert st.ctx.timings()["native/collide+verlet"][1] == 3
    st.close(); sim.close()


def test_native_mouse_blob_hands_over_without_error(gpe):
    """Mouse attraction packs the cloud into a blob while steps are in flight: windows that overfill before the
    (lagged) density statistic moves the context to the compat kernels spill to the arena.  Never an error, and
    the same bits as a compat-only run."""
    n = 60_000
    world = gpe.scenes.world_for(n)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=9)
    a = gpe.State(pos, rad, world=world, mode=gpe.MODE_NATIVE)
    b = gpe.State(pos, rad, world=world, mode=gpe.MODE_COMPAT)
    for st in (a, b):
        st.particles.mouse_click_callback(True, (world[0] * 0.5, world[1] * 0.5))
    for chunk in range(8):
        a.run(1 / 60, 100, resort_every=240, resort_first=(chunk == 0))
        b.run(1 / 60, 100, resort_every=240, resort_first=(chunk == 0))
        assert np.array_equal(a.positions(), b.positions()), "after %d steps" % ((chunk + 1) * 100)
    a.ctx.sync()
    a.close(); b.close()


@pytest.mark.parametrize("sizes", [(90, 150, 240), (300, 640, 1000), (30, 1100, 257)])
def test_native_crushed_cells_are_exact(gpe, oracle, sizes):
    """Cells of ~100 members (a crushed pile): whole-wave resolution (9..64 members) and its blocked form (65..1024:
    the floor corners of the 100 M gravity-on scene hold 600 by step 2450) in the sub-tile and spill windows, one lane
    beyond that; kept on the native kernels by GPE_FLAG_NATIVE_FORCE -- same bits as the oracle."""
    rng = np.random.default_rng(21)
    world = (40.0, 40.0)
    # a sparse background plus three blobs of particles inside one cell each
    bg = (rng.random((1500, 2), dtype=np.float32) * np.float32(40.0)).astype(np.float32)
    blobs = [np.array(c, np.float32) + rng.random((k, 2), dtype=np.float32) * np.float32(0.9)
             for c, k in zip(((11.1, 11.1), (22.1, 16.6), (30.9, 30.9)), sizes)]
    pos = np.concatenate([bg] + blobs).astype(np.float32)
    rad = np.full(len(pos), 0.5, np.float32)
    st = _native(gpe, pos, rad, world, flags=gpe._lib.FLAG_NATIVE_FORCE)
    sim = oracle.Sim(pos, rad, oracle.default_params(world[0], world[1], 0.5))
    for s in range(3):
        st.update(1 / 60, resort=(s == 0)); sim.step(1 / 60, resort=(s == 0))
        _assert_positions(st.positions(), sim.pos, "crushed cells, step %d" % s)
    st.ctx.sync()
    st.close(); sim.close()


@pytest.mark.parametrize("n", [2400, 3000, 4200])
def test_native_cells_of_nine_to_sixteen_members_are_exact(gpe, oracle, n):
    """A compressed region at 2.2 .. 3.9 particles per cell: with their phantom memberships most collision cells hold
    9..16 members -- the sub-tile windows resolve those by the sixteen lanes of a DPP row, four cells per wave
    (resolve_row), beside lane groups (4..8) and whole-wave cells (17 and more); 16x16 windows at the lowest density,
    8x8 windows above.  Same bits as the oracle."""
    world = (33.0, 33.0)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=77 + n)
    st = _native(gpe, pos, rad, world, flags=gpe._lib.FLAG_NATIVE_FORCE)
    sim = oracle.Sim(pos, rad, oracle.default_params(world[0], world[1], 0.5))
    for s in range(4):
        st.update(1 / 60, resort=(s == 0)); sim.step(1 / 60, resort=(s == 0))
        _assert_positions(st.positions(), sim.pos, "cells of 9..16 members, step %d" % s)
    st.ctx.sync()
    info = st.ctx.pipeline_info()
    assert info["native_steps"] == 4 and info["overflow_tiles"] > 0
    st.close(); sim.close()


def _threads():
    import os
    try:
        return max(1, min(16, len(os.sched_getaffinity(0))))
    except AttributeError:
        return max(1, min(16, os.cpu_count() or 1))


def test_config0_100k_particles_1000_steps_match_oracle(gpe, oracle):
    """BASELINE.json configs[0]: 100 k particles, gravity off, 1000 steps -- the workload the reference's own
    CPU-adapter run would take.  The HIP NATIVE path against the oracle (OpenMP over the loops that are GPU
    threads in the WGSL: same bits as the serial oracle, tests/test_oracle_golden.py), re-sort every 240 steps
    (particle_system.rs:13-14 at 60 Hz), bit-exact at steps 1, 240, 241 (either side of a re-sort), 720 and 1000:
    the run crosses four re-sorts."""
    n = 100_000
    world = gpe.scenes.world_for(n)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=0x5EED)
    st = _native(gpe, pos, rad, world)
    oracle.set_threads(_threads())
    try:
        sim = oracle.Sim(pos, rad, oracle.default_params(world[0], world[1], 0.5))
        check_at = {1, 240, 241, 720, 1000}
        for s in range(1000):
            resort = (s % 240) == 0                                    # step 0 and every 240th step re-sort
            st.update(1 / 60, resort=resort)
            sim.step(1 / 60, resort=resort)
            if s + 1 in check_at:
                _assert_positions(st.positions(), sim.pos, "positions after step %d" % (s + 1))
                _assert_positions(st.previous_positions(), sim.prev, "previous positions after step %d" % (s + 1))
        assert np.array_equal(st.particles.download_particle_ids(), sim.particle_ids)
        st.ctx.sync()
        st.close(); sim.close()
    finally:
        oracle.set_threads(1)


def test_config1_1m_particles_match_oracle_directly(gpe, oracle):
    """BASELINE.json configs[1] size (1 M particles, the reference's 3048 x 1048 scene): NATIVE against the oracle
    itself (not against the compat kernels), 3 steps, the first one re-sorting."""
    n = 1_000_000
    world = gpe.scenes.REF_WORLD
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=0x5EED)
    st = _native(gpe, pos, rad, world)
    oracle.set_threads(_threads())
    try:
        sim = oracle.Sim(pos, rad, oracle.default_params(world[0], world[1], 0.5))
        for s in range(3):
            st.update(1 / 60, resort=(s == 0)); sim.step(1 / 60, resort=(s == 0))
        _assert_positions(st.positions(), sim.pos, "positions (1 M, 3 steps)")
        _assert_positions(st.previous_positions(), sim.prev, "previous positions (1 M, 3 steps)")
        assert np.array_equal(st.particles.download_particle_ids(), sim.particle_ids)
        st.ctx.sync()
        st.close(); sim.close()
    finally:
        oracle.set_threads(1)


def test_gravity_soak_piles_up_without_error(gpe):
    """Gravity on, run long enough for the cloud to fall, pile up and be crushed at the floor (cells of dozens of
    members, windows beyond the LDS capacity: sub-tile windows, whole-wave cells, spill arena, possibly the handover to
    the compat kernels and the way back).  No error at any point, and the same bits as a compat-only run."""
    n = 2_000_000
    world = gpe.scenes.world_for(n)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=0x50A1)
    g = (0.0, -9.81)
    a = gpe.State(pos, rad, world=world, gravity=g, mode=gpe.MODE_NATIVE)
    b = gpe.State(pos, rad, world=world, gravity=g, mode=gpe.MODE_COMPAT)
    done = 0
    for chunk in (600, 600, 600):
        a.run(1 / 60, chunk, resort_every=240, resort_first=(done == 0))
        b.run(1 / 60, chunk, resort_every=240, resort_first=(done == 0))
        done += chunk
        a.ctx.sync()                                   # reports any sticky device error
        assert np.array_equal(a.positions(), b.positions()), "after %d steps" % done
    assert np.isfinite(a.positions()).all()
    a.close(); b.close()


def _fuzz_scene(seed):
    """A random scene from a seed: world shape (down to less than one tile high), density, radii, gravity, time
    step, re-sort pattern, and a share of particles put exactly on cell boundaries, on the walls and on top of
    each other."""
    rng = np.random.default_rng(0xF00D + seed)
    n = int(rng.choice([1, 3, 70, 900, 5000, 12000, 30000]))
    kind = rng.choice(["one", "mixed", "cont"])
    if kind == "one":
        rad = np.full(n, np.float32(rng.choice([0.5, 0.25, 2.0])), np.float32)
    elif kind == "mixed":
        rad = rng.choice(np.array([0.5, 1.0, 2.0, 3.0], np.float32), n).astype(np.float32)
    else:
        rad = (0.3 + 2.2 * rng.random(n, dtype=np.float32)).astype(np.float32)
    max_r = float(np.abs(rad).max())
    cell = float(np.float32(max_r) * np.float32(2.2))
    density = float(rng.choice([0.03, 0.2, 0.6, 1.2]))                # particles per cell
    aspect = float(rng.choice([1.0, 2.9, 17.0, 1 / 9.0, 60.0]))
    cells = max(4.0, n / density)
    h_cells = max(1.3, float(np.sqrt(cells / aspect)))
    w_cells = max(1.3, cells / h_cells)
    world = (float(np.float32(w_cells * cell + rng.random())), float(np.float32(h_cells * cell + rng.random())))
    pos = (rng.random((n, 2), dtype=np.float32) * np.array(world, np.float32)).astype(np.float32)
    k = n // 5
    if k:
        idx = rng.permutation(n)
        a, b, c, d = idx[:k // 4], idx[k // 4:k // 2], idx[k // 2:3 * k // 4], idx[3 * k // 4:k]
        pos[a, 0] = (np.floor(pos[a, 0] / np.float32(cell)) * np.float32(cell)).astype(np.float32)   # on a cell edge
        pos[b, 1] = (np.floor(pos[b, 1] / np.float32(cell)) * np.float32(cell)).astype(np.float32)
        pos[c, 0] = np.where(rng.random(len(c)) < 0.5, rad[c], np.float32(world[0]) - rad[c])        # on a wall
        if len(d) > 1:
            pos[d[1:]] = pos[d[0]] + (rng.random((len(d) - 1, 2), dtype=np.float32) * np.float32(1e-3)
                                      * (rng.random() < 0.5)).astype(np.float32)                    # a clump
    pos[:, 0] = np.clip(pos[:, 0], 0, np.float32(world[0]))
    pos[:, 1] = np.clip(pos[:, 1], 0, np.float32(world[1]))
    gravity = [(0.0, 0.0), (0.0, -9.81), (3.0, 7.5), (-20.0, 0.0)][int(rng.integers(4))]
    dt = float(rng.choice([1 / 60, 1 / 30, 1 / 144]))
    steps = int(rng.integers(5, 11))
    resorts = set(int(x) for x in rng.choice(steps, size=int(rng.integers(1, 3)), replace=False)) | {0}
    mouse = (world[0] * float(rng.random()), world[1] * float(rng.random())) if rng.random() < 0.25 else None
    return pos, rad, world, max_r, gravity, dt, steps, resorts, mouse


@pytest.mark.parametrize("seed", range(24))
def test_native_fuzz_scenes_match_oracle(gpe, oracle, seed):
    """Seeded random scenes (ragged box sizes, thin worlds, boundary placements, clumps, several radii laws, gravity
    in any direction, three time steps, re-sorts at random steps, mouse on in a quarter of them): bit-exact
    positions, previous positions and particle ids after every step.  Even seeds pin the scene to the native
    kernels (GPE_FLAG_NATIVE_FORCE: clumps then go through the sub-tile and spill windows) and check that every step
    ran there; odd seeds leave the choice to the step policy."""
    pos, rad, world, max_r, gravity, dt, steps, resorts, mouse = _fuzz_scene(seed)
    st = _native(gpe, pos, rad, world, gravity=gravity, flags=gpe._lib.FLAG_NATIVE_FORCE if seed % 2 == 0 else 0)
    st.ctx.set_profiling(True)
    p = oracle.default_params(world[0], world[1], max_r, gravity=gravity)
    if mouse is not None:
        p.mouse_pressed, p.mouse_x, p.mouse_y = 1, mouse[0], mouse[1]
        st.particles.mouse_click_callback(True, mouse)
    sim = oracle.Sim(pos, rad, p)
    for s in range(steps):
        st.update(dt, resort=(s in resorts)); sim.step(dt, resort=(s in resorts))
        _assert_positions(st.positions(), sim.pos, "seed %d, step %d (n=%d, world %s)" % (seed, s, len(rad), world))
    _assert_positions(st.previous_positions(), sim.prev, "seed %d previous positions" % seed)
    assert np.array_equal(st.particles.download_particle_ids(), sim.particle_ids)
    st.ctx.sync()
    if seed % 2 == 0:
        assert st.ctx.timings().get("native/collide+verlet", (0, 0))[1] == steps
    st.close(); sim.close()


@pytest.mark.parametrize("world,n", [((30000.0, 30000.0), 200_000),      # 27 k x 27 k cells, 11.6 M blocks: 3 radix passes
                                     ((70000.0, 900.0), 150_000),        # 63 636 cell columns: just inside the 16-bit box
                                     ((72000.0, 900.0), 50_000)])        # 65 455 columns: outside it -> compat kernels
def test_native_sparse_huge_worlds(gpe, oracle, world, n):
    """Worlds whose cell box approaches the 16-bit cell coordinates of the reference's Morton ids (grid.wgsl:101-108):
    sparse clouds plus a dense patch so that pairs do collide; bit-exact against the oracle either side of the
    eligibility limit (native_configure: more than 65000 columns or rows -> the compat kernels)."""
    rng = np.random.default_rng(int(world[0]))
    pos = (rng.random((n, 2), dtype=np.float32) * np.array(world, np.float32)).astype(np.float32)
    k = n // 4                                                            # a patch at the far corner, ~1 per cell
    side = float(np.sqrt(k) * 1.1)
    pos[:k] = (np.array(world, np.float32) - np.float32(side) * rng.random((k, 2), dtype=np.float32)).astype(np.float32)
    pos[:, 1] = np.clip(pos[:, 1], 0, np.float32(world[1]))
    rad = np.full(n, 0.5, np.float32)
    st = _native(gpe, pos, rad, world, gravity=(0.0, -9.81))
    st.ctx.set_profiling(True)
    sim = oracle.Sim(pos, rad, oracle.default_params(world[0], world[1], 0.5, gravity=(0.0, -9.81)))
    for s in range(5):
        st.update(1 / 60, resort=(s in (0, 3))); sim.step(1 / 60, resort=(s in (0, 3)))
    _assert_positions(st.positions(), sim.pos, "huge sparse world %s" % (world,))
    _assert_positions(st.previous_positions(), sim.prev, "previous positions")
    assert np.array_equal(st.particles.download_particle_ids(), sim.particle_ids)
    st.ctx.sync()
    ran_native = st.ctx.timings().get("native/collide+verlet", (0, 0))[1]
    assert ran_native == (5 if world[0] < 71000 else 0)
    info = st.ctx.pipeline_info()
    if world[0] < 71000:
        assert (info["pipeline"], info["reason"]) == (gpe._lib.PIPELINE_NATIVE, gpe._lib.REASON_NONE), info
        blocks = (int(world[0] / 1.1) // 8 + 1) * (int(world[1] / 1.1) // 8 + 1)
        assert info["sort_passes"] == ((blocks - 1).bit_length() + 7) // 8 == 3
    else:
        assert (info["pipeline"], info["reason"]) == (gpe._lib.PIPELINE_COMPAT, gpe._lib.REASON_GRID_TOO_WIDE), info
    st.close(); sim.close()


def test_native_growth_then_module_calls(gpe, oracle):
    """A NATIVE context grows (add_particles beyond its capacity) while the reference's 4N arrays do not exist yet,
    steps on the native kernels, and only then is asked for Grid::update and the collision-cell list: those arrays
    are allocated at the grown capacity and hold what the oracle's hold (used slots)."""
    n = 3000
    world = (260.0, 150.0)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=61)
    extra_pos, extra_rad = gpe.scenes.uniform_cloud(5000, world, seed=62)          # > capacity: the buffers grow
    st = _native(gpe, pos, rad, world)
    st.update(1 / 60, resort=True)
    cur, prev, perm = st.positions(), st.previous_positions(), st.particles.download_particle_ids()
    st.add_particles(extra_pos, extra_rad)
    sim = oracle.Sim(np.concatenate([cur, extra_pos]), np.concatenate([rad[perm], extra_rad]),
                     oracle.default_params(world[0], world[1], 0.5), prev=np.concatenate([prev, extra_pos]))
    for s in range(3):
        st.update(1 / 60, resort=(s == 1)); sim.step(1 / 60, resort=(s == 1))
    _assert_positions(st.positions(), sim.pos, "positions after growth (native)")
    st.grid.update(); st.collision_system.build_collision_cells()
    sim.grid_build(); sim.grid_sort(); sim.build_collision_cells()
    cells = st.grid.download_cell_ids().ravel()
    assert np.array_equal(cells, sim.cell_ids.ravel())
    used = cells != 0xFFFFFFFF
    assert np.array_equal(st.grid.download_object_ids().ravel()[used], sim.object_ids.ravel()[used])
    k = sim.num_collision_cells
    assert st.collision_system.num_collision_cells() == k
    assert np.array_equal(st.collision_system.download_collision_cells().ravel()[:k], sim.collision_cells.ravel()[:k])
    st.close(); sim.close()


def test_relaxed_cloud_long_run_native_equals_compat(gpe):
    """Gravity off, the headline scene run long (6000 steps at 1 M): the cloud relaxes into touching clusters, the
    densest tiles approach (and, later in such runs, exceed) the LDS capacity.  Same bits as the compat kernels at
    every checkpoint, no error."""
    n = 1_000_000
    world = gpe.scenes.world_for(n)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=0x5EED)
    a = gpe.State(pos, rad, world=world, mode=gpe.MODE_NATIVE)
    b = gpe.State(pos, rad, world=world, mode=gpe.MODE_COMPAT)
    done = 0
    for chunk in (2000, 2000, 2000):
        a.run(1 / 60, chunk, resort_every=240, resort_first=(done == 0))
        b.run(1 / 60, chunk, resort_every=240, resort_first=(done == 0))
        done += chunk
        a.ctx.sync()
        assert np.array_equal(a.positions(), b.positions()), "after %d steps" % done
        assert np.array_equal(a.previous_positions(), b.previous_positions()), "after %d steps" % done
    a.close(); b.close()


def test_kept_block_table_equals_sorting_every_step(gpe, oracle):
    """The radix passes run only when a particle has left the reach of the block table they last produced
    (k_native_hash / kDrift*).  Same bits as sorting every step (GPE_FLAG_SORT_EVERY_STEP) and as the oracle, with
    gravity pulling the whole cloud across block boundaries, a Morton re-sort in the middle and particles added on the
    way; and the passes did run on a fraction of the steps only."""
    n = 120_000
    world = gpe.scenes.world_for(n)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=33)
    g = (0.0, -9.81)
    a = gpe.State(pos, rad, world=world, gravity=g, mode=gpe.MODE_NATIVE)
    b = gpe.State(pos, rad, world=world, gravity=g, mode=gpe.MODE_NATIVE, flags=gpe._lib.FLAG_SORT_EVERY_STEP)
    sim = oracle.Sim(pos, rad, oracle.default_params(world[0], world[1], 0.5, gravity=g))
    oracle.set_threads(8)
    try:
        for s in range(90):
            rs = s in (0, 50)
            a.update(1 / 60, resort=rs); b.update(1 / 60, resort=rs); sim.step(1 / 60, resort=rs)
            if s % 10 == 9 or s in (0, 50, 51):
                pa = a.positions()
                assert np.array_equal(pa, b.positions()), "kept table vs sorting every step, step %d" % s
                _assert_positions(pa, sim.pos, "kept table vs oracle, step %d" % s)
    finally:
        oracle.set_threads(1)
    ia, ib = a.ctx.pipeline_info(), b.ctx.pipeline_info()
    assert ia["pipeline"] == gpe._lib.PIPELINE_NATIVE and ia["reason"] == gpe._lib.REASON_NONE
    assert ia["native_steps"] == 90 and ib["native_steps"] == 90
    assert ib["native_sorts"] >= 90                     # (+ the configuration-time sort)
    assert 2 <= ia["native_sorts"] <= 45, ia            # first step, re-sort step, and whenever the fall demands it
    a.close(); b.close(); sim.close()


def test_kept_block_table_survives_drift_in_every_direction(gpe, oracle):
    """Particles leaving their sorted block to the left, right, up and down by different amounts per step (a cloud
    with a velocity field set through the previous positions), gravity off: exact against the oracle at every step,
    while sorts happen only now and then."""
    n = 40_000
    world = gpe.scenes.world_for(n)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=5)
    rng = np.random.default_rng(6)
    vel = (rng.random((n, 2), dtype=np.float32) - np.float32(0.5)) * np.float32(1.6)     # up to 0.8 units per step
    prev = (pos - vel).astype(np.float32)
    st = gpe.State(pos, rad, world=world, mode=gpe.MODE_NATIVE, prev=prev)
    sim = oracle.Sim(pos, rad, oracle.default_params(world[0], world[1], 0.5), prev=prev)
    for s in range(40):
        st.update(1 / 60, resort=(s == 0)); sim.step(1 / 60, resort=(s == 0))
        _assert_positions(st.positions(), sim.pos, "drifting cloud, step %d" % s)
    info = st.ctx.pipeline_info()
    assert info["native_steps"] == 40 and 2 <= info["native_sorts"] < 40, info
    st.close(); sim.close()


def test_pipeline_info_reports_mode_table_and_default(gpe):
    """gpe_get_pipeline_info: a context created with the defaults runs the NATIVE kernels (gpe_config_default); COMPAT
    by request says so; a context without particles has nothing to run.  (GPE_REASON_TABLE_TOO_LARGE cannot be
    reached while GPE_REASON_GRID_TOO_WIDE caps an axis at 65000 cells: 8125 x 8125 blocks < 2^27.)"""
    L = gpe._lib
    pos, rad = gpe.scenes.uniform_cloud(5000, (300.0, 200.0), seed=2)
    st = gpe.State(pos, rad, world=(300.0, 200.0))                     # no mode given
    i = st.ctx.pipeline_info()
    assert (i["pipeline"], i["reason"]) == (L.PIPELINE_NATIVE, L.REASON_NONE) and i["sort_passes"] == 2, i
    st.ctx.call("gpe_set_mode", L.MODE_COMPAT)
    i = st.ctx.pipeline_info()
    assert (i["pipeline"], i["reason"]) == (L.PIPELINE_COMPAT, L.REASON_MODE_COMPAT), i
    st.close()
    ctx = gpe.Context()
    i = ctx.pipeline_info()
    assert (i["pipeline"], i["reason"]) == (L.PIPELINE_COMPAT, L.REASON_NO_PARTICLES), i
    ctx.close()


@pytest.mark.parametrize("flag", ["FLAG_COUNTING_SORT_TILES", "FLAG_XCD_EIGHTHS", "FLAG_FUSED_HISTOGRAMS"])
def test_other_tile_forms_give_the_same_bits(gpe, oracle, flag):
    """The dense launch has two forms of its tile -- 32 x 32 cells with direct cell slots (the default) and the same
    tile with counting-sort member lists (rounds 1-2, GPE_FLAG_COUNTING_SORT_TILES; what sharded order-key windows and
    crowded scenes run): bit-identical to each other and to the oracle, with gravity moving the cloud across tile
    boundaries, stragglers, a re-sort, and a crowded corner whose windows go to the over-capacity launch.  (A third
    form, 64 x 32 cells on 1024 threads, was measured slower in round 3 and left the product in round 4:
    profiles/r04/wide_tiles_64x32_removed.patch.)  FLAG_XCD_EIGHTHS: the tiles dealt to the XCDs as one contiguous
    eighth of the rows each (rounds 1-3) instead of interleaved bands of rows.  FLAG_FUSED_HISTOGRAMS: the radix digits
    counted by the hash kernel every step (rounds 1-3) instead of by a gated launch on the steps that sort."""
    n = 150_000
    world = gpe.scenes.world_for(n)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=17)
    rng = np.random.default_rng(18)
    pos[:4000] = (rng.random((4000, 2), dtype=np.float32) * np.float32(40.0) + np.float32(3.0)).astype(np.float32)   # 2.5 / unit^2
    g = (3.0, -9.81)
    a = gpe.State(pos, rad, world=world, gravity=g)
    b = gpe.State(pos, rad, world=world, gravity=g, flags=getattr(gpe._lib, flag))
    sim = oracle.Sim(pos, rad, oracle.default_params(world[0], world[1], 0.5, gravity=g))
    oracle.set_threads(8)
    try:
        for s in range(40):
            rs = s in (0, 25)
            a.update(1 / 60, resort=rs); b.update(1 / 60, resort=rs); sim.step(1 / 60, resort=rs)
            if s % 8 == 7 or s in (0, 25, 39):
                pa = a.positions()
                assert np.array_equal(pa, b.positions()), "%s, step %d" % (flag, s)
                _assert_positions(pa, sim.pos, "default tiles vs oracle, step %d" % s)
    finally:
        oracle.set_threads(1)
    assert np.array_equal(a.previous_positions(), b.previous_positions())
    a.ctx.sync(); b.ctx.sync()
    a.close(); b.close(); sim.close()


def test_half_tiles_take_what_runs_over_the_direct_slot_form(gpe, oracle):
    """A patch of 3 x 3 tiles at 1.9 x the benchmark density (they keep ~1090 particles: more than the direct-slot form's
    928) and a small blob crowded far beyond that, in an otherwise ordinary cloud -- fewer than 2 % of the tiles, so the
    dense launch stays on direct-slot tiles and hands those few on: the half-tile launch (round 4) resolves the patch as
    32 x 16 direct-slot halves and passes the blob's halves on to the 16 x 16 / 8 x 8 windows.  Bit-identical to a run
    without that launch (GPE_FLAG_NO_HALF_TILES: rounds 1-3, where every such tile went to the windows) and to the oracle;
    gravity moves the cloud across tile and half boundaries."""
    n = 200_000
    world = gpe.scenes.world_for(n)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=23)
    rng = np.random.default_rng(24)
    side = np.float32(3 * 32 * 1.1)
    extra = int(0.9 * 0.3131 * float(side) ** 2)           # + 0.9 x the benchmark density on the patch
    patch = (np.array([400.0, 150.0], np.float32) + rng.random((extra, 2), dtype=np.float32) * side).astype(np.float32)
    blob = (np.array([900.0, 40.0], np.float32) + rng.random((1500, 2), dtype=np.float32) * np.float32(20.0)).astype(np.float32)
    pos = np.concatenate([pos, patch, blob]).astype(np.float32)
    rad = np.full(len(pos), 0.5, np.float32)
    g = (2.0, -9.81)
    a = gpe.State(pos, rad, world=world, gravity=g)
    b = gpe.State(pos, rad, world=world, gravity=g, flags=gpe._lib.FLAG_NO_HALF_TILES)
    sim = oracle.Sim(pos, rad, oracle.default_params(world[0], world[1], 0.5, gravity=g))
    oracle.set_threads(8)
    over = []
    try:
        for s in range(30):
            rs = s in (0, 20)
            a.update(1 / 60, resort=rs); b.update(1 / 60, resort=rs); sim.step(1 / 60, resort=rs)
            a.ctx.sync()
            over.append(a.ctx.pipeline_info()["overflow_tiles"])
            if s % 6 == 5:
                pa = a.positions()
                assert np.array_equal(pa, b.positions()), "step %d" % s
                _assert_positions(pa, sim.pos, "half tiles vs oracle, step %d" % s)
    finally:
        oracle.set_threads(1)
    tiles = (int(world[0] / 1.1) // 32 + 1) * (int(world[1] / 1.1) // 32 + 1)
    assert max(over) >= 6 and max(over) <= tiles // 50 + 4, (over, tiles)    # tiles were handed on, and few enough
    assert np.array_equal(a.previous_positions(), b.previous_positions())
    assert a.ctx.pipeline_info()["compat_steps"] == 0
    a.close(); b.close(); sim.close()


def test_more_hinted_tiles_than_the_front_workgroups_take(gpe):
    """A patch of 7 x 7 tiles at 1.9 x the benchmark density in the 1 M cloud: 36 and more tiles run over the direct-slot
    form, fewer than 2 % of the 2610 -- the first 32 are redone as halves by the dense launch's front workgroups from the
    second step on (they registered themselves: kCtlHints), the others go through the half-tile launch behind it, step
    after step; a re-sort in between drops the hints.  Bit-identical to a run without half tiles and hints."""
    n = 1_000_000
    world = gpe.scenes.world_for(n)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=31)
    rng = np.random.default_rng(32)
    side = np.float32(7 * 32 * 1.1)
    extra = int(0.9 * 0.3131 * float(side) ** 2)
    patch = (np.array([1200.0, 400.0], np.float32) + rng.random((extra, 2), dtype=np.float32) * side).astype(np.float32)
    pos = np.concatenate([pos, patch]).astype(np.float32)
    rad = np.full(len(pos), 0.5, np.float32)
    g = (3.0, -9.81)
    a = gpe.State(pos, rad, world=world, gravity=g)
    b = gpe.State(pos, rad, world=world, gravity=g, flags=gpe._lib.FLAG_NO_HALF_TILES)
    over = []
    for s in range(24):
        rs = s in (0, 13)
        a.update(1 / 60, resort=rs); b.update(1 / 60, resort=rs)
        a.ctx.sync()
        over.append(a.ctx.pipeline_info()["overflow_tiles"])
        if s % 4 == 3:
            assert np.array_equal(a.positions(), b.positions()), "step %d" % s
    assert np.array_equal(a.previous_positions(), b.previous_positions())
    assert max(over) > 32 and a.ctx.pipeline_info()["compat_steps"] == 0, over
    a.close(); b.close()


def test_hinted_tiles_when_a_host_collides_twice_on_one_grid(gpe):
    """Tiles that ran over are registered for the NEXT collide launch (kCtlHints: numbered by launch, not by step).  A
    host that calls the modules one by one may solve collisions twice on one grid build, or build grids without
    colliding: the registered tiles must be taken exactly once by every launch.  NATIVE against COMPAT, same calls."""
    n = 200_000
    world = gpe.scenes.world_for(n)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=51)
    rng = np.random.default_rng(52)
    side = np.float32(3 * 32 * 1.1)
    extra = int(0.9 * 0.3131 * float(side) ** 2)
    patch = (np.array([500.0, 200.0], np.float32) + rng.random((extra, 2), dtype=np.float32) * side).astype(np.float32)
    pos = np.concatenate([pos, patch]).astype(np.float32)
    rad = np.full(len(pos), 0.5, np.float32)
    g = (1.0, -9.81)
    a = gpe.State(pos, rad, world=world, gravity=g, mode=gpe.MODE_NATIVE)
    b = gpe.State(pos, rad, world=world, gravity=g, mode=gpe.MODE_COMPAT)
    both = (a, b)
    for st in both:
        st.run(1 / 60, 6, resort_every=0, resort_first=True)            # the patch's tiles run over and register
    for rep in range(3):
        for st in both:
            st.grid.update(); st.collision_system.solve_collisions(); st.collision_system.solve_collisions()
            st.particles.update_positions(1 / 60)
            st.grid.update(); st.grid.update(); st.collision_system.solve_collisions(); st.particles.update_positions(1 / 60)
            st.update(1 / 60)
        assert np.array_equal(a.positions(), b.positions()), "after round %d of module calls" % rep
    for st in both:
        st.run(1 / 60, 6, resort_every=0, resort_first=False)
    assert np.array_equal(a.positions(), b.positions())
    assert np.array_equal(a.previous_positions(), b.previous_positions())
    info = a.ctx.pipeline_info()
    assert info["compat_steps"] == 0 and info["overflow_tiles"] > 0, info
    a.close(); b.close()


def test_stragglers_flying_into_empty_space_are_not_lost(gpe, oracle):
    """A few very fast particles shot out of a compact cloud into an otherwise empty world: their tiles look nothing up
    (no block of the kept table lies near them), so they exist for those tiles only through the straggler lists.  Exact
    against the oracle at every step, and the cloud itself does not force a sort every step."""
    rng = np.random.default_rng(77)
    world = (700.0, 500.0)
    n = 30_000
    pos = (np.array([60.0, 60.0], np.float32) + rng.random((n, 2), dtype=np.float32) * np.float32(300.0)).astype(np.float32)
    rad = np.full(n, 0.5, np.float32)
    prev = pos.copy()
    fast = rng.choice(n, 40, replace=False)
    ang = rng.random(40) * 2 * np.pi
    speed = 6.0 + 9.0 * rng.random(40)                                   # 5-14 cells per step, in every direction
    prev[fast] = (pos[fast] - np.stack([np.cos(ang) * speed, np.sin(ang) * speed], 1)).astype(np.float32)
    st = gpe.State(pos, rad, world=world, prev=prev)
    sim = oracle.Sim(pos, rad, oracle.default_params(world[0], world[1], 0.5), prev=prev)
    for s in range(45):
        st.update(1 / 60, resort=(s == 0)); sim.step(1 / 60, resort=(s == 0))
        _assert_positions(st.positions(), sim.pos, "fast particles in empty space, step %d" % s)
    info = st.ctx.pipeline_info()
    assert info["native_steps"] == 45 and info["native_sorts"] < 30, info
    st.close(); sim.close()


def test_roster_stamp_follows_every_sort(gpe):
    """The tile rosters (ids written down by the tiles of a sort step) are valid only under the sort count they were
    stamped with; the tiles compare against their own copy of that count, which every sort must bump -- whoever ran it:
    a step that needed it, a step that sorts unconditionally (no roster is written then), the configuration's own sort.
    (A copy that lags would let a step use the ids of an older grouping.)"""
    n = 60_000
    world = gpe.scenes.world_for(n)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=21)
    for flags in (0, gpe._lib.FLAG_SORT_EVERY_STEP):
        st = gpe.State(pos, rad, world=world, gravity=(0.0, -9.81), mode=gpe.MODE_NATIVE, flags=flags)
        info = st.ctx.pipeline_info()
        assert info["roster_stamp"] == info["native_sorts"] >= 1           # the configuration sorted once already
        for chunk in range(4):
            st.run(1 / 60, 30, resort_every=50, resort_first=(chunk == 0))
            info = st.ctx.pipeline_info()
            assert info["roster_stamp"] == info["native_sorts"] & 0xFFFFFFFF, (flags, chunk, info)
        if flags:
            assert info["native_sorts"] >= 120
        st.close()


def test_table_is_kept_again_after_a_spell_of_sorting_every_step(gpe):
    """A cloud that flies 4.5 cells per step makes every step sort; after 128 such steps the library stops trying to keep
    the block table (it sorts unconditionally for 256 steps: no old keys, no drift test, no rosters written).  The test
    then stops the particles (prev = pos, written through the device pointer: the positions do not change).  When the
    spell ends the table must be kept again -- few sorts from there on -- and the tiles must not start from rosters
    written before the spell: the particles have moved hundreds of cells since.  Bit-identical to sorting every step."""
    n = 60_000
    world = (2600.0, 260.0)
    rng = np.random.default_rng(77)
    pos = np.empty((n, 2), np.float32)
    pos[:, 0] = rng.random(n, dtype=np.float32) * np.float32(700.0) + np.float32(1.0)
    pos[:, 1] = rng.random(n, dtype=np.float32) * np.float32(258.0) + np.float32(1.0)
    rad = np.full(n, 0.5, np.float32)
    prev = pos.copy()
    prev[:, 0] -= np.float32(5.0)                                         # 5 units = 4.5 cells per step to the right

    def run(flags):
        st = gpe.State(pos, rad, world=world, gravity=(0.0, 0.0), mode=gpe.MODE_NATIVE, prev=prev, flags=flags)
        marks = []
        st.run(1 / 60, 140, resort_every=0, resort_first=True)
        marks.append(st.ctx.pipeline_info()["native_sorts"])
        # stop: prev = pos on the device (gpe_device_ptr + gpe_buffer_upload; positions untouched)
        now = np.ascontiguousarray(st.positions())
        ptr, nbytes = st.ctx.device_ptr(gpe._lib.PREV)
        assert nbytes == now.nbytes
        st.ctx.call("gpe_buffer_upload", ptr, now.ctypes.data_as(__import__("ctypes").c_void_p), now.nbytes)
        st.run(1 / 60, 460, resort_every=0, resort_first=False)          # the rest of the spell, and the count afresh
        marks.append(st.ctx.pipeline_info()["native_sorts"])
        st.run(1 / 60, 200, resort_every=0, resort_first=False)
        info = st.ctx.pipeline_info()
        marks.append(info["native_sorts"])
        out = (st.positions().copy(), st.previous_positions().copy(), marks, info)
        st.close()
        return out

    p_keep, q_keep, marks, info = run(0)
    p_sort, q_sort, marks_sort, _ = run(gpe._lib.FLAG_SORT_EVERY_STEP)
    assert info["native_steps"] == 800 and info["compat_steps"] == 0
    assert marks[0] >= 130                                                # the flight: (nearly) every step sorted
    assert marks[2] - marks[1] <= 40, marks                               # the last 200 steps: the table is kept again
    assert marks_sort[2] - marks_sort[1] == 200
    assert info["roster_stamp"] == info["native_sorts"]
    assert np.array_equal(p_keep, p_sort) and np.array_equal(q_keep, q_sort)


@pytest.mark.parametrize("seed", list(range(1, 13)))
def test_random_api_sequences_native_equals_compat(gpe, seed):
    """The same random sequence of boundary calls -- fused steps with and without a re-sort, gpe_run, the module calls one
    by one (Grid::update / solve_collisions / update_positions), Morton re-sorts, mouse and gravity changes, new
    particles (same and larger radii: a new cell size), a device pointer handed out, downloads -- on a NATIVE and on a
    COMPAT context.  The native pipeline carries state from step to step (sorted ids, block table, straggler lists,
    tile rosters, lagged policies); whatever a host does in between, the two contexts must hold the same bits."""
    import ctypes
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(6_000, 14_000))
    world = (float(rng.integers(160, 300)), float(rng.integers(90, 170)))
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=seed)
    if seed % 2 == 0:
        # a region at four times the density: its tiles hold more than the direct-slot form stages (over-capacity
        # launch, rosters that say "too many"), and it spreads over the neighbouring tiles as the run goes on
        m = n // 2
        corner = np.array([rng.random() * (world[0] - 60.0) + 5.0, rng.random() * (world[1] - 50.0) + 5.0], np.float32)
        blob = (rng.random((m, 2), dtype=np.float32) * np.array([50.0, 40.0], np.float32) + corner).astype(np.float32)
        pos = np.concatenate([pos, blob]); rad = np.concatenate([rad, np.full(m, 0.5, np.float32)])
    a = gpe.State(pos, rad, world=world, mode=gpe.MODE_NATIVE)
    b = gpe.State(pos, rad, world=world, mode=gpe.MODE_COMPAT)
    both = (a, b)
    log = []
    for op_index in range(60):
        op = int(rng.integers(0, 12))
        if op <= 2:
            k = int(rng.integers(1, 6)); resort = bool(rng.integers(0, 4) == 0)
            for st in both:
                for s in range(k):
                    st.update(1 / 60, resort=(resort and s == 0))
            log.append("step x%d%s" % (k, " resort" if resort else ""))
        elif op == 3:
            k = int(rng.integers(5, 40)); every = int(rng.choice([0, 7, 16]))
            for st in both:
                st.run(1 / 60, k, resort_every=every, resort_first=False)
            log.append("run %d every %d" % (k, every))
        elif op == 4:
            for st in both:
                st.grid.update(); st.collision_system.solve_collisions(); st.particles.update_positions(1 / 60)
            log.append("module calls")
        elif op == 5:
            for st in both:
                st.particles.sort_by_cell_id()
            log.append("morton resort")
        elif op == 6:
            pressed = bool(rng.integers(0, 2)); at = (float(rng.random() * world[0]), float(rng.random() * world[1]))
            for st in both:
                st.particles.mouse_click_callback(pressed, at)
            log.append("mouse %s" % pressed)
        elif op == 7:
            g = (float(rng.choice([0.0, 3.0, -3.0])), float(rng.choice([0.0, -9.81, 9.81])))
            for st in both:
                st.ctx.call("gpe_set_gravity", g[0], g[1])
            log.append("gravity %s" % (g,))
        elif op == 8:
            m = int(rng.integers(1, 300))
            big = rng.integers(0, 5) == 0
            p_new, r_new = gpe.scenes.mixed_radius_cloud(m, world, seed=int(rng.integers(1 << 30)),
                                                         radii=(0.5, 1.0, 1.5) if big else (0.5,))
            # (inside the box whatever the radius: the integration clamps to [r, world - r])
            p_new = np.clip(p_new, 2.0, np.array(world, np.float32) - 2.0).astype(np.float32)
            for st in both:
                st.add_particles(p_new, r_new)
            log.append("add %d%s" % (m, " mixed radii" if big else ""))
        elif op == 9:
            for st in both:
                ptr, nbytes = st.ctx.device_ptr(gpe._lib.POS)
                assert nbytes == 8 * st.particles.len()
            log.append("device ptr")
        elif op == 10:
            # a host that writes through the pointer it was given: stop every particle (prev = pos)
            for st in both:
                now = np.ascontiguousarray(st.positions())
                ptr, _ = st.ctx.device_ptr(gpe._lib.PREV)
                st.ctx.call("gpe_buffer_upload", ptr, now.ctypes.data_as(ctypes.c_void_p), now.nbytes)
            log.append("prev = pos")
        elif op == 11 and op_index % 3 == 0:
            # ... and one that moves particles: a tenth of them teleported to random places inside the box
            cnt = a.particles.len()
            who = rng.choice(cnt, size=max(1, cnt // 10), replace=False)
            where = (rng.random((len(who), 2), dtype=np.float32) * (np.array(world, np.float32) - 4.0) + 2.0).astype(np.float32)
            for st in both:
                now = np.ascontiguousarray(st.positions())
                now[who] = where
                for what in (gpe._lib.POS, gpe._lib.PREV):
                    ptr, _ = st.ctx.device_ptr(what)
                    st.ctx.call("gpe_buffer_upload", ptr, now.ctypes.data_as(ctypes.c_void_p), now.nbytes)
            log.append("teleport %d" % len(who))
        else:
            pa, pb = a.positions(), b.positions()
            assert np.array_equal(pa, pb), (seed, op_index, log)
            log.append("download")
    assert np.array_equal(a.positions(), b.positions()), (seed, log)
    assert np.array_equal(a.previous_positions(), b.previous_positions()), (seed, log)
    assert np.array_equal(a.radii(), b.radii())
    info = a.ctx.pipeline_info()
    assert info["roster_stamp"] == info["native_sorts"] & 0xFFFFFFFF
    assert info["native_steps"] > 0, log
    a.close(); b.close()
