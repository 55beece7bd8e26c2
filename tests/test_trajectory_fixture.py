"""Committed trajectory fixture (tests/golden/oracle_trajectory.json, written by make_trajectory_fixture.py).
The float results of the collision response and of the integration are not pinned by any reference test
(SURVEY.md 8c): "parity unpinned", oracle-defined.  The fixture freezes them: the CPU test checks that the oracle
still produces these bits, the GPU tests that both HIP pipelines do."""
import importlib.util
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _gen():
    spec = importlib.util.spec_from_file_location("make_trajectory_fixture",
                                                  os.path.join(HERE, "golden", "make_trajectory_fixture.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def fixture():
    with open(os.path.join(HERE, "golden", "oracle_trajectory.json")) as f:
        return json.load(f)


def _check(gen, fixture, pos, prev):
    got, want = gen.digest(pos, prev), fixture["expected"]
    assert got["sample_pos_hex"] == want["sample_pos_hex"]
    assert got["sha256_pos"] == want["sha256_pos"]
    assert got["sha256_prev"] == want["sha256_prev"]


def test_oracle_reproduces_the_fixture(oracle, fixture):
    gen = _gen()
    pos, prev, _ = gen.run_oracle(fixture["spec"])
    _check(gen, fixture, pos, prev)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["compat", "native"])
def test_hip_pipelines_reproduce_the_fixture(gpe, fixture, mode):
    gen = _gen()
    spec = fixture["spec"]
    pos, rad = gen.scene(spec)
    st = gpe.State(pos, rad, world=tuple(spec["world"]), gravity=tuple(spec["gravity"]),
                   mode=gpe.MODE_NATIVE if mode == "native" else gpe.MODE_COMPAT)
    st.particles.mouse_click_callback(True, (spec["mouse"]["x"], spec["mouse"]["y"]))
    for s in range(spec["steps"]):
        st.update(spec["dt"], resort=(s in spec["resort_at"]))
    _check(gen, fixture, st.positions(), st.previous_positions())
    st.close()


@pytest.mark.gpu
def test_snapshot_restore_continues_bit_exact(gpe, tmp_path):
    """State.save / State.load (checkpoint of pos, prev, radius + world, gravity): the restored run and the
    uninterrupted run produce the same bits."""
    n = 20_000
    world = gpe.scenes.world_for(n)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=77)
    a = gpe.State(pos, rad, world=world, gravity=(0.0, -9.81), mode=gpe.MODE_NATIVE)
    for s in range(6):
        a.update(1 / 60, resort=(s == 0))
    path = str(tmp_path / "snap.npz")
    a.save(path)
    b = gpe.State.load(path, mode=gpe.MODE_NATIVE)
    for s in range(6):
        a.update(1 / 60, resort=(s == 3)); b.update(1 / 60, resort=(s == 3))
    assert np.array_equal(a.positions(), b.positions())
    assert np.array_equal(a.previous_positions(), b.previous_positions())
    a.close(); b.close()


@pytest.mark.gpu
def test_chrome_trace_has_the_reference_scope_names(gpe, tmp_path):
    """gpe_get_trace / Context.write_chrome_trace: one complete event per recorded scope, named like the
    reference's profiler scopes (grid.rs:324,329; collision_cell_builder.rs:216,227,233; collision_solver.rs:226;
    particle_integration.rs:81), in stream order."""
    n = 5000
    world = gpe.scenes.world_for(n)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=3)
    st = gpe.State(pos, rad, world=world, mode=gpe.MODE_COMPAT, profiling=True)
    st.ctx.reset_timings()
    for s in range(3):
        st.update(1 / 60, resort=(s == 0))
    path = str(tmp_path / "benchmark.json")
    count = st.ctx.write_chrome_trace(path)
    with open(path) as f:
        ev = json.load(f)["traceEvents"]
    assert count == len(ev) and count > 0
    names = {e["name"] for e in ev}
    for want in ("Build cell ids", "Sort map", "Collision cell prefix sum", "Solve Collisions - Color 1",
                 "Solve Collisions - Color 4", "Particle integration pass", "Particle sort"):
        assert want in names, (want, sorted(names))
    assert all(e["ph"] == "X" and e["dur"] >= 0 for e in ev)
    top = [e for e in ev if e["name"] == "Particle integration pass"]
    assert len(top) == 3 and top[0]["ts"] < top[1]["ts"] < top[2]["ts"]
    tot = st.ctx.timings()
    assert abs(sum(e["dur"] for e in top) / 1e3 - tot["Particle integration pass"][0]) < 1e-3
    st.close()
