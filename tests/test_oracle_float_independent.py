"""CPU: an independent second derivation of the float results the reference holds no vector for (K11 collision
response, K12 Verlet), compared bit for bit with the oracle.

The oracle (oracle/gpe_oracle.c) is a C restatement; this file re-derives the same step for a handful of hand-built
scenes in numpy float32 SCALAR arithmetic, written directly from the WGSL text -- membership from grid.wgsl:39-129,
the pair loop from collision_solver.wgsl:66-118 (response :91-111), colours from collision_solver.rs:224 /
.wgsl:55-58, integration from particle_integration.wgsl:25-77 -- with none of the oracle's code or data layout
(dictionaries of Python lists instead of sorted 4N arrays).  numpy float32 scalars round every operation to IEEE
binary32 (no FMA, correctly rounded sqrt and divide), which is the arithmetic the oracle fixes.  Scenes: equal and
unequal radii, a pair closer than 1e-4 (the `distance > 0.0001` guard), three members in one cell (order of the
pairs), particles sharing several cells (the pair is resolved once per shared collision cell, in colour order), both
wall clamps, gravity and mouse attraction.  This pins the oracle's float path to a second, independently written
computation; it is still "oracle-defined" with respect to a real wgpu run (driver-rounded sqrt and divide there).
"""
import numpy as np
import pytest

F = np.float32
STIFFNESS = F(0.6)                     # collision_solver.wgsl:2
MOUSE_STRENGTH = F(150.0)              # particle_integration.wgsl:22


def _split(n):
    x = n & 0x0000FFFF
    x = (x | (x << 8)) & 0x00FF00FF
    x = (x | (x << 4)) & 0x0F0F0F0F
    x = (x | (x << 2)) & 0x33333333
    x = (x | (x << 1)) & 0x55555555
    return x


def _morton(cx, cy):                   # grid.wgsl:112-114, u32(i32) is a bit cast
    return (_split(cx & 0xFFFFFFFF) | (_split(cy & 0xFFFFFFFF) << 1)) & 0xFFFFFFFF


def _in_cell(px, py, sq_r, cx, cy, cs):        # grid.wgsl:117-129
    lox, loy = F(cx) * cs, F(cy) * cs
    hix, hiy = lox + cs, loy + cs
    qx = min(max(px, lox), hix)
    qy = min(max(py, loy), hiy)
    dx, dy = px - qx, py - qy
    return (dx * dx + dy * dy) < sq_r


def numpy_step(pos, prev, rad, world, cs, dt, gravity=(0.0, 0.0), mouse=None):
    """One State::update without re-sort, in numpy float32 scalars."""
    n = len(rad)
    pos = [[F(p[0]), F(p[1])] for p in pos]
    prev = [[F(p[0]), F(p[1])] for p in prev]
    rad = [F(r) for r in rad]
    cs = F(cs)
    # membership, frozen for the step (grid.wgsl:39-97): home first, then neighbours y-major; key -> members in
    # ascending object index (the stable sort of (cell, object) pairs leaves them so)
    cells = {}
    for i in range(n):
        px, py = pos[i]
        hx, hy = int(np.floor(px / cs)), int(np.floor(py / cs))
        keys = [_morton(hx, hy)]
        for y in (-1, 0, 1):
            for x in (-1, 0, 1):
                if x == 0 and y == 0:
                    continue
                if _in_cell(px, py, rad[i] * rad[i], hx + x, hy + y, cs):
                    keys.append(_morton(hx + x, hy + y))
        assert len(keys) <= 4
        for k in keys:
            cells.setdefault(k, []).append(i)
    runs = sorted((k, m) for k, m in cells.items() if len(m) >= 2 and k != 0xFFFFFFFF)

    def unsplit(v):
        x = v & 0x55555555
        x = (x | (x >> 1)) & 0x33333333
        x = (x | (x >> 2)) & 0x0F0F0F0F
        x = (x | (x >> 4)) & 0x00FF00FF
        x = (x | (x >> 8)) & 0x0000FFFF
        return x

    for colour in (1, 2, 3, 4):                                        # collision_solver.rs:224
        for key, members in runs:
            if 1 + (unsplit(key) % 2) + (unsplit(key >> 1) % 2) * 2 != colour:      # .wgsl:55-58
                continue
            for a in range(len(members)):                              # .wgsl:68-118
                for b in range(a + 1, len(members)):
                    i, j = members[a], members[b]
                    vx, vy = pos[i][0] - pos[j][0], pos[i][1] - pos[j][1]           # :91, live positions
                    d = np.sqrt(vx * vx + vy * vy)                                  # :93 length()
                    rs = rad[i] + rad[j]
                    if rs * rs > d * d and d > F(0.0001):                           # :95
                        depth = rs - d                                              # :97
                        cx, cy = ((vx / d) * depth) * STIFFNESS, ((vy / d) * depth) * STIFFNESS   # :98-101
                        inv1, inv2 = F(1.0) / rad[i], F(1.0) / rad[j]                # :103-104
                        w1, w2 = inv1 / (inv1 + inv2), inv2 / (inv1 + inv2)          # :107-108
                        pos[i][0] = pos[i][0] + cx * w1; pos[i][1] = pos[i][1] + cy * w1   # :110
                        pos[j][0] = pos[j][0] - cx * w2; pos[j][1] = pos[j][1] - cy * w2   # :111
    gx, gy = F(gravity[0]), F(gravity[1])
    dt2 = F(dt) * F(dt)
    W, H = F(world[0]), F(world[1])
    for i in range(n):                                                 # particle_integration.wgsl:25-77
        cx, cy = pos[i]
        vx, vy = cx - prev[i][0], cy - prev[i][1]
        ax, ay = gx, gy
        if mouse is not None:
            dx, dy = F(mouse[0]) - cx, F(mouse[1]) - cy
            ln = np.sqrt(dx * dx + dy * dy)
            ax, ay = ax + (dx / ln) * MOUSE_STRENGTH, ay + (dy / ln) * MOUSE_STRENGTH
        nx, ny = (cx + vx) + ax * dt2, (cy + vy) + ay * dt2
        prev[i] = [cx, cy]
        r = rad[i]
        pos[i] = [min(max(nx, r), W - r), min(max(ny, r), H - r)]
    return np.array(pos, np.float32), np.array(prev, np.float32)


SCENES = {
    # name: (positions, prev or None, radii, world, gravity, mouse)
    "equal radii, one shared cell": ([[5.2, 5.3], [5.9, 5.6]], None, [0.5, 0.5], (40.0, 30.0), (0, 0), None),
    "unequal radii (inverse-radius weights)": ([[20.0, 20.0], [23.0, 21.0], [21.5, 24.0]], None, [1.0, 3.0, 2.0],
                                               (60.0, 50.0), (0, 0), None),
    "closer than 1e-4: no response": ([[7.0, 7.0], [7.00005, 7.0], [7.4, 7.3]], None, [0.5, 0.5, 0.5], (40.0, 30.0), (0, 0), None),
    "three members, pair order": ([[8.9, 8.9], [9.3, 9.2], [9.1, 9.6], [9.7, 9.0]], None, [0.5] * 4, (40.0, 30.0), (0, 0), None),
    "several shared cells, colour order": ([[11.05, 11.02], [10.95, 10.98], [11.3, 10.9], [10.7, 11.2]], None, [0.5] * 4,
                                           (40.0, 30.0), (0, 0), None),
    "both wall clamps + gravity": ([[0.2, 0.3], [0.8, 0.4], [39.9, 29.8], [39.4, 29.9]],
                                   [[0.6, 0.9], [0.8, 0.4], [39.2, 29.1], [39.4, 29.9]], [0.5] * 4, (40.0, 30.0), (0.0, -9.81), None),
    "mouse attraction": ([[15.0, 15.0], [15.6, 15.2], [22.0, 9.0]], None, [0.5, 0.5, 0.5], (40.0, 30.0), (0.0, -9.81), (18.0, 12.0)),
}


@pytest.mark.parametrize("name", sorted(SCENES))
def test_numpy_f32_derivation_matches_the_oracle_bit_for_bit(oracle, name):
    pos, prev, rad, world, gravity, mouse = SCENES[name]
    pos = np.array(pos, np.float32)
    prev = pos.copy() if prev is None else np.array(prev, np.float32)
    rad = np.array(rad, np.float32)
    max_r = float(np.abs(rad).max())
    cs = np.float32(max_r) * np.float32(2.2)                            # grid.rs:159-161
    p = oracle.default_params(world[0], world[1], max_r, gravity=gravity)
    if mouse is not None:
        p.mouse_pressed, p.mouse_x, p.mouse_y = 1, mouse[0], mouse[1]
    sim = oracle.Sim(pos, rad, p, prev=prev)
    cur_np, prev_np = pos, prev
    moved = False
    for _ in range(4):
        sim.step(1.0 / 60.0, resort=False)
        cur_np, prev_np = numpy_step(cur_np, prev_np, rad, world, cs, 1.0 / 60.0, gravity, mouse)
        assert np.array_equal(sim.pos.view(np.uint32), cur_np.view(np.uint32)), (name, sim.pos, cur_np)
        assert np.array_equal(sim.prev.view(np.uint32), prev_np.view(np.uint32)), name
        moved = moved or not np.array_equal(cur_np, prev_np)
    assert moved
    sim.close()


def test_the_scenes_exercise_what_they_claim(oracle):
    """The guard scene really has a pair under 1e-4 that overlaps; the unequal-radii scene really collides."""
    pos, _, rad, world, _, _ = SCENES["closer than 1e-4: no response"]
    d = np.hypot(pos[0][0] - pos[1][0], pos[0][1] - pos[1][1])
    assert d < 1e-4 and d > 0
    two = np.array(pos[:2], np.float32)                                 # the close pair alone: overlapping, yet untouched
    after, _ = numpy_step(two, two, rad[:2], world, np.float32(1.1), 1 / 60)
    assert np.array_equal(after, two)
    pos, _, rad, world, _, _ = SCENES["unequal radii (inverse-radius weights)"]
    a0 = np.array(pos, np.float32)
    a1, _ = numpy_step(a0, a0, rad, world, np.float32(3.0) * np.float32(2.2), 1 / 60)
    assert not np.array_equal(a0, a1)
    # the lighter (smaller) particle moves further: weights are 1/r
    assert np.hypot(*(a1[0] - a0[0])) > np.hypot(*(a1[1] - a0[1]))
