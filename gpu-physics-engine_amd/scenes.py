"""Seeded synthetic scenes for tests and bench.

The reference fills its scene from an unseeded thread RNG (particle_system.rs:106-127:
pos ~ U[0,W) x U[0,H), prev = pos, radius 0.5), so its runs are not reproducible; these
generators keep the same distribution with a fixed numpy PCG64 seed.
"""
import numpy as np

REF_WORLD = (3048.0, 1048.0)          # state.rs:35
REF_PARTICLES = 1_000_000             # particle_system.rs:28
REF_RADIUS = 0.5                      # particle_system.rs:117
REF_DENSITY = REF_PARTICLES / (REF_WORLD[0] * REF_WORLD[1])   # 0.3131 particles / unit^2


def world_for(n, aspect=REF_WORLD[0] / REF_WORLD[1], density=REF_DENSITY):
    """World size holding n particles at the reference scene's density and aspect ratio.
    1M -> 3048 x 1048 exactly; 100M -> 30480 x 10480."""
    if n == REF_PARTICLES:
        return REF_WORLD
    area = n / density
    h = float(np.sqrt(area / aspect))
    w = float(aspect * h)
    return (float(np.float32(round(w, 1))), float(np.float32(round(h, 1))))


def uniform_cloud(n, world, seed=0x5EED, radius=REF_RADIUS, chunk=1 << 24):
    """pos ~ U[0,W) x U[0,H) as f32 (n,2); radius (n,) constant."""
    rng = np.random.default_rng(seed)
    pos = np.empty((n, 2), np.float32)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        u = rng.random((hi - lo, 2), dtype=np.float32)
        pos[lo:hi, 0] = u[:, 0] * np.float32(world[0])
        pos[lo:hi, 1] = u[:, 1] * np.float32(world[1])
    rad = np.full(n, radius, np.float32)
    return pos, rad


def mixed_radius_cloud(n, world, seed=7, radii=(0.5, 1.0, 2.0, 3.0)):
    """Heterogeneous radii as after State::add_particles (particle_system.rs:189: 1..=3)."""
    rng = np.random.default_rng(seed)
    pos = (rng.random((n, 2), dtype=np.float32) * np.array(world, np.float32)).astype(np.float32)
    rad = rng.choice(np.array(radii, np.float32), n).astype(np.float32)
    return pos, rad
