# quick parity of the wide tiles against the default tiles (bit-identical), several scenes
import importlib, sys, numpy as np
sys.path.insert(0, '.')
gpe = importlib.import_module("gpu-physics-engine_amd")
L = gpe._lib
for n, grav, seed, dens in ((200_000, (0.0, 0.0), 3, None), (300_000, (0.0, -9.81), 4, None), (60_000, (5.0, -30.0), 5, 0.8), (1_000_000, (0.0, 0.0), 6, None)):
    world = gpe.scenes.world_for(n) if dens is None else gpe.scenes.world_for(n, density=dens)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=seed)
    a = gpe.State(pos, rad, world=world, gravity=grav)
    b = gpe.State(pos, rad, world=world, gravity=grav, flags=L.FLAG_WIDE_TILES)
    for k in range(6):
        a.run(1/60, 20, resort_every=0, resort_first=(k == 0)); b.run(1/60, 20, resort_every=0, resort_first=(k == 0))
        ok = np.array_equal(a.positions(), b.positions()) and np.array_equal(a.previous_positions(), b.previous_positions())
        print("n=%d grav=%s after %d steps: %s" % (n, grav, 20*(k+1), "identical" if ok else "DIFFERENT"), flush=True)
        assert ok
    a.ctx.sync(); b.ctx.sync()
    print(a.ctx.pipeline_info(), b.ctx.pipeline_info())
    a.close(); b.close()
print("wide tiles ok")
