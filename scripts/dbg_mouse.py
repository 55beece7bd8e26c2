"""Mouse attraction pulls the cloud into a dense blob: the native path must hand over to the compat kernels before
its LDS windows overfill (lagged window statistic), never raise, and stay bit-identical to the compat pipeline."""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
gpe = importlib.import_module("gpu-physics-engine_amd")
n = 120_000
world = gpe.scenes.world_for(n)
pos, rad = gpe.scenes.uniform_cloud(n, world, seed=9)
a = gpe.State(pos, rad, world=world, mode=gpe.MODE_NATIVE, profiling=True)
b = gpe.State(pos, rad, world=world, mode=gpe.MODE_COMPAT)
for st in (a, b):
    st.particles.mouse_click_callback(True, (world[0] * 0.5, world[1] * 0.5))
for chunk in range(16):
    a.run(1 / 60, 100, resort_every=240, resort_first=(chunk == 0))
    b.run(1 / 60, 100, resort_every=240, resort_first=(chunk == 0))
    pa, pb = a.positions(), b.positions()
    a.ctx.sync()
    tim = a.ctx.timings()
    d = np.hypot(pa[:, 0] - world[0] / 2, pa[:, 1] - world[1] / 2)
    print("steps %5d  equal %s  native steps %d  compat-kernel steps %d  median dist to mouse %.1f  within 30: %d" %
          ((chunk + 1) * 100, np.array_equal(pa, pb), tim.get("native/collide", (0, 0))[1], tim.get("Sort map", (0, 0))[1],
           np.median(d), int((d < 30).sum())), flush=True)
