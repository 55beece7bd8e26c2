"""ms/step of the native step as the scene evolves: python scripts/time_evolution.py N chunks steps_per_chunk [on]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
gpe = importlib.import_module("gpu-physics-engine_amd")
n = int(sys.argv[1]); chunks = int(sys.argv[2]); per = int(sys.argv[3])
grav = (0.0, -9.81) if (len(sys.argv) > 4 and sys.argv[4] == "on") else (0.0, 0.0)
world = gpe.scenes.world_for(n)
pos, rad = gpe.scenes.uniform_cloud(n, world, seed=0x5EED)
st = gpe.State(pos, rad, world=world, gravity=grav, mode=gpe.MODE_NATIVE)
st.run(1 / 60, 10, resort_every=240, resort_first=True)
st.ctx.sync()
done = 10
for c in range(chunks):
    t0 = time.perf_counter()
    st.run(1 / 60, per, resort_every=240, resort_first=False)
    st.ctx.sync()
    wall = (time.perf_counter() - t0) / per * 1e3
    done += per
    p = st.positions(); q = st.previous_positions()
    v = np.sqrt(((p - q) ** 2).sum(1))
    print("steps %6d  %.4f ms/step   mean |v| %.4f  max |v| %.3f  y-mean %.1f" % (done, wall, v.mean(), v.max(), p[:, 1].mean()), flush=True)
st.ctx.set_profiling(True); st.ctx.reset_timings()
st.run(1 / 60, 100, resort_every=240, resort_first=False)
st.ctx.sync()
tim = st.ctx.timings()
print("relaxed state: " + "  ".join("%s %.1fus" % (k, v[0] / max(1, v[1]) * 1e3) for k, v in sorted(tim.items(), key=lambda kv: -kv[1][0])), flush=True)
