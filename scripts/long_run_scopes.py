"""Scopes of the 1 M gravity-off step late in a long run (the undamped cloud has clumped: a few tiles run over the
direct-slot form): python scripts/long_run_scopes.py [N] [steps before] [flags]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gpe = importlib.import_module("gpu-physics-engine_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
before = int(sys.argv[2]) if len(sys.argv) > 2 else 1680
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
world = gpe.scenes.world_for(n)
pos, rad = gpe.scenes.uniform_cloud(n, world, seed=0x5EED + 1)
st = gpe.State(pos, rad, world=world, mode=gpe.MODE_NATIVE, flags=flags)
done = 0
while done < before:
    st.run(1 / 60, 240, resort_every=240, resort_first=True); done += 240
st.ctx.sync()
for rep in range(2):
    t0 = time.perf_counter(); st.run(1 / 60, 200, resort_every=0, resort_first=False); st.ctx.sync()
    wall = (time.perf_counter() - t0) / 200 * 1e3
    st.ctx.set_profiling(True); st.ctx.reset_timings()
    st.run(1 / 60, 100, resort_every=0, resort_first=False); st.ctx.sync()
    tim = st.ctx.timings(); st.ctx.set_profiling(False)
    pi = st.ctx.pipeline_info()
    print("n=%d after %d steps: wall %.4f ms/step, over-capacity tiles %d (sub-tiles %d) | " % (n, done, wall, pi["overflow_tiles"], pi["overflow_subtiles"]) +
          "  ".join("%s %.1fus" % (k, v[0] / max(1, v[1]) * 1e3) for k, v in sorted(tim.items(), key=lambda kv: -kv[1][0])[:5]), flush=True)
    done += 300
