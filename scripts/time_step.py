"""Per-scope GPU time of the native step for one particle count: python scripts/time_step.py N [steps] [gravity] [flags]
(TIME_STEP_SKIP=k in the environment: advance k steps first)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gpe = importlib.import_module("gpu-physics-engine_amd")
n = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
grav = (0.0, -9.81) if (len(sys.argv) > 3 and sys.argv[3] == "on") else (0.0, 0.0)
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # gpe_config.flags, e.g. 2 = GPE_FLAG_SORT_EVERY_STEP
world = gpe.scenes.world_for(n)
pos, rad = gpe.scenes.uniform_cloud(n, world, seed=0x5EED)
st = gpe.State(pos, rad, world=world, gravity=grav, mode=gpe.MODE_NATIVE, flags=flags)
st.run(1 / 60, 10, resort_every=240, resort_first=True)
skip = int(os.environ.get("TIME_STEP_SKIP", "0"))             # steps to advance before anything is timed (a clumped cloud: 1600)
if skip: st.run(1 / 60, skip, resort_every=240, resort_first=False)
st.ctx.sync()
t0 = time.perf_counter()
st.run(1 / 60, steps, resort_every=240, resort_first=False)
st.ctx.sync()
wall = (time.perf_counter() - t0) / steps * 1e3
st.ctx.set_profiling(True); st.ctx.reset_timings()
st.run(1 / 60, steps, resort_every=240, resort_first=False)
st.ctx.sync()
tim = st.ctx.timings()
tag = " ".join("%s=%s" % (k, os.environ[k]) for k in sorted(os.environ) if k.startswith("GPE_"))
pi = st.ctx.pipeline_info()
tag += " flags=%d sorts %d of %d steps, over-capacity tiles %d" % (flags, pi["native_sorts"], pi["native_steps"], pi["overflow_tiles"])
print("n=%d %s  wall %.4f ms/step | " % (n, tag, wall) +
      "  ".join("%s %.1fus" % (k, v[0] / max(1, v[1]) * 1e3) for k, v in sorted(tim.items(), key=lambda kv: -kv[1][0])), flush=True)
