"""Per-scope times of the 100M (or N) gravity-on scene after S steps: python scripts/soak_profile.py N S"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gpe = importlib.import_module("gpu-physics-engine_amd")
n = int(sys.argv[1]); S = int(sys.argv[2])
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
world = gpe.scenes.world_for(n)
pos, rad = gpe.scenes.uniform_cloud(n, world, seed=0x5EED)
st = gpe.State(pos, rad, world=world, gravity=(0.0, -9.81), mode=gpe.MODE_NATIVE, flags=flags)
done = 0
while done < S:
    k = min(240 - done % 240, S - done)
    st.run(1 / 60, k, resort_every=0, resort_first=(done % 240 == 0)); done += k
st.ctx.sync()
st.ctx.set_profiling(True); st.ctx.reset_timings()
t0 = time.perf_counter(); st.run(1 / 60, 30, resort_every=0, resort_first=False); st.ctx.sync()
print("n=%d flags=%d after %d steps: %.3f ms/step  %s" % (n, flags, S, (time.perf_counter() - t0) / 30 * 1e3, st.ctx.pipeline_info()))
for k, v in sorted(st.ctx.timings().items(), key=lambda kv: -kv[1][0]):
    print("   %-40s %9.3f ms/call x %d" % (k, v[0] / max(1, v[1]), v[1]))
