"""What the parts of the counting-sort windows' colour passes cost in the crushed 100 M scene (library built with
-DGPE_DBG_RT: scripts/build_variant.sh dbgrt "-DGPE_DBG_RT"): the product's kernels up to the mark, then a few steps each
with a part switched off (results wrong from there on; the scene barely changes in a few steps).
python scripts/overflow_phases.py N mark [mark ...]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gpe = importlib.import_module("gpu-physics-engine_amd")
n, marks = int(sys.argv[1]), [int(v) for v in sys.argv[2:]]
world = gpe.scenes.world_for(n)
pos, rad = gpe.scenes.uniform_cloud(n, world, seed=0x5EED)
st = gpe.State(pos, rad, world=world, gravity=(0.0, -9.81), mode=gpe.MODE_NATIVE)
skip = gpe._lib.load().gpe_debug_skip
done = 0
def advance(k):
    global done
    while k > 0:
        c = min(k, 240 - done % 240)
        st.run(1 / 60, c, resort_every=0, resort_first=(done % 240 == 0)); done += c; k -= c
for m in marks:
    advance(m - done)
    snap = st.save() if hasattr(st, "save_state") else None
    for mask, name in ((0, "all"), (4, "without whole-wave cells"), (1, "without lane groups"), (2, "without one-lane cells"), (7, "without colour passes"), (0, "all again")):
        assert skip(st.ctx.h, mask) == 0
        st.ctx.set_profiling(True); st.ctx.reset_timings()
        advance(4)
        st.ctx.sync()
        tim = st.ctx.timings()
        st.ctx.set_profiling(False)
        print("step %d %-26s " % (m, name) + "  ".join("%s %.2fms" % (k.replace("native/", ""), v[0] / max(1, v[1])) for k, v in sorted(tim.items(), key=lambda kv: -kv[1][0])[:4]), flush=True)
    skip(st.ctx.h, 0)
print(st.ctx.pipeline_info())
