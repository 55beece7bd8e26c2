"""Why tiles run over the direct-slot form in the long 1 M gravity-off run (library built with -DGPE_TILE_CYCLES):
python scripts/bail_reasons.py [N] [steps]"""
import ctypes as C, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gpe = importlib.import_module("gpu-physics-engine_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
world = gpe.scenes.world_for(n)
pos, rad = gpe.scenes.uniform_cloud(n, world, seed=0x5EED + 1)
st = gpe.State(pos, rad, world=world, mode=gpe.MODE_NATIVE)
fn = gpe._lib.load().gpe_debug_bail_reasons
out = (C.c_uint32 * 8)()
done = 0
while done < steps:
    t0 = time.perf_counter()
    st.run(1 / 60, 200, resort_every=240, resort_first=(done == 0)); st.ctx.sync()
    el = time.perf_counter() - t0
    done += 200
    fn(st.ctx.h, out)
    pi = st.ctx.pipeline_info()
    print("steps %4d-%4d: %.4f ms/step  over-capacity tiles now %d | per step: looked-up>1536 %.2f  kept>928 %.2f  side list>96 %.2f  cell>64 (lanes) %.2f  wave cells>16 (lanes) %.2f" %
          (done - 200, done, el / 200 * 1e3, pi["overflow_tiles"], out[1] / 200, out[2] / 200, out[3] / 200, out[4] / 200, out[5] / 200), flush=True)
