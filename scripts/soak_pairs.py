"""The gravity-on scene later in its run (BASELINE.json configs[2]: it falls, piles up and is crushed): ms/step over
`window` steps at each mark and -- with a library built with -DGPE_COUNT_PAIRS (scripts/build_variant.sh pairs
"-DGPE_COUNT_PAIRS") -- the pairs the colour passes walk and resolve per step, i.e. ms per 10^9 pairs.
python scripts/soak_pairs.py N window mark [mark ...]"""
import ctypes as C, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gpe = importlib.import_module("gpu-physics-engine_amd")
n, window, marks = int(sys.argv[1]), int(sys.argv[2]), [int(v) for v in sys.argv[3:]]
world = gpe.scenes.world_for(n)
pos, rad = gpe.scenes.uniform_cloud(n, world, seed=0x5EED)
st = gpe.State(pos, rad, world=world, gravity=(0.0, -9.81), mode=gpe.MODE_NATIVE, flags=int(os.environ.get("SOAK_FLAGS", "0")))
lib = gpe._lib.load()
counter = getattr(lib, "gpe_debug_pair_counts", None)
done = 0
def advance(k):
    global done
    while k > 0:
        c = min(k, 240 - done % 240)
        st.run(1 / 60, c, resort_every=0, resort_first=(done % 240 == 0)); done += c; k -= c
for m in marks:
    advance(m - window // 2 - done)
    st.ctx.sync()
    out = (C.c_uint64 * 2)()
    if counter: counter(st.ctx.h, out, 1)
    s0 = st.ctx.pipeline_info()["native_sorts"]
    t0 = time.perf_counter(); advance(window); st.ctx.sync(); el = time.perf_counter() - t0
    msg = "n=%d around step %5d: %8.3f ms/step  sorts %3d of %d" % (n, m, el / window * 1e3, st.ctx.pipeline_info()["native_sorts"] - s0, window)
    if counter:
        counter(st.ctx.h, out, -1)
        msg += "  pairs walked %.4g / step, resolved %.4g / step (diagnostic build: the time is not the product's)" % (out[0] / window, out[1] / window)
    print(msg, flush=True)
print(st.ctx.pipeline_info())
