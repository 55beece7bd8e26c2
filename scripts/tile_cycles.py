"""Where the time of the dense and the over-capacity launch goes, tile by tile, in the gravity-on scene at a given step
(library built with -DGPE_TILE_CYCLES: scripts/build_variant.sh cycles "-DGPE_TILE_CYCLES").  Prints, per band of tile
rows (the scene is stratified in y: pile, compression zone, free fall), the tiles by outcome and their cycles.
python scripts/tile_cycles.py N step [step ...]"""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gpe = importlib.import_module("gpu-physics-engine_amd")
n, marks = int(sys.argv[1]), [int(v) for v in sys.argv[2:]]
world = gpe.scenes.world_for(n)
pos, rad = gpe.scenes.uniform_cloud(n, world, seed=0x5EED)
st = gpe.State(pos, rad, world=world, gravity=(0.0, -9.81), mode=gpe.MODE_NATIVE)
lib = gpe._lib.load()
fn = lib.gpe_debug_tile_cycles
done = 0
def advance(k):
    global done
    while k > 0:
        c = min(k, 240 - done % 240)
        st.run(1 / 60, c, resort_every=0, resort_first=(done % 240 == 0)); done += c; k -= c
for m in marks:
    advance(m - done)
    tx, ty = C.c_uint32(), C.c_uint32()
    assert fn(st.ctx.h, None, 0, C.byref(tx), C.byref(ty)) == 0
    advance(1)
    tiles = tx.value * ty.value
    out = np.zeros((tiles, 4), np.uint32)
    assert fn(st.ctx.h, out.ctypes.data_as(C.POINTER(C.c_uint32)), out.size, C.byref(tx), C.byref(ty)) == 0
    out = out.reshape(ty.value, tx.value, 4).astype(np.int64)
    cyc, outcome, qcyc, looked = out[..., 0], out[..., 1], out[..., 2], out[..., 3]
    print("step %d: %d x %d tiles; dense launch %.3g Gcycles of workgroup time, over-capacity launch %.3g; outcomes done %d, handed on %d, held %d" %
          (m, tx.value, ty.value, cyc.sum() / 1e9, qcyc.sum() / 1e9, (outcome == 0).sum(), (outcome == 1).sum(), (outcome == 2).sum()))
    bands = 12
    edges = np.linspace(0, ty.value, bands + 1).astype(int)
    for b in range(bands):
        sl = slice(edges[b], edges[b + 1])
        c, o, q, l = cyc[sl], outcome[sl], qcyc[sl], looked[sl]
        nd = max(1, (o == 0).sum())
        print("   tile rows %3d-%3d: looked up %6.0f / tile | done %6d tiles, %7.0f cycles each | handed on %6d (%7.0f cycles wasted each) | held %6d | quarters %8.0f cycles per over-capacity tile" %
              (edges[b], edges[b + 1] - 1, l.mean(), (o == 0).sum(), c[o == 0].sum() / nd, (o == 1).sum(),
               c[o == 1].mean() if (o == 1).any() else 0.0, (o == 2).sum(), q[o != 0].mean() if (o != 0).any() else 0.0))
    d = cyc[outcome == 0]
    if d.size:
        print("   done tiles: cycles percentiles 10/50/90/99/max = %s" % " / ".join("%d" % v for v in np.percentile(d, [10, 50, 90, 99, 100])))
    q = qcyc[qcyc > 0]
    if q.size:
        # the over-capacity launch ends with its slowest workgroup: the tail of the tiles' quarter cycles
        top = np.sort(q)[::-1]
        print("   over-capacity tiles: quarter cycles percentiles 50/90/99/99.9/max = %s; the 10 slowest: %s; all of them / 768 workgroups = %.3g cycles" %
              (" / ".join("%d" % v for v in np.percentile(q, [50, 90, 99, 99.9, 100])), " ".join("%.3g" % v for v in top[:10]), q.sum() / 768.0))
print(st.ctx.pipeline_info())
