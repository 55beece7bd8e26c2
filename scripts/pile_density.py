"""How dense does a pile get?  Max particles per cell and per 24x24-cell window as a gravity scene evolves."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
gpe = importlib.import_module("gpu-physics-engine_amd")
n = int(sys.argv[1]); chunks = int(sys.argv[2]); per = int(sys.argv[3])
world = gpe.scenes.world_for(n)
pos, rad = gpe.scenes.uniform_cloud(n, world, seed=0x5EED)
st = gpe.State(pos, rad, world=world, gravity=(0.0, -9.81), mode=gpe.MODE_NATIVE)
cs = np.float32(1.1)
gx, gy = int(world[0] / cs) + 1, int(world[1] / cs) + 1
done = 0
for c in range(chunks):
    st.run(1 / 60, per, resort_every=240, resort_first=(c == 0))
    done += per
    p = st.positions()
    cx = np.floor(p[:, 0] / cs).astype(np.int64); cy = np.floor(p[:, 1] / cs).astype(np.int64)
    cells = np.bincount(cy * gx + cx, minlength=gx * gy).reshape(gy, gx)
    bl = cells[: gy // 8 * 8, : gx // 8 * 8].reshape(gy // 8, 8, gx // 8, 8).sum((1, 3))
    w = bl[:-2, :-2] + bl[:-2, 1:-1] + bl[:-2, 2:] + bl[1:-1, :-2] + bl[1:-1, 1:-1] + bl[1:-1, 2:] + bl[2:, :-2] + bl[2:, 1:-1] + bl[2:, 2:]
    print("steps %6d  max/cell %3d  max 8x8 block %5d  max 24x24 window %5d (%.2f per cell)  windows > 1536: %d of %d"
          % (done, cells.max(), bl.max(), w.max(), w.max() / 576.0, int((w > 1536).sum()), w.size), flush=True)
