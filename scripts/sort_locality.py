"""Is the onesweep pass sensitive to the order of its input?  100M (key, id) pairs: random 22-bit keys, sorted keys,
and the step's own pattern (row-major block index along a Morton-ordered particle array)."""
import importlib, os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
gpe = importlib.import_module("gpu-physics-engine_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ctx = gpe.Context()
rng = np.random.default_rng(1)
bx, by = 3464, 1191
def morton_pattern():
    # particles in Morton order of their block; key = row-major block index
    b = np.arange(bx * by, dtype=np.uint64)
    x, y = b % bx, b // bx
    def spread(v):
        v = v & 0xFFFF; v = (v | (v << 8)) & 0x00FF00FF; v = (v | (v << 4)) & 0x0F0F0F0F
        v = (v | (v << 2)) & 0x33333333; v = (v | (v << 1)) & 0x55555555; return v
    order = np.argsort(spread(x) | (spread(y) << 1), kind="stable")
    per = n // (bx * by) + 1
    return np.repeat(b[order].astype(np.uint32), per)[:n]
cases = {"random 22-bit": rng.integers(0, bx * by, n, dtype=np.uint32),
         "sorted": np.sort(rng.integers(0, bx * by, n, dtype=np.uint32)),
         "row-major key along Morton order": morton_pattern()}
vals = np.arange(n, dtype=np.uint32)
for name, keys in cases.items():
    kb, vb = gpe.GpuBuffer(ctx, keys), gpe.GpuBuffer(ctx, vals)
    ctx.call("gpe_sort_pairs_u32", kb.dptr, vb.dptr, n); ctx.sync()          # warm-up (sorted afterwards!)
    kb.free(); vb.free()
    kb, vb = gpe.GpuBuffer(ctx, keys), gpe.GpuBuffer(ctx, vals)
    ctx.set_profiling(True); ctx.reset_timings()
    ctx.call("gpe_sort_pairs_u32", kb.dptr, vb.dptr, n); ctx.sync()
    t = ctx.timings()
    print("%-36s %s" % (name, "  ".join("%s %.3f ms/call x%d" % (k, v[0] / max(1, v[1]), v[1]) for k, v in t.items())), flush=True)
    ctx.set_profiling(False)
    kb.free(); vb.free()
ctx.close()
