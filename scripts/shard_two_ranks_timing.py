"""Two (or more) ranks sharing cuda:0 over gloo: wall time per sharded step, with and without the kept block table
(gpe_config.flags = 2: sort every step).  python scripts/shard_two_ranks_timing.py [N total] [ranks] [steps]"""
import importlib, os, socket, sys, time
import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, ws, port, n, steps, flags):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    gpe = importlib.import_module("gpu-physics-engine_amd")
    sharded = importlib.import_module("gpu-physics-engine_amd.sharded")
    world = gpe.scenes.world_for(n)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=0x5EED)
    dec = sharded.Decomposition(world, np.float32(0.5) * np.float32(2.2), ws)
    mine = np.nonzero(dec.owner_of(pos) == rank)[0]
    eng = sharded.GpeEngine(np.ascontiguousarray(pos[mine]), np.ascontiguousarray(rad[mine]), mine, world, gravity=(0.0, -9.81), device=0, flags=flags)
    st = sharded.ShardedState(eng, dec, rank, device_exchange=True)
    del pos, rad
    if os.environ.get("GPE_TIMING") == "1" and rank == 0:
        def timed(obj, name):
            f = getattr(obj, name)
            def g(*a, **k):
                eng.sync(); t = time.perf_counter(); r = f(*a, **k); eng.sync()
                print("      %s: %.1f ms" % (name, (time.perf_counter() - t) * 1e3), flush=True); return r
            setattr(obj, name, g)
        for nm in ("rebalance", "resort", "_plan_device_exchange", "_densest_rank_per_block", "_run_steps"):
            timed(st, nm)
        for nm in ("set_counts", "morton_resort", "shard_counts", "make_tables", "set_active_cells"):
            timed(eng, nm)
        oc = eng.ctx.call
        def call(name, *a):
            if name in ("gpe_shard_begin", "gpe_shard_exchange", "gpe_shard_unpack", "gpe_shard_configure", "gpe_shard_run"):
                eng.sync(); t = time.perf_counter(); r = oc(name, *a); eng.sync()
                print("      %s: %.1f ms" % (name, (time.perf_counter() - t) * 1e3), flush=True); return r
            return oc(name, *a)
        eng.ctx.call = call
    for s in range(steps):
        eng.sync(); dist.barrier(); t0 = time.perf_counter()
        st.update(1 / 60, resort=(s in (0, steps - 2)))
        eng.sync(); dist.barrier()
        if rank == 0:
            print("flags %d step %d: %.1f ms" % (flags, s, (time.perf_counter() - t0) * 1e3), flush=True)
    if rank == 0:
        eng.ctx.set_profiling(True); eng.ctx.reset_timings()
    st.update(1 / 60, resort=False); eng.sync()
    if rank == 0:
        print("   " + "  ".join("%s %.2fms" % (k, v[0] / max(1, v[1])) for k, v in sorted(eng.ctx.timings().items(), key=lambda kv: -kv[1][0])[:10]), flush=True)
        print("   pipeline", eng.ctx.pipeline_info(), flush=True)
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
    ws = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    for flags in ((0,) if os.environ.get("GPE_TIMING") == "1" else (2, 0)):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        mp.spawn(worker, args=(ws, port, n, steps, flags), nprocs=ws, join=True)
