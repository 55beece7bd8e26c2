"""Per-step cost of the sharded step loop (gpe_shard_run: RCCL exchange + unpack + step + pack) against the plain
gpe_run on ONE GPU: a one-rank communicator whose only neighbour slot is the rank itself (no ghosts, no migrants), so
the difference is the fixed cost of the exchange machinery.  python scripts/shard_overhead_one_rank.py [N] [steps]"""
import ctypes as C, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gpe = importlib.import_module("gpu-physics-engine_amd"); L = gpe._lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # gpe_config.flags, e.g. 256 = GPE_FLAG_SHARD_OVERLAP
world = gpe.scenes.world_for(n)
pos, rad = gpe.scenes.uniform_cloud(n, world, seed=4)

plain = gpe.State(pos, rad, world=world, mode=gpe.MODE_NATIVE)
plain.run(1 / 60, 20, resort_every=0, resort_first=True); plain.ctx.sync()
t0 = time.perf_counter(); plain.run(1 / 60, steps, resort_every=0, resort_first=False); plain.ctx.sync()
t_plain = (time.perf_counter() - t0) / steps
plain.close()

st = gpe.State(pos, rad, world=world, mode=gpe.MODE_NATIVE, flags=flags); ctx = st.ctx
st.update(1 / 60, resort=True)
ctx.call("gpe_use_order_keys", 1)
cs = np.float32(0.5) * np.float32(2.2)
gx = int(np.floor(np.float32(world[0]) / cs)) + 1; gy = int(np.floor(np.float32(world[1]) / cs)) + 1
bx, by = (gx + 7) // 8, (gy + 7) // 8
ctx.call("gpe_set_active_cells", 0, 0, gx - 1, gy - 1)
def dev(nbytes):
    p = C.c_void_p(); ctx.call("gpe_buffer_alloc", nbytes, C.byref(p))
    z = np.zeros(nbytes, np.uint8); ctx.call("gpe_buffer_upload", p, z.ctypes.data_as(C.c_void_p), nbytes); return p
cap_mig, cap_gho = 4096, 16384
words = 4 + 6 * cap_mig + 4 * cap_gho; own_words = 4 + 4 * cap_mig
owner, mask = dev(bx * by), dev(bx * by * 4)
send, recv = dev((words + own_words + 16) * 4), dev((words + 16) * 4)
plan = L.GpeShardPlan(); plan.struct_size = C.sizeof(L.GpeShardPlan)
plan.rank, plan.world_size, plan.n_slots = 0, 1, 2
plan.blocks_x, plan.blocks_y = bx, by
plan.d_owner_of_block, plan.d_dest_mask_of_block = owner.value, mask.value
plan.slot_rank[0], plan.send_off[0], plan.send_cap_mig[0], plan.send_cap_gho[0] = 0, 0, cap_mig, cap_gho
plan.recv_off[0], plan.recv_cap_mig[0], plan.recv_cap_gho[0] = 0, cap_mig, cap_gho
plan.slot_rank[1], plan.send_off[1], plan.send_cap_mig[1], plan.send_cap_gho[1] = 0, words, 0, cap_mig
plan.recv_off[1], plan.recv_cap_mig[1], plan.recv_cap_gho[1] = words, 0, 0
plan.d_send, plan.d_recv = send.value, recv.value
plan.own_x0, plan.own_y0, plan.own_x1, plan.own_y1 = 0, 0, bx, by     # the rank's rectangle: the tiles pack, no pack kernel
ctx.call("gpe_shard_configure", C.byref(plan))
ident = (C.c_uint8 * L.COMM_ID_BYTES)(); assert L.load().gpe_comm_unique_id(ident) == 0
ctx.call("gpe_shard_comm_init", ident, 0, 1)
kp, nb = C.c_void_p(), C.c_uint64(); ctx.call("gpe_device_ptr", L.ORDER_KEYS, C.byref(kp), C.byref(nb))
keys = np.arange(n, dtype=np.uint32); ctx.call("gpe_buffer_upload", kp, keys.ctypes.data_as(C.c_void_p), keys.nbytes)
ctx.call("gpe_shard_begin")
ctx.call("gpe_shard_run", 1.0 / 60.0, 20); ctx.sync()
t0 = time.perf_counter(); ctx.call("gpe_shard_run", 1.0 / 60.0, steps); ctx.sync()
t_shard = (time.perf_counter() - t0) / steps
ctx.set_profiling(True); ctx.reset_timings()
ctx.call("gpe_shard_run", 1.0 / 60.0, 50); ctx.sync()
tim = ctx.timings()
print("flags %d: " % flags, end="")
print("n=%d  plain %.4f ms/step   sharded loop (RCCL self exchange, %d KB segment) %.4f ms/step   overhead %.1f us" %
      (n, t_plain * 1e3, words * 4 // 1024, t_shard * 1e3, (t_shard - t_plain) * 1e6))
print("   " + "  ".join("%s %.1fus" % (k, v[0] / max(1, v[1]) * 1e3) for k, v in sorted(tim.items(), key=lambda kv: -kv[1][0])))
print("   ", ctx.pipeline_info())
