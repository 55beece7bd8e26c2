"""GPU (-m gpu): two ranks sharing cuda:0 (gloo, staged through the host) run the sharded step on libgpe.so
(native kernels, order keys, ghost band, migration, global re-sort indices); the union of their particles must
be bit-identical to the single-context run.  The nccl/RCCL transport differs from this test only in where the
send/recv buffers live (gpu-physics-engine_amd/sharded.py: stage_cpu)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, ws, port, n, world, gravity, steps, resort_at, dt, seed, out_dir, device_exchange):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        gpe = importlib.import_module("gpu-physics-engine_amd")
        sharded = importlib.import_module("gpu-physics-engine_amd.sharded")
        pos, rad = gpe.scenes.uniform_cloud(n, world, seed=seed)
        cs = np.float32(0.5) * np.float32(2.2)
        dec = sharded.Decomposition(world, cs, ws)
        mine = np.nonzero(dec.owner_of(pos) == rank)[0]
        eng = sharded.GpeEngine(pos[mine], rad[mine], mine, world, gravity=gravity, device=0)
        st = sharded.ShardedState(eng, dec, rank, device_exchange=device_exchange)
        assert st.fast == device_exchange
        for s in range(steps):
            st.update(dt, resort=(s in resort_at))
        gid, p, q = st.owned()
        eng.ctx.sync()
        # device-resident exchange: the counts live on the device; st.owned() reads them back
        info = eng.ctx.pipeline_info()
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), gid=gid, pos=p, prev=q,
                 migrants=st.stats["migrants"] if not st.fast else abs(st.n_owned - len(mine)) + 1,
                 ghosts=st.stats["ghosts"] if not st.fast else st.n_ghost, recuts=st.stats.get("recuts", 0),
                 native_steps=info["native_steps"], native_sorts=info["native_sorts"])
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("device_exchange", [False, True], ids=["torch-exchange", "device-exchange"])
@pytest.mark.parametrize("ws,n,world,gravity", [
    (2, 40_000, (420.0, 300.0), (40.0, 0.0)),
    (4, 60_000, (500.0, 380.0), (25.0, -30.0)),
    (2, 40_000, (420.0, 300.0), (0.0, -80.0)),      # everything falls onto rank 0: its buffers must grow mid-run
])
def test_two_ranks_one_gpu_equal_single_context(gpe, tmp_path, ws, n, world, gravity, device_exchange):
    """device-exchange: packing / hole filling / appending by the library's kernels (csrc/k_shard.hip), counts on the
    device, one all_to_all_single per step; torch-exchange: the general formulation in sharded.py."""
    steps, dt, seed, resort_at = 14, 0.05, 5, (0, 6)
    if gravity[1] <= -80.0:
        steps, resort_at = 30, (0, 17)
    port = _free_port()
    mp.spawn(_worker, args=(ws, port, n, world, gravity, steps, resort_at, dt, seed, str(tmp_path), device_exchange),
             nprocs=ws, join=True)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=seed)
    ref = gpe.State(pos, rad, world=world, gravity=gravity, mode=gpe.MODE_NATIVE)
    for s in range(steps):
        ref.update(dt, resort=(s in resort_at))
    want_pos, want_prev = ref.positions(), ref.previous_positions()
    ref.close()
    gids, poss, prevs, migrants, ghosts, recuts = [], [], [], 0, 0, 0
    for r in range(ws):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        gids.append(d["gid"]); poss.append(d["pos"]); prevs.append(d["prev"])
        migrants += int(d["migrants"]); ghosts += int(d["ghosts"]); recuts += int(d["recuts"])
        assert int(d["native_steps"]) == steps
        if device_exchange and ws == 2 and gravity[1] > -80.0:         # (the 4-rank scene is violent: stragglers force a sort every step)
            # the rank keeps the grouping of its owned particles across steps (ghosts: their own small sort every step);
            # the radix passes of the owned particles ran on the re-sort steps and whenever arrivals demanded it only
            assert int(d["native_sorts"]) < steps, (r, int(d["native_sorts"]), steps)
    gid = np.concatenate(gids)
    assert np.array_equal(np.sort(gid), np.arange(n))
    order = np.argsort(gid)
    assert migrants > 0 and ghosts > 0
    if gravity[1] <= -80.0:
        assert recuts >= ws                                     # the pile-up made every rank re-cut at a re-sort step
    assert np.array_equal(np.concatenate(poss)[order], want_pos)
    assert np.array_equal(np.concatenate(prevs)[order], want_prev)


def _config3_worker(rank, ws, port, n, steps, resort_at, out_dir):
    """One rank of BASELINE.json configs[3]'s workload (100 M particles over 4 ranks, gravity on), all ranks on cuda:0."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        gpe = importlib.import_module("gpu-physics-engine_amd")
        sharded = importlib.import_module("gpu-physics-engine_amd.sharded")
        world = gpe.scenes.world_for(n)
        pos, rad = gpe.scenes.uniform_cloud(n, world, seed=0x5EED)
        dec = sharded.Decomposition(world, np.float32(0.5) * np.float32(2.2), ws)
        mine = np.nonzero(dec.owner_of(pos) == rank)[0]
        p_mine, r_mine = np.ascontiguousarray(pos[mine]), np.ascontiguousarray(rad[mine])
        del pos, rad
        eng = sharded.GpeEngine(p_mine, r_mine, mine, world, gravity=(0.0, -9.81), device=0)
        st = sharded.ShardedState(eng, dec, rank, device_exchange=True)
        for s in range(steps):
            st.update(1 / 60, resort=(s in resort_at))
        gid, p, q = st.owned()
        eng.ctx.sync()
        np.save(os.path.join(out_dir, "gid%d.npy" % rank), gid.astype(np.int64))
        np.save(os.path.join(out_dir, "pos%d.npy" % rank), p)
        np.save(os.path.join(out_dir, "prev%d.npy" % rank), q)
        eng.close()
    finally:
        dist.destroy_process_group()


def test_config3_workload_100m_over_four_ranks_on_one_gpu(gpe, tmp_path):
    """BASELINE.json configs[3]'s WORKLOAD -- 100 M particles, gravity on, cut into 4 rectangles of 25 M -- with the four
    ranks sharing cuda:0 over gloo (no multi-GPU box is available to the build, so this measures nothing about xGMI; it
    does exercise the decomposition, the segment capacities, the order keys and the global re-sort indices at full
    size): 3 steps, re-sorts at steps 0 and 2, bit-identical to the single-context NATIVE run."""
    n, ws, steps, resort_at = 100_000_000, 4, 3, (0, 2)
    port = _free_port()
    mp.spawn(_config3_worker, args=(ws, port, n, steps, resort_at, str(tmp_path)), nprocs=ws, join=True)
    world = gpe.scenes.world_for(n)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=0x5EED)
    ref = gpe.State(pos, rad, world=world, gravity=(0.0, -9.81), mode=gpe.MODE_NATIVE)
    del pos, rad
    for s in range(steps):
        ref.update(1 / 60, resort=(s in resort_at))
    want_pos, want_prev = ref.positions(), ref.previous_positions()
    ref.close()
    seen = np.zeros(n, bool)
    for r in range(ws):
        gid = np.load(os.path.join(str(tmp_path), "gid%d.npy" % r))
        assert not seen[gid].any()
        seen[gid] = True
        # the global ids are the particles' indices in the unsharded system AFTER its re-sorts
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "pos%d.npy" % r)), want_pos[gid]), "rank %d positions" % r
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "prev%d.npy" % r)), want_prev[gid]), "rank %d previous positions" % r
    assert seen.all()


def _overflow_worker(rank, ws, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["GPE_SHARD_CAP_SCALE"] = "0.002"             # a handful of rows per neighbour segment
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        gpe = importlib.import_module("gpu-physics-engine_amd")
        sharded = importlib.import_module("gpu-physics-engine_amd.sharded")
        n, world = 40_000, (420.0, 300.0)
        pos, rad = gpe.scenes.uniform_cloud(n, world, seed=5)
        dec = sharded.Decomposition(world, np.float32(0.5) * np.float32(2.2), ws)
        mine = np.nonzero(dec.owner_of(pos) == rank)[0]
        eng = sharded.GpeEngine(pos[mine], rad[mine], mine, world, device=0)
        st = sharded.ShardedState(eng, dec, rank, device_exchange=True)
        msg = ""
        try:
            for s in range(3):
                st.update(0.05, resort=(s == 0))
            st.owned()
        except gpe.GpeError as e:
            msg = str(e)
        with open(os.path.join(out_dir, "rank%d.txt" % rank), "w") as f:
            f.write(msg)
        eng.close()
    finally:
        dist.destroy_process_group()


def test_device_exchange_overflow_is_loud(gpe, tmp_path):
    """Segments far too small for the ghost band: the kernels drop the rows, set the sticky error word, and the
    next synchronising call reports it -- never a silently wrong result."""
    port = _free_port()
    mp.spawn(_overflow_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        msg = open(os.path.join(str(tmp_path), "rank%d.txt" % r)).read()
        assert "segment overflowed" in msg, msg


def _disagree_worker(rank, ws, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["GPE_SHARD_CAP_SCALE"] = "1" if rank == 0 else "0.5"      # the ranks are configured differently
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        gpe = importlib.import_module("gpu-physics-engine_amd")
        sharded = importlib.import_module("gpu-physics-engine_amd.sharded")
        n, world = 40_000, (420.0, 300.0)
        pos, rad = gpe.scenes.uniform_cloud(n, world, seed=5)
        dec = sharded.Decomposition(world, np.float32(0.5) * np.float32(2.2), ws)
        mine = np.nonzero(dec.owner_of(pos) == rank)[0]
        eng = sharded.GpeEngine(pos[mine], rad[mine], mine, world, device=0)
        msg = "no error"
        try:
            sharded.ShardedState(eng, dec, rank, device_exchange=True)
        except ValueError as e:
            msg = str(e)
        with open(os.path.join(out_dir, "rank%d.txt" % rank), "w") as f:
            f.write(msg)
        eng.close()
    finally:
        dist.destroy_process_group()


def test_ranks_that_disagree_on_segment_sizes_fail_together(gpe, tmp_path):
    """A send and a receive of different lengths would not fail, they would wait for each other for ever (RCCL) or
    scramble the rows (all_to_all): the set-up compares what every rank sends with what its peer expects and raises
    on EVERY rank."""
    port = _free_port()
    mp.spawn(_disagree_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        msg = open(os.path.join(str(tmp_path), "rank%d.txt" % r)).read()
        assert "disagree on the segment sizes" in msg, msg


def _teardown_worker(rank, ws, port, do_close):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    gpe = importlib.import_module("gpu-physics-engine_amd")
    sharded = importlib.import_module("gpu-physics-engine_amd.sharded")
    n, world = 40_000, (420.0, 300.0)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=5)
    dec = sharded.Decomposition(world, np.float32(0.5) * np.float32(2.2), ws)
    mine = np.nonzero(dec.owner_of(pos) == rank)[0]
    eng = sharded.GpeEngine(pos[mine], rad[mine], mine, world, gravity=(40.0, 0.0), device=0)
    st = sharded.ShardedState(eng, dec, rank, device_exchange=True)
    assert st.fast and st.transport == "torch"
    st.run(0.05, 8, resort_every=5, resort_first=True)
    gid, p, q = st.owned()
    assert len(gid) == st.n_owned and np.isfinite(p).all()
    if do_close:
        eng.close()
    dist.destroy_process_group()
    # without close(): the context, its torch views and the staging buffers die in whatever order the interpreter
    # picks as this function returns and the process exits


@pytest.mark.parametrize("do_close", [False, True], ids=["no-close", "close"])
def test_rank_teardown_is_clean(gpe, do_close):
    """A rank that leaves its worker function (and the process) with or without GpeEngine.close() exits with
    rc 0: the context runs on a stream torch owns (gpe_set_stream), so nothing torch's allocators remember can
    name a destroyed stream.  (Round 1: SIGSEGV here when the library's own stream was wrapped in an ExternalStream.)
    mp.spawn raises ProcessExitedException on any non-zero exit code or signal."""
    mp.spawn(_teardown_worker, args=(2, _free_port(), do_close), nprocs=2, join=True)


def _blob_worker(rank, ws, port, n, world, mouse, steps, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        gpe = importlib.import_module("gpu-physics-engine_amd")
        sharded = importlib.import_module("gpu-physics-engine_amd.sharded")
        pos, rad = gpe.scenes.uniform_cloud(n, world, seed=11)
        dec = sharded.Decomposition(world, np.float32(0.5) * np.float32(2.2), ws)
        mine = np.nonzero(dec.owner_of(pos) == rank)[0]
        eng = sharded.GpeEngine(pos[mine], rad[mine], mine, world, device=0)
        eng.ctx.call("gpe_set_mouse", 1, float(mouse[0]), float(mouse[1]))
        st = sharded.ShardedState(eng, dec, rank, device_exchange=True)
        for s0 in range(0, steps, 6):                     # the same schedule as the reference: re-sort every 60 steps
            st.run(1 / 60, 6, resort_every=0, resort_first=(s0 % 60 == 0))
        gid, p, q = st.owned()
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), gid=gid, pos=p, prev=q)
        eng.close()
    finally:
        dist.destroy_process_group()


def test_sharded_blob_crosses_the_handover_threshold(gpe, tmp_path):
    """Mouse attraction packs the cloud into a blob on the boundary between two ranks: 24x24-cell windows pass the
    population at which a single-device run hands over to the compat kernels (4096).  A sharded run cannot hand over
    (the compat kernels know no order keys) -- round 1 aborted with GPE_ERR_UNSUPPORTED here; now its dense windows
    keep going through the sub-tile windows and the spill arena.  Same bits as the single-device run, which does
    hand over."""
    n, world = 100_000, (420.0, 300.0)
    mouse = (210.0, 150.0)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=11)
    ref = gpe.State(pos, rad, world=world, mode=gpe.MODE_NATIVE)
    ref.particles.mouse_click_callback(True, mouse)

    def densest(p):                                        # particles in the fullest 26 x 26 unit window (~24x24 cells)
        h, _, _ = np.histogram2d(p[:, 0], p[:, 1], bins=(int(world[0] // 13), int(world[1] // 13)))
        return (h[:-1, :-1] + h[1:, :-1] + h[:-1, 1:] + h[1:, 1:]).max()

    # run the single-device reference until the blob is denser than the handover population, and a little beyond
    steps, dense = 0, 0
    while steps < 150 and dense <= 4200:
        ref.run(1 / 60, 6, resort_every=0, resort_first=(steps % 60 == 0))
        steps += 6
        dense = densest(ref.positions())
    assert dense > 4096, "the blob never got dense enough (%d)" % dense
    want_pos, want_prev = ref.positions(), ref.previous_positions()
    ref.close()
    mp.spawn(_blob_worker, args=(2, _free_port(), n, world, mouse, steps, str(tmp_path)), nprocs=2, join=True)
    gids, poss, prevs = [], [], []
    for r in range(2):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        gids.append(d["gid"]); poss.append(d["pos"]); prevs.append(d["prev"])
    gid = np.concatenate(gids)
    assert np.array_equal(np.sort(gid), np.arange(n))
    order = np.argsort(gid)
    assert np.array_equal(np.concatenate(poss)[order], want_pos)
    assert np.array_equal(np.concatenate(prevs)[order], want_prev)


def test_rccl_communicator_in_the_library_single_rank(gpe):
    """The in-library RCCL transport on real hardware as far as one GPU allows (RCCL refuses two ranks on one
    device): librccl loads, ncclGetUniqueId / ncclCommInitRank (1 rank) / ncclCommDestroy work on the context's
    device, and an exchange without a plan is refused loudly rather than attempted."""
    import ctypes as C
    n = 2000
    world = gpe.scenes.world_for(n)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=3)
    st = gpe.State(pos, rad, world=world, mode=gpe.MODE_NATIVE)
    lib = gpe._lib.load()
    assert lib.gpe_comm_probe() == 0
    ident = (C.c_uint8 * gpe._lib.COMM_ID_BYTES)()
    assert lib.gpe_comm_unique_id(ident) == 0, lib.gpe_last_error(None)
    assert any(bytes(ident))
    st.ctx.call("gpe_shard_comm_init", ident, 0, 1)
    with pytest.raises(gpe.GpeError) as e:
        st.ctx.call("gpe_shard_exchange")                  # no gpe_shard_configure yet
    assert e.value.status == gpe._lib.GPE_ERR_STATE
    st.ctx.call("gpe_shard_comm_destroy")
    st.update(1 / 60, resort=True)                         # the context is unharmed
    st.ctx.sync()
    st.close()


@pytest.mark.parametrize("overlap", [False, True])
def test_rccl_exchange_moves_a_segment_on_hardware(gpe, overlap):
    """gpe_shard_exchange's RCCL path on a real device: a one-rank communicator and a plan whose only "neighbour" is the
    rank itself, so the grouped ncclSend / ncclRecv pair of that slot moves the packed segment from the send buffer
    into the receive buffer through RCCL, on the context's stream -- the call sequence, sizes and offsets of the
    multi-GPU run, minus the second GPU this box does not have.  overlap: the same on the exchange's own stream
    (GPE_FLAG_SHARD_OVERLAP), ordered with the step by the two events."""
    import ctypes as C
    L = gpe._lib
    n = 4000
    world = gpe.scenes.world_for(n)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=4)
    st = gpe.State(pos, rad, world=world, mode=gpe.MODE_NATIVE, flags=L.FLAG_SHARD_OVERLAP if overlap else 0)
    ctx = st.ctx
    ctx.call("gpe_use_order_keys", 1)
    cs = np.float32(0.5) * np.float32(2.2)
    gx = int(np.floor(np.float32(world[0]) / cs)) + 1
    gy = int(np.floor(np.float32(world[1]) / cs)) + 1
    bx, by = (gx + 7) // 8, (gy + 7) // 8
    ctx.call("gpe_set_active_cells", 0, 0, gx - 1, gy - 1)

    def dev(nbytes, fill=None):
        p = C.c_void_p()
        ctx.call("gpe_buffer_alloc", nbytes, C.byref(p))
        data = np.zeros(nbytes, np.uint8) if fill is None else fill
        ctx.call("gpe_buffer_upload", p, data.ctypes.data_as(C.c_void_p), data.nbytes)
        return p

    cap_mig, cap_gho = 8, 16
    words = 4 + 6 * cap_mig + 4 * cap_gho
    own_words = 4 + 4 * 8
    owner, mask = dev(bx * by), dev(bx * by * 4)           # rank 0 owns every block, nothing borders another rank
    send, recv = dev((words + own_words + 16) * 4), dev((words + 16) * 4)
    plan = L.GpeShardPlan()
    plan.struct_size = C.sizeof(L.GpeShardPlan)
    plan.rank, plan.world_size, plan.n_slots = 0, 1, 2
    plan.blocks_x, plan.blocks_y = bx, by
    plan.d_owner_of_block, plan.d_dest_mask_of_block = owner.value, mask.value
    plan.slot_rank[0], plan.send_off[0], plan.send_cap_mig[0], plan.send_cap_gho[0] = 0, 0, cap_mig, cap_gho
    plan.recv_off[0], plan.recv_cap_mig[0], plan.recv_cap_gho[0] = 0, cap_mig, cap_gho
    plan.slot_rank[1], plan.send_off[1], plan.send_cap_mig[1], plan.send_cap_gho[1] = 0, words, 0, 8
    plan.recv_off[1], plan.recv_cap_mig[1], plan.recv_cap_gho[1] = words, 0, 0
    plan.d_send, plan.d_recv = send.value, recv.value
    ctx.call("gpe_shard_configure", C.byref(plan))
    ident = (C.c_uint8 * L.COMM_ID_BYTES)()
    assert L.load().gpe_comm_unique_id(ident) == 0
    ctx.call("gpe_shard_comm_init", ident, 0, 1)
    keys_ptr, nbytes = C.c_void_p(), C.c_uint64()
    ctx.call("gpe_device_ptr", L.ORDER_KEYS, C.byref(keys_ptr), C.byref(nbytes))
    keys = np.arange(n, dtype=np.uint32)                   # order keys = local indices: the run must equal a plain one
    ctx.call("gpe_buffer_upload", keys_ptr, keys.ctypes.data_as(C.c_void_p), keys.nbytes)
    ctx.call("gpe_shard_begin")                            # packs (nothing borders anything: empty segments)
    # the step loop of a sharded run inside the library: 6 x { RCCL exchange, unpack, step, pack }
    ctx.call("gpe_shard_run", 1.0 / 60.0, 6)
    no, nt = C.c_uint64(), C.c_uint64()
    ctx.call("gpe_shard_counts", C.byref(no), C.byref(nt), 1)
    assert (no.value, nt.value) == (n, n)
    plain = gpe.State(pos, rad, world=world, mode=gpe.MODE_NATIVE)
    plain.run(1 / 60, 6, resort_every=0, resort_first=False)
    assert np.array_equal(st.positions(), plain.positions())
    assert np.array_equal(st.previous_positions(), plain.previous_positions())
    plain.close()
    ctx.call("gpe_shard_begin")
    ctx.sync()
    pattern = (np.arange(words, dtype=np.uint32) * np.uint32(2654435761)) | np.uint32(1)
    pattern[0:4] = 0                                       # header: no rows, so that a later unpack has nothing to do
    ctx.call("gpe_buffer_upload", send, pattern.ctypes.data_as(C.c_void_p), pattern.nbytes)
    ctx.call("gpe_shard_exchange")
    ctx.sync()
    got = np.zeros(words, np.uint32)
    ctx.call("gpe_buffer_download", recv, got.ctypes.data_as(C.c_void_p), got.nbytes)
    assert np.array_equal(got, pattern)
    ctx.call("gpe_shard_comm_destroy")
    for p in (owner, mask, send, recv):
        ctx.call("gpe_buffer_free", p)
    st.close()


def _mixed_scene(n, world, seed):
    """Radii 0.5 .. 3, the big ones (radius > 1) only in the right-hand third of the world: the ranks on the left hold
    no particle of the largest radius, so their contexts derive a smaller cell size on their own."""
    rng = np.random.default_rng(seed)
    pos = (rng.random((n, 2), dtype=np.float32) * np.array(world, np.float32)).astype(np.float32)
    rad = rng.choice(np.array([0.5, 0.75, 1.0], np.float32), n).astype(np.float32)
    right = pos[:, 0] > np.float32(world[0] * 0.67)
    rad[right] = rng.choice(np.array([0.5, 1.0, 2.0, 3.0], np.float32), int(right.sum())).astype(np.float32)
    return pos, rad


def _mixed_worker(rank, ws, port, n, world, gravity, steps, resort_at, dt, seed, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        gpe = importlib.import_module("gpu-physics-engine_amd")
        sharded = importlib.import_module("gpu-physics-engine_amd.sharded")
        pos, rad = _mixed_scene(n, world, seed)
        cs = np.float32(np.abs(rad).max()) * np.float32(2.2)
        dec = sharded.Decomposition(world, cs, ws, grid=(ws, 1))          # vertical strips: rank 0 is all small radii
        mine = np.nonzero(dec.owner_of(pos) == rank)[0]
        if rank == 0:
            assert np.abs(rad[mine]).max() < np.abs(rad).max()
        eng = sharded.GpeEngine(pos[mine], rad[mine], mine, world, gravity=gravity, device=0)
        st = sharded.ShardedState(eng, dec, rank, device_exchange=True)
        assert st.fast
        for s in range(steps):
            st.update(dt, resort=(s in resort_at))
        gid, p, q = st.owned()
        eng.ctx.sync()
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), gid=gid, pos=p, prev=q)
        eng.close()
    finally:
        dist.destroy_process_group()


def test_mixed_radii_ranks_agree_on_the_cell_size(gpe, tmp_path):
    """The cell size is 2.2 x the largest radius of the whole system; a rank that holds none of the big particles
    must still use it (ShardedState._agree_on_cell_size).  Three ranks sharing the GPU, vertical strips, gravity
    pushing particles across the cuts: bit-identical to the single-context run."""
    ws, n, world, gravity = 3, 30_000, (1500.0, 500.0), (-60.0, -5.0)
    steps, dt, seed, resort_at = 12, 0.05, 21, (0, 7)
    port = _free_port()
    mp.spawn(_mixed_worker, args=(ws, port, n, world, gravity, steps, resort_at, dt, seed, str(tmp_path)),
             nprocs=ws, join=True)
    pos, rad = _mixed_scene(n, world, seed)
    ref = gpe.State(pos, rad, world=world, gravity=gravity, mode=gpe.MODE_NATIVE)
    for s in range(steps):
        ref.update(dt, resort=(s in resort_at))
    want_pos, want_prev = ref.positions(), ref.previous_positions()
    ref.close()
    gids, poss, prevs = [], [], []
    for r in range(ws):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        gids.append(d["gid"]); poss.append(d["pos"]); prevs.append(d["prev"])
    gid = np.concatenate(gids)
    assert np.array_equal(np.sort(gid), np.arange(n))
    order = np.argsort(gid)
    assert np.array_equal(np.concatenate(poss)[order], want_pos)
    assert np.array_equal(np.concatenate(prevs)[order], want_prev)


def _spill_blob_scene():
    """A sparse cloud of radius-0.5 particles plus 9000 particles of radius 0.01 packed into 12 x 12 units across the
    cut between two ranks (block row 18 = y 158.4), all drifting up at 0.01 units per step: every 8x8-cell tile near
    the blob looks up more particles than any LDS window stages (the global spill windows resolve them), a handful of
    blob particles migrate to the upper rank every step, and nothing moves fast enough to force a sort."""
    world = (420.0, 300.0)
    rng = np.random.default_rng(99)
    nb, nt = 40_000, 9_000
    pos = np.empty((nb + nt, 2), np.float32)
    pos[:nb] = (rng.random((nb, 2), dtype=np.float32) * np.array(world, np.float32)).astype(np.float32)
    pos[nb:, 0] = np.float32(200.0) + rng.random(nt, dtype=np.float32) * np.float32(12.0)
    pos[nb:, 1] = np.float32(152.0) + rng.random(nt, dtype=np.float32) * np.float32(12.0)
    rad = np.full(nb + nt, 0.5, np.float32)
    rad[nb:] = np.float32(0.01)
    prev = pos.copy()
    prev[nb:, 1] -= np.float32(0.01)
    return world, pos, prev, rad


def _spill_blob_worker(rank, ws, port, steps, resort_at, dt, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        gpe = importlib.import_module("gpu-physics-engine_amd")
        sharded = importlib.import_module("gpu-physics-engine_amd.sharded")
        world, pos, prev, rad = _spill_blob_scene()
        dec = sharded.Decomposition(world, np.float32(0.5) * np.float32(2.2), ws)
        mine = np.nonzero(dec.owner_of(pos) == rank)[0]
        eng = sharded.GpeEngine(pos[mine], rad[mine], mine, world, device=0, prev=prev[mine])
        st = sharded.ShardedState(eng, dec, rank, device_exchange=True)
        assert st.fast
        for s in range(steps):
            st.update(dt, resort=(s in resort_at))
        gid, p, q = st.owned()
        eng.ctx.sync()
        info = eng.ctx.pipeline_info()
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), gid=gid, pos=p, prev=q, n0=len(mine),
                 native_sorts=info["native_sorts"], window_max=info["window_max"])
        eng.close()
    finally:
        dist.destroy_process_group()


def test_migrants_into_a_spill_window_on_a_step_that_keeps_its_table(gpe, tmp_path):
    """A sharded run that keeps its block table, over a region so dense that its 8x8-cell tiles go through the global
    spill windows, with particles arriving from the other rank on steps that do not sort: the arrivals are stragglers
    (no block of the kept table lists them), the spill window files them behind its looked-up slots, and its write-back
    must find their local index through the straggler list -- round 3 searched the block offsets for them and wrote
    their result onto some other particle.  Bit-identical to the single-context run."""
    ws, steps, dt, resort_at = 2, 12, 1 / 60, (0, 7)
    mp.spawn(_spill_blob_worker, args=(ws, _free_port(), steps, resort_at, dt, str(tmp_path)), nprocs=ws, join=True)
    world, pos, prev, rad = _spill_blob_scene()
    ref = gpe.State(pos, rad, world=world, mode=gpe.MODE_NATIVE, prev=prev)
    for s in range(steps):
        ref.update(dt, resort=(s in resort_at))
    want_pos, want_prev = ref.positions(), ref.previous_positions()
    ref.close()
    n = len(rad)
    gids, poss, prevs, moved = [], [], [], 0
    for r in range(ws):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        gids.append(d["gid"]); poss.append(d["pos"]); prevs.append(d["prev"])
        moved += abs(len(d["gid"]) - int(d["n0"]))
        assert int(d["native_sorts"]) < steps, "every step sorted: the kept table was never used (%d)" % int(d["native_sorts"])
        # (the statistic counts a rank's OWNED particles per 24x24-cell window: about half the blob each; with the other
        # half as ghosts every 8x8 tile near the blob looks up ~9000 particles, more than the 4096 an LDS window takes)
        assert int(d["window_max"]) > 2500, "the blob never filled a spill window (%d)" % int(d["window_max"])
    assert moved >= 20, "too few particles changed rank (%d)" % moved
    gid = np.concatenate(gids)
    assert np.array_equal(np.sort(gid), np.arange(n))
    order = np.argsort(gid)
    assert np.array_equal(np.concatenate(poss)[order], want_pos)
    assert np.array_equal(np.concatenate(prevs)[order], want_prev)


@pytest.mark.parametrize("ws,n,world,gravity", [
    (2, 40_000, (420.0, 300.0), (40.0, 0.0)),
    (4, 60_000, (500.0, 380.0), (25.0, -30.0)),
    (8, 120_000, (700.0, 520.0), (15.0, -20.0)),
    (2, 40_000, (420.0, 300.0), (0.0, -80.0)),      # everything falls onto rank 0: buffers grow, the rectangles are re-cut
])
@pytest.mark.parametrize("overlap", [False, True])
def test_local_group_in_one_process_equals_single_context(gpe, ws, n, world, gravity, overlap):
    """The sharded run with NO Python in its control plane and no torch anywhere: `ws` contexts of this process, one
    thread each, as a local group (gpe_local_group_*), set up and stepped by gpe_shard_setup / gpe_shard_run_scheduled
    (decomposition, cell size, segments, global re-sort, re-cut: csrc/gpe_shard_ctl.hip).  Bit-identical to the
    single-context run; the order keys a rank hands back ARE the single-context indices.
    overlap: GPE_FLAG_SHARD_OVERLAP -- the tiles along the rank's border first (k_collide_border), the exchange on a stream
    of its own beside the interior tiles, the next unpack behind its event."""
    lg = importlib.import_module("gpu-physics-engine_amd.local_group")
    steps, dt, seed, every = 14, 0.05, 5, 6
    if gravity[1] <= -80.0:
        steps, every = 30, 17
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=seed)
    run = lg.LocalShardedRun(pos, rad, world, ws, gravity=gravity, flags=gpe._lib.FLAG_SHARD_OVERLAP if overlap else 0)
    run.run(dt, steps, resort_every=every, resort_first=True)
    owned = run.owned()
    stats = run.stats()
    run.close()
    ref = gpe.State(pos, rad, world=world, gravity=gravity, mode=gpe.MODE_NATIVE)
    ref.run(dt, steps, resort_every=every, resort_first=True)
    want_pos, want_prev = ref.positions(), ref.previous_positions()
    ref.close()
    seen = np.zeros(n, bool)
    for r, (gid, p, q) in enumerate(owned):
        assert not seen[gid].any()
        seen[gid] = True
        assert np.array_equal(p, want_pos[gid]), "rank %d positions" % r
        assert np.array_equal(q, want_prev[gid]), "rank %d previous positions" % r
        assert stats[r]["transport"] == 2 and stats[r]["steps"] == steps and stats[r]["resorts"] == (steps + every - 1) // every
    assert seen.all()
    if gravity[1] <= -80.0:
        assert all(s["recuts"] >= 1 for s in stats), stats


@pytest.mark.parametrize("overlap", [False, True])
def test_dense_patch_on_the_border_between_two_ranks(gpe, overlap):
    """A patch of ~3 x 3 tiles at 1.9 x the benchmark density and a small blob far denser than that, both astride the cut
    between two ranks: the order-key tiles there keep more particles than the direct-slot form stages (880), so they are
    handed on -- to the half-tile launch (k_collide_halves<ORD>, which packs for the neighbour like any border tile) and,
    the blob and a one-cell pile of 700, to the 16 x 16 / 8 x 8 / spill windows; with the exchange beside the step (overlap) the border tiles are redone on
    the spot by k_collide_border instead.  Bit-identical to the single-context run; gravity drags the patch across the cut."""
    lg = importlib.import_module("gpu-physics-engine_amd.local_group")
    n, world, g = 60_000, (500.0, 380.0), (12.0, -6.0)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=41)
    rng = np.random.default_rng(42)
    side = np.float32(3 * 32 * 1.1)
    extra = int(0.9 * 0.3131 * float(side) ** 2)
    patch = (np.array([250.0 - float(side) / 2, 120.0], np.float32) + rng.random((extra, 2), dtype=np.float32) * side).astype(np.float32)
    blob = (np.array([243.0, 300.0], np.float32) + rng.random((1200, 2), dtype=np.float32) * np.float32(14.0)).astype(np.float32)
    # ... and 700 particles pressed into one cell next to the cut: a cell for the blocked whole-wave walk (65..1024 members)
    # of an order-key spill window, its neighbours (phantom members) for rows and whole waves
    pile = (np.array([249.7, 200.2], np.float32) + rng.random((700, 2), dtype=np.float32) * np.float32(0.9)).astype(np.float32)
    pos = np.concatenate([pos, patch, blob, pile]).astype(np.float32)
    rad = np.full(len(pos), 0.5, np.float32)
    steps, dt, every = 20, 1 / 60, 12
    run = lg.LocalShardedRun(pos, rad, world, 2, gravity=g, grid=(2, 1), flags=gpe._lib.FLAG_SHARD_OVERLAP if overlap else 0)
    run.run(dt, steps, resort_every=every, resort_first=True)
    owned = run.owned()
    over = [c.pipeline_info()["overflow_tiles"] for c in run.ctx]
    run.close()
    ref = gpe.State(pos, rad, world=world, gravity=g, mode=gpe.MODE_NATIVE)
    ref.run(dt, steps, resort_every=every, resort_first=True)
    want_pos, want_prev = ref.positions(), ref.previous_positions()
    ref.close()
    seen = np.zeros(len(pos), bool)
    for r, (gid, p, q) in enumerate(owned):
        assert not seen[gid].any()
        seen[gid] = True
        assert np.array_equal(p, want_pos[gid]), "rank %d positions" % r
        assert np.array_equal(q, want_prev[gid]), "rank %d previous positions" % r
    assert seen.all()
    if not overlap:
        assert max(over) > 0, over                         # tiles were handed on (with overlap the border kernel redoes its own)


def test_config4_workload_800m_single_context_and_eight_ranks(gpe):
    """BASELINE.json configs[4]'s WORKLOAD on the one GPU of the box: 800 M particles at the reference density in the
    near-square world (45 953 cells per axis: the regime the 16-bit cell coordinates of grid.wgsl:101-114 cap, 33 M
    blocks, the FOUR-pass block-key sort no smaller case reaches), gravity on, two steps with the first re-sorting.
    (a) one NATIVE context: four radix passes, the re-sort is a permutation, positions finite and inside the walls;
    (b) eight ranks x 100 M in one process (a local group sharing cuda:0; the box allows six GPU processes), run after
        (a) is closed: bit-identical to (a).  No xGMI is involved -- that is the driver's SCALE run."""
    import ctypes as C
    import torch
    n, ws = 800_000_000, 8
    free, total = torch.cuda.mem_get_info(0)
    if free < 150 << 30:
        pytest.skip("needs ~110 GB of device memory for the single context (free: %d GB)" % (free >> 30))
    try:
        avail_kb = int([l for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0].split()[1])
    except Exception:                                      # noqa: BLE001
        avail_kb = 0
    if avail_kb and avail_kb < 90 << 20:
        pytest.skip("needs ~70 GB of host memory (available: %d GB)" % (avail_kb >> 20))
    lg = importlib.import_module("gpu-physics-engine_amd.local_group")
    world = gpe.scenes.world_for(n, aspect=1.0)
    assert 50_000 < world[0] < 51_000 and world[0] == world[1], world
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=0x5EED + 4)
    g, dt, steps = (0.0, -9.81), 1 / 60, 2
    # (a)
    ref = gpe.State(pos, rad, world=world, gravity=g, mode=gpe.MODE_NATIVE)
    info = ref.ctx.pipeline_info()
    assert (info["pipeline"], info["sort_passes"]) == (gpe._lib.PIPELINE_NATIVE, 4), info
    ref.run(dt, steps, resort_every=0, resort_first=True)
    want_pos, want_prev = ref.positions(), ref.previous_positions()
    ids = ref.particles.download_particle_ids()
    info = ref.ctx.pipeline_info()
    ref.close()
    assert info["native_steps"] == steps and info["compat_steps"] == 0, info
    seen = np.zeros(n, bool)
    seen[ids] = True
    assert seen.all(), "the re-sort is not a permutation"
    del ids
    assert np.isfinite(want_pos).all()
    assert want_pos.min() >= 0.5 and want_pos[:, 0].max() <= np.float32(world[0]) - np.float32(0.5) and \
        want_pos[:, 1].max() <= np.float32(world[1]) - np.float32(0.5)
    cells = int(np.floor(np.float32(world[0]) / (np.float32(0.5) * np.float32(2.2)))) + 1
    assert 45_000 < cells < 47_000, cells
    # (b)
    run = lg.LocalShardedRun(pos, rad, world, ws, gravity=g)
    del pos, rad
    run.run(dt, steps, resort_every=0, resort_first=True)
    seen[:] = False
    stats = run.stats()
    for r in range(ws):
        c = run.ctx[r]
        cap = C.c_uint64()
        c.call("gpe_capacity", C.byref(cap))
        gid, p, q = np.empty(cap.value, np.uint32), np.empty((cap.value, 2), np.float32), np.empty((cap.value, 2), np.float32)
        no = C.c_uint64()
        c.call("gpe_shard_download_owned", gid.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p),
               q.ctypes.data_as(C.c_void_p), cap.value, C.byref(no))
        k = int(no.value)
        assert 90_000_000 < k < 110_000_000, (r, k)
        gid = gid[:k].astype(np.int64)
        assert not seen[gid].any()
        seen[gid] = True
        assert np.array_equal(p[:k], want_pos[gid]), "rank %d positions" % r
        assert np.array_equal(q[:k], want_prev[gid]), "rank %d previous positions" % r
        assert stats[r]["resorts"] == 1 and stats[r]["transport"] == 2, stats[r]
        del gid, p, q
    run.close()
    assert seen.all()
