"""Multi-GPU particle step: one process per GPU, spatial shards, point-to-point halo exchange.

The reference is single-device.  Here every rank owns the particles whose HOME CELL lies in its part of the
world (ownership is per 8x8-cell block, the granule of the library's block table).  Each step:

  1. classify   one streaming pass (library kernel `gpe_shard_classify`) lists the owned particles that
                now sit in another rank's block (migrants) or in a block bordering other ranks (ghosts);
  2. exchange   counts by all_gather, then ONE message per neighbouring rank pair over
                torch.distributed point-to-point (`batch_isend_irecv`; backend nccl = RCCL send/recv over
                xGMI on GPUs, gloo on CPU): migrants travel with (pos, prev, radius, order key), ghosts
                with (pos, radius, order key);
  3. step       the rank runs the ordinary step on owned + ghost particles; only owned ones are
                integrated (gpe_set_counts), ghosts are dropped afterwards.

Exactness.  A ghost band of one block (8 cells) covers the dependency cone of the four colour passes
(colour-k cells within 5-k cells of an owned particle's home, their members' homes within 5 cells), and
the members of a cell are ordered by ORDER KEY = the particle's index in the unsharded system
(gpe_use_order_keys), so every rank resolves its own particles with exactly the operations the
single-device run performs: the sharded result is bit-identical (tests/test_sharded_cpu.py runs two gloo
ranks against the single-process oracle).  The Morton re-sort assigns the new indices globally: new index =
(particles of all ranks in Morton blocks before mine) + rank inside the block by (cell key, old index).

Device-resident path (default on GPUs, `ShardedState(..., device_exchange=True)`): the same protocol inside the
library, control plane included -- packing, hole filling and appending by kernels on fixed-size neighbour segments, the
particle counts kept on the device, the decomposition's tables, the cell-size agreement, the global re-sort and the
re-cut behind `gpe_shard_setup` / `gpe_shard_run_scheduled` (csrc/gpe_shard_ctl.hip), so that a host in any language
does what this file does with a handful of calls.  Here that path only chooses who carries the collectives: the
library's own RCCL communicator, or callbacks into torch.distributed.  The torch formulation below stays as the general
path (any displacement, CPU tests with a pluggable engine) and as the specification the kernels are tested against.

Load balance: at re-sort steps the ranks compare their particle counts and, when they have drifted apart (a pile
under gravity), re-cut the rectangles at the particle quantiles of the all-reduced block-column / block-row
histograms and move the particles to their new owners (`ShardedState.rebalance`; `gpe_shard_recut` in the library).
Buffers of a rank that fills up between re-sorts grow by themselves (library side, gpe_shard_step).

The engine behind a rank is pluggable (`engine` argument): `GpeEngine` drives libgpe.so on the rank's GPU;
the CPU tests plug in an oracle-backed engine (tests only) to exercise this file without a GPU.
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib as L

BLOCK = 8                    # cells per ownership block edge == the library's block-table granule


# ------------------------------------------------------------------------------------------------------
# decomposition
# ------------------------------------------------------------------------------------------------------
def _factor(world_size):
    """Px x Py process grid, as square as possible, Px <= Py."""
    px = int(np.floor(np.sqrt(world_size)))
    while world_size % px:
        px -= 1
    return px, world_size // px


class Decomposition:
    """Rectangles of 8x8-cell blocks: owner[block], dest_mask[block] (ranks whose region is within one block)."""

    def __init__(self, world, cell_size, world_size, grid=None, xcuts=None, ycuts=None):
        self.world = (float(world[0]), float(world[1]))
        self.cell_size = np.float32(cell_size)
        # same arithmetic as the library's native_configure: largest home coordinate = floor(world / cell)
        self.gx = int(np.floor(np.float32(world[0]) / self.cell_size)) + 1
        self.gy = int(np.floor(np.float32(world[1]) / self.cell_size)) + 1
        self.bx = (self.gx + BLOCK - 1) // BLOCK
        self.by = (self.gy + BLOCK - 1) // BLOCK
        self.world_size = world_size
        self.px, self.py = grid if grid is not None else _factor(world_size)
        if self.px * self.py != world_size:
            raise ValueError("process grid does not match world_size")
        if world_size > 26:
            raise ValueError("destination masks are 26 bits wide")
        if self.px > self.bx or self.py > self.by:
            raise ValueError("world too small for this many ranks")
        # block columns / rows where the rectangles are cut: equal widths unless given (ShardedState.rebalance)
        self.xcuts = [int(v) for v in xcuts] if xcuts is not None else [round(i * self.bx / self.px) for i in range(self.px + 1)]
        self.ycuts = [int(v) for v in ycuts] if ycuts is not None else [round(j * self.by / self.py) for j in range(self.py + 1)]
        if (len(self.xcuts) != self.px + 1 or len(self.ycuts) != self.py + 1 or self.xcuts[0] != 0 or self.ycuts[0] != 0 or
                self.xcuts[-1] != self.bx or self.ycuts[-1] != self.by or
                any(b <= a for a, b in zip(self.xcuts[:-1], self.xcuts[1:])) or
                any(b <= a for a, b in zip(self.ycuts[:-1], self.ycuts[1:]))):
            raise ValueError("cuts must rise from 0 to the block count")
        col = np.zeros(self.bx, np.int64)
        row = np.zeros(self.by, np.int64)
        for i in range(self.px):
            col[self.xcuts[i]:self.xcuts[i + 1]] = i
        for j in range(self.py):
            row[self.ycuts[j]:self.ycuts[j + 1]] = j
        owner = (row[:, None] * self.px + col[None, :]).astype(np.uint8)           # [by, bx]
        bits = (np.uint32(1) << owner.astype(np.uint32))
        pad = np.zeros((self.by + 2, self.bx + 2), np.uint32)
        pad[1:-1, 1:-1] = bits
        mask = np.zeros_like(bits)
        for dy in range(3):
            for dx in range(3):
                mask |= pad[dy:dy + self.by, dx:dx + self.bx]
        self.owner = np.ascontiguousarray(owner)
        self.dest_mask = np.ascontiguousarray(mask & ~bits)

    def rect_blocks(self, rank):
        i, j = rank % self.px, rank // self.px
        return self.xcuts[i], self.ycuts[j], self.xcuts[i + 1], self.ycuts[j + 1]      # half-open, blocks

    def rect_units(self, rank):
        """World-space rectangle whose particles (by home cell) this rank owns."""
        x0, y0, x1, y1 = self.rect_blocks(rank)
        cs = float(self.cell_size) * BLOCK
        return x0 * cs, y0 * cs, min(x1 * cs, self.world[0]), min(y1 * cs, self.world[1])

    def active_cells(self, rank):
        """Inclusive cell box holding the rank's own blocks plus the one-block ghost ring."""
        x0, y0, x1, y1 = self.rect_blocks(rank)
        return (max(0, (x0 - 1) * BLOCK), max(0, (y0 - 1) * BLOCK),
                min(self.gx - 1, (x1 + 1) * BLOCK - 1), min(self.gy - 1, (y1 + 1) * BLOCK - 1))

    def owner_of(self, pos):
        """Owner rank of host positions (numpy, same float32 arithmetic as the kernels)."""
        pos = np.asarray(pos, np.float32).reshape(-1, 2)
        cx = np.floor(pos[:, 0] / self.cell_size).astype(np.int64) >> 3
        cy = np.floor(pos[:, 1] / self.cell_size).astype(np.int64) >> 3
        cx = np.clip(cx, 0, self.bx - 1)
        cy = np.clip(cy, 0, self.by - 1)
        return self.owner[cy, cx].astype(np.int64)

    def neighbours(self, rank):
        """Ranks whose rectangle lies within one block of `rank`'s (ascending)."""
        x0, y0, x1, y1 = self.rect_blocks(rank)
        m = int(np.bitwise_or.reduce(self.dest_mask[y0:y1, x0:x1].reshape(-1))) if x1 > x0 and y1 > y0 else 0
        return [p for p in range(self.world_size) if (m >> p) & 1 and p != rank]

    def border_blocks(self, src, dst):
        """How many of src's blocks lie within one block of dst's rectangle (their particles are dst's ghosts)."""
        x0, y0, x1, y1 = self.rect_blocks(src)
        return int(((self.dest_mask[y0:y1, x0:x1] >> np.uint32(dst)) & 1).sum())

    def segment_caps(self, src, dst, per_block, scale=1.0):
        """(migrant rows, ghost rows) of the segment src -> dst of the device-resident exchange: three times the
        mean population of src's blocks bordering dst, plus slack.  A pure function of the decomposition and of
        `per_block`, so the sender and the receiver size the segment alike."""
        gho = int((self.border_blocks(src, dst) * per_block * 3.0 + 2048) * scale)
        return gho // 4 + int(512 * scale) + 1, gho + 1

    @staticmethod
    def segment_words(cap_mig, cap_gho):
        return 4 + 6 * cap_mig + 4 * cap_gho                    # header, migrant rows, ghost rows (u32 words)

    def min_region_blocks(self):
        return min(min(b - a for a, b in zip(self.xcuts[:-1], self.xcuts[1:])),
                   min(b - a for a, b in zip(self.ycuts[:-1], self.ycuts[1:])))

    def morton_entries(self):
        def split(n):
            x = n & 0xFFFF
            x = (x | (x << 8)) & 0x00FF00FF
            x = (x | (x << 4)) & 0x0F0F0F0F
            x = (x | (x << 2)) & 0x33333333
            x = (x | (x << 1)) & 0x55555555
            return x
        max_key = split(self.gx - 1) | (split(self.gy - 1) << 1)
        return (max_key >> 6) + 1


def quantile_cuts(hist, parts, min_width=2):
    """Cut `len(hist)` block columns (rows) into `parts` runs of about equal particle count, every run at least
    `min_width` blocks wide (the device-resident exchange needs two).  Pure function of its arguments: every rank
    derives the same cuts from the all-reduced histogram."""
    hist = np.asarray(hist, np.float64)
    nb = len(hist)
    if parts * min_width > nb:
        min_width = max(1, nb // parts)
    c = np.cumsum(hist)
    total = c[-1] if nb else 0.0
    cuts = [0]
    for i in range(1, parts):
        cut = int(np.searchsorted(c, total * i / parts, side="left")) + 1 if total > 0 else round(i * nb / parts)
        cut = max(cut, cuts[-1] + min_width)
        cut = min(cut, nb - (parts - i) * min_width)
        cuts.append(cut)
    cuts.append(nb)
    return cuts


# ------------------------------------------------------------------------------------------------------
# GPU engine (libgpe.so)
# ------------------------------------------------------------------------------------------------------
class _DevArray:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2}


class GpeEngine:
    """One rank's particles inside a gpe context (native mode, order keys on)."""

    def __init__(self, pos, rad, gid, world, gravity=(0.0, 0.0), device=0, capacity=None, profiling=False, flags=0, prev=None):
        from .engine import Context
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.ctx = Context(world=world, gravity=gravity, mode=L.MODE_NATIVE, device=device, profiling=profiling, flags=flags)
        # The context runs on a stream torch owns (torch never destroys its pool streams): torch's caching allocators
        # and ProcessGroupNCCL remember the stream of every buffer they handle past the context's lifetime, so the
        # stream must outlive the context -- lend it one instead of borrowing the library's (gpe_set_stream).
        self.stream = torch.cuda.Stream(device=self.device)
        self.ctx.call("gpe_set_stream", C.c_void_p(self.stream.cuda_stream))
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 2)
        rad = np.ascontiguousarray(rad, np.float32)
        n = pos.shape[0]
        if n == 0:
            raise ValueError("every rank needs at least one particle at start")
        if prev is not None:
            prev = np.ascontiguousarray(prev, np.float32).reshape(-1, 2)
        self.ctx.call("gpe_set_particles", pos.ctypes.data_as(C.c_void_p),
                      prev.ctypes.data_as(C.c_void_p) if prev is not None else None, rad.ctypes.data_as(C.c_void_p), n)
        self.ctx.call("gpe_use_order_keys", 1)
        self.reserve(capacity or int(n * 1.3) + 4096)
        self.n_owned = n
        with torch.cuda.stream(self.stream):
            self.arrays()["gid"][:n] = torch.as_tensor(np.ascontiguousarray(gid, np.int64), device=self.device).to(torch.int32)
        self.ctx.sync()

    # -- storage ------------------------------------------------------------------------------------
    def capacity(self):
        cap = C.c_uint64()
        self.ctx.call("gpe_capacity", C.byref(cap))
        return cap.value

    def reserve(self, capacity):
        self.ctx.call("gpe_reserve", int(capacity))

    def _view(self, what, shape, typestr):
        ptr, nbytes = C.c_void_p(), C.c_uint64()
        self.ctx.call("gpe_device_ptr", what, C.byref(ptr), C.byref(nbytes))
        return torch.as_tensor(_DevArray(ptr.value, shape, typestr), device=self.device)

    def arrays(self):
        """Views over the context's buffers at full capacity (they move at every step / reserve / resort)."""
        cap = self.capacity()
        return {"pos": self._view(L.POS, (cap, 2), "<f4"), "prev": self._view(L.PREV, (cap, 2), "<f4"),
                "radius": self._view(L.RADIUS, (cap,), "<f4"), "gid": self._view(L.ORDER_KEYS, (cap,), "<i4")}

    def set_counts(self, n_total, n_owned):
        self.ctx.call("gpe_set_counts", int(n_total), int(n_owned))

    # -- per step ------------------------------------------------------------------------------------
    def classify(self, dec, rank, tables):
        owner_t, mask_t, out_idx, out_info, out_cnt = tables
        out_cnt.zero_()
        self.ctx.call("gpe_shard_classify", owner_t.data_ptr(), mask_t.data_ptr(), dec.bx, dec.by, rank,
                      out_idx.data_ptr(), out_info.data_ptr(), out_cnt.data_ptr(), out_idx.numel())
        k = int(out_cnt[0].item())                   # the one host sync of the exchange
        if k > out_idx.numel():
            raise RuntimeError("shard classify overflow: %d boundary particles, capacity %d" % (k, out_idx.numel()))
        return out_idx[:k].long(), out_info[:k]

    def make_tables(self, dec, cap):
        dev = self.device
        return (torch.as_tensor(dec.owner.reshape(-1), device=dev), torch.as_tensor(dec.dest_mask.reshape(-1).astype(np.int32), device=dev),
                torch.empty(cap, dtype=torch.int32, device=dev), torch.empty(cap, dtype=torch.int32, device=dev),
                torch.zeros(4, dtype=torch.int32, device=dev))

    def step(self, dt):
        self.ctx.call("gpe_step", float(dt), 0)

    def shard_peek(self):
        no, nt = C.c_uint64(), C.c_uint64()
        self.ctx.call("gpe_shard_peek", C.byref(no), C.byref(nt))
        return no.value, nt.value

    def shard_counts(self, leave=False):
        no, nt = C.c_uint64(), C.c_uint64()
        self.ctx.call("gpe_shard_counts", C.byref(no), C.byref(nt), 1 if leave else 0)
        return no.value, nt.value

    def morton_resort(self):
        """K1 + stable sort by home-cell key + K4 on the owned particles (ties keep the current order).
        Returns (sorted keys, permutation) as int64 tensors."""
        self.ctx.call("gpe_morton_resort")
        n = self.n_owned
        keys = self._view(L.HOME_CELL_IDS, (n,), "<i4").long() & 0xFFFFFFFF
        perm = self._view(L.PARTICLE_IDS, (n,), "<i4").long() & 0xFFFFFFFF
        return keys, perm

    def argsort_u32(self, keys):
        """Permutation that sorts `keys` (a device int32 tensor read as unsigned) ascending, stable: the library's own
        radix sort (gpe_sort_pairs_u32) on the engine's stream.  (torch.argsort loads its sort kernels on first use --
        seconds, and minutes when several ranks on one box do it at once.)"""
        n = int(keys.numel())
        k = keys.contiguous().clone()
        v = torch.arange(n, dtype=torch.int32, device=keys.device)
        self.ctx.call("gpe_sort_pairs_u32", C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()), n)
        return v.long()

    def set_active_cells(self, box):
        self.ctx.call("gpe_set_active_cells", *[int(v) for v in box])

    def refresh(self):
        self.ctx.call("gpe_refresh")

    def sync(self):
        self.ctx.sync()

    def stream_ctx(self):
        return torch.cuda.stream(self.stream)

    def close(self):
        """Everything enqueued on the lent stream (the library's kernels, torch's copies, RCCL) has finished before
        the context frees its buffers; the stream itself stays torch's."""
        if getattr(self.ctx, "h", None):
            torch.cuda.synchronize(self.device)
            self.ctx.close()


# ------------------------------------------------------------------------------------------------------
# the sharded state
# ------------------------------------------------------------------------------------------------------
class ShardedState:
    """`State` for one rank of a sharded run.  `engine` holds this rank's owned particles (first n_owned slots).

    device_exchange (default on GPUs): everything is inside the library -- the decomposition's tables and segment
    buffers, the cell-size agreement, the per-step exchange, the global re-sort and the re-cut (csrc/gpe_shard_ctl.hip,
    `gpe_shard_setup` / `gpe_shard_run_scheduled`); this class only says who carries the collectives: the library's own
    RCCL communicator (process group nccl), or callbacks into torch.distributed (gloo rehearsals on a one-GPU box,
    staged through host memory).  Otherwise the torch formulation below runs (any displacement, CPU tests)."""

    def __init__(self, engine, dec, rank, group=None, device_exchange=None, rebalance=1.25):
        self.e, self.dec, self.rank, self.group = engine, dec, rank, group
        self.rebalance_above = rebalance       # re-cut at re-sort steps when max/mean owned exceeds it (None: never)
        self.ws = dec.world_size
        self.n_owned = engine.n_owned
        self.n_ghost = 0
        self.stats = {"migrants": 0, "ghosts": 0, "steps": 0}
        # gloo moves host tensors only: a GPU engine under a gloo group (tests on a one-GPU box) stages on the host;
        # with the nccl (= RCCL) backend the device buffers go straight to send/recv
        self.stage_cpu = (self.ws > 1 and engine.device.type == "cuda" and dist.get_backend(group) == "gloo")
        # device-resident exchange (csrc/k_shard.hip): needs the GPU engine, every rectangle >= 2 blocks wide
        # (a particle that leaves a rank then only concerns that rank's own neighbours) and <= 8 neighbours
        if device_exchange is None:
            device_exchange = os.environ.get("GPE_SHARD_EXCHANGE", "device") != "torch"
        self.fast = bool(device_exchange and self.ws > 1 and isinstance(engine, GpeEngine) and
                         dec.min_region_blocks() >= 2 and len(dec.neighbours(rank)) <= 8)
        self.transport = None
        if self.fast:
            self._setup_control_plane()
        else:
            self._agree_on_cell_size()
            self.tables = engine.make_tables(dec, max(1 << 16, engine.capacity() // 4))
            engine.set_active_cells(dec.active_cells(rank))

    def _agree_on_cell_size(self):
        """The cell size is 2.2 x the largest radius of the WHOLE system (grid.rs:159-161); a rank's context only
        saw its own particles.  Every rank takes the maximum over the ranks (Grid::new's max_obj_radius) and checks
        that the decomposition was cut with that cell size -- a rank on a different grid would exchange nonsense.
        (The device-resident path does the same inside gpe_shard_setup.)"""
        e = self.e
        if not isinstance(e, GpeEngine):
            return
        r = C.c_float(0.0)
        e.ctx.call("gpe_max_radius", C.byref(r))
        m = torch.tensor([abs(float(r.value))], dtype=torch.float64)
        if self.ws > 1:
            if dist.get_backend(self.group) == "nccl":
                m = m.to(e.device)
            dist.all_reduce(m, op=dist.ReduceOp.MAX, group=self.group)
        gmax = float(np.float32(m.item()))
        if gmax != abs(float(r.value)):
            e.ctx.call("gpe_grid_set_max_radius", gmax)
        cs = C.c_float(0.0)
        e.ctx.call("gpe_cell_size", C.byref(cs))
        if np.float32(cs.value) != np.float32(self.dec.cell_size):
            raise ValueError("sharded run: the decomposition was cut with cell size %r, the system's is %r "
                             "(2.2 x the largest radius over all ranks)" % (float(self.dec.cell_size), float(cs.value)))

    # -- the control plane inside the library ------------------------------------------------------------------
    def _setup_control_plane(self):
        e, dec = self.e, self.dec
        lay = L.GpeShardLayout()
        xc = (C.c_int32 * (dec.px + 1))(*dec.xcuts)
        yc = (C.c_int32 * (dec.py + 1))(*dec.ycuts)
        L.check(L.load().gpe_shard_layout_build(dec.world[0], dec.world[1], float(dec.cell_size), self.ws, dec.px, dec.py,
                                                xc, yc, C.byref(lay)))
        self._install_collectives()
        scale = float(os.environ.get("GPE_SHARD_CAP_SCALE", "1"))            # tests shrink the segments to see the error
        try:
            e.ctx.call("gpe_shard_setup", C.byref(lay), self.rank, scale)
        except L.GpeError as exc:
            self._raise_transport_error()
            if "disagree on the segment sizes" in str(exc) or "cell size" in str(exc):
                raise ValueError(str(exc)) from exc
            raise

    def _install_collectives(self):
        """Who carries the control plane's collectives and the per-step segment exchange:
        "rccl"  -- the library itself: a communicator of its own (gpe_shard_comm_init; the 128-byte id travels over
                   torch.distributed once).  The whole run then has no Python in it.  Default when the process group is nccl.
        "torch" -- gpe_shard_set_collectives with callbacks into torch.distributed (all_reduce / all_to_all_single): gloo
                   rehearsals on a one-GPU box, staged through host memory.
        A communicator that cannot be set up (every rank sees the same verdict) leaves the run on the callbacks."""
        import sys
        e = self.e
        want = os.environ.get("GPE_SHARD_TRANSPORT",
                              "rccl" if (not self.stage_cpu and dist.get_backend(self.group) == "nccl") else "torch")
        if want == "rccl":
            try:
                # every rank takes part in the broadcast whatever happened on rank 0: a failure there travels as an
                # all-zero id, so that all ranks leave this transport together instead of waiting for each other
                ident = torch.zeros(L.COMM_ID_BYTES, dtype=torch.uint8)
                if dist.get_rank(self.group) == 0:
                    raw = (C.c_uint8 * L.COMM_ID_BYTES)()
                    if L.load().gpe_comm_unique_id(raw) == L.GPE_OK:
                        ident = torch.tensor(list(raw), dtype=torch.uint8)
                comm_dev = e.device if dist.get_backend(self.group) == "nccl" else torch.device("cpu")
                ident = ident.to(comm_dev)
                src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
                dist.broadcast(ident, src=src, group=self.group)
                ident = ident.cpu()
                if not bool(ident.any()):
                    raise RuntimeError("rank 0 could not create an RCCL id: %s" % (L.load().gpe_last_error(None) or b"").decode())
                raw = (C.c_uint8 * L.COMM_ID_BYTES)(*ident.tolist())
                torch.cuda.synchronize(e.device)
                # The choice must be the SAME on every rank (a rank that left for the callbacks while its peers sit in
                # ncclRecv would hang the job at the first step): every rank reports how its own communicator set-up
                # went, and the minimum decides.
                ok, why = 1, None
                try:
                    e.ctx.call("gpe_shard_comm_init", raw, self.rank, self.ws)      # collective
                    e.sync()
                except Exception as exc:                               # noqa: BLE001 -- any backend error
                    ok, why = 0, exc
                flag = torch.tensor([ok], dtype=torch.int32, device=comm_dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
                if int(flag.item()) == 1:
                    self.transport = "rccl"
                    return
                try:
                    e.ctx.call("gpe_shard_comm_destroy")
                except Exception:                                      # noqa: BLE001
                    pass
                raise RuntimeError("a rank could not set the communicator up" + (": %s" % why if why else " (not this one)"))
            except Exception as exc:                                   # noqa: BLE001 -- any backend error
                print("[gpe sharded] in-library RCCL transport unavailable (%s: %s); carrying the collectives with "
                      "torch.distributed" % (type(exc).__name__, exc), file=sys.stderr, flush=True)
        self._transport_error = None
        coll = L.GpeShardCollectives()
        coll.struct_size = C.sizeof(L.GpeShardCollectives)
        self._cb = (L.ALL_REDUCE_FN(self._cb_all_reduce), L.ALL_TO_ALL_FN(self._cb_all_to_all))   # keep the thunks alive
        coll.all_reduce_u32, coll.all_to_all_u32 = self._cb
        e.ctx.call("gpe_shard_set_collectives", C.byref(coll))
        self.transport = "torch"

    def _dev_words(self, ptr, count):
        return torch.as_tensor(_DevArray(ptr, (int(count),), "<i4"), device=self.e.device)

    def _cb_all_reduce(self, user, d_buf, count, op, stream):
        """gpe_shard_collectives.all_reduce_u32 over torch.distributed (u32 words travel as int32: a sum wraps the same
        way; a maximum is taken on the values read as unsigned)."""
        try:
            with self.e.stream_ctx():
                buf = self._dev_words(d_buf, count)
                comm_cpu = self.stage_cpu or dist.get_backend(self.group) != "nccl"
                t = buf.cpu() if comm_cpu else buf
                if op == L.REDUCE_MAX:
                    w = t.to(torch.int64) & 0xFFFFFFFF
                    dist.all_reduce(w, op=dist.ReduceOp.MAX, group=self.group)
                    t = w.to(torch.int32)               # (values above 2^31 wrap into the negative half: the same bits)
                else:
                    dist.all_reduce(t, group=self.group)
                buf.copy_(t)
                self.e.stream.synchronize()
            return 0
        except Exception as exc:                                       # noqa: BLE001 -- must not unwind into C
            self._transport_error = exc
            return 1

    def _cb_all_to_all(self, user, d_send, send_off, send_cnt, d_recv, recv_off, recv_cnt, stream):
        """gpe_shard_collectives.all_to_all_u32: one all_to_all_single (RCCL: a send / recv pair per peer in one group)."""
        try:
            ws = self.ws
            so, sc = [int(send_off[p]) for p in range(ws)], [int(send_cnt[p]) for p in range(ws)]
            ro, rc = [int(recv_off[p]) for p in range(ws)], [int(recv_cnt[p]) for p in range(ws)]
            with self.e.stream_ctx():
                comm_cpu = self.stage_cpu or dist.get_backend(self.group) != "nccl"
                dev = torch.device("cpu") if comm_cpu else self.e.device
                send = self._dev_words(d_send, max(1, max(o + n for o, n in zip(so, sc))))
                recv = self._dev_words(d_recv, max(1, max(o + n for o, n in zip(ro, rc))))
                # all_to_all_single wants the pieces back to back in rank order
                pieces = [send[so[p]:so[p] + sc[p]] for p in range(ws)]
                flat = torch.cat(pieces).to(dev) if sum(sc) else torch.empty(0, dtype=torch.int32, device=dev)
                got = torch.empty(sum(rc), dtype=torch.int32, device=dev)
                if comm_cpu:
                    self.e.stream.synchronize()
                dist.all_to_all_single(got, flat, rc, sc, group=self.group)
                o = 0
                for p in range(ws):
                    if rc[p]:
                        recv[ro[p]:ro[p] + rc[p]].copy_(got[o:o + rc[p]])
                    o += rc[p]
                self.e.stream.synchronize()
            return 0
        except Exception as exc:                                       # noqa: BLE001 -- must not unwind into C
            self._transport_error = exc
            return 1

    def _raise_transport_error(self):
        exc = getattr(self, "_transport_error", None)
        if exc is not None:
            self._transport_error = None
            raise exc

    def _shard_stats(self):
        st = L.GpeShardStats()
        st.struct_size = C.sizeof(L.GpeShardStats)
        self.e.ctx.call("gpe_shard_get_stats", C.byref(st))
        return st

    def _run_scheduled(self, dt, steps, resort_every, resort_first):
        """State::update `steps` times inside the library: re-sorts (with re-cut) where the schedule says, gpe_shard_run
        in between."""
        try:
            self.e.ctx.call("gpe_shard_run_scheduled", float(dt), int(steps), int(resort_every), 1 if resort_first else 0)
        except L.GpeError:
            self._raise_transport_error()
            raise
        st = self._shard_stats()
        self.stats["recuts"] = int(st.recuts)
        self.n_owned = self.e.n_owned = int(st.n_owned)

    # -- helpers -------------------------------------------------------------------------------------
    def _ensure_capacity(self, need):
        if need > self.e.capacity():
            self.e.reserve(int(need * 1.25) + 4096)
            self.e.set_counts(max(self.n_owned, 1), self.n_owned)

    def _gather_counts(self, mine):
        """mine: int64 [ws, 2] (migrants, ghosts) this rank sends to each peer -> [src, dst, 2] of all."""
        if self.ws == 1:
            return mine[None]
        if self.stage_cpu:
            mine = mine.cpu()
        out = [torch.empty_like(mine) for _ in range(self.ws)]
        dist.all_gather(out, mine, group=self.group)
        return torch.stack(out)

    def _exchange(self, send_bufs, counts):
        """send_bufs[peer] = float32 payload or None; counts = [src, dst, 2] host tensor."""
        recv_bufs, ops = {}, []
        comm_dev = torch.device("cpu") if self.stage_cpu else self.e.device
        for peer in range(self.ws):
            if peer == self.rank:
                continue
            nm, ng = int(counts[peer, self.rank, 0]), int(counts[peer, self.rank, 1])
            if nm + ng:
                recv_bufs[peer] = torch.empty(nm * 6 + ng * 4, dtype=torch.float32, device=comm_dev)
                ops.append(dist.P2POp(dist.irecv, recv_bufs[peer], peer, group=self.group))
            if send_bufs.get(peer) is not None:
                ops.append(dist.P2POp(dist.isend, send_bufs[peer].to(comm_dev), peer, group=self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        if self.stage_cpu:
            recv_bufs = {p: b.to(self.e.device) for p, b in recv_bufs.items()}
        return recv_bufs

    # -- one step --------------------------------------------------------------------------------------
    def exchange(self):
        """Migrate particles that left this rank's blocks, then (re)build the ghost layer.
        Written to need two host syncs only (the classify count and the count matrix): no boolean-mask
        indexing -- the (peer, kind) groups come from one sort, their sizes from the gathered counts."""
        e, rank, ws = self.e, self.rank, self.ws
        dev = e.device
        with e.stream_ctx():
            a = e.arrays()
            e.set_counts(max(self.n_owned, 1), self.n_owned)               # ghosts of the last step are gone
            idx, info = e.classify(self.dec, rank, self.tables)            # host sync 1: k boundary particles
            k = int(idx.numel())
            peers = torch.arange(ws, device=dev, dtype=info.dtype)
            owner1 = (info >> 26) & 31                                     # 1 + new owner, 0 = stays here
            mask = info & 0x03FFFFFF                                       # ranks bordering its current block
            # kind 0: migrant to `peer`; kind 1: ghost for `peer` (a migrant is a ghost for every bordering rank
            # except its new owner, including this rank itself: it stays behind as one of its own ghosts)
            mig = owner1[:, None] == peers[None, :] + 1                                        # [k, ws]
            gho = (((mask[:, None] >> peers[None, :]) & 1) != 0) & ~mig                        # [k, ws]
            sel = torch.stack([mig, gho], 0)                                                   # [2, k, ws]
            mine = sel.sum(1).to(torch.int64).t().contiguous()                                 # [ws, 2]
            counts = self._gather_counts(mine).cpu()                       # host sync 2: [src, dst, 2]
            cm = [int(counts[rank, p, 0]) for p in range(ws)]              # migrants I send to p
            cg = [int(counts[rank, p, 1]) for p in range(ws)]              # ghosts I send to p (p == rank: local)
            total = sum(cm) + sum(cg)
            send = {}
            local_ghosts = None
            n_mig = sum(cm)
            holes = None
            if total:
                # group the selected (kind, peer, entry) triples: key = kind * ws + peer, entries in index order
                key = torch.arange(2 * ws, device=dev).view(2, 1, ws).expand(2, k, ws)
                key = torch.where(sel, key, torch.full_like(key, 4 * ws)).permute(0, 2, 1).reshape(-1)   # [2*ws*k]
                order = torch.argsort(key, stable=True)[:total]
                src = idx[order % k]                                       # local particle of every triple
                gid_f = a["gid"].view(torch.float32)
                rows = torch.cat([a["pos"][src], a["prev"][src], a["radius"][src, None], gid_f[src, None]], 1)
                holes = src[:n_mig]
                off_m = np.concatenate([[0], np.cumsum(cm)])
                off_g = n_mig + np.concatenate([[0], np.cumsum(cg)])
                for p in range(ws):
                    g_rows = rows[off_g[p]:off_g[p + 1]][:, [0, 1, 4, 5]]
                    if p == rank:
                        local_ghosts = g_rows.clone()
                        continue
                    if cm[p] + cg[p]:
                        send[p] = torch.cat([rows[off_m[p]:off_m[p + 1]].reshape(-1), g_rows.reshape(-1)]).contiguous()
            recv = self._exchange(send, counts)

            n_in = sum(int(counts[p, rank, 0]) for p in range(ws) if p != rank)
            n_gh = sum(int(counts[p, rank, 1]) for p in range(ws) if p != rank) + cg[rank]
            self._ensure_capacity(self.n_owned - n_mig + n_in + n_gh + 2)
            a = e.arrays()
            # drop the migrants: fill their holes from the tail (order is free: members sort by order key).
            # Fixed-length formulation (no sync): the i-th hole in front of the tail takes the i-th tail survivor;
            # the unused pairs copy a spare slot onto itself.
            if n_mig:
                n0 = self.n_owned
                spare = e.capacity() - 1
                hs = torch.sort(holes).values
                tail = torch.arange(n0 - n_mig, n0, device=dev)
                tail_is_hole = torch.isin(tail, hs)
                survivors = tail[torch.argsort(tail_is_hole.to(torch.int8), stable=True)]
                n_move = (~tail_is_hole).sum()
                lanes = torch.arange(n_mig, device=dev)
                dst = torch.where((hs < n0 - n_mig) & (lanes < n_move), hs, torch.full_like(hs, spare))
                srcm = torch.where(lanes < n_move, survivors, torch.full_like(survivors, spare))
                for name in ("pos", "prev", "radius", "gid"):
                    a[name][dst] = a[name][srcm]
                self.n_owned = n0 - n_mig
            # arrivals: migrants become owned, ghosts follow all owned particles
            o = self.n_owned
            for p in sorted(recv):
                c = int(counts[p, rank, 0])
                if c:
                    r6 = recv[p][:c * 6].view(c, 6)
                    a["pos"][o:o + c] = r6[:, 0:2]
                    a["prev"][o:o + c] = r6[:, 2:4]
                    a["radius"][o:o + c] = r6[:, 4]
                    a["gid"][o:o + c] = r6[:, 5].contiguous().view(torch.int32)
                    o += c
            self.n_owned = o
            for p in sorted(recv):
                c0, c = int(counts[p, rank, 0]), int(counts[p, rank, 1])
                if c:
                    r4 = recv[p][c0 * 6:c0 * 6 + c * 4].view(c, 4)
                    a["pos"][o:o + c] = r4[:, 0:2]
                    a["radius"][o:o + c] = r4[:, 2]
                    a["gid"][o:o + c] = r4[:, 3].contiguous().view(torch.int32)
                    o += c
            if local_ghosts is not None and cg[rank]:
                c = cg[rank]
                a["pos"][o:o + c] = local_ghosts[:, 0:2]
                a["radius"][o:o + c] = local_ghosts[:, 2]
                a["gid"][o:o + c] = local_ghosts[:, 3].contiguous().view(torch.int32)
                o += c
            self.n_ghost = o - self.n_owned
            if self.n_owned == 0:
                raise RuntimeError("rank %d owns no particle any more (unsupported)" % rank)
            e.n_owned = self.n_owned
            e.set_counts(self.n_owned + self.n_ghost, self.n_owned)
            self.stats["migrants"] += n_mig
            self.stats["ghosts"] += self.n_ghost

    # -- load balance ---------------------------------------------------------------------------------------
    def _all_reduce_host(self, t):
        """Sum a small tensor over the ranks; returns it on the host."""
        if self.ws > 1:
            if dist.get_backend(self.group) == "nccl":
                t = t.to(self.e.device)
                dist.all_reduce(t, group=self.group)
            else:
                t = t.cpu()
                dist.all_reduce(t, group=self.group)
        return t.cpu()

    def rebalance(self):
        """Re-cut the rectangles so that every rank owns about the same number of particles, and move the particles
        to their new owners.  Collective; called at re-sort steps, when every particle sits on its owner and the
        ghosts are dropped.  The result of the run does not depend on the cuts (order keys + ghost band), so a
        re-cut run stays bit-identical to the single-device run.  Returns True when the cuts changed."""
        if self.rebalance_above is None or self.ws == 1:
            return False
        e, dec, rank, ws = self.e, self.dec, self.rank, self.ws
        n = self.n_owned
        with e.stream_ctx():
            mine = torch.zeros(ws, dtype=torch.int64)
            mine[rank] = n
            owned = self._all_reduce_host(mine).numpy()
            if owned.max() <= self.rebalance_above * owned.mean():
                return False
            a = e.arrays()
            pos = a["pos"][:n]
            cs = float(dec.cell_size)
            # same float32 arithmetic as the kernels: floor(pos / cell_size) >> 3
            bxi = torch.clamp(torch.floor(pos[:, 0] / np.float32(cs)).long() >> 3, 0, dec.bx - 1)
            byi = torch.clamp(torch.floor(pos[:, 1] / np.float32(cs)).long() >> 3, 0, dec.by - 1)
            hx = self._all_reduce_host(torch.bincount(bxi, minlength=dec.bx)).numpy()
            hy = self._all_reduce_host(torch.bincount(byi, minlength=dec.by)).numpy()
            xcuts, ycuts = quantile_cuts(hx, dec.px), quantile_cuts(hy, dec.py)
            if xcuts == dec.xcuts and ycuts == dec.ycuts:
                return False
            new = Decomposition(dec.world, dec.cell_size, ws, grid=(dec.px, dec.py), xcuts=xcuts, ycuts=ycuts)
            # new owner of every particle, particles grouped by it
            owner_t = torch.as_tensor(new.owner.astype(np.int64), device=pos.device)
            dest = owner_t[byi, bxi]
            order = torch.argsort(dest, stable=True)
            send_counts = torch.bincount(dest, minlength=ws).cpu()
            gid_f = a["gid"][:n].view(torch.float32)
            rows = torch.cat([a["pos"][:n], a["prev"][:n], a["radius"][:n, None], gid_f[:, None]], 1)[order].contiguous()
            # counts matrix, then the rows themselves: one all_to_all each
            comm_cpu = self.stage_cpu or e.device.type == "cpu"
            sc = send_counts if comm_cpu else send_counts.to(e.device)
            rc = torch.empty_like(sc)
            dist.all_to_all_single(rc, sc, group=self.group)
            rc = rc.cpu()
            n_new = int(rc.sum())
            send = rows.cpu() if self.stage_cpu else rows
            recv = torch.empty((n_new, 6), dtype=torch.float32, device=send.device)
            dist.all_to_all_single(recv, send, [int(v) for v in rc], [int(v) for v in send_counts], group=self.group)
            if self.stage_cpu:
                recv = recv.to(e.device)
            if n_new == 0:
                raise RuntimeError("rank %d owns no particle after the re-cut (unsupported)" % rank)
            self._ensure_capacity(int(n_new * 1.3) + 4096)
            a = e.arrays()
            a["pos"][:n_new] = recv[:, 0:2]
            a["prev"][:n_new] = recv[:, 2:4]
            a["radius"][:n_new] = recv[:, 4]
            a["gid"][:n_new] = recv[:, 5].contiguous().view(torch.int32)
            self.n_owned, self.n_ghost = n_new, 0
            e.n_owned = n_new
            e.set_counts(n_new, n_new)
        self.dec = new
        self.tables = e.make_tables(new, max(1 << 16, e.capacity() // 4))
        e.set_active_cells(new.active_cells(rank))
        self.stats["recuts"] = self.stats.get("recuts", 0) + 1
        return True

    def resort(self):
        """The reference's Morton re-sort (particle_sort.rs:58-69) with GLOBAL new indices."""
        e = self.e
        with e.stream_ctx():
            n = self.n_owned
            e.set_counts(n, n)
            a = e.arrays()
            order = e.argsort_u32(a["gid"][:n])                           # ties of the key sort = old index order
            for name in ("pos", "prev", "radius", "gid"):
                a[name][:n] = a[name][:n][order]
            e.n_owned = n
            keys, _ = e.morton_resort()                                    # arrays now sorted by (key, old index)
            entries = self.dec.morton_entries()
            mb = keys >> 6
            local = torch.bincount(mb, minlength=entries)
            total = local.clone()
            if self.ws > 1:
                if self.stage_cpu:
                    t = total.cpu()
                    dist.all_reduce(t, group=self.group)
                    total = t.to(e.device)
                else:
                    dist.all_reduce(total, group=self.group)
            base = torch.cumsum(total, 0) - total                          # particles of all ranks in earlier blocks
            first = torch.cumsum(local, 0) - local                         # first local position of each block
            new_gid = base[mb] + (torch.arange(n, device=e.device) - first[mb])
            a = e.arrays()
            a["gid"][:n] = new_gid.to(torch.int32)

    def update(self, dt, resort=False):
        """state.rs:115-131 for one rank: [re-sort] -> exchange -> collide (owned + ghosts) -> integrate owned."""
        if self.fast:
            self._run_scheduled(dt, 1, 0, resort)
            self.stats["steps"] += 1
            return
        if resort:
            # every Morton block's particles must sit on their owner before indices are assigned
            if self.ws > 1:
                self.exchange()
                self.e.set_counts(self.n_owned, self.n_owned)             # ghosts dropped
                self.rebalance()
            self.resort()
        if self.ws > 1:
            self.exchange()
        else:
            self.e.set_counts(self.n_owned, self.n_owned)
        self.e.step(dt)
        self.stats["steps"] += 1

    def run(self, dt, steps, resort_every=0, resort_first=True):
        """state.rs:115-131 `steps` times; re-sorts on the first step (resort_first) and every resort_every steps.
        With the device-resident exchange the whole schedule is ONE library call (gpe_shard_run_scheduled)."""
        if self.fast:
            self._run_scheduled(dt, steps, resort_every, resort_first)
            self.stats["steps"] += steps
            return
        for s in range(steps):
            self.update(dt, resort=bool((s == 0 and resort_first) or (resort_every and s > 0 and s % resort_every == 0)))

    def owned(self):
        """(gid, pos, prev) of the owned particles as host arrays."""
        if self.fast:
            cap = self.e.capacity()
            gid = np.empty(cap, np.uint32)
            pos, prev = np.empty((cap, 2), np.float32), np.empty((cap, 2), np.float32)
            no = C.c_uint64()
            self.e.ctx.call("gpe_shard_download_owned", gid.ctypes.data_as(C.c_void_p), pos.ctypes.data_as(C.c_void_p),
                            prev.ctypes.data_as(C.c_void_p), cap, C.byref(no))
            n = int(no.value)
            self.n_owned = self.e.n_owned = n
            self.n_ghost = int(self._shard_stats().n_ghost)
            return gid[:n].astype(np.int64), pos[:n].copy(), prev[:n].copy()
        self.e.sync()
        with self.e.stream_ctx():
            a = self.e.arrays()
            n = self.n_owned
            out = (a["gid"][:n].cpu().numpy().astype(np.int64), a["pos"][:n].cpu().numpy(), a["prev"][:n].cpu().numpy())
        return out
