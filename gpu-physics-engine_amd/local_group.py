"""Several contexts of ONE process as the ranks of a sharded run (gpe_local_group_*, csrc/gpe_group.hip): one host
thread per context, segments and collectives moved between the contexts' buffers inside the library.

The reference owns one device from one process (renderer/wgpu_context.rs:42-49); this is the smallest step from there to
a multi-GPU node -- no launcher, no torch, no collective library.  Every call below is a thin ctypes call into libgpe.so
(the GIL is released while a rank waits in a collective, so plain Python threads do).  Also how the tests put more
ranks on one GPU than the box allows processes (8 x 100 M particles = BASELINE.json configs[4]'s workload).
"""
import ctypes as C
import threading

import numpy as np

from . import _lib as L
from .engine import Context, _ptr


def build_layout(world, cell_size, world_size, grid=None, xcuts=None, ycuts=None):
    """gpe_shard_layout_build: the world cut into px x py rectangles of 8x8-cell blocks."""
    lay = L.GpeShardLayout()
    px, py = grid if grid is not None else (0, 0)
    xc = (C.c_int32 * len(xcuts))(*xcuts) if xcuts is not None else None
    yc = (C.c_int32 * len(ycuts))(*ycuts) if ycuts is not None else None
    L.check(L.load().gpe_shard_layout_build(float(world[0]), float(world[1]), float(cell_size), int(world_size), px, py, xc, yc,
                                            C.byref(lay)))
    return lay


def owner_of(layout, pos):
    """gpe_shard_layout_owner_of: owner rank of each host position (the kernels' own f32 arithmetic)."""
    pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 2)
    out = np.empty(len(pos), np.uint8)
    L.check(L.load().gpe_shard_layout_owner_of(C.byref(layout), _ptr(pos), len(pos), _ptr(out)))
    return out


class LocalShardedRun:
    """`world_size` contexts on `devices` (default: all on device 0), each owning the particles of its rectangle.

    run(dt, steps, resort_every, resort_first) is State::update x steps on every rank (gpe_shard_run_scheduled);
    owned() returns per rank (order keys, positions, previous positions): the order keys are the particles' indices in
    the single-context system after its own re-sorts, so `want_pos[keys] == pos` bit for bit."""

    def __init__(self, pos, rad, world, world_size, gravity=(0.0, 0.0), devices=None, prev=None, grid=None, flags=0,
                 capacity_scale=1.0, capacity=None):
        self.ws = int(world_size)
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 2)
        rad = np.ascontiguousarray(rad, np.float32)
        cs = np.float32(np.abs(rad).max()) * np.float32(2.2)
        self.layout = build_layout(world, cs, self.ws, grid=grid)
        own = owner_of(self.layout, pos)
        self.group = C.c_void_p()
        L.check(L.load().gpe_local_group_create(self.ws, C.byref(self.group)))
        self.ctx = [None] * self.ws
        devices = devices or [0] * self.ws

        def start(r):
            mine = np.nonzero(own == r)[0]
            if len(mine) == 0:
                raise ValueError("rank %d owns no particle at start" % r)
            c = Context(world=world, gravity=gravity, mode=L.MODE_NATIVE, device=devices[r], flags=flags)
            self.ctx[r] = c
            p, q = np.ascontiguousarray(pos[mine]), (np.ascontiguousarray(prev[mine], np.float32) if prev is not None else None)
            c.call("gpe_shard_set_particles", _ptr(p), _ptr(q) if q is not None else None, _ptr(np.ascontiguousarray(rad[mine])),
                   _ptr(mine.astype(np.uint32)), len(mine), int(capacity or 0))
            c.call("gpe_local_group_join", self.group, r)
            c.call("gpe_shard_setup", C.byref(self.layout), r, float(capacity_scale))
        self._each(start)

    def _each(self, fn):
        """fn(rank) on one thread per rank; the first failure aborts the group (the other ranks leave their collectives
        with an error instead of waiting for ever) and is re-raised."""
        errs = [None] * self.ws
        out = [None] * self.ws

        def body(r):
            try:
                out[r] = fn(r)
            except BaseException as e:                                  # noqa: BLE001 -- reported below
                errs[r] = e
                L.load().gpe_local_group_abort(self.group)
        th = [threading.Thread(target=body, args=(r,)) for r in range(self.ws)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        first = [e for e in errs if e is not None and "aborted" not in str(e)] or [e for e in errs if e is not None]
        if first:
            raise first[0]
        return out

    def run(self, dt, steps, resort_every=0, resort_first=True):
        self._each(lambda r: self.ctx[r].call("gpe_shard_run_scheduled", float(dt), int(steps), int(resort_every),
                                              1 if resort_first else 0))

    def owned(self):
        def get(r):
            c = self.ctx[r]
            cap = C.c_uint64()
            c.call("gpe_capacity", C.byref(cap))
            cap = cap.value
            gid, p, q = np.empty(cap, np.uint32), np.empty((cap, 2), np.float32), np.empty((cap, 2), np.float32)
            no = C.c_uint64()
            c.call("gpe_shard_download_owned", _ptr(gid), _ptr(p), _ptr(q), cap, C.byref(no))
            n = int(no.value)
            return gid[:n].astype(np.int64), p[:n].copy(), q[:n].copy()
        return self._each(get)

    def stats(self):
        out = []
        for c in self.ctx:
            st = L.GpeShardStats()
            st.struct_size = C.sizeof(L.GpeShardStats)
            c.call("gpe_shard_get_stats", C.byref(st))
            out.append({k: getattr(st, k) for k, _ in L.GpeShardStats._fields_ if k != "struct_size"})
        return out

    def close(self):
        for c in self.ctx:
            if c is not None:
                c.close()
        self.ctx = []
        if self.group:
            L.load().gpe_local_group_destroy(self.group)
            self.group = None
