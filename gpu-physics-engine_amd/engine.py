"""Host-side mirror of the reference's module API over the C-ABI (include/gpe.h).

Same names and argument meaning as the Rust types the reference's State::update() and tests use
(citations into /root/reference/src), so the parity tests read like tests/*.rs:

    WgpuContext      -> Context            (renderer/wgpu_context.rs:8; here: one HIP stream)
    GpuBuffer<T>     -> GpuBuffer          (utils/gpu_buffer.rs:7)
    ParticleSystem   -> ParticleSystem     (particles/particle_system.rs:16)
    Grid             -> Grid               (grid/grid.rs:24)
    CollisionSystem  -> CollisionSystem    (physics/collision_system.rs:9)
    GPUSorter        -> GPUSorter          (utils/radix_sort/radix_sort.rs:44)
    PrefixSum        -> PrefixSum          (utils/prefix_sum/prefix_sum.rs:11)
    State            -> State              (state.rs:21; update() == gpe_step)

One Context owns one gpe_ctx, i.e. one particle system + grid + collision system, which is how
State composes them (state.rs:34-70).  Errors surface as GpeError (the reference panics).
numpy is used only to hold host arrays; every call goes through ctypes to libgpe.so.
"""
import ctypes as C
import sys

import numpy as np

from . import _lib as L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """WgpuContext::new_for_test() stand-in (wgpu_context.rs:73-101): device + queue == HIP stream."""

    def __init__(self, world=(3048.0, 1048.0), gravity=(0.0, 0.0), mode=None, device=-1,
                 profiling=False, stream=None, flags=0):
        self.lib = L.load()
        cfg = L.GpeConfig()
        L.check(self.lib.gpe_config_default(C.byref(cfg)))
        cfg.device = device
        cfg.world_width, cfg.world_height = world
        cfg.gravity_x, cfg.gravity_y = gravity
        if mode is not None:            # default: gpe_config_default's (NATIVE)
            cfg.mode = mode
        cfg.profiling = int(profiling)
        cfg.flags = int(flags)          # L.FLAG_*
        h = C.c_void_p()
        L.check(self.lib.gpe_create(C.byref(cfg), C.byref(h)))
        self.h = h
        if stream is not None:          # a hipStream_t the caller owns and keeps alive (gpe_set_stream)
            self.call("gpe_set_stream", C.c_void_p(int(stream)))

    def close(self):
        if getattr(self, "h", None):
            self.lib.gpe_destroy(self.h)
            self.h = None

    def __del__(self):
        # not during interpreter shutdown: the HIP runtime may already have been torn down by then, and the
        # process is about to release everything anyway
        try:
            if sys is None or sys.is_finalizing():
                return
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def call(self, name, *args):
        L.check(getattr(self.lib, name)(self.h, *args), self.h)

    def sync(self):
        self.call("gpe_sync")

    def pipeline_info(self):
        """gpe_get_pipeline_info: which kernels the next step runs (L.PIPELINE_*), why (L.REASON_*), and the counts."""
        info = L.GpePipelineInfo()
        info.struct_size = C.sizeof(L.GpePipelineInfo)
        self.call("gpe_get_pipeline_info", C.byref(info))
        return {k: getattr(info, k) for k, _ in L.GpePipelineInfo._fields_ if k not in ("struct_size", "reserved")}

    def device_ptr(self, what):
        """gpe_device_ptr: (address, bytes) of a particle / grid array on the device (render hand-off, state.rs:150-176)."""
        ptr, nbytes = C.c_void_p(), C.c_uint64()
        self.call("gpe_device_ptr", what, C.byref(ptr), C.byref(nbytes))
        return ptr, nbytes.value

    def download(self, what, dtype, shape=None):
        nbytes = C.c_uint64()
        self.call("gpe_array_bytes", what, C.byref(nbytes))
        out = np.empty(nbytes.value // np.dtype(dtype).itemsize, dtype)
        self.call("gpe_download", what, _ptr(out), nbytes.value)
        return out.reshape(shape) if shape else out

    def timings(self):
        cnt = C.c_uint32(0)
        self.call("gpe_get_timings", None, C.byref(cnt))
        arr = (L.GpeTiming * max(1, cnt.value))()
        cnt2 = C.c_uint32(cnt.value)
        self.call("gpe_get_timings", arr, C.byref(cnt2))
        return {arr[i].name.decode(): (arr[i].total_ms, arr[i].calls) for i in range(min(cnt.value, cnt2.value))}

    def trace(self):
        """[(scope name, start ms, duration ms)] of the recorded scope instances, oldest first (gpe_get_trace)."""
        cnt = C.c_uint32(0)
        self.call("gpe_get_trace", None, C.byref(cnt))
        arr = (L.GpeTraceEvent * max(1, cnt.value))()
        cnt2 = C.c_uint32(cnt.value)
        self.call("gpe_get_trace", arr, C.byref(cnt2))
        return [(arr[i].name.decode(), arr[i].start_ms, arr[i].duration_ms) for i in range(min(cnt.value, cnt2.value))]

    def write_chrome_trace(self, path):
        """The reference's `benchmark.json` (state.rs:108-112): one complete event ("ph": "X", microseconds) per
        recorded scope, loadable by chrome://tracing / Perfetto."""
        import json
        events = [{"name": n, "cat": "gpu", "ph": "X", "ts": t0 * 1e3, "dur": d * 1e3, "pid": 0, "tid": 0}
                  for n, t0, d in self.trace()]
        with open(path, "w") as f:
            json.dump({"traceEvents": events, "displayTimeUnit": "ms"}, f)
        return len(events)

    def reset_timings(self):
        self.call("gpe_reset_timings")

    def set_profiling(self, on):
        """False/0 off, True/1 every scope, k > 1: the scopes of every k-th step only (sampled)."""
        self.call("gpe_set_profiling", int(on))


class GpuBuffer:
    """utils/gpu_buffer.rs:7-29 for u32 data: device buffer + host mirror (`data()`).  Like the reference's, the device
    buffer grows by doubling (gpu_buffer.rs:54-76) and keeps its DEVICE contents when it does."""

    def __init__(self, ctx, data):
        self.ctx = ctx
        self._data = np.ascontiguousarray(data, dtype=np.uint32).copy()
        self._cap_bytes = max(1, self._data.nbytes)
        self.dptr = C.c_void_p()
        ctx.call("gpe_buffer_alloc", self._cap_bytes, C.byref(self.dptr))
        ctx.call("gpe_buffer_upload", self.dptr, _ptr(self._data), self._data.nbytes)

    def len(self):
        return len(self._data)

    def data(self):
        return self._data

    def capacity_bytes(self):
        return self._cap_bytes

    def _at(self, index):
        return C.c_void_p(self.dptr.value + 4 * int(index))

    def download(self):
        """gpu_buffer.rs:96-175: read the device buffer back into the host mirror."""
        self.ctx.call("gpe_buffer_download", self.dptr, _ptr(self._data), self._data.nbytes)
        return self._data

    def download_last(self):
        """gpu_buffer.rs:177-262: the last element as the DEVICE holds it (None for an empty buffer); the mirror stays."""
        if len(self._data) == 0:
            return None
        out = np.zeros(1, np.uint32)
        self.ctx.call("gpe_buffer_download", self._at(len(self._data) - 1), _ptr(out), 4)
        return int(out[0])

    def _append(self, values):
        # gpu_buffer.rs:49-87 (`upload`): a buffer that is too small is replaced by one of twice the needed size, the old
        # device contents are carried over (buffer-to-buffer copy there; through the host here: the C-ABI has no
        # device-to-device copy and this is off the step path), then only the new tail is written
        values = np.ascontiguousarray(values, np.uint32).reshape(-1)
        old_n = len(self._data)
        need = 4 * (old_n + len(values))
        if need > self._cap_bytes:
            kept = np.zeros(old_n, np.uint32)
            self.ctx.call("gpe_buffer_download", self.dptr, _ptr(kept), kept.nbytes)
            fresh = C.c_void_p()
            self.ctx.call("gpe_buffer_alloc", 2 * max(need, 1), C.byref(fresh))
            self.ctx.call("gpe_buffer_upload", fresh, _ptr(kept), kept.nbytes)
            self.ctx.call("gpe_buffer_free", self.dptr)
            self.dptr, self._cap_bytes = fresh, 2 * max(need, 1)
        self._data = np.concatenate([self._data, values])
        self.ctx.call("gpe_buffer_upload", self._at(old_n), _ptr(values), values.nbytes)

    def push(self, value):
        """gpu_buffer.rs:30-33."""
        self._append([value])

    def push_all(self, values):
        """gpu_buffer.rs:35-38."""
        self._append(values)

    def replace_elem(self, new_data, index):
        """gpu_buffer.rs:264-275: one element of the mirror and of the device buffer."""
        if index < 0 or index >= len(self._data):
            raise IndexError("Index out of bounds")                    # (the reference panics)
        self._data[index] = np.uint32(new_data)
        one = np.array([new_data], np.uint32)
        self.ctx.call("gpe_buffer_upload", self._at(index), _ptr(one), 4)

    def free(self):
        if self.dptr:
            self.ctx.call("gpe_buffer_free", self.dptr)
            self.dptr = C.c_void_p()


class ParticleSystem:
    """particles/particle_system.rs:16-24."""

    SORT_INTERVAL = 4.0      # seconds (particle_system.rs:13-14)

    def __init__(self, ctx):
        import time
        self.ctx = ctx
        self._mouse = (False, (0.0, 0.0))
        self._clock = time.monotonic
        self.last_sort_time = self._clock() - self.SORT_INTERVAL     # :45,97: the first frame sorts

    def is_it_time_to_sort(self):
        """particle_system.rs:229-231 (wall clock, like the reference)."""
        return self._clock() - self.last_sort_time >= self.SORT_INTERVAL

    def reset_last_sort_time(self):
        """particle_system.rs:233-235."""
        self.last_sort_time = self._clock()

    @classmethod
    def new_from_buffers(cls, ctx, positions, radii, prev=None):
        """particle_system.rs:49-99 (the test constructor; prev = cur; integrator world 1920x1080 there --
        here the world is the Context's, set it explicitly to mirror that quirk)."""
        self = cls(ctx)
        pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 2)
        rad = np.ascontiguousarray(radii, np.float32).reshape(-1)
        if pos.shape[0] != rad.shape[0]:
            raise ValueError("positions and radii differ in length")
        pp = None
        if prev is not None:
            pv = np.ascontiguousarray(prev, np.float32).reshape(-1, 2)
            pp = _ptr(pv)
        ctx.call("gpe_set_particles", _ptr(pos), pp, _ptr(rad), pos.shape[0])
        return self

    def add_particles(self, positions, radii):
        """particle_system.rs:163-220 (the reference draws 100 random ones around the mouse)."""
        pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 2)
        rad = np.ascontiguousarray(radii, np.float32).reshape(-1)
        self.ctx.call("gpe_add_particles", _ptr(pos), _ptr(rad), pos.shape[0])

    def len(self):
        n = C.c_uint64()
        self.ctx.call("gpe_len", C.byref(n))
        return n.value

    def get_max_radius(self):
        r = C.c_float()
        self.ctx.call("gpe_max_radius", C.byref(r))
        return r.value

    def sort_by_cell_id(self, cell_size=None):
        """particle_system.rs:236-243.  The cell size is the Grid's (state.rs:123)."""
        self.ctx.call("gpe_morton_resort")

    def update_positions(self, dt):
        """particle_system.rs:245-247."""
        self.ctx.call("gpe_integrate", float(dt))

    def mouse_click_callback(self, pressed, position):
        """particle_system.rs:221-224 -> particle_integration.rs:176-181."""
        self._mouse = (bool(pressed), (float(position[0]), float(position[1])))
        self.ctx.call("gpe_set_mouse", 1 if pressed else 0, float(position[0]), float(position[1]))

    def mouse_move_callback(self, position):
        """particle_system.rs:225-227 -> particle_integration.rs:182-185: the position moves, the button state stays."""
        self._mouse = (self._mouse[0], (float(position[0]), float(position[1])))
        self.ctx.call("gpe_set_mouse", 1 if self._mouse[0] else 0, float(position[0]), float(position[1]))

    def download_home_cell_ids(self):
        return self.ctx.download(L.HOME_CELL_IDS, np.uint32)

    def download_particle_ids(self):
        return self.ctx.download(L.PARTICLE_IDS, np.uint32)

    def download_particle_buffers(self):
        """particle_system.rs:258-265 -> (current_positions, previous_positions, radii)."""
        return (self.ctx.download(L.POS, np.float32, (-1, 2)),
                self.ctx.download(L.PREV, np.float32, (-1, 2)),
                self.ctx.download(L.RADIUS, np.float32))


class Grid:
    """grid/grid.rs:24-33."""

    def __init__(self, ctx, particle_system):
        """Grid::new (grid.rs:66-71): cell size from the particle system's max radius."""
        self.ctx = ctx

    @classmethod
    def new_without_camera(cls, ctx, max_obj_radius, particle_system):
        """grid.rs:74-149."""
        self = cls(ctx, particle_system)
        ctx.call("gpe_grid_set_max_radius", float(max_obj_radius))
        return self

    @staticmethod
    def compute_cell_size(max_obj_radius):
        """grid.rs:159-161."""
        return float(L.load().gpe_compute_cell_size(float(max_obj_radius)))

    def cell_size(self):
        cs = C.c_float()
        self.ctx.call("gpe_cell_size", C.byref(cs))
        return cs.value

    def build_cell_ids(self):
        self.ctx.call("gpe_grid_build")        # grid.rs:296-306

    def sort_map(self):
        self.ctx.call("gpe_grid_sort")         # grid.rs:310-312

    def update(self):
        self.ctx.call("gpe_grid_update")       # grid.rs:322-332

    def download_cell_ids(self):
        return self.ctx.download(L.CELL_IDS, np.uint32)      # grid.rs:314

    def download_object_ids(self):
        return self.ctx.download(L.OBJECT_IDS, np.uint32)    # grid.rs:318


class CollisionSystem:
    """physics/collision_system.rs:9-12."""

    def __init__(self, ctx, dim, particle_system, grid):
        if dim != 2:
            raise ValueError("2-D only (state.rs:18 DIMENSION = 2)")
        self.ctx = ctx

    def solve_collisions(self):
        self.ctx.call("gpe_solve_collisions")  # collision_system.rs:30-39

    def build_collision_cells(self):
        self.ctx.call("gpe_build_collision_cells")

    def download_collision_cells(self):
        return self.ctx.download(L.COLLISION_CELLS, np.uint32)   # collision_system.rs:41

    def num_collision_cells(self):
        return int(self.ctx.download(L.NUM_COLLISION_CELLS, np.uint32)[0])


NUM_BLOCKS_PER_WORKGROUP = 45      # radix_sort.rs:40
RADIX_SORT_BUCKETS = 256           # radix_sort.rs:30


class GPUSorter:
    """utils/radix_sort/radix_sort.rs:44-48 over two caller-owned GpuBuffers."""

    def __init__(self, ctx, length, keys, payload):
        self.ctx, self.length, self.keys, self.payload = ctx, int(length), keys, payload
        self.keys_b = GpuBuffer(ctx, np.zeros(self.length, np.uint32))      # radix_sort.rs:260-276
        self.payload_b = GpuBuffer(ctx, np.zeros(self.length, np.uint32))
        self.histogram = GpuBuffer(ctx, np.zeros(RADIX_SORT_BUCKETS, np.uint32))

    def sort(self, sort_first_n=None):
        n = self.length if sort_first_n is None else int(sort_first_n)      # radix_sort.rs:202
        self.ctx.call("gpe_sort_pairs_u32", self.keys.dptr, self.payload.dptr, n)

    def build_histogram(self, num_elements, current_shift):
        """radix_sort.rs:180-188 (ping = true: reads keys_a)."""
        self.ctx.call("gpe_sort_histogram_u32", self.keys.dptr, int(num_elements), int(current_shift),
                      self.histogram.dptr)

    def scatter(self, num_elements, current_shift):
        """radix_sort.rs:190-198 (ping = true: keys_a/payload_a -> keys_b/payload_b)."""
        self.ctx.call("gpe_sort_scatter_pass_u32", self.keys.dptr, self.payload.dptr, self.keys_b.dptr,
                      self.payload_b.dptr, int(num_elements), int(current_shift))

    def get_keys_b(self):
        return self.keys_b.download()

    def get_payload_b(self):
        return self.payload_b.download()

    def get_histogram(self):
        return self.histogram.download()


class PrefixSum:
    """utils/prefix_sum/prefix_sum.rs:11-18."""

    def __init__(self, ctx, buffer):
        self.ctx, self.buffer = ctx, buffer

    def execute(self, num_items):
        self.ctx.call("gpe_inclusive_scan_u32", self.buffer.dptr, int(num_items))   # prefix_sum.rs:143-160

    def update_buffers(self, buffer):
        self.buffer = buffer                                                         # prefix_sum.rs:172


class State:
    """state.rs:21-31 without window/renderer: particles + grid + collision system and update()."""

    def __init__(self, positions, radii, world=(3048.0, 1048.0), gravity=(0.0, 0.0), mode=None,
                 prev=None, device=-1, profiling=False, flags=0):
        self.world, self.gravity, self.mode = tuple(map(float, world)), tuple(map(float, gravity)), mode
        self.ctx = Context(world=world, gravity=gravity, mode=mode, device=device, profiling=profiling, flags=flags)
        self.particles = ParticleSystem.new_from_buffers(self.ctx, positions, radii, prev=prev)
        self.grid = Grid(self.ctx, self.particles)
        self.collision_system = CollisionSystem(self.ctx, 2, self.particles, self.grid)

    def update(self, dt, resort=False):
        """state.rs:115-131 (dt is explicit instead of wall clock, the re-sort an explicit flag)."""
        self.ctx.call("gpe_step", float(dt), L.STEP_RESORT if resort else 0)

    def update_wallclock(self, dt):
        """state.rs:115-131 with the reference's own re-sort policy: every SORT_INTERVAL seconds of wall clock
        (particle_system.rs:229-235), first frame included."""
        resort = self.particles.is_it_time_to_sort()
        self.update(dt, resort=resort)
        if resort:
            self.particles.reset_last_sort_time()
        return resort

    def run(self, dt, steps, resort_every=0, resort_first=True):
        self.ctx.call("gpe_run", float(dt), int(steps), int(resort_every), 1 if resort_first else 0)

    def add_particles(self, positions, radii):
        """state.rs:187-200."""
        self.particles.add_particles(positions, radii)

    def positions(self):
        return self.ctx.download(L.POS, np.float32, (-1, 2))

    def previous_positions(self):
        return self.ctx.download(L.PREV, np.float32, (-1, 2))

    def radii(self):
        return self.ctx.download(L.RADIUS, np.float32)

    # Checkpoint / restore (the reference's only state dump is download_particle_buffers,
    # particle_system.rs:258-265): the three arrays the step evolves plus the constants a step depends on.
    def save(self, path):
        """Binary snapshot (numpy .npz, no pickle): positions, previous positions, radii, world, gravity."""
        np.savez(path, format=np.array([1], np.int32), pos=self.positions(), prev=self.previous_positions(),
                 radius=self.radii(), world=np.array(self.world, np.float32), gravity=np.array(self.gravity, np.float32))

    @classmethod
    def load(cls, path, mode=None, device=-1):
        """A State that continues from a snapshot written by save(): the next update() yields the same bits as
        the saved run's next update() would have (the step has no hidden state beyond these arrays)."""
        with np.load(path, allow_pickle=False) as d:
            if int(d["format"][0]) != 1:
                raise ValueError("unknown snapshot format")
            return cls(d["pos"], d["radius"], world=tuple(d["world"]), gravity=tuple(d["gravity"]), mode=mode,
                       prev=d["prev"], device=device)

    def close(self):
        self.ctx.close()
