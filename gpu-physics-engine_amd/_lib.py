"""ctypes binding of libgpe.so (include/gpe.h).  No torch, no numpy arrays in the signatures:
plain pointers and sizes, exactly what a Rust `extern "C"` block would bind."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libgpe.so")

GPE_OK = 0
GPE_ERR_INVALID_ARG = -1
GPE_ERR_HIP = -2
GPE_ERR_OOM = -3
GPE_ERR_STATE = -4
GPE_ERR_UNSUPPORTED = -5
GPE_ERR_NO_DEVICE = -6

MODE_COMPAT = 0
MODE_NATIVE = 1
FLAG_NATIVE_FORCE = 1
FLAG_SORT_EVERY_STEP = 2
FLAG_NATIVE_STATS = 4
FLAG_SAFE_SORT = 8
FLAG_COUNTING_SORT_TILES = 16
FLAG_XCD_EIGHTHS = 64
FLAG_NO_HALF_TILES = 128
FLAG_SHARD_OVERLAP = 256
FLAG_FUSED_HISTOGRAMS = 512
PIPELINE_COMPAT, PIPELINE_NATIVE = 0, 1
(REASON_NONE, REASON_MODE_COMPAT, REASON_NO_PARTICLES, REASON_OUT_OF_BOX, REASON_GRID_TOO_WIDE,
 REASON_TABLE_TOO_LARGE, REASON_DENSE_WINDOWS) = range(7)
STEP_RESORT = 1

UNUSED_CELL_ID = 0xFFFFFFFF
MAX_CELLS_PER_OBJECT = 4
COUNTING_CHUNK_SIZE = 4

# gpe_array
POS, PREV, RADIUS, HOME_CELL_IDS, PARTICLE_IDS, CELL_IDS, OBJECT_IDS, COLLISION_CELLS, \
    NUM_COLLISION_CELLS, CHUNK_OBJ_COUNT, INDIRECT_ARGS, ORDER_KEYS = range(12)


class GpeConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("device", C.c_int32),
        ("world_width", C.c_float), ("world_height", C.c_float),
        ("gravity_x", C.c_float), ("gravity_y", C.c_float),
        ("cell_size_multiplier", C.c_float), ("stiffness", C.c_float),
        ("mouse_strength", C.c_float), ("mode", C.c_uint32), ("profiling", C.c_uint32),
        ("flags", C.c_uint32), ("reserved", C.c_uint32 * 4),
    ]


class GpePipelineInfo(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("pipeline", C.c_uint32), ("reason", C.c_uint32),
                ("sort_passes", C.c_uint32), ("native_steps", C.c_uint64), ("compat_steps", C.c_uint64),
                ("native_sorts", C.c_uint64), ("window_max", C.c_uint32), ("roster_stamp", C.c_uint32),
                ("overflow_tiles", C.c_uint32), ("overflow_subtiles", C.c_uint32), ("overflow_spills", C.c_uint32),
                ("arena_slots", C.c_uint32)]


class GpeTiming(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("total_ms", C.c_double), ("calls", C.c_uint64)]


class GpeTraceEvent(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("start_ms", C.c_double), ("duration_ms", C.c_double)]


class GpeShardPlan(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("rank", C.c_uint32), ("world_size", C.c_uint32), ("n_slots", C.c_uint32),
        ("blocks_x", C.c_int32), ("blocks_y", C.c_int32),
        ("d_owner_of_block", C.c_void_p), ("d_dest_mask_of_block", C.c_void_p),
        ("slot_rank", C.c_uint32 * 9),
        ("send_off", C.c_uint32 * 9), ("send_cap_mig", C.c_uint32 * 9), ("send_cap_gho", C.c_uint32 * 9),
        ("recv_off", C.c_uint32 * 9), ("recv_cap_mig", C.c_uint32 * 9), ("recv_cap_gho", C.c_uint32 * 9),
        ("d_send", C.c_void_p), ("d_recv", C.c_void_p),
        ("own_x0", C.c_int32), ("own_y0", C.c_int32), ("own_x1", C.c_int32), ("own_y1", C.c_int32),
    ]


# gpe_shard_transport_fn
SHARD_TRANSPORT_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
COMM_ID_BYTES = 128
SHARD_MAX_RANKS = 26
REDUCE_SUM, REDUCE_MAX = 0, 1


class GpeShardLayout(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("world_size", C.c_uint32), ("px", C.c_uint32), ("py", C.c_uint32),
        ("world_width", C.c_float), ("world_height", C.c_float), ("cell_size", C.c_float),
        ("cells_x", C.c_int32), ("cells_y", C.c_int32), ("blocks_x", C.c_int32), ("blocks_y", C.c_int32),
        ("xcuts", C.c_int32 * 27), ("ycuts", C.c_int32 * 27),
    ]


# gpe_shard_collectives.all_reduce_u32 / .all_to_all_u32
ALL_REDUCE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p)
ALL_TO_ALL_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p,
                            C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p)


class GpeShardCollectives(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("reserved", C.c_uint32), ("user", C.c_void_p),
                ("all_reduce_u32", ALL_REDUCE_FN), ("all_to_all_u32", ALL_TO_ALL_FN)]


class GpeShardStats(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("recuts", C.c_uint32), ("resorts", C.c_uint64), ("steps", C.c_uint64),
                ("n_owned", C.c_uint64), ("n_ghost", C.c_uint64), ("n_neighbours", C.c_uint32), ("transport", C.c_uint32)]


class GpeError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("gpe status %d: %s" % (status, message))
        self.status = status


# every symbol include/gpe.h declares: (name, restype, argtypes)
_VP, _U64, _U32, _I32, _F = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32, C.c_float
SYMBOLS = [
    ("gpe_abi_version", _U32, []),
    ("gpe_config_default", _I32, [C.POINTER(GpeConfig)]),
    ("gpe_create", _I32, [C.POINTER(GpeConfig), C.POINTER(_VP)]),
    ("gpe_destroy", _I32, [_VP]),
    ("gpe_last_error", C.c_char_p, [_VP]),
    ("gpe_set_particles", _I32, [_VP, _VP, _VP, _VP, _U64]),
    ("gpe_add_particles", _I32, [_VP, _VP, _VP, _U64]),
    ("gpe_len", _I32, [_VP, C.POINTER(_U64)]),
    ("gpe_max_radius", _I32, [_VP, C.POINTER(_F)]),
    ("gpe_morton_resort", _I32, [_VP]),
    ("gpe_integrate", _I32, [_VP, _F]),
    ("gpe_set_mouse", _I32, [_VP, _I32, _F, _F]),
    ("gpe_set_world", _I32, [_VP, _F, _F]),
    ("gpe_set_gravity", _I32, [_VP, _F, _F]),
    ("gpe_compute_cell_size", _F, [_F]),
    ("gpe_grid_set_max_radius", _I32, [_VP, _F]),
    ("gpe_cell_size", _I32, [_VP, C.POINTER(_F)]),
    ("gpe_grid_build", _I32, [_VP]),
    ("gpe_grid_sort", _I32, [_VP]),
    ("gpe_grid_update", _I32, [_VP]),
    ("gpe_solve_collisions", _I32, [_VP]),
    ("gpe_build_collision_cells", _I32, [_VP]),
    ("gpe_step", _I32, [_VP, _F, _U32]),
    ("gpe_run", _I32, [_VP, _F, _U64, _U64, _I32]),
    ("gpe_sync", _I32, [_VP]),
    ("gpe_set_mode", _I32, [_VP, _U32]),
    ("gpe_get_pipeline_info", _I32, [_VP, C.POINTER(GpePipelineInfo)]),
    ("gpe_download", _I32, [_VP, C.c_int, _VP, _U64]),
    ("gpe_array_bytes", _I32, [_VP, C.c_int, C.POINTER(_U64)]),
    ("gpe_device_ptr", _I32, [_VP, C.c_int, C.POINTER(_VP), C.POINTER(_U64)]),
    ("gpe_buffer_alloc", _I32, [_VP, _U64, C.POINTER(_VP)]),
    ("gpe_buffer_free", _I32, [_VP, _VP]),
    ("gpe_buffer_upload", _I32, [_VP, _VP, _VP, _U64]),
    ("gpe_buffer_download", _I32, [_VP, _VP, _VP, _U64]),
    ("gpe_sort_pairs_u32", _I32, [_VP, _VP, _VP, _U64]),
    ("gpe_sort_histogram_u32", _I32, [_VP, _VP, _U64, _U32, _VP]),
    ("gpe_sort_scatter_pass_u32", _I32, [_VP, _VP, _VP, _VP, _VP, _U64, _U32]),
    ("gpe_inclusive_scan_u32", _I32, [_VP, _VP, _U64]),
    ("gpe_reserve", _I32, [_VP, _U64]),
    ("gpe_capacity", _I32, [_VP, C.POINTER(_U64)]),
    ("gpe_set_counts", _I32, [_VP, _U64, _U64]),
    ("gpe_use_order_keys", _I32, [_VP, _I32]),
    ("gpe_set_active_cells", _I32, [_VP, _I32, _I32, _I32, _I32]),
    ("gpe_stream_handle", _I32, [_VP, C.POINTER(_VP)]),
    ("gpe_set_stream", _I32, [_VP, _VP]),
    ("gpe_refresh", _I32, [_VP]),
    ("gpe_shard_classify", _I32, [_VP, _VP, _VP, _I32, _I32, _U32, _VP, _VP, _VP, _U64]),
    ("gpe_shard_configure", _I32, [_VP, C.POINTER(GpeShardPlan)]),
    ("gpe_shard_begin", _I32, [_VP]),
    ("gpe_shard_unpack", _I32, [_VP]),
    ("gpe_shard_step", _I32, [_VP, _F]),
    ("gpe_shard_peek", _I32, [_VP, C.POINTER(_U64), C.POINTER(_U64)]),
    ("gpe_shard_counts", _I32, [_VP, C.POINTER(_U64), C.POINTER(_U64), _I32]),
    ("gpe_comm_probe", _I32, []),
    ("gpe_comm_unique_id", _I32, [_VP]),
    ("gpe_shard_comm_init", _I32, [_VP, _VP, _U32, _U32]),
    ("gpe_shard_comm_attach", _I32, [_VP, _VP]),
    ("gpe_shard_comm_destroy", _I32, [_VP]),
    ("gpe_shard_set_transport", _I32, [_VP, _VP, _VP]),
    ("gpe_shard_exchange", _I32, [_VP]),
    ("gpe_shard_run", _I32, [_VP, _F, _U64]),
    ("gpe_shard_layout_build", _I32, [_F, _F, _F, _U32, _U32, _U32, _VP, _VP, C.POINTER(GpeShardLayout)]),
    ("gpe_shard_layout_owner_of", _I32, [C.POINTER(GpeShardLayout), _VP, _U64, _VP]),
    ("gpe_shard_quantile_cuts", _I32, [_VP, _U32, _U32, _U32, _VP]),
    ("gpe_shard_set_collectives", _I32, [_VP, C.POINTER(GpeShardCollectives)]),
    ("gpe_local_group_create", _I32, [_U32, C.POINTER(_VP)]),
    ("gpe_local_group_destroy", _I32, [_VP]),
    ("gpe_local_group_join", _I32, [_VP, _VP, _U32]),
    ("gpe_local_group_abort", _I32, [_VP]),
    ("gpe_shard_set_particles", _I32, [_VP, _VP, _VP, _VP, _VP, _U64, _U64]),
    ("gpe_shard_setup", _I32, [_VP, C.POINTER(GpeShardLayout), _U32, _F]),
    ("gpe_shard_get_layout", _I32, [_VP, C.POINTER(GpeShardLayout)]),
    ("gpe_shard_resort", _I32, [_VP]),
    ("gpe_shard_recut", _I32, [_VP, _F, C.POINTER(_I32)]),
    ("gpe_shard_run_scheduled", _I32, [_VP, _F, _U64, _U64, _I32]),
    ("gpe_shard_download_owned", _I32, [_VP, _VP, _VP, _VP, _U64, C.POINTER(_U64)]),
    ("gpe_shard_get_stats", _I32, [_VP, C.POINTER(GpeShardStats)]),
    ("gpe_set_profiling", _I32, [_VP, _U32]),
    ("gpe_reset_timings", _I32, [_VP]),
    ("gpe_get_timings", _I32, [_VP, C.POINTER(GpeTiming), C.POINTER(_U32)]),
    ("gpe_get_trace", _I32, [_VP, C.POINTER(GpeTraceEvent), C.POINTER(_U32)]),
]

_lib = None


def load():
    """dlopen libgpe.so.  There is no CPU or pure-Python fallback: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python gpu-physics-engine_amd/build.py` "
            "(or __graft_entry__.build()); the HIP extension is the only implementation" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the library does not export it
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status, ctx=None):
    if status != GPE_OK:
        msg = load().gpe_last_error(ctx)
        raise GpeError(status, msg.decode() if msg else "")
