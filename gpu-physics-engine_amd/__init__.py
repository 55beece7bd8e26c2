"""gpu-physics-engine_amd -- MI355X-native particle step behind the reference's module API.

The package holds only what the hot path needs: csrc/ (hand-written HIP for gfx950 + the C-ABI of
include/gpe.h), the ctypes binding (_lib), the host-side mirror of the reference's
particles/grid/physics/utils API (engine) and seeded synthetic scenes (scenes).
Import with importlib.import_module("gpu-physics-engine_amd") -- the directory name has a hyphen.
"""
from . import _lib, scenes  # noqa: F401
from ._lib import (GpeError, MODE_COMPAT, MODE_NATIVE, STEP_RESORT, UNUSED_CELL_ID,  # noqa: F401
                   MAX_CELLS_PER_OBJECT, COUNTING_CHUNK_SIZE)
from .engine import (Context, GpuBuffer, ParticleSystem, Grid, CollisionSystem, GPUSorter,  # noqa: F401
                     PrefixSum, State, NUM_BLOCKS_PER_WORKGROUP, RADIX_SORT_BUCKETS)


def _build_library(force=False):
    from .build import build_library       # importing the submodule rebinds the package attribute `build` ...
    globals()["build"] = _build_library    # ... so put the callable back
    return build_library(force=force)


build = _build_library
