"""The C++ host mirror (gpu-physics-engine_amd/host/gpe_host.hpp) and the C++ restatement of the reference's five
integration test files (tests/cpp/reference_tests.cpp).  CPU: it compiles, links against libgpe.so and lists the
reference's test names.  GPU: it runs and every test passes."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "reference_tests.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "reference_tests")
LIBDIR = os.path.join(ROOT, "gpu-physics-engine_amd")

REFERENCE_TEST_NAMES = [
    # tests/grid.rs
    "test_grid_build_cell_ids_with_multiple_particles", "test_grid_build_cell_ids_and_sort",
    "test_grid_build_cell_ids_sort_and_build_empty_collision_cells_list",
    "test_grid_build_cell_ids_sort_and_build_collision_cells_list",
    # tests/particle_sort.rs
    "sort_particles_test",
    # tests/radix_sort.rs
    "sort_test", "sort_test_small_sized_array",
    # tests/prefix_sum.rs
    "inclusive_prefix_sum_test", "inclusive_prefix_sum_same_values_test", "inclusive_prefix_sum_all_zero_test",
    "inclusive_prefix_sum_random_test", "inclusive_prefix_sum_resize_test",
]


def _build(gpe):
    gpe.build()
    deps = [SRC, os.path.join(LIBDIR, "host", "gpe_host.hpp"), os.path.join(ROOT, "include", "gpe.h")]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", SRC, "-L" + LIBDIR, "-lgpe",
                               "-Wl,-rpath," + LIBDIR, "-o", EXE])
    return EXE


def test_cpp_host_compiles_and_names_the_reference_tests(gpe):
    exe = _build(gpe)
    names = subprocess.check_output([exe, "--list"], text=True).split()
    for n in REFERENCE_TEST_NAMES:
        assert n in names


@pytest.mark.gpu
def test_cpp_reference_tests_pass_on_gpu(gpe):
    exe = _build(gpe)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    for n in REFERENCE_TEST_NAMES:
        assert "test %s ... ok" % n in r.stdout


SHARD_SRC = os.path.join(ROOT, "tests", "cpp", "sharded_two_contexts.cpp")
SHARD_EXE = os.path.join(ROOT, "tests", "cpp", "sharded_two_contexts")
SHARD_TESTS = ["two_contexts_local_group_equal_one_context", "two_contexts_caller_collectives_equal_one_context",
               "four_contexts_local_group_equal_one_context"]


def _build_sharded(gpe):
    gpe.build()
    deps = [SHARD_SRC, os.path.join(ROOT, "include", "gpe.h")]
    if not os.path.exists(SHARD_EXE) or any(os.path.getmtime(d) > os.path.getmtime(SHARD_EXE) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-pthread", SHARD_SRC, "-L" + LIBDIR, "-lgpe",
                               "-Wl,-rpath," + LIBDIR, "-o", SHARD_EXE])
    return SHARD_EXE


def test_cpp_sharded_host_compiles_against_the_header_alone(gpe):
    """tests/cpp/sharded_two_contexts.cpp includes include/gpe.h and nothing else of ours: no Python, no torch, no HIP
    header -- the sharded run's control plane is behind the C-ABI."""
    exe = _build_sharded(gpe)
    assert subprocess.check_output([exe, "--list"], text=True).split() == SHARD_TESTS
    src = open(SHARD_SRC).read()
    assert "hip/" not in src and "torch" not in src.replace("no torch", "")


@pytest.mark.gpu
def test_cpp_sharded_host_equals_one_context_on_gpu(gpe):
    """Several gpe_ctx in one process (a thread each) through 30 steps with two re-sorts, over the library's local group
    and over the host's own collectives: bit-identical to one context."""
    exe = _build_sharded(gpe)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    for n in SHARD_TESTS:
        assert "test %s ... ok" % n in r.stdout
