"""CPU: the collision-cell list as a PREDICATE.  The HIP kernels (csrc/k_collision.hip `starts_run`, and the native
tiles' "cells of two or more members") do not walk the WGSL's per-chunk state machine
(collision_cell_builder.wgsl:27-85 count, :112-189 build); they use its effect:

    collision_cells[0..K) = ascending start indices of the maximal runs of equal keys with key != UNUSED, length >= 2.

The oracle follows the state machine line by line (oracle/gpe_oracle.c orc_count_objects_per_chunk /
orc_build_collision_cells), so comparing the oracle with the predicate on random sorted key arrays -- runs of every
length straddling the 4-entry chunk boundaries, UNUSED tails, total not a multiple of 4 -- checks that equivalence
(SURVEY.md 8c recorded it from a throw-away simulation; this is the kept test).  Also the per-chunk counts: chunk c
counts the runs whose FIRST entry lies in it."""
import numpy as np
import pytest

UNUSED = 0xFFFFFFFF


def starts_run(prev, cur, nxt, has_next):                 # csrc/k_collision.hip: starts_run
    return cur != UNUSED and prev != cur and has_next and nxt == cur


def predicate_cells(keys):
    n = len(keys)
    out = []
    for i in range(n):
        prev = int(keys[i - 1]) if i >= 1 else UNUSED      # :40 select(UNUSED, cell_ids[first_idx - 1], first_idx >= 1)
        nxt = int(keys[i + 1]) if i + 1 < n else UNUSED
        if starts_run(prev, int(keys[i]), nxt, i + 1 < n):
            out.append(i)
    return np.array(out, np.uint32)


def random_sorted_keys(rng, total, max_run, unused_tail):
    keys = []
    k = int(rng.integers(0, 5))
    while len(keys) < total - unused_tail:
        run = int(rng.integers(1, max_run + 1))
        keys.extend([k] * run)
        k += int(rng.integers(1, 4))
    keys = keys[:total - unused_tail] + [UNUSED] * unused_tail
    return np.array(keys, np.uint32)


@pytest.mark.parametrize("seed", range(40))
def test_state_machine_equals_the_run_predicate(oracle, seed):
    rng = np.random.default_rng(seed)
    total = int(rng.integers(1, 400))
    keys = random_sorted_keys(rng, total, max_run=int(rng.integers(1, 9)), unused_tail=int(rng.integers(0, min(total, 12) + 1)))
    counts = oracle.count_objects_per_chunk(keys)
    want = predicate_cells(keys)
    # per-chunk counts: the runs that START in the chunk
    by_chunk = np.bincount(want // 4, minlength=len(counts)).astype(np.uint32) if len(want) else np.zeros(len(counts), np.uint32)
    assert np.array_equal(counts, by_chunk)
    scanned = oracle.inclusive_scan(counts.copy())
    cells, k, indirect = oracle.build_collision_cells(keys, scanned)
    assert k == len(want)
    assert np.array_equal(cells[:k], want)
    assert (cells[k:] == UNUSED).all()                     # untouched entries keep the initial fill
    assert indirect[0] == (k + 63) // 64 and indirect[1] == 1 and indirect[2] == 1     # :96-109


def test_reference_546_particle_vector_through_the_predicate(golden):
    """tests/grid.rs:286-290: 546 coincident particles of radius 10 occupy four cells -> starts [0, 546, 1092, 1638]."""
    g = golden["grid_case_2"]
    keys = np.repeat(np.array(sorted(set(g.get("sorted_cell_ids_unique", []))) or [0, 1, 2, 3], np.uint32), g["num_particles"])
    assert list(predicate_cells(keys)) == [0, 546, 1092, 1638]
