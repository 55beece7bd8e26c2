"""Generates tests/golden/oracle_trajectory.json with the CPU oracle (oracle/gpe_oracle.c).

The reference holds no vector for the float results of K11 / K12 ("oracle-defined"); this fixture
freezes what the oracle computes for one small seeded scene -- mixed radii, gravity, mouse attraction (button held), a Morton
re-sort in the middle -- so that (a) a change to the oracle that moves these bits is noticed and (b) the HIP
pipelines are compared against committed data, not only against the oracle as built that day.
Inputs are regenerated from the seed (numpy PCG64 is stable across versions); outputs are kept as a SHA-256 of
the final position / previous-position bytes plus 16 sampled values in hex.

    python tests/golden/make_trajectory_fixture.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

SPEC = {"n": 3000, "seed": 20261004, "world": [110.0, 90.0], "gravity": [3.0, -9.81], "dt": 1.0 / 60.0,
        "steps": 24, "resort_at": [0, 12], "mouse": {"x": 40.0, "y": 55.0},
        "radii": [0.5, 0.35, 0.5, 0.42]}


def scene(spec):
    rng = np.random.default_rng(spec["seed"])
    n = spec["n"]
    pos = np.empty((n, 2), np.float32)
    pos[:, 0] = rng.random(n, dtype=np.float32) * np.float32(spec["world"][0])
    pos[:, 1] = rng.random(n, dtype=np.float32) * np.float32(spec["world"][1])
    rad = np.array(spec["radii"], np.float32)[np.arange(n) % len(spec["radii"])]
    return pos, rad


def run_oracle(spec):
    from oracle import oracle as orc
    pos, rad = scene(spec)
    p = orc.default_params(spec["world"][0], spec["world"][1], float(rad.max()), gravity=tuple(spec["gravity"]))
    p.mouse_pressed, p.mouse_x, p.mouse_y = 1, spec["mouse"]["x"], spec["mouse"]["y"]      # pressed throughout
    sim = orc.Sim(pos, rad, p)
    for s in range(spec["steps"]):
        sim.step(spec["dt"], resort=(s in spec["resort_at"]))
    out = (sim.pos.copy(), sim.prev.copy(), sim.radius.copy())
    sim.close()
    return out


def digest(pos, prev):
    idx = np.linspace(0, len(pos) - 1, 16).astype(np.int64)
    return {"sha256_pos": hashlib.sha256(np.ascontiguousarray(pos, np.float32).tobytes()).hexdigest(),
            "sha256_prev": hashlib.sha256(np.ascontiguousarray(prev, np.float32).tobytes()).hexdigest(),
            "sample_index": idx.tolist(),
            "sample_pos_hex": [[float(v).hex() for v in pos[i]] for i in idx]}


if __name__ == "__main__":
    pos, prev, _ = run_oracle(SPEC)
    out = {"_about": "oracle-defined trajectory fixture; see make_trajectory_fixture.py", "spec": SPEC,
           "expected": digest(pos, prev)}
    with open(os.path.join(ROOT, "tests", "golden", "oracle_trajectory.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote fixture:", out["expected"]["sha256_pos"][:16])
