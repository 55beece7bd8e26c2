"""CPU (gloo, 2 ranks): `python bench.py --gpus N` with no launcher environment starts its own N ranks -- children
spawned before the parent makes any GPU call --, reports the number of ranks that actually joined as `n_gpus`, and
exits non-zero when a rank is missing.  `--rehearse` runs the launcher and the barrier / max-over-ranks timing
protocol around an empty loop: no GPU, no physics, the line is marked as a rehearsal and carries no value."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env, capture_output=True,
                          text=True, timeout=240)


def test_bare_gpus_2_launches_two_ranks():
    r = _run({}, "--gpus", "2", "--rehearse", "--steps", "3")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["self_launched"] is True and d["rehearsal"] is True and d["value"] is None


def test_missing_rank_is_a_failure_not_a_smaller_run():
    r = _run({"GPE_BENCH_FAIL_RANK": "1"}, "--gpus", "2", "--rehearse", "--steps", "3")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_launcher_environment_must_match_gpus():
    """Under an external launcher (torch.distributed.run sets RANK / WORLD_SIZE) a --gpus that disagrees with
    WORLD_SIZE is refused: the JSON would misreport the GPU count."""
    r = _run({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"},
             "--gpus", "4", "--rehearse")
    assert r.returncode != 0 and "WORLD_SIZE=1 but --gpus 4" in r.stderr
