#!/usr/bin/env python3
"""bench.py -- physics steps/s of the particle step on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  N > 1 either under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N:
  RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment) or bare: `python bench.py --gpus N` then
  starts its own N ranks -- child processes, spawned BEFORE this process makes any GPU call -- one cuda:<local_rank>
  each over nccl (= RCCL), and exits non-zero unless all N joined.  `n_gpus` in the JSON is the number of ranks
  that ran.

A "step" is one State::update() (state.rs:115-131): [Morton re-sort] -> pair list + sort ->
collision-cell list -> 4 colour passes -> Verlet, over one synthetic uniform-random particle cloud
that is already resident in HBM when the timed region starts.  Default workload = BASELINE.json
configs[1]: 1M particles, gravity off, the reference's 3048 x 1048 world.  The 100M-particle
configuration (configs[2]) is measured in the same run and reported under "extra_workloads".

`value` is what a long-lived host sees: the K-step windows follow each other from step W of the run until 2000
steps have been timed (re-sorts and radix sorts inside), value = all timed steps / all timed seconds.  The figure
of a fresh cloud (the same K steps restarted from a snapshot, as rounds 1-3 reported) is `value_fresh_cloud`.

One JSON line on stdout (rank 0).  `roofline` prices the dominant kernel: algorithmic bytes per launch
("Kernels and rooflines") / its mean launch time from hipEvent pairs recorded on the
library's own stream inside the timed region.  `cpu_baseline` is the CPU oracle (a port of the
reference algorithm, oracle/gpe_oracle.c) timed on this box's host cores on a bounded sample.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
ALGO_BYTES_PER_PARTICLE = 148    # hash 12 + sort 68 + grid 8 + collision 24 + integrate 36
RESORT_EVERY = 240               # 4 s at 60 Hz (particle_system.rs:13-14)
PROFILE_EVERY = 10               # kernels of every 10th timed step are bracketed by HIP events (an event pair
                                 # per kernel on every step costs ~35% at 1M particles)
HEADLINE_STEPS = 2000            # the K-step windows FOLLOW each other until this many steps of the run have been timed
MAX_HEADLINE_SECONDS = 20.0      # ... or this much time (big particle counts); `value` = all timed steps / all timed seconds
FRESH_SECONDS = 0.25             # N = 1: afterwards, windows restarted from the state after the warm-up (`value_fresh_cloud`)
MAX_WINDOWS = 400


class Schedule:
    """The run's global step counter: a Morton re-sort on step 0 and on every RESORT_EVERY-th step of the RUN
    (particle_system.rs:13-14, :45), wherever the timed windows happen to start."""

    def __init__(self, run):
        self.run, self.done = run, 0          # run(dt, steps, resort_every, resort_first)

    def advance(self, dt, k):
        while k > 0:
            to_resort = (RESORT_EVERY - self.done % RESORT_EVERY) % RESORT_EVERY
            if to_resort == 0:
                self.run(dt, 1, 0, True)
                self.done += 1; k -= 1
                continue
            c = min(k, to_resort)
            self.run(dt, c, 0, False)
            self.done += c; k -= c


def timed_windows(sched, dt, steps, sync, dist, torch, min_seconds=0.0, restore=None, min_steps=0,
                  max_seconds=MAX_HEADLINE_SECONDS):
    """EXACTLY `steps` steps per window, each window bracketed by barrier + synchronize on both sides and reduced
    with MAX over the ranks.  Returns the list of window times (seconds).  Every rank runs the same number of windows
    (the count follows from all-reduced times).
    restore is None: the windows FOLLOW each other -- the run a long-lived host sees, re-sorts every 240 steps of the
    run included -- until min_steps steps have been timed (or max_seconds).
    restore: called before every window -- puts the system back to the state the first window starts from, so that
    every window times the SAME steps of the run (a cloud without damping relaxes: at 1M the step is 20 % slower after
    4000 steps); repeated until min_seconds have been measured."""
    times = []
    while True:
        if restore is not None:
            restore()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sched.advance(dt, steps)
        sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        times.append(el)
        if len(times) >= MAX_WINDOWS or sum(times) >= max_seconds:
            return times
        if restore is not None:
            if sum(times) >= min_seconds:
                return times
        elif len(times) * steps >= min_steps:
            return times


def window_stats(times, steps):
    import statistics
    med = statistics.median(times)
    return {"windows": len(times), "steps_per_window": steps, "timed_steps_total": steps * len(times),
            "timed_seconds_total": round(sum(times), 4), "mean_window_ms": round(sum(times) / len(times) * 1e3, 4),
            "median_window_ms": round(med * 1e3, 4),
            "first_window_ms": round(times[0] * 1e3, 4), "min_window_ms": round(min(times) * 1e3, 4),
            "max_window_ms": round(max(times) * 1e3, 4)}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--extra-tail", type=int, nargs="*", default=[500, 750, 1000],
                    help="N = 1: the 100M run goes on to these step marks after its timed window; ms/step of every "
                         "stretch is reported (the gravity-on scene falls, piles up and slows down)")
    ap.add_argument("--particles", type=int, default=1_000_000, help="particles per GPU")
    ap.add_argument("--mode", choices=["compat", "native"], default=os.environ.get("GPE_BENCH_MODE", "native"))
    ap.add_argument("--gravity", choices=["off", "on"], default="off")
    ap.add_argument("--no-extra", action="store_true", help="skip the 100M-particle extra workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-window", action="store_true",
                    help="time ONE window of --steps steps (profiler runs: scripts/gpu_profile_r03.sh)")
    ap.add_argument("--extra-particles", type=int, default=100_000_000)
    ap.add_argument("--extra-steps", type=int, default=250,
                    help="timed steps of the 100M legs (the re-sort at step 240 falls inside the window)")
    ap.add_argument("--soak", action="store_true", help="N = 1: also run the 100M gravity-on scene for 3000 steps "
                                                        "and report steps/s around steps 500 / 1500 / 2500")
    ap.add_argument("--rehearse", action="store_true",
                    help="launcher rehearsal on CPU (gloo): ranks join, run the timing protocol around an empty loop, "
                         "print a line marked rehearsal -- no GPU, no physics, never a result")
    return ap.parse_args()


def kernel_rooflines(timings, n_particles, mode):
    """Per-kernel algorithmic traffic (bytes per launch) for the kernels this build launches."""
    n = n_particles
    pairs = 4 * n if mode == "compat" else n
    table = {
        # name: (bytes per launch, description)
        "sort/scatter": (16 * pairs, "R key+payload 8 B, W key+payload 8 B per pair"),
        "sort/count": (4 * pairs, "R key 4 B per pair"),
        "sort/onesweep": (16 * pairs, "one radix pass: R key+payload 8 B, W key+payload 8 B per pair"),
        "sort/hist": (4 * pairs, "R key 4 B per pair"),
        "native/hash": (12 * n, "R pos 8 B, W cell key 4 B per particle"),
        "native/collide": (24 * n, "R pos 8 + radius 4 + id 4, W pos 8 per particle (SURVEY 8d collision row)"),
        "native/collide+verlet": (40 * n, "collision row (R pos 8 + radius 4 + id 4, W pos 8) + fused integration "
                                          "(R prev 8, W prev 8) per particle"),
        "Build cell ids": (12 * n + 32 * n, "R pos 8 + radius 4, W 4 cell ids + 4 object ids"),
        "Particle integration pass": (36 * n, "R pos 8 + prev 8 + radius 4, W pos 8 + prev 8"),
        "Collision cell count objects per chunk": (16 * n + 4 * n, "R 4N keys, W N counts"),
        "Build collision cells": (4 * n + 16 * n + 4 * n, "R counts + keys, W ~N starts"),
    }
    out = {}
    for name, (bytes_per_launch, desc) in table.items():
        if name in timings and timings[name][1] > 0:
            ms = timings[name][0] / timings[name][1]
            out[name] = {"avg_ms": ms, "bytes": bytes_per_launch, "GBps": bytes_per_launch / (ms * 1e-3) / 1e9,
                         "calls": timings[name][1], "total_ms": timings[name][0], "what": desc}
    return out


def run_workload(gpe, torch, dist, rank, world_size, n, steps, warmup, mode, gravity, device, headline_steps=HEADLINE_STEPS,
                 fresh_seconds=0.0, tail=()):
    """Returns (mean seconds per `steps` steps over the timed windows, max over ranks; timings dict of rank 0; world;
    window statistics; pipeline info; fresh-cloud statistics or None; tail marks).
    The windows follow each other from step `warmup` of the run on (headline_steps in all; 0: one window).
    fresh_seconds > 0: afterwards the state after `warmup` steps is put back again and again and the same `steps` steps
    are timed for that long -- the figure of rounds 1-3 (no radix sort inside a window, freshly written rosters).
    tail: after the windows, the run goes on to each of these step marks; ms/step of every stretch is reported."""
    import numpy as np
    world = gpe.scenes.world_for(n)
    t0 = time.time()
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=0x5EED + rank)
    log("[rank %d] scene: %d particles in %.1f x %.1f (%.1fs)" % (rank, n, world[0], world[1], time.time() - t0))
    g = (0.0, -9.81) if gravity == "on" else (0.0, 0.0)
    st = gpe.State(pos, rad, world=world, gravity=g, device=device,
                   mode=gpe.MODE_NATIVE if mode == "native" else gpe.MODE_COMPAT)
    del pos, rad
    dt = 1.0 / 60.0
    sched = Schedule(lambda d, k, every, first: st.run(d, k, resort_every=every, resort_first=first))
    snap = None
    if fresh_seconds > 0.0 and warmup >= 1:
        # the state after warmup - 1 steps, kept on the host for the fresh-cloud windows below
        sched.advance(dt, warmup - 1)
        st.ctx.sync()
        snap = (np.ascontiguousarray(st.positions()), np.ascontiguousarray(st.previous_positions()),
                np.ascontiguousarray(st.radii()), sched.done)
        sched.advance(dt, 1)
    else:
        sched.advance(dt, warmup)           # first frame re-sorts
    st.ctx.sync()
    st.ctx.set_profiling(PROFILE_EVERY)     # HIP-event pairs around the kernels of every k-th step
    st.ctx.reset_timings()
    times = timed_windows(sched, dt, steps, st.ctx.sync, dist, torch, min_steps=headline_steps)
    elapsed = sum(times) / len(times)
    timings = st.ctx.timings()
    pipe = st.ctx.pipeline_info()
    marks = []
    for m in tail:                          # the same run, further on (the 100M gravity scene falls and piles up)
        k = m - sched.done
        if k <= 0:
            continue
        st.ctx.sync()
        t1 = time.perf_counter()
        sched.advance(dt, k)
        st.ctx.sync()
        el = time.perf_counter() - t1
        marks.append({"steps": [m - k, m], "ms_per_step": round(el / k * 1e3, 4), "steps_per_sec": round(k / el, 2)})
        log("  steps %d-%d: %.3f ms/step" % (m - k, m, el / k * 1e3))
    fresh = None
    if snap is not None:
        import ctypes
        st.ctx.set_profiling(0)
        pv = lambda a: a.ctypes.data_as(ctypes.c_void_p)

        def restore():
            # gpe_set_particles (ParticleSystem::new_from_buffers), then the last warm-up step again, untimed (like any
            # first step on fresh buffers it sorts)
            st.ctx.call("gpe_set_particles", pv(snap[0]), pv(snap[1]), pv(snap[2]), len(snap[2]))
            sched.done = snap[3]
            sched.advance(dt, 1)
            st.ctx.sync()
        import statistics
        ft = timed_windows(sched, dt, steps, st.ctx.sync, dist, torch, min_seconds=fresh_seconds, restore=restore)
        fresh = dict(window_stats(ft, steps), seconds_per_window=statistics.median(ft))
    p = st.positions()
    assert np.isfinite(p).all(), "non-finite positions after the run"
    st.close()
    return elapsed, timings, world, window_stats(times, steps), pipe, fresh, marks


def run_sharded(gpe, torch, dist, rank, world_size, n_per_gpu, steps, warmup, gravity, device, headline_steps=HEADLINE_STEPS):
    """N > 1: one shard per GPU (gpu-physics-engine_amd/sharded.py): block ownership, one-block ghost band and
    migration over RCCL point-to-point every step.  Weak scaling: every rank fills its own rectangle of the
    world with n_per_gpu particles at the reference density."""
    import numpy as np
    sharded = importlib.import_module("gpu-physics-engine_amd.sharded")
    px, py = sharded._factor(world_size)
    w1 = gpe.scenes.world_for(n_per_gpu)
    world = (w1[0] * px, w1[1] * py)
    cs = np.float32(gpe.scenes.REF_RADIUS) * np.float32(2.2)
    dec = sharded.Decomposition(world, cs, world_size, grid=(px, py))
    x0, y0, x1, y1 = dec.rect_units(rank)
    rng = np.random.default_rng(0x5EED + rank)
    pos = np.empty((n_per_gpu, 2), np.float32)
    pos[:, 0] = x0 + rng.random(n_per_gpu, dtype=np.float32) * np.float32((x1 - x0) * 0.999999)
    pos[:, 1] = y0 + rng.random(n_per_gpu, dtype=np.float32) * np.float32((y1 - y0) * 0.999999)
    keep = dec.owner_of(pos) == rank                      # float rounding at the cut: keep what this rank owns
    pos = np.ascontiguousarray(pos[keep])
    rad = np.full(len(pos), gpe.scenes.REF_RADIUS, np.float32)
    gid = np.arange(len(pos), dtype=np.int64) + rank * n_per_gpu
    log("[rank %d] shard: %d particles in [%.1f,%.1f)x[%.1f,%.1f) of %.1f x %.1f" % (rank, len(pos), x0, x1, y0, y1, world[0], world[1]))
    g = (0.0, -9.81) if gravity == "on" else (0.0, 0.0)
    eng = sharded.GpeEngine(pos, rad, gid, world, gravity=g, device=device)
    del pos, rad, gid
    st = sharded.ShardedState(eng, dec, rank)
    dt = 1.0 / 60.0
    sched = Schedule(lambda d, k, every, first: st.run(d, k, resort_every=every, resort_first=first))
    sched.advance(dt, warmup)
    eng.sync()
    eng.ctx.set_profiling(PROFILE_EVERY)
    eng.ctx.reset_timings()
    st.stats = {"migrants": 0, "ghosts": 0, "steps": 0}
    times = timed_windows(sched, dt, steps, eng.sync, dist, torch, min_steps=headline_steps)
    elapsed = sum(times) / len(times)
    timings = eng.ctx.timings()
    _, p, _ = st.owned()
    info = {"owned": st.n_owned, "process_grid": [px, py],
            "exchange": ("device-resident (k_shard.hip pack/unpack, no host sync); segments moved by "
                         + ("grouped ncclSend/ncclRecv inside the library (gpe_shard_run)" if st.transport == "rccl"
                            else "a torch.distributed callback (all_to_all_single, staged through the host under gloo)"))
                        if st.fast else "torch (two host syncs per step)"}
    if st.fast:
        info["ghosts"] = st.n_ghost
    else:
        info["ghosts_per_step"] = st.stats["ghosts"] / max(1, st.stats["steps"])
        info["migrants_per_step"] = st.stats["migrants"] / max(1, st.stats["steps"])
    assert np.isfinite(p).all(), "non-finite positions after the run"
    info["windows"] = window_stats(times, steps)
    eng.close()
    return elapsed, timings, world, info


def cpu_baseline(gpe, n, budget_s=10.0):
    """The CPU oracle (port of the reference algorithm) on this box's host cores: all the threads of the box's CPU
    share (OpenMP over the loops that are GPU threads in the WGSL; same bits as the serial oracle,
    tests/test_oracle_golden.py), with the single-thread rate beside it."""
    from oracle import oracle as orc      # checker / baseline only
    world = gpe.scenes.world_for(n)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=0x5EED)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(16, avail))      # a one-GPU box's CPU share

    def rate(th, budget):
        orc.set_threads(th)
        sim = orc.Sim(pos, rad, orc.default_params(world[0], world[1], 0.5))
        sim.step(1.0 / 60.0, resort=True)     # warm-up step (includes the re-sort)
        steps, t0 = 0, time.perf_counter()
        while True:
            sim.step(1.0 / 60.0, resort=False)
            steps += 1
            el = time.perf_counter() - t0
            if el >= budget or steps >= 1000:
                break
        sim.close()
        return steps / el, steps

    try:
        single, s1 = rate(1, budget_s * 0.5)
        multi, sm = rate(threads, budget_s) if threads > 1 else (single, s1)
    finally:
        orc.set_threads(1)
    return {"value": multi, "unit": "steps/s", "cores": threads, "kind": "port",
            "sample": "%d particles x %d steps of oracle/gpe_oracle.c (reference algorithm: 4N pairs, LSD radix "
                      "sort, chunk count, scan, 4 colour passes, Verlet) on %d OpenMP threads of %d host cores; "
                      "1 thread: %.2f steps/s (%d steps)" % (n, sm, threads, os.cpu_count() or 0, single, s1)}


# ------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without a launcher environment starts its own N ranks
# ------------------------------------------------------------------------------------------------------
def launch_ranks(args):
    """Spawn one child per rank BEFORE this process touches the GPU (a process that has initialised the GPU must
    never fork/exec GPU work on this pool; this parent only waits).  The children inherit stdout: rank 0 prints the
    JSON line.  Returns the exit code: 0 only if every rank exited 0."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "GPE_BENCH_SELF_LAUNCHED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = list(procs)

    def stop_children(signum=None, frame=None):       # exact PIDs of our own children, nothing else
        for q in procs:
            if q.poll() is None:
                q.terminate()
        for q in procs:
            try:
                q.wait(timeout=10)
            except Exception:
                q.kill()
        if signum is not None:
            raise SystemExit(128 + signum)

    import signal
    signal.signal(signal.SIGTERM, stop_children)
    signal.signal(signal.SIGINT, stop_children)
    deadline = time.time() + float(os.environ.get("GPE_BENCH_DEADLINE_S", "3000"))
    while alive:
        time.sleep(0.2)
        if time.time() > deadline:
            log("bench.py: the ranks did not finish within the deadline -- stopping them")
            stop_children()
            return 124
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                log("bench.py: rank %d exited with %d -- stopping the other ranks" % (procs.index(p), code))
                for q in alive:                       # exact PIDs of our own children, nothing else
                    q.terminate()
    if rc != 0:
        for q in procs:
            try:
                q.wait(timeout=10)
            except Exception:
                q.kill()
    return rc


def rehearse(args, rank, world_size):
    """The launcher and the timing protocol with no GPU and no physics (CPU test of `--gpus N`)."""
    import torch
    import torch.distributed as dist
    if os.environ.get("GPE_BENCH_FAIL_RANK") == str(rank):
        raise SystemExit(3)                           # test hook: a rank that never joins
    if world_size > 1:
        dist.init_process_group(backend="gloo")
        joined = dist.get_world_size()
    else:
        joined = 1
    if joined != args.gpus:
        raise SystemExit("bench.py: %d ranks joined, --gpus %d" % (joined, args.gpus))
    if world_size > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    if world_size > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world_size > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "physics_steps_per_sec", "value": None, "unit": "steps/s", "n_gpus": joined,
                          "steps": args.steps, "warmup": args.warmup, "rehearsal": True,
                          "self_launched": os.environ.get("GPE_BENCH_SELF_LAUNCHED") == "1",
                          "note": "launcher rehearsal on CPU (gloo): no GPU, no physics -- not a result"}), flush=True)
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()


def kernel_source_digest():
    """sha256 (16 hex digits) over the sources of the profiled kernels (hash, radix passes, collide: k_native.hip,
    k_onesweep.hip and the shared header): profiles/traffic.json records the digest its PMC numbers were measured on,
    so a stale traffic figure is flagged instead of silently reported."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "gpu-physics-engine_amd", "csrc")
    for f in ("gpe_internal.h", "k_native.hip", "k_onesweep.hip"):
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def load_traffic():
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        return json.load(open(tpath))
    except Exception:
        return {}


def roofline_block(timings, n, mode, tj):
    """The `roofline` object of one workload: the dominant kernel (largest total time among the kernels with an
    algorithmic byte count) priced against HBM, with the two figures that say what actually binds it."""
    roofs = kernel_rooflines(timings, n, mode)
    if not roofs:
        return None, roofs
    name, d = max(roofs.items(), key=lambda kv: kv[1]["total_ms"])
    traffic = tj.get(mode, {}).get(name, {}).get(str(n))
    note = None
    if traffic is not None:
        measured_on = tj.get("_measured_on", {})
        digest = kernel_source_digest()
        if measured_on.get("csrc_sha16") != digest:
            note = ("PMC figures were measured on csrc %s (%s); this build is %s -- re-run scripts/gpu_profile_r03.sh"
                    % (measured_on.get("csrc_sha16"), measured_on.get("commit"), digest))
            log("warning: " + note)
    valu, issue_frac = None, None
    sq = tj.get("sq", {}).get(name, {}).get(str(n))
    if sq and sq.get("GRBM_GUI_ACTIVE"):
        cycles = sq["GRBM_GUI_ACTIVE"] / 8.0                      # the counter is summed over the 8 XCDs
        per_simd = sq["SQ_INSTS_VALU"] / 1024.0                   # 256 CUs x 4 SIMDs
        # SQ_ACTIVE_INST_VALU counts quad-cycles in which a SIMD is executing a VALU instruction: x 4 / (1024 SIMDs x
        # kernel cycles) is the fraction of the kernel during which the VALU pipes are busy -- the roofline this kernel
        # actually sits on; HBM `frac` is the contract's figure.
        issue_frac = round(sq.get("SQ_ACTIVE_INST_VALU", 0) * 4.0 / (1024.0 * cycles), 3)
        # one SIMD issues a wave64 VALU instruction per ~2.2 cycles (fma / integer add) to ~4.1 cycles (compare +
        # select), measured with 4-8 waves per SIMD: profiles/r02/valu_issue_probe.txt
        valu = {"valu_wave_insts_per_launch": sq["SQ_INSTS_VALU"], "kernel_cycles": round(cycles),
                "issue_busy_frac_at_2.2_cycles": round(per_simd * 2.2 / cycles, 3),
                "issue_busy_frac_at_4.1_cycles": round(per_simd * 4.1 / cycles, 3),
                "lanes_active_per_valu_inst": round(sq.get("SQ_THREAD_CYCLES_VALU", 0) / (64.0 * max(1, sq.get("SQ_ACTIVE_INST_VALU", 1))), 3),
                "wave_cycles_waiting_frac": round(sq.get("SQ_WAIT_ANY", 0) / max(1, sq.get("SQ_WAVE_CYCLES", 1)), 3),
                "source": "SQ counters recorded in profiles/traffic.json (rocprofv3 --pmc; not measured by this run)"}
    block = {"bound": "hbm", "kernel": name, "achieved": round(d["GBps"], 1), "peak": HBM_PEAK_GBS,
             "unit": "GB/s", "frac": round(d["GBps"] / HBM_PEAK_GBS, 4), "traffic": traffic,
             "traffic_is": "HBM bytes per launch from the rocprofv3 PMC passes recorded in profiles/traffic.json "
                           "(not measured by this run)" + ("; STALE: " + note if note else ""),
             "bytes_per_launch": d["bytes"], "avg_launch_ms": round(d["avg_ms"], 5),
             "launches": d["calls"], "launches_are": "the launches of every %dth timed step (hipEvent pairs on the "
                                                     "library's stream)" % PROFILE_EVERY,
             "issue_frac": issue_frac,
             "issue_frac_is": "VALU-busy fraction of the kernel: SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel cycles), "
                              "from the same PMC record -- the limit this kernel is closest to (not HBM)",
             "valu_issue": valu}
    return block, roofs


def log_timings(timings, roofs):
    for name, (tot, calls) in sorted(timings.items(), key=lambda kv: -kv[1][0]):
        extra = "  %.0f GB/s algorithmic" % roofs[name]["GBps"] if name in roofs else ""
        log("  %-44s %9.3f ms total  %6d calls  %8.4f ms/call%s" % (name, tot, calls, tot / max(1, calls), extra))


def extra_entry(name, run, n_per_gpu, ngpu, steps, mode, resort_note, tj):
    el, tim, world, wstats, info = run[:5]
    marks = run[6] if len(run) > 6 else None
    sps = steps / el
    roofline, roofs = roofline_block(tim, n_per_gpu, mode, tj)
    log_timings(tim, roofs)
    return {
        "workload": name, "n_gpus": ngpu, "particles_per_gpu": n_per_gpu, "particles_total": n_per_gpu * ngpu,
        "world": [world[0], world[1]], "steps": steps, "schedule": resort_note, "windows": wstats,
        "system_steps_per_sec": round(sps, 3), "shard_steps_per_sec": round(sps * ngpu, 3),
        "ms_per_step": round(1e3 / sps, 4),
        "particle_steps_per_sec": round(sps * n_per_gpu * ngpu, 1),
        "step_algorithmic_GBps": round(ALGO_BYTES_PER_PARTICLE * n_per_gpu * ngpu * sps / 1e9, 1),
        "step_frac_of_hbm_roofline_per_gpu": round(ALGO_BYTES_PER_PARTICLE * n_per_gpu * sps / 1e9 / HBM_PEAK_GBS, 4),
        "roofline": roofline,
        "dominant_kernel": roofline["kernel"] if roofline else None,
        "dominant_kernel_avg_ms": roofline["avg_launch_ms"] if roofline else None,
        "dominant_kernel_GBps": roofline["achieved"] if roofline else None,
        "dominant_kernel_frac": roofline["frac"] if roofline else None,
        "pipeline": info if isinstance(info, dict) and "pipeline" in info else None,
        "sharding": info if isinstance(info, dict) and "pipeline" not in info else None,
        "later_in_the_same_run": marks or None,
    }


def run_soak(gpe, torch, n, mode, device, total=3000, window=100, marks=(250, 500, 750, 1000, 1250, 1500, 2000, 2500)):
    """The 100M gravity-on scene does not stay a fresh cloud: it falls, piles up and is crushed.  Steps/s over
    `window` steps around each mark, so the steady state is reported next to the fresh-cloud figure."""
    import numpy as np
    world = gpe.scenes.world_for(n)
    pos, rad = gpe.scenes.uniform_cloud(n, world, seed=0x5EED)
    st = gpe.State(pos, rad, world=world, gravity=(0.0, -9.81), device=device,
                   mode=gpe.MODE_NATIVE if mode == "native" else gpe.MODE_COMPAT)
    del pos, rad
    dt = 1.0 / 60.0
    out, done = [], 0

    def advance(k):
        nonlocal done
        # keep the global schedule "re-sort at step 0 and every 240th step"
        while k > 0:
            to_resort = (RESORT_EVERY - done % RESORT_EVERY) % RESORT_EVERY
            if to_resort == 0:
                st.run(dt, 1, resort_every=0, resort_first=True)
                done += 1; k -= 1
                continue
            c = min(k, to_resort)
            st.run(dt, c, resort_every=0, resort_first=False)
            done += c; k -= c

    for m in marks:
        advance(m - window // 2 - done)
        st.ctx.sync()
        t0 = time.perf_counter()
        advance(window)
        st.ctx.sync()
        el = time.perf_counter() - t0
        out.append({"around_step": m, "steps": window, "ms_per_step": round(el / window * 1e3, 3),
                    "steps_per_sec": round(window / el, 2), "native_sorts_so_far": st.ctx.pipeline_info()["native_sorts"]})
        log("soak: around step %d: %.3f ms/step" % (m, el / window * 1e3))
    advance(total - done)
    st.ctx.sync()
    pipe = st.ctx.pipeline_info()
    p = st.positions()
    assert np.isfinite(p).all(), "non-finite positions after the soak"
    st.close()
    return {"workload": "%d particles, gravity on, %d steps, re-sort every %d" % (n, total, RESORT_EVERY), "marks": out,
            "pipeline": pipe}


def main():
    args = parse()
    have_launcher_env = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not have_launcher_env:
        raise SystemExit(launch_ranks(args))          # this parent never touches the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d: the JSON would misreport the GPU count" %
                         (world_size, args.gpus))
    if args.rehearse:
        return rehearse(args, rank, world_size)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    # GPE_BENCH_BACKEND=gloo GPE_BENCH_SHARE_GPU=1: rehearse the N > 1 path on a one-GPU box (all ranks on cuda:0,
    # exchange staged through the host); the real run is one rank per GPU over nccl (= RCCL).
    backend = os.environ.get("GPE_BENCH_BACKEND", "nccl")
    if os.environ.get("GPE_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    elif world_size > 1 and torch.cuda.device_count() < world_size:
        raise SystemExit("bench.py: --gpus %d but only %d GPUs are visible" % (world_size, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dist = None
    if world_size > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist_mod.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist_mod.init_process_group(backend=backend)
        dist = dist_mod
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: %d ranks joined, --gpus %d" % (dist.get_world_size(), args.gpus))
    ngpu = dist.get_world_size() if dist is not None else 1          # the ranks that actually run
    gpe = importlib.import_module("gpu-physics-engine_amd")
    gpe._lib.load()

    n = args.particles
    tj = load_traffic()
    # The GPU legs first, back to back (a coarse utilisation sampler then sees them as one busy stretch); the CPU
    # baseline (rank 0, N = 1) last.
    shard_info, pipe, fresh = None, None, None
    if world_size > 1:
        elapsed, timings, world, shard_info = run_sharded(gpe, torch, dist, rank, world_size, n, args.steps,
                                                          args.warmup, args.gravity, local_rank,
                                                          headline_steps=0 if args.single_window else HEADLINE_STEPS)
        wstats = shard_info.pop("windows")
    else:
        elapsed, timings, world, wstats, pipe, fresh, _ = run_workload(
            gpe, torch, dist, rank, world_size, n, args.steps, args.warmup, args.mode, args.gravity, local_rank,
            headline_steps=0 if args.single_window else HEADLINE_STEPS,
            fresh_seconds=0.0 if args.single_window else FRESH_SECONDS)
    # The 100M legs, shaped like BASELINE.json configs[2..4]: gravity on, warm-up 10 steps (the first re-sorts),
    # then timed windows with the re-sort every 240 steps of the run INSIDE them.
    extras = []
    if not args.no_extra and args.extra_particles != n:
        ne, xs = args.extra_particles, args.extra_steps
        sched = ("warm-up 10 steps (first one re-sorts), windows of %d timed steps (median reported), re-sort every %d "
                 "steps of the run" % (xs, RESORT_EVERY))
        if world_size > 1:
            per = max(1, ne // world_size)
            log("extra workload: %d particles in all over %d GPUs (%d per GPU), gravity on ..." % (per * world_size, world_size, per))
            r = run_sharded(gpe, torch, dist, rank, world_size, per, xs, 10, "on", local_rank, headline_steps=0)
            extras.append(("%d particles over %d GPUs (%d per GPU), gravity on (0,-9.81)%s" %
                           (per * world_size, world_size, per, " = BASELINE.json configs[3]" if world_size == 4 else ""),
                           (r[0], r[1], r[2], r[3].pop("windows"), r[3]), per, sched, xs))
            log("extra workload: %d particles per GPU (%d in all), gravity on ..." % (ne, ne * world_size))
            r = run_sharded(gpe, torch, dist, rank, world_size, ne, xs, 10, "on", local_rank, headline_steps=0)
            extras.append(("%d particles per GPU, %d in all, gravity on (0,-9.81)%s" %
                           (ne, ne * world_size, " = BASELINE.json configs[4]" if world_size == 8 else ""),
                           (r[0], r[1], r[2], r[3].pop("windows"), r[3]), ne, sched, xs))
        else:
            log("extra workload: %d particles, gravity on ..." % ne)
            extras.append(("%d particles, gravity on (0,-9.81) = BASELINE.json configs[2]" % ne,
                           run_workload(gpe, torch, None, 0, 1, ne, xs, 10, args.mode, "on", local_rank, headline_steps=0,
                                        tail=tuple(m for m in args.extra_tail if m > 10 + xs)),
                           ne, sched, xs))
    soak = None
    if args.soak and world_size == 1:
        soak = run_soak(gpe, torch, args.extra_particles, args.mode, local_rank)
    cpu = None
    if ngpu == 1 and not args.no_cpu_baseline:
        log("cpu baseline (oracle) ...")
        cpu = cpu_baseline(gpe, min(n, 1_000_000))
        cpu["value"] = round(cpu["value"], 4)
    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    steps_per_s = args.steps / elapsed
    mode = "native" if world_size > 1 else args.mode
    roofline, roofs = roofline_block(timings, n, mode, tj)
    log_timings(timings, roofs)

    # Whole-job value: every GPU advances one shard of `particles_per_gpu` particles per step, so the job
    # completes n_gpus shard-steps per step (== plain steps/s at n_gpus = 1).
    result = {
        "metric": "physics_steps_per_sec", "value": round(steps_per_s * ngpu, 3), "unit": "steps/s",
        "n_gpus": ngpu, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32+u32",
        "data": "synthetic",
        "timed_steps_total": wstats["timed_steps_total"],
        "timing": dict(wstats, value_is="ALL timed steps / ALL timed seconds: windows of `steps` steps (each bracketed by "
                                        "barrier + synchronize, max over ranks) FOLLOW each other from step `warmup` of the run "
                                        "until %d steps have been timed -- the Morton re-sort every %d steps of the run and "
                                        "every radix sort a step needs fall inside the windows" % (HEADLINE_STEPS, RESORT_EVERY)),
        "config": {"workload": "%d particles per GPU, gravity %s, world %.1f x %.1f, radius 0.5, uniform random "
                               "(BASELINE.json configs[1] at 1M per GPU)" % (n, args.gravity, world[0], world[1]),
                   "particles_per_gpu": n, "particles_total": n * ngpu, "mode": mode, "resort_every": RESORT_EVERY,
                   "dt": 1.0 / 60.0,
                   "value_is": "steps/s of the whole system x n_gpus shards (one shard = particles_per_gpu particles)",
                   "system_steps_per_sec": round(steps_per_s, 3),
                   "particle_steps_per_sec": round(steps_per_s * n * ngpu, 1),
                   "step_algorithmic_GBps": round(ALGO_BYTES_PER_PARTICLE * n * ngpu * steps_per_s / 1e9, 2),
                   "step_frac_of_hbm_roofline": round(ALGO_BYTES_PER_PARTICLE * n * steps_per_s / 1e9 / HBM_PEAK_GBS, 5),
                   "reference_frame_ms_rx6800xt_incl_render": 3.66 if n == 1_000_000 else None,
                   "launched_by": "bench.py itself (child ranks spawned before any GPU call)"
                                  if os.environ.get("GPE_BENCH_SELF_LAUNCHED") == "1" else
                                  ("torch.distributed.run / external launcher" if world_size > 1 else "single process"),
                   "pipeline": pipe,
                   "sharding": shard_info},
        "roofline": roofline,
    }
    if fresh is not None:
        # the figure rounds 1-3 reported as `value`: the same `steps` steps right behind the warm-up, restarted from a
        # snapshot again and again (no radix sort inside a window, rosters freshly written)
        result["value_fresh_cloud"] = round(args.steps / fresh["seconds_per_window"], 3)
        result["fresh_cloud"] = dict({k: v for k, v in fresh.items() if k != "seconds_per_window"},
                                     value_is="median window of `steps` steps, every window restarted from the state after "
                                              "`warmup` steps (%.2f s of them)" % FRESH_SECONDS)
    if cpu is not None:
        result["cpu_baseline"] = cpu
    if extras:
        result["extra_workloads"] = [extra_entry(e[0], e[1], e[2], ngpu, e[4], mode, e[3], tj) for e in extras]
    if soak is not None:
        result["soak"] = soak
    print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
