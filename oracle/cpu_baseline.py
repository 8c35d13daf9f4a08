"""bench.py's cpu_baseline leg: the CPU oracle timed on this host's cores on a bounded sample of the benchmark
workload (random-action rollouts with auto-reset, LiDAR every sub-step).  TEST/BENCH INFRASTRUCTURE - a reported
baseline, never the product path.

What is timed: the plain-C port on ONE thread and on as many threads as the process really gets, in the
same leg, with the measured speed-up between them; the vectorised NumPy port at B = 4 096 (one step); and the B = 1 step
of BASELINE.json configs[0].  "As many as the process really gets": a container's CPU quota does not show in
sched_getaffinity (round 2 reported 256 "cores" on a lease that ran 11), so the share is read from the cgroup files where
they exist AND measured - a fixed spin on one thread against the same spin on every thread at once.
"""
from __future__ import annotations

import math
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import c_oracle
from . import racecar_oracle as ro


def cgroup_cpu_quota():
    """CPU quota of this process's cgroup in cores (float), or None when there is none / it cannot be read."""
    try:                                                    # cgroup v2
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            return float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:                                                    # cgroup v1
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            quota = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = float(f.read())
        if quota > 0:
            return quota / period
    except (OSError, ValueError):
        pass
    return None


def measured_cpu_share(threads, iterations=200_000_000):
    """How many cores `threads` concurrent spinners really get: threads x t(one spinner) / t(all spinners); the better
    of two tries (freshly started threads take the scheduler a while to spread over the cores)."""
    lib = c_oracle.load()
    lib.oc_spin(1000)
    t0 = time.perf_counter()
    lib.oc_spin(iterations)
    t1 = time.perf_counter() - t0
    if threads <= 1:
        return 1.0, t1
    best = 0.0
    with ThreadPoolExecutor(threads) as pool:
        list(pool.map(lib.oc_spin, [iterations // 4] * threads))       # threads started and spread
        for _ in range(2):
            t0 = time.perf_counter()
            list(pool.map(lib.oc_spin, [iterations] * threads))
            best = max(best, threads * t1 / (time.perf_counter() - t0))
    return best, t1


def effective_cores():
    """(cores_effective, detail dict): min of the affinity mask, the cgroup quota and the measured share."""
    affinity = len(os.sched_getaffinity(0))
    quota = cgroup_cpu_quota()
    cap = affinity if quota is None else max(1, min(affinity, int(math.ceil(quota))))
    share, _ = measured_cpu_share(min(cap, 64))
    if cap > 64 and share > 0.9 * 64:                       # 64 spinners all ran in parallel: probe the full width
        share, _ = measured_cpu_share(cap)
    eff = max(1, min(cap, int(round(share))))
    return eff, {"sched_getaffinity": affinity, "os_cpu_count": os.cpu_count(),
                 "cgroup_cpu_quota_cores": quota, "measured_parallel_share_cores": round(share, 2)}


def _time_rollout(env, repeat, target_s, max_steps=2000):
    env.reset(mode=ro.RESET_RANDOM, seed=0)
    for k in range(2):                                       # warm-up
        env.step(env.random_actions(1, k), repeat=repeat, outputs=False)
    steps, t0 = 0, time.perf_counter()
    while True:
        env.step(env.random_actions(1, 2 + steps), repeat=repeat, outputs=False)
        steps += 1
        dt = time.perf_counter() - t0
        if dt >= target_s or steps >= max_steps:
            return steps, dt


def run(track, cars=1, occupancy=False, repeat=1, n_envs=0, target_s=9.0):
    cores, detail = effective_cores()
    n_envs = n_envs or 256 * cores
    cfg = ro.OracleConfig(num_envs=n_envs, cars_per_env=cars, auto_reset=True, render_occupancy=occupancy)
    make = lambda threads: c_oracle.COracleEnv(track.occ, track.drivable, track.progress, track.centerline, track.origin,
                                               track.resolution, cfg, threads=threads)
    s1, d1 = _time_rollout(make(1), repeat, target_s / 3.0)
    v1 = n_envs * s1 * repeat / d1
    if cores > 1:
        sn, dn = _time_rollout(make(cores), repeat, target_s)
        vn = n_envs * sn * repeat / dn
    else:
        sn, dn, vn = s1, d1, v1
    out = {"value": vn, "unit": "env-steps/s", "cores": cores, "cores_effective": cores, "kind": "port",
           "one_thread_value": v1, "speedup_all_over_one_thread": vn / v1,
           "parallel_efficiency": vn / v1 / cores,
           "sample": f"{n_envs} envs x {sn} steps of the same workload (track {track.name}, {cars} car/env, "
                     f"{'lidar+occupancy' if occupancy else 'lidar'}, repeat {repeat}) in {dn:.1f} s on {cores} threads, "
                     f"and x {s1} steps in {d1:.1f} s on one; plain-C oracle (oracle/racecar_oracle.c, gcc -O2), env "
                     f"ranges handed to a thread pool; the upstream PyBullet env is not installable here and cannot be timed",
           "cpu_share": detail}
    return out


def run_numpy_batch(track, n_envs=4096):
    """the vectorised NumPy oracle at B = 4 096, single process - ONE step of the same workload (a step takes
    seconds: the reset's own scan is skipped, the state it would produce is installed directly)."""
    cfg = ro.OracleConfig(num_envs=n_envs, auto_reset=True)
    env = ro.OracleRaceEnv(track.occ, track.drivable, track.progress, track.centerline, track.origin, track.resolution, cfg)
    env.seed, env.mode = 0, ro.RESET_RANDOM
    env._reset_envs(np.arange(n_envs))                       # reset() without its observation pass
    act = ro.random_actions(1, 0, n_envs)
    t0 = time.perf_counter()
    env.step(act)
    dt = time.perf_counter() - t0
    return {"value": n_envs / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{n_envs} envs x 1 step (track {track.name}, lidar) in {dt:.1f} s; vectorised NumPy oracle "
                      f"(oracle/racecar_oracle.py), one process"}


def run_single_env(track_name="columbia", steps=3000):
    """BASELINE.json configs[0] - 1 env, columbia, obs_type=lidar, CPU step() - as a timing: the oracle stepped one env
    at a time (B = 1, one core), the scalar C port and the vectorised NumPy port, random actions with auto-reset."""
    from racing_dreamer_amd.track_assets import load_track
    t = load_track(track_name)
    out = {"workload": f"configs[0]: 1 env, {track_name}, obs_type=lidar, CPU step() (B = 1, one core)"}
    cfg = ro.OracleConfig(num_envs=1, auto_reset=True)
    for name, make, n in (("c_port_steps_per_s", lambda: c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin,
                                                                              t.resolution, cfg, threads=1), steps),
                          ("numpy_port_steps_per_s", lambda: ro.OracleRaceEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin,
                                                                              t.resolution, cfg), max(steps // 10, 50))):
        env = make()
        env.reset(mode=ro.RESET_RANDOM, seed=0)
        acts = [ro.random_actions(1, k, 1) for k in range(n)]
        env.step(acts[0])
        t0 = time.perf_counter()
        for a in acts:
            env.step(a)
        out[name] = n / (time.perf_counter() - t0)
    return out
