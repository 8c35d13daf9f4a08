"""bench.py's cpu_baseline leg: the C oracle timed on this host's cores on a bounded sample of the
benchmark workload (random-action rollouts with auto-reset, LiDAR every sub-step).  TEST/BENCH
INFRASTRUCTURE - a reported baseline, never the product path."""
from __future__ import annotations

import os
import time

from . import c_oracle
from . import racecar_oracle as ro


def run(track, cars=1, occupancy=False, repeat=1, n_envs=0, target_s=12.0):
    cores = len(os.sched_getaffinity(0))
    n_envs = n_envs or 512 * cores
    cfg = ro.OracleConfig(num_envs=n_envs, cars_per_env=cars, auto_reset=True, render_occupancy=occupancy)
    env = c_oracle.COracleEnv(track.occ, track.drivable, track.progress, track.centerline, track.origin,
                              track.resolution, cfg, threads=cores)
    env.reset(mode=ro.RESET_RANDOM, seed=0)
    for k in range(2):                                   # warm-up
        env.step(env.random_actions(1, k), repeat=repeat)
    steps, t0 = 0, time.perf_counter()
    while True:
        env.step(env.random_actions(1, 2 + steps), repeat=repeat)
        steps += 1
        dt = time.perf_counter() - t0
        if dt >= target_s or steps >= 2000:
            break
    return {"value": n_envs * steps * repeat / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n_envs} envs x {steps} steps of the same workload (track {track.name}, {cars} car/env, "
                      f"{'lidar+occupancy' if occupancy else 'lidar'}, repeat {repeat}) in {dt:.1f} s; plain-C oracle "
                      f"(oracle/racecar_oracle.c, gcc -O2) on {cores} threads; the upstream PyBullet env is not "
                      f"installable here and cannot be timed",
            "os_cpu_count": os.cpu_count()}


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
