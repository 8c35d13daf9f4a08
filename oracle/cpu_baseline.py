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
