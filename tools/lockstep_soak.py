#!/usr/bin/env python3
"""Long lockstep run of the HIP env against the C oracle (GPU box): python tools/lockstep_soak.py [steps] [envs]

The parity tests compare every output of every step, but over tens of steps (the NumPy oracle is slow); a soak that
checks only the last scan says little.  This compares EVERY output of EVERY step over thousands of steps with the C
port of the oracle (oracle/racecar_oracle.c, itself held bit-identical to the NumPy oracle by tests/test_c_oracle.py) on the
configurations where rare events live: finish-line crossings and `lap > laps` (a follow-the-gap driver that laps), resets of all
three modes behind every kind of episode end, 2-4 cars with inter-car rays / collisions / clash fallbacks of the start law,
secondary agents' n-step reward windows, the time-limit wrapper with remapped actions, the max_speed task, lidar_occupancy.
Exit code 1 at the first difference (prints where)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import EXACT_FLOAT, EXACT_INT  # noqa: E402
from oracle import c_oracle, racecar_oracle as ro  # noqa: E402
from racing_dreamer_amd import spec  # noqa: E402
from racing_dreamer_amd.batched_env import BatchedRaceEnv  # noqa: E402
from racing_dreamer_amd.track_assets import load_track  # noqa: E402

TASK_IDS = {None: -1, "maximize_progress": 0, "max_speed": 1, "n_step_progress": 2}
CONFIGS = [
    # name, track, cars, policy, kwargs of both envs, reset mode, repeat
    ("random actions + render", "austria", 1, "random", dict(obs="lidar_occupancy"), "random", 1),
    ("a driver that laps, one lap per episode", "circle", 1, "ftg", dict(laps=1), "random", 4),
    ("a driver that laps, two laps, grid start", "columbia", 1, "ftg", dict(laps=2), "grid", 4),
    ("two cars, close starts", "treitlstrasse_v2", 2, "forward", dict(), "random_ball", 2),
    ("four cars, secondary agents on n_step_progress, a driver", "columbia", 4, "ftg",
     dict(car_tasks=["maximize_progress", "n_step_progress", "n_step_progress", "n_step_progress"], n_steps=7), "random_ball", 2),
    ("time limit + remapped actions", "barcelona", 1, "random", dict(time_limit_steps=50, remap=True), "grid", 4),
    ("three cars, max_speed task", "gbr", 3, "forward", dict(task="max_speed"), "random", 1),
    ("no termination on contact", "Treitlstrasse_3-U_v3", 2, "forward", dict(terminate_on_collision=False, time_limit=6.0), "random_ball", 3),
    # round 6: the start law where bins are skipped and headings held (boxes on the track, one-cell corridors), three cars
    ("narrow map: skipped bins, held headings", "torino", 3, "forward", dict(), "random_ball", 2),
    ("a lobby, not a loop: 26 unusable bins, grid start of four", "levinelobby", 4, "forward", dict(time_limit=3.0), "grid", 2),
    # round 6: the reference's own lidar_occupancy arithmetic in the step (binary64 on both sides; a twentieth of the envs and a tenth
    # of the steps: the C port renders 9 ms per car and thread)
    ("random actions + the reference render", "columbia", 1, "random", dict(obs="lidar_occupancy_reference", scale=(0.1, 0.05)), "random", 2),
]


def run(name, track_name, cars, policy, kw, mode, repeat, steps, n):
    t = load_track(track_name)
    occ = {None: False, "lidar_occupancy": True, "lidar_occupancy_reference": "reference"}[kw.get("obs")]
    if "scale" in kw:
        steps, n = max(int(steps * kw["scale"][0]), 20), max(int(n * kw["scale"][1]), 16)
    common = dict(laps=kw.get("laps", 10), time_limit=kw.get("time_limit", 180.0), terminate_on_collision=kw.get("terminate_on_collision", True),
                  time_limit_steps=kw.get("time_limit_steps", 0), n_steps=kw.get("n_steps", 10))
    env = BatchedRaceEnv(t, n, cars, obs_type=kw.get("obs") or "lidar", auto_reset=True, task=kw.get("task", "maximize_progress"),
                         remap_actions=kw.get("remap", False), car_tasks=kw.get("car_tasks"), **common)
    cfg = ro.OracleConfig(num_envs=n, cars_per_env=cars, auto_reset=True, render_occupancy=occ,
                          task=spec.TASK_MAX_SPEED if kw.get("task") == "max_speed" else 0, remap_actions=kw.get("remap", False),
                          car_tasks=None if kw.get("car_tasks") is None else [TASK_IDS[x] for x in kw["car_tasks"]], **common)
    ora = c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg, threads=min(16, os.cpu_count() or 1))
    if occ == "reference":
        ora.set_frame(t)
    dv = env.reset(mode=mode, seed=5)
    ov = ora.reset(mode=spec.RESET_MODES[mode], seed=5)
    names = EXACT_INT + EXACT_FLOAT + (["lidar_occupancy"] if occ else [])

    def check(k):
        torch.cuda.synchronize()
        for f in names:
            d = dv[f].cpu().numpy().reshape(-1)
            o = np.asarray(ov[f]).reshape(-1)
            if d.dtype != o.dtype:
                o = o.astype(d.dtype)
            if not np.array_equal(d.view(np.uint8), o.view(np.uint8)):
                bad = np.nonzero(d != o)[0]
                per = d.size // (n * cars)
                print(f"  DIFFERENCE at step {k} in `{f}`: {bad.size} elements; first at car {bad[0] // per if bad.size else '?'} "
                      f"(element {bad[0] % per if bad.size else '?'}): device {d[bad[:4]]} oracle {o[bad[:4]]}", flush=True)
                return False
        return True

    if not check("reset"):
        return False
    t0 = time.perf_counter()
    dones = laps_done = walls = opps = 0
    for k in range(steps):
        if policy == "ftg":
            act = env.follow_the_gap().reshape(-1, 2).cpu().numpy().copy()        # the device's follow-the-gap law drives both
            if k % 97 < 3:                                                         # ... with a swerve now and then: contacts happen too
                act[:, 1] = ro.random_actions(3, k, n * cars)[:, 1]
        else:
            act = ora.random_actions(3, k)
            if policy == "forward":
                act[:, 0] = np.abs(act[:, 0])
        dv = env.step(torch.from_numpy(act).to(env.device), repeat=repeat)
        ov = ora.step(act, repeat=repeat)
        if not check(k):
            return False
        dones += int(ov["done"].sum())
        laps_done += int((ov["lap"] > 1).sum() > 0)
        walls += int(ov["wall_collision"].sum())
        opps += int(ov["opponent_collision"].sum())
    dt = time.perf_counter() - t0
    print(f"  identical over {steps} steps x {n} envs x {cars} car(s), repeat {repeat} ({dt:.0f} s): {dones} episode ends, steps with a car past its first lap "
          f"{laps_done}, wall contacts {walls}, car contacts {opps}, max lap seen {int(ov['lap'].max())}", flush=True)
    env.close()
    return True


def run_mixed(steps, n):
    """BASELINE configs[4]'s track mix in ONE batch (MixedTrackEnv: one launch per kernel over all blocks, rc_step_group) against
    one C oracle per block with that block's global env offset."""
    from racing_dreamer_amd.batched_env import MixedTrackEnv
    names, cars = ["columbia", "austria", "barcelona"], 2
    sizes = [n // 3, n // 3, n - 2 * (n // 3)]
    env = MixedTrackEnv(names, sizes, cars_per_env=cars, auto_reset=True)
    oras = []
    for nm, m, (a, b) in zip(names, sizes, env.blocks):
        t = load_track(nm)
        oras.append(c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution,
                                        ro.OracleConfig(num_envs=m, cars_per_env=cars, auto_reset=True, first_env=a), threads=min(16, os.cpu_count() or 1)))
    dv = env.reset(mode="random_ball", seed=21)
    ovs = [o.reset(mode=spec.RESET_RANDOM_BALL, seed=21) for o in oras]
    names_out = EXACT_INT + EXACT_FLOAT
    t0, dones = time.perf_counter(), 0
    for k in range(-1, steps):
        if k >= 0:
            act = ro.random_actions(3, k, n * cars)
            act[:, 0] = np.abs(act[:, 0])
            dv = env.step(torch.from_numpy(act).to(env.device).view(n, cars, 2), repeat=2)
            ovs = [o.step(act[cars * a:cars * b], repeat=2) for o, (a, b) in zip(oras, env.blocks)]
            dones += sum(int(ov["done"].sum()) for ov in ovs)
        torch.cuda.synchronize()
        for (a, b), ov, nm in zip(env.blocks, ovs, names):
            for f in names_out:
                d = dv[f][a:b].cpu().numpy().reshape(-1)
                o = np.asarray(ov[f]).reshape(-1).astype(d.dtype)
                if not np.array_equal(d.view(np.uint8), o.view(np.uint8)):
                    bad = np.nonzero(d != o)[0]
                    print(f"  DIFFERENCE at step {k} in `{f}` of block [{a}, {b}) on {nm}: {bad.size} elements, first {bad[:4]}: device {d[bad[:4]]} oracle {o[bad[:4]]}", flush=True)
                    return False
    print(f"  identical over {steps} steps x {n} envs x {cars} cars in three track blocks, repeat 2 ({time.perf_counter() - t0:.0f} s): {dones} episode ends", flush=True)
    env.close()
    return True


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    only = sys.argv[3] if len(sys.argv) > 3 else None
    ok = True
    for c in CONFIGS:
        if only is not None and only not in c[0]:
            continue
        print(f"{c[0]}: {c[1]}, {c[2]} car(s), policy {c[3]}, reset {c[5]}", flush=True)
        if not run(*c, steps=steps, n=n):
            ok = False
            break
    if ok and (only is None or only in "track mix"):
        print("track mix in one batch (MixedTrackEnv): columbia / austria / barcelona, 2 cars, policy forward, reset random_ball", flush=True)
        ok = run_mixed(steps, n)
    print("lockstep soak ok" if ok else "lockstep soak FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
