#!/usr/bin/env python3
"""obs_type lidar_occupancy_reference on EVERY compiled map (GPU box): the HIP kernels against the C oracle's exact render
(oracle/racecar_oracle.c, oc_patch_exact_range) from `n` poses per map - half near the centre line, the rest anywhere on the grid
and beyond it, an eighth on cell corners with axis-aligned and diagonal headings.  python tools/exact_all_tracks.py [n]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle, racecar_oracle as ro  # noqa: E402
from racing_dreamer_amd.batched_env import BatchedRaceEnv  # noqa: E402
from racing_dreamer_amd.track_assets import available_tracks, load_track  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 768
total = bad_total = 0
t0 = time.time()
for k, name in enumerate(sorted(available_tracks())):
    t = load_track(name)
    rng = np.random.default_rng(1000 + k)
    x = t.origin[0] + rng.uniform(-0.5, t.width * 0.05 + 0.5, n)
    y = t.origin[1] + rng.uniform(-0.5, t.height * 0.05 + 0.5, n)
    th = rng.uniform(-np.pi, np.pi, n)
    h = n // 2
    cl = t.centerline[rng.integers(0, len(t.centerline), h)]
    x[:h], y[:h] = cl[:, 0] + rng.uniform(-0.5, 0.5, h), cl[:, 1] + rng.uniform(-0.5, 0.5, h)
    q = n // 8
    x[:q] = t.origin[0] + rng.integers(0, t.width, q) * 0.05
    y[:q] = t.origin[1] + rng.integers(0, t.height, q) * 0.05
    th[:q] = rng.choice([0.0, np.pi / 2, np.pi, -np.pi / 2, np.pi / 4, -np.pi / 4, 3 * np.pi / 4], q)
    poses = np.stack([x, y, th], 1).astype(np.float32)
    cfg = ro.OracleConfig(num_envs=n, cars_per_env=1, render_occupancy="reference")
    ora = c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg, threads=16)
    ora.set_frame(t)
    ora.reset()
    ora.arr["x"][:], ora.arr["y"][:], ora.arr["theta"][:] = poses[:, 0], poses[:, 1], poses[:, 2]
    ora.arr["st"][:], ora.arr["ct"][:] = ro.sincos32(poses[:, 2])
    ora.arr["fresh"][:] = 0
    ora._observe()
    want = ora.patch.reshape(n, 64, 64)
    env = BatchedRaceEnv(t, n, 1, obs_type="lidar_occupancy_reference")
    env.reset()
    got = env.set_pose(poses)["lidar_occupancy"]
    torch.cuda.synchronize()
    got = got.cpu().numpy().reshape(n, 64, 64)
    env.close()
    bad = int((got != want).reshape(n, -1).any(1).sum())
    total += n
    bad_total += bad
    print(f"{name:28s} {t.height:5d} x {t.width:<5d} {n} poses: {bad} patches differ; mean drivable share of a patch {want.mean():.3f}", flush=True)
print(f"{total} patches on {k + 1} maps in {time.time() - t0:.0f} s: {bad_total} differ")
sys.exit(1 if bad_total else 0)
