#!/usr/bin/env python3
"""obs_type lidar_occupancy_reference on EVERY compiled map (GPU box): the HIP kernels against the C oracle's exact render
(oracle/racecar_oracle.c, oc_patch_exact_range) from `n` poses per map - half near the centre line, the rest anywhere on the grid
and beyond it, an eighth on cell corners with axis-aligned and diagonal headings; and the binary32 estimate of the sampling pass
against the binary64 sum on every pixel of those renders (rc_selftest_exact_estimate).  python tools/exact_all_tracks.py [n]"""
import ctypes
import os
import struct
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle, racecar_oracle as ro  # noqa: E402
from racing_dreamer_amd import _lib as L  # noqa: E402
from racing_dreamer_amd.batched_env import BatchedRaceEnv  # noqa: E402
from racing_dreamer_amd.track_assets import available_tracks, load_track  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 768
total = bad_total = 0
tot, worst = [0, 0, 0], 0.0
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
    chk = (ctypes.c_uint64 * 4)()                                     # both sums for every pixel (rc_selftest_exact_estimate)
    L.check(env._lib.rc_selftest_exact_estimate(env._h, chk))
    err = struct.unpack("f", struct.pack("I", chk[3] & 0xffffffff))[0]
    tot = [a + b for a, b in zip(tot, (chk[0], chk[1], chk[2]))]
    worst = max(worst, err)
    env.close()
    bad = int((got != want).reshape(n, -1).any(1).sum())
    total += n
    bad_total += bad
    print(f"{name:28s} {t.height:5d} x {t.width:<5d} {n} poses: {bad} patches differ; estimate: {chk[1] / max(chk[0], 1) * 100:.4f} % of the pixels in the band, "
          f"{chk[2]} decided wrongly, largest error {err:.2e}", flush=True)
print(f"{total} patches on {k + 1} maps in {time.time() - t0:.0f} s: {bad_total} differ; the binary32 estimate: {tot[0]} pixels inside the array, "
      f"{tot[1]} ({tot[1] / max(tot[0], 1) * 100:.4f} %) sent to the binary64 sum by the band, {tot[2]} the estimate alone would have got wrong, "
      f"largest |estimate - binary64 sum| {worst:.2e} (bound 1.1e-4, band 1e-3)")
sys.exit(1 if bad_total or tot[2] else 0)
