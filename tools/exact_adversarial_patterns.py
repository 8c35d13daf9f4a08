#!/usr/bin/env python3
"""The exact render's binary32 estimate where the spline coefficients are as large as a 0 / 1 image can make them (GPU box): the
drivable bitmap of columbia replaced by a checkerboard, stripes, noise, 2 x 2 blocks; 2 048 poses each; rc_selftest_exact_estimate
computes the estimate AND the binary64 sum for every pixel.  python tools/exact_adversarial_patterns.py"""
import ctypes as C, dataclasses, struct, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from racing_dreamer_amd import _lib as L
from racing_dreamer_amd.batched_env import BatchedRaceEnv
from racing_dreamer_amd.track_assets import load_track, pack_words
t0 = load_track("columbia")
yy, xx = np.mgrid[0:t0.height, 0:t0.width]
rng = np.random.default_rng(3)
for name, drv in (("checkerboard", (yy + xx) % 2 == 0), ("stripes", xx % 2 == 0), ("noise", rng.random((t0.height, t0.width)) < 0.5), ("2x2 blocks", ((yy // 2) + (xx // 2)) % 2 == 0)):
    drv = drv.copy(); drv[0, :] = drv[-1, :] = drv[:, 0] = drv[:, -1] = False
    t = dataclasses.replace(t0, drv_words=pack_words(drv, t0.pitch))
    n = 2048
    poses = np.stack([t.origin[0] + rng.uniform(0, t.width * 0.05, n), t.origin[1] + rng.uniform(0, t.height * 0.05, n), rng.uniform(-np.pi, np.pi, n)], 1).astype(np.float32)
    env = BatchedRaceEnv(t, n, 1, obs_type="lidar_occupancy_reference"); env.reset(); env.set_pose(poses)
    out = (C.c_uint64 * 4)(); L.check(env._lib.rc_selftest_exact_estimate(env._h, out)); env.close()
    err = struct.unpack("f", struct.pack("I", out[3] & 0xffffffff))[0]
    print(f"{name:14s} pixels {out[0]}  in the band {out[1]} ({out[1] / out[0] * 100:.3f} %)  decided wrongly {out[2]}  largest error {err:.2e}")
