#!/usr/bin/env python3
"""Parity of whatever libracecar_hip.so is installed, at a batch large enough for the production scan path (one wave per car,
no split): 16 384 envs on austria, three random-action steps, every LiDAR row against the C oracle.  Used by the A/B sessions
(tools/ab_bench.sh times variants; this says whether a variant still computes the spec).  GPU box."""
import sys
import os
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import c_oracle, racecar_oracle as ro            # noqa: E402
from racing_dreamer_amd.batched_env import BatchedRaceEnv    # noqa: E402
from racing_dreamer_amd.track_assets import load_track       # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "?"
n = 16384
t = load_track("austria")
env = BatchedRaceEnv(t, n, 1, auto_reset=True)
ora = c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, ro.OracleConfig(num_envs=n, auto_reset=True), threads=16)
d = env.reset(mode="random", seed=3)
o = ora.reset(mode=ro.RESET_RANDOM, seed=3)
worst, bad = 0.0, 0
for k in range(4):
    a, b = d["lidar"].cpu().numpy().reshape(n, -1), np.asarray(o["lidar"]).reshape(n, -1)
    worst = max(worst, float(np.abs(a - b).max()))
    bad += int((a != b).sum())
    act = ro.random_actions(7, k, n)
    d = env.step(torch.from_numpy(act).cuda())
    o = ora.step(act)
print(f"ab_check {name}: scan kernel {env.scan_kernel_name()}: {bad} of {4 * n * 1080} ranges differ from the C oracle, max |diff| {worst:.3g} m", flush=True)
env.close()
