#!/usr/bin/env python3
"""The production step on every compiled track at 65 536 envs (GPU box): python tools/scan_all_tracks.py
Scan / dynamics duration on the launch-attached events, the scan's algorithmic bytes over the HBM peak, grid size."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from racing_dreamer_amd.batched_env import BatchedRaceEnv  # noqa: E402
from racing_dreamer_amd.track_assets import available_tracks, load_track  # noqa: E402

n, steps = 65536, 200
print(f"{'track':38s} {'cells':>12s} {'scan [ms]':>10s} {'frac':>6s} {'dynamics':>9s} {'M env-steps/s':>14s}")
rows = []
for name in available_tracks():
    t = load_track(name)
    env = BatchedRaceEnv(t, n, 1, auto_reset=True)
    torch.cuda.set_stream(env.stream)
    env.reset(mode="random", seed=0)
    for k in range(150):
        env.step_random(0, k)
    env.reset_kernel_times()
    env.set_profiling(True)
    for k in range(steps):
        env.step_random(0, 1000 + k)
    env.stream.synchronize()
    kt = env.kernel_times()
    env.set_profiling(False)
    scan, dyn = kt["rc_raycast_kernel"]["avg_ms"], kt["rc_dynamics_kernel"]["avg_ms"]
    frac = (4 * 1080 + 16) * n / (scan * 1e-3) / 8e12
    rows.append((name, scan, frac))
    print(f"{name:38s} {t.height:5d} x {t.width:4d} {scan:10.4f} {frac:6.3f} {dyn:9.4f} {n / (scan + dyn) / 1e3:14.1f}", flush=True)
    env.close()
s = np.array([r[1] for r in rows])
print(f"{len(rows)} tracks: scan {s.min():.4f} ({rows[int(s.argmin())][0]}) .. {s.max():.4f} ms ({rows[int(s.argmax())][0]}), median {np.median(s):.4f}; "
      f"fraction of the HBM peak {min(r[2] for r in rows):.3f} .. {max(r[2] for r in rows):.3f}")
