#!/usr/bin/env python3
"""The scan's duration against `ray_split` at small batches (run on the GPU box): python tools/split_sweep.py [--track columbia]

Production kernel, launch-attached events, mean of --steps launches per point.  The rule in racecar_abi.hip (rc_step's launch
set-up) is read off this table."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from racing_dreamer_amd.batched_env import BatchedRaceEnv  # noqa: E402
from racing_dreamer_amd.track_assets import load_track  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--track", default="columbia")
    ap.add_argument("--envs", default="256,1024,2048,4096,8192,16384,32768")
    ap.add_argument("--cars", type=int, default=1)
    ap.add_argument("--splits", default="0,1,2,3,4,5,6,8,9,17")
    ap.add_argument("--steps", type=int, default=300)
    a = ap.parse_args()
    splits = [int(x) for x in a.splits.split(",")]
    print(f"scan duration [us] on {a.track}, {a.cars} car(s) per env; columns: ray_split (0 = the library's rule); then dynamics")
    print(f"{'envs':>7s} " + " ".join(f"{s:>7d}" for s in splits) + "     dyn")
    for n in [int(x) for x in a.envs.split(",")]:
        env = BatchedRaceEnv(load_track(a.track), n, a.cars, auto_reset=True)
        env.reset(mode="random", seed=0)
        for k in range(60):
            env.step_random(0, k)
        row, dyn = [], 0.0
        for s in splits:
            env.debug_set("ray_split", s)
            for k in range(10):
                env.step_random(0, 100 + k)
            env.reset_kernel_times()
            env.set_profiling(True)
            for k in range(a.steps):
                env.step_random(0, 200 + k)
            torch.cuda.synchronize()
            t = env.kernel_times()
            env.set_profiling(False)
            row.append(t["rc_raycast_kernel"]["avg_ms"] * 1e3)
            dyn = t["rc_dynamics_kernel"]["avg_ms"] * 1e3
        env.debug_set("ray_split", 0)
        print(f"{n:>7d} " + " ".join(f"{v:7.2f}" for v in row) + f" {dyn:7.2f}", flush=True)
        del env


if __name__ == "__main__":
    main()
