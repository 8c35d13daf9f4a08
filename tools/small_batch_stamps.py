#!/usr/bin/env python3
"""Where the scan's time goes at a SMALL batch (run on the GPU box): python tools/small_batch_stamps.py [--envs 4096] [--track columbia]

At 65 536 cars the scan is throughput-bound (tools/scan_stamps.py); at 4 096 it is not: its 23 us are the same whether a car's
17 rounds go to 1, 3 or 17 waves (racecar_abi.hip, the split's measurements).  This tool runs the instrumented scan
(`rc_debug_scan_stamps`) at a small batch for several splits and prints (the shader clocks of two CUs are not
comparable - s_memtime is a per-CU counter -, so slots 6 / 7 hold the chip-wide 100 MHz clock at entry and flush): when waves ENTER
relative to the first entry (the dispatcher's ramp), how long a wave lives and in which phase, when the last flush is issued and
how many waves are in flight along the way - next to the production kernel's duration on the launch-attached events.  Analysis only."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from racing_dreamer_amd.batched_env import BatchedRaceEnv  # noqa: E402
from racing_dreamer_amd.track_assets import load_track  # noqa: E402


def pct(x, q):
    return float(np.percentile(x, q))


def one(env, envs, split, steps, n_marks=12):
    env.debug_set("ray_split", split)
    for k in range(8):
        env.step_random(0, 100 + k)
    # the production kernel's own duration at this split (launch-attached events)
    env.reset_kernel_times()
    env.set_profiling(True)
    for k in range(steps):
        env.step_random(0, 200 + k)
    torch.cuda.synchronize()
    prof = {k: v["avg_ms"] for k, v in env.kernel_times().items()}
    env.set_profiling(False)
    n_waves = envs * split
    st = env.debug_scan_stamps(n_waves)
    env.step_random(0, 999)
    torch.cuda.synchronize()
    s = st.cpu().numpy().astype(np.int64)
    env.debug_scan_stamps(0)
    s = s[s[:, 0] != 0]
    print(f"--- split {split}: {len(s)} waves of {n_waves}; production kernel {prof.get('rc_raycast_kernel', float('nan')) * 1e3:.2f} us, dynamics {prof.get('rc_dynamics_kernel', float('nan')) * 1e3:.2f} us (launch-attached events, mean of {steps})")
    # the chip-wide clock (100 MHz: 10 ns ticks) places the waves in the kernel's life; the per-CU shader clock times the phases
    t0 = s[:, 6].min()
    ent = (s[:, 6] - t0) * 0.01
    end = (s[:, 7] - t0) * 0.01
    life_us = (s[:, 7] - s[:, 6]) * 0.01
    life = s[:, 21] - s[:, 0]
    ghz = life.sum() / max(life_us.sum(), 1e-9) / 1e3
    span = float(end.max())
    print(f"    in-kernel span, first entry -> last flush issued: {span:.2f} us; shader clock = {ghz:.2f} GHz (lifetimes on both clocks)")

    def line(name, v, unit="us"):
        print(f"    {name:<52s} mean {v.mean():8.2f}  p10 {pct(v, 10):8.2f}  p50 {pct(v, 50):8.2f}  p90 {pct(v, 90):8.2f}  max {v.max():8.2f} {unit}")

    cyc = 1.0 / (ghz * 1e3)
    line("entry after the first entry (dispatch ramp)", ent)
    line("wave lifetime (entry -> flush issued)", life * cyc)
    line("  entry -> car state arrived", (s[:, 1] - s[:, 0]) * cyc)
    line("  -> first-trip line staged, first round prepared", (s[:, 2] - s[:, 1]) * cyc)
    line(f"  its rounds ({17 / split:.1f} on average)", (s[:, 20] - s[:, 2]) * cyc)
    line("  flush issued", (s[:, 21] - s[:, 20]) * cyc)
    line("flush issued after the first entry", end)
    # who is resident when: waves in flight at 1 us marks
    marks = np.arange(0.0, span, max(span / n_marks, 0.5))
    print("    waves in flight at " + " ".join(f"{m:.1f}" for m in marks) + " us: " + " ".join(str(int(((ent <= m) & (end > m)).sum())) for m in marks))
    cu = (s[:, 24] >> 8) & 0xff | ((s[:, 5] & 15) << 8)
    print(f"    distinct (XCD, SE, CU) seen: {len(np.unique(cu))}; waves per CU: mean {len(s) / len(np.unique(cu)):.1f}, max {np.bincount(np.unique(cu, return_inverse=True)[1]).max()}")
    trips = s[:, 22]
    print(f"    wave-level trips per wave: mean {trips.mean():.1f}; rounds' cycles per trip {(s[:, 20] - s[:, 2]).sum() / max(trips.sum(), 1):.0f}")
    return span, prof.get("rc_raycast_kernel", float("nan"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--track", default="columbia")
    ap.add_argument("--splits", default="1,3,6,17")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--marks", type=int, default=12)
    a = ap.parse_args()
    env = BatchedRaceEnv(load_track(a.track), a.envs, 1, auto_reset=True)
    env.reset(mode="random", seed=0)
    for k in range(40):
        env.step_random(0, k)
    print(f"{a.envs} envs on {a.track}; shader cycles (s_memtime)")
    for split in [int(x) for x in a.splits.split(",")]:
        span, ms = one(env, a.envs, split, a.steps, a.marks)
        print(f"    => {span:.2f} us of in-kernel span (instrumented build) against {ms * 1e3:.2f} us of the production kernel on the events")
    env.debug_set("ray_split", 0)


if __name__ == "__main__":
    main()
