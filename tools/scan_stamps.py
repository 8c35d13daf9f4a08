#!/usr/bin/env python3
"""Where a wave of the default scan spends its life (run on the GPU box): python tools/scan_stamps.py [--envs N] [--track T]

Runs the benchmark workload with the instrumented build of the one-wave-per-car kernel (`rc_debug_scan_stamps`): every
wave stamps the shader clock (s_memtime) at fixed points - entry, car state arrived, first round prepared, end of each of
its 17 rounds, rounds done, flush issued - and counts its wave-level trips.  Prints the mean / percentiles of each phase
in shader cycles and the cycles per wave-level trip; analysis only.  Read shares and ratios off it, not times: the instrumented
build holds 6 waves per SIMD instead of 8 and reads the clock six times per round - its launch takes 1.9 x the production kernel's
(316 us against 166 at 65 536 cars, tools/small_batch_stamps.py); and `s_memtime` is a per-CU counter (stamps of two CUs are not
comparable: slots 6 / 7 hold the chip-wide 100 MHz clock for that)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from racing_dreamer_amd.batched_env import BatchedRaceEnv  # noqa: E402
from racing_dreamer_amd.track_assets import load_track  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--track", default="austria")
    ap.add_argument("--steps", type=int, default=40)
    a = ap.parse_args()
    env = BatchedRaceEnv(load_track(a.track), a.envs, 1, auto_reset=True)
    env.reset(mode="random", seed=0)
    for k in range(a.steps):
        env.step_random(0, k)
    st = env.debug_scan_stamps(a.envs)
    env.step_random(0, a.steps)
    torch.cuda.synchronize()
    s = st.cpu().numpy().astype(np.int64)
    env.debug_scan_stamps(0)
    s = s[s[:, 0] != 0]
    life = s[:, 21] - s[:, 0]
    trips = s[:, 22]

    def line(name, x):
        print(f"{name:<44s} mean {x.mean():9.0f}   p10 {np.percentile(x, 10):8.0f}   p50 {np.percentile(x, 50):8.0f}   p90 {np.percentile(x, 90):8.0f}")

    print(f"{len(s)} waves, track {a.track}; shader cycles")
    line("wave lifetime (entry -> flush issued)", life)
    line("  entry -> car state arrived", s[:, 1] - s[:, 0])
    line("  -> first-trip line staged, round 0 prepared", s[:, 2] - s[:, 1])
    line("  17 rounds", s[:, 20] - s[:, 2])
    line("  flush (5 stores issued)", s[:, 21] - s[:, 20])
    line("wave-level trips per car", trips)
    line("  of which took the exact path", s[:, 23])
    print(f"cycles of the 17 rounds per wave-level trip: {(s[:, 20] - s[:, 2]).sum() / trips.sum():.0f}")
    # a linear fit: round time = a + b * trips needs trips per round; approximate from the per-car totals
    t_total = (s[:, 20] - s[:, 2]).astype(np.float64)
    A = np.stack([np.ones_like(t_total), trips.astype(np.float64)], 1)
    coef, *_ = np.linalg.lstsq(A, t_total, rcond=None)
    print(f"least squares over cars: rounds time = {coef[0]:.0f} + {coef[1]:.0f} x trips  (i.e. {coef[0] / 17:.0f} per round + {coef[1]:.0f} per trip)")
    tot = (s[:, 20] - s[:, 2]).astype(np.float64).sum()
    for name, slot in (("wait for the previous round's loads", 27), ("prepare the next round", 28), ("traversal", 29), ("inter-car returns, transform", 3), ("range to LDS", 4), ("round loop control", 30)):
        print(f"  phase '{name}': {s[:, slot].sum() / tot * 100:5.1f} % of the rounds' time, {s[:, slot].mean() / 17:7.0f} cycles per round")
    # per round: time against that round's own trips
    nib = np.stack([(s[:, 25] >> (4 * i)) & 15 for i in range(16)] + [s[:, 26] & 15], axis=1).astype(np.float64)
    x = nib.reshape(-1)
    print("trips per round: mean %.2f; share of rounds with 1..7 trips: %s" % (x.mean(), " ".join(f"{(x == n).mean() * 100:.1f}%" for n in range(1, 8))))
    # traversal time of a car against its trips
    c3, *_ = np.linalg.lstsq(A, s[:, 29].astype(np.float64), rcond=None)
    print(f"least squares over cars: traversal time = {c3[0]:.0f} + {c3[1]:.0f} x trips  (i.e. {c3[0] / 17:.0f} per round + {c3[1]:.0f} per trip)")
    span = s[:, 21].max() - s[:, 0].min()
    print(f"first entry -> last flush: {span} cycles; sum of lifetimes / span = {life.sum() / span:.0f} waves in flight on average")
    hw = s[:, 24]
    print(f"distinct HW_ID values seen: {len(np.unique(hw))}")


if __name__ == "__main__":
    main()
