#!/usr/bin/env python3
"""Round 5: ALL FOUR shipped Dreamer checkpoints as probes of the integrator's free parameters.

Rounds 3-4 calibrated handedness and steering lock with two agents (austria_dreamer, treitlstrasse_dreamer: G10).  The reference
ships two more (treitlstrasse_dreamer_20210220 / _20210224, trained with the lidar_occupancy reconstruction); one of them drives
at 3 m/s - the speed the published 2.00 laps on treitlstrasse imply - and turns into a wall at 0.28 of the lap in the spec's env.
This runs every agent under the reference's test protocol (tools/analysis/eval_protocol.py, protocol A: grid start, action_repeat
4, 40 s, episode over at a wall) on the C oracle over a grid of (top speed at full throttle, steering lock, steering rate) and
prints, per point and agent: episodes that end without a wall contact, progress in laps, mean speed.
    python tools/analysis/agent_calibration.py [--episodes 8] > profiles/r05_j_agent_calibration.txt"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import eval_protocol as ep                      # noqa: E402
from oracle import c_oracle                     # noqa: E402

AGENTS = (("austria", "austria"), ("treitlstrasse", "treitlstrasse_v2"), ("treitlstrasse_20210220", "treitlstrasse_v2"),
          ("treitlstrasse_occupancy", "treitlstrasse_v2"))
PUBLISHED = {"austria": 1.31, "treitlstrasse_v2": 2.00}


def point(max_vel, lock, rate, accel, episodes):
    c_oracle.set_dynamics(accel_max=accel, drag=accel / max_vel, max_vel=max_vel, steer_gain=-lock, steer_step=rate * 0.01)
    cells = []
    for agent, track in AGENTS:
        a = ep.run_episodes(track, agent, episodes, repeat=4, max_agent_steps=1000, laps=10)
        clean = int((a["ended"] == "limit").sum())
        cells.append(f"{clean}/{episodes} {np.median(a['progress']):.2f} laps {a['mean_speed'].mean():.2f} m/s")
    c_oracle.set_dynamics()
    return cells


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--episodes", type=int, default=8)
    ap.add_argument("--grid", default="coarse")
    a = ap.parse_args()
    print("clean episodes / median progress / mean speed under protocol A; published best: austria 1.31, treitlstrasse_v2 2.00")
    print(f"{'top speed':>9s} {'lock':>5s} {'rate':>5s} {'accel':>5s} | " + " | ".join(f"{ag:>28s}" for ag, _ in AGENTS))
    grid = [(5.0, 0.19, 3.2, 4.0)]
    if a.grid == "coarse":
        grid += [(mv, 0.19, 3.2, 4.0) for mv in (4.5, 4.25, 4.0, 3.75)]
        grid += [(5.0, lk, 3.2, 4.0) for lk in (0.17, 0.21, 0.23, 0.26)]
        grid += [(5.0, 0.19, rt, 4.0) for rt in (1.6, 6.4)]
        grid += [(5.0, 0.19, 3.2, ac) for ac in (3.0, 5.0)]
        grid += [(4.25, lk, 3.2, 4.0) for lk in (0.21, 0.23)]
    for mv, lk, rt, ac in grid:
        cells = point(mv, lk, rt, ac, a.episodes)
        print(f"{mv:9.2f} {lk:5.2f} {rt:5.1f} {ac:5.1f} | " + " | ".join(f"{c:>28s}" for c in cells), flush=True)
