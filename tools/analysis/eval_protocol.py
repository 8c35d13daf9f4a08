#!/usr/bin/env python3
"""G12 - this build's env under the reference's own EVALUATION protocols, beside the only task-performance numbers the reference
holds (dreamer/plotting/structs.py:23-28: best max-progress Dreamer 1.31 / 2.23 / 2.00 laps on austria / columbia /
treitlstrasse_v2; best model-free 0.36-0.38 on austria).   python tools/analysis/eval_protocol.py [--episodes N] [--backend c|hip]

Protocol A (dreamer/dream.py:55,58,120-121, make_test_env): scenario max_progress (laps 10, task time limit 180 s, an episode
ends at the first wall contact), reset(mode='grid'), action_repeat 4, TimeLimit(4000 / 4 = 1000 agent steps = 40 s); the figure is
`lap + progress - 1` at the end of the episode (dreamer/wrappers.py:218, tools.py:195).
Protocol B (dreamer/evaluations/run_evaluation.py:76, make_env.py:9-15): scenario eval (laps 1), grid start, action_repeat 8, no
TimeLimit wrapper: the episode ends when the lap is done, at a wall, or after 180 s.
The agents are the reference's shipped checkpoints (ros_agent/checkpoints/*_dreamer; fixtures tests/golden/dreamer_policy_*.npz)
run by oracle/dreamer_policy_port.py with the reference's own sampling (posterior sample, best-of-100 action)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import racecar_oracle as ro                       # noqa: E402
from oracle.dreamer_policy_port import DreamerPolicy          # noqa: E402
from racing_dreamer_amd.track_assets import load_track        # noqa: E402

PUBLISHED = {"austria": 1.31, "columbia": 2.23, "treitlstrasse_v2": 2.00}        # dreamer/plotting/structs.py:26-28
GOLDEN = os.path.join(ROOT, "tests", "golden")


def make_env(track_name, n, laps, backend):
    t = load_track(track_name)
    if backend == "hip":
        from racing_dreamer_amd.batched_env import BatchedRaceEnv
        return BatchedRaceEnv(t, n, 1, auto_reset=False, remap_actions=True, laps=laps, time_limit=180.0), True
    from oracle import c_oracle
    cfg = ro.OracleConfig(num_envs=n, auto_reset=False, remap_actions=True, laps=laps, time_limit=180.0)
    return c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg, threads=8), False


def run_episodes(track_name, agent, n, repeat, max_agent_steps, laps, backend="c", sample=True, seed=0):
    """n episodes side by side (one env each, the policy's random stream differs per env).  Returns per episode: progress in
    laps (`lap + progress - 1`), sim time [s], ended by 'wall' | 'laps' | 'limit', mean speed."""
    env, hip = make_env(track_name, n, laps, backend)
    policy = DreamerPolicy(np.load(os.path.join(GOLDEN, f"dreamer_policy_{agent}.npz")), sample=sample, seed=seed)
    get = (lambda o, k: o[k].cpu().numpy()) if hip else (lambda o, k: np.asarray(o[k]))
    out = env.reset(mode="grid" if hip else ro.RESET_GRID, seed=1)
    state = policy.initial(n)
    alive = np.ones(n, bool)
    prog, time, how, dist = np.zeros(n), np.zeros(n), np.array(["limit"] * n, dtype=object), np.zeros(n)
    for k in range(max_agent_steps):
        scan = get(out, "lidar").reshape(n, ro.N_BEAMS)
        action, state = policy.act(scan, state)
        if hip:
            import torch
            out = env.step(torch.from_numpy(action).to(env.device).view(n, 1, 2), repeat=repeat)
        else:
            out = env.step(action, repeat=repeat)
        done = get(out, "done").reshape(n) != 0
        p = get(out, "progress_total").reshape(n)
        t = get(out, "time").reshape(n)
        wall = get(out, "wall_collision").reshape(n) != 0
        upd = alive
        prog[upd], time[upd] = p[upd], t[upd]
        dist[upd] += get(out, "speed").reshape(n)[upd] * repeat * 0.01
        ended = alive & done
        how[ended & wall] = "wall"
        how[ended & ~wall] = "laps"
        alive &= ~done
        if not alive.any():
            break
    if hip:
        env.close()
    return dict(progress=prog, time=time, ended=how, mean_speed=dist / np.maximum(time, 1e-9))


def table(episodes, backend):
    rows = []
    for agent, track in (("austria", "austria"), ("treitlstrasse", "treitlstrasse_v2"), ("austria", "columbia"), ("treitlstrasse", "columbia"),
                         ("austria", "barcelona")):
        a = run_episodes(track, agent, episodes, repeat=4, max_agent_steps=1000, laps=10, backend=backend)
        b = run_episodes(track, agent, episodes, repeat=8, max_agent_steps=2250, laps=1, backend=backend)
        rows.append((agent, track, a, b))
        pa, pb = a["progress"], b["progress"]
        lap_done = b["ended"] == "laps"
        print(f"{agent:14s} on {track:17s} A: progress {pa.mean():.2f} laps (min {pa.min():.2f}, max {pa.max():.2f}; {int((a['ended'] == 'wall').sum())}/{episodes} ended at a wall; "
              f"{a['mean_speed'].mean():.2f} m/s)  published best {PUBLISHED.get(track, float('nan')):.2f} | "
              f"B: {int(lap_done.sum())}/{episodes} laps completed" + (f" in {b['time'][lap_done].mean():.1f} s" if lap_done.any() else "")
              + f", progress {pb.mean():.2f}", flush=True)
    return rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--episodes", type=int, default=10)
    ap.add_argument("--backend", default="c", choices=["c", "hip"])
    a = ap.parse_args()
    table(a.episodes, a.backend)
