#!/usr/bin/env python3
"""Round 6: the reference's OWN mapping from a policy command to a bicycle model, held against this env's spec.

The ROS nodes that deploy the shipped agents (`ros_agent/agents/dreamer/src/agent.py:96-119`, `acme/src/agent.py:84-96`,
`sb3/src/agent.py:84-96`) feed the simulator-trained policy's command into an Ackermann drive message - on the real car and in
f1tenth_simulator, a kinematic single-track model.  What they do with it is the only in-tree statement of what a command of
the reference's simulator means on a bicycle model:

    steering = 0 - action['steering'] * 0.6 * 0.42      # "working better in hardware"           (dreamer node :111)
    steering = 0 - action['steering'] * 0.7 * 0.42      # "working better in simulation"         (dreamer node :112)
    steering = 0 - action['steering'] * 0.4 * 0.42      # acme / sb3 nodes :91
    self._steering = self._steering * 1/6 + steering * 5/6      # "lowpass in simulation"        (:119, acme :95)
    decisions at most every 0.079 s ("limit to approx. 10Hz", :59-60)
    motor: target speed += 0.065 if action['motor'] >= 0.5 else -= 0.05 per decision, clipped to [1.7, 5] m/s   (:96-107)

i.e. (i) the NEGATION against ROS's left-positive steering angle: a positive simulator command steers RIGHT; (ii) the effective
full lock on a bicycle model is 0.4-0.7 x 0.42 = 0.168-0.294 rad, not 0.42; (iii) a motor command of 0.5 means "hold the speed".
This tool runs the shipped agents under the reference's test protocol A (tools/analysis/eval_protocol.py) on the C oracle with

    spec        this build's spec: lock 0.19 rad, the command fed straight in, a decision every 4 sub-steps (action_repeat 4)
    sim-0.7     the authors' simulation mapping as an agent-side filter: the env's lock set to 0.42 rad (the nominal scale the
                node multiplies by), the command scaled by 0.7 (-> 0.294 rad) and low-passed 1/6 : 5/6, a decision every 10
                sub-steps (the node's ~10 Hz) - and the same at action_repeat 4 and 8 for comparison
    hw-0.6      scale 0.6 (0.252 rad), the hardware low-pass with its scan-dependent weight (:114-118)
    acme-0.4    scale 0.4 (0.168 rad), low-pass 1 / (1.5 v + 0.5) of the acme / sb3 nodes (:93-94)
    + motor     any of them with the node's bang-bang speed law driving this env's throttle (m = target speed / 5 m/s)

and prints, per agent and mapping: episodes without a wall contact, progress in laps (published best beside it), mean speed.
    python tools/analysis/deployment_mapping.py [--episodes 8] > profiles/r06_b_deployment_mapping.txt"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import c_oracle                                    # noqa: E402
from oracle import racecar_oracle as ro                       # noqa: E402
from oracle.deployment_port import NodeFilter                # noqa: E402
from oracle.dreamer_policy_port import DreamerPolicy          # noqa: E402
from racing_dreamer_amd import spec                           # noqa: E402
from racing_dreamer_amd.track_assets import load_track        # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
PUBLISHED = {"austria": 1.31, "columbia": 2.23, "treitlstrasse_v2": 2.00}        # dreamer/plotting/structs.py:26-28
NOMINAL = spec.MAX_STEER                                      # 0.42: ros_agent/models/dreamer/racing_dreamer.py:14


def run(track_name, agent, n, repeat, seconds, mapping, seed=0):
    t = load_track(track_name)
    cfg = ro.OracleConfig(num_envs=n, auto_reset=False, remap_actions=True, laps=10, time_limit=180.0)
    env = c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg, threads=8)
    policy = DreamerPolicy(np.load(os.path.join(GOLDEN, f"dreamer_policy_{agent}.npz")), sample=True, seed=seed)
    filt = None
    if mapping is not None:
        kind, motor_law = mapping
        c_oracle.set_dynamics(steer_gain=-NOMINAL)
        filt = NodeFilter(n, kind, motor_law, max_vel=spec.MAX_VEL)
    out = env.reset(mode=ro.RESET_GRID, seed=1)
    state = policy.initial(n)
    alive = np.ones(n, bool)
    prog, time, wall_end, dist = np.zeros(n), np.zeros(n), np.zeros(n, bool), np.zeros(n)
    for _ in range(int(round(seconds / (repeat * spec.DT)))):
        scan = np.asarray(out["lidar"]).reshape(n, ro.N_BEAMS)
        action, state = policy.act(scan, state)
        if filt is not None:
            action = filt(action, scan.astype(np.float64))
        out = env.step(action, repeat=repeat)
        done = np.asarray(out["done"]).reshape(n) != 0
        prog[alive] = np.asarray(out["progress_total"]).reshape(n)[alive]
        time[alive] = np.asarray(out["time"]).reshape(n)[alive]
        dist[alive] += np.asarray(out["speed"]).reshape(n)[alive] * repeat * spec.DT
        wall_end |= alive & done & (np.asarray(out["wall_collision"]).reshape(n) != 0)
        alive &= ~done
        if not alive.any():
            break
    c_oracle.set_dynamics()
    return prog, wall_end, dist / np.maximum(time, 1e-9)


MAPPINGS = [
    ("spec: lock 0.19, direct, repeat 4", None, 4),
    ("sim-0.7: 0.294 rad, 5/6 low-pass, ~10 Hz (repeat 10)", ("sim", False), 10),
    ("sim-0.7 at repeat 8", ("sim", False), 8),
    ("sim-0.7 at repeat 4 (the training cadence)", ("sim", False), 4),
    ("0.7 x 0.42 direct feed, repeat 4 (lock 0.294 alone)", ("direct:0.7", False), 4),
    ("hw-0.6: 0.252 rad, hardware low-pass, repeat 10", ("hw", False), 10),
    ("0.6 x 0.42 direct feed, repeat 4 (lock 0.252 alone)", ("direct:0.6", False), 4),
    ("acme-0.4: 0.168 rad, low-pass 1/(1.5 v + 0.5), repeat 10", ("acme", False), 10),
    ("0.4 x 0.42 direct feed, repeat 4 (lock 0.168 alone)", ("direct:0.4", False), 4),
    ("sim-0.7 + the node's speed law, repeat 10", ("sim", True), 10),
    ("hw-0.6 + the node's speed law, repeat 10", ("hw", True), 10),
    ("0.4 x 0.42 direct + the node's speed law, repeat 10", ("direct:0.4", True), 10),
]
AGENTS = (("austria", "austria"), ("austria", "columbia"), ("treitlstrasse", "Treitlstrasse_3-U_v3"), ("treitlstrasse", "columbia"))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--episodes", type=int, default=8)
    a = ap.parse_args()
    print("protocol A (grid start, 40 s, episode over at a wall) on the C oracle; cell = clean episodes / median progress [laps] / mean speed")
    print("published best (dreamer/plotting/structs.py:26-28): austria 1.31, columbia 2.23; treitlstrasse 2.00 is for v2 (the shipped agent is a v3 agent)")
    print(f"{'mapping':58s} | " + " | ".join(f"{ag + ' on ' + tr:>34s}" for ag, tr in AGENTS))
    for name, mapping, repeat in MAPPINGS:
        cells = []
        for agent, track in AGENTS:
            p, wall, v = run(track, agent, a.episodes, repeat, 40.0, mapping)
            cells.append(f"{int((~wall).sum())}/{a.episodes} {np.median(p):.2f} laps {v.mean():.2f} m/s")
        print(f"{name:58s} | " + " | ".join(f"{c:>34s}" for c in cells), flush=True)
