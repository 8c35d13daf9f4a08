"""TEST INFRASTRUCTURE (not shipped, not on the product path): what the reference's deployment nodes do between a policy's
command and the drive message of a bicycle model - the only in-tree statement of what a command of the reference's simulator
means on such a model.

Restated from
  * ros_agent/agents/dreamer/src/agent.py:59-60    decisions at most every 0.079 s ("limit to approx. 10Hz")
  * ros_agent/agents/dreamer/src/agent.py:86-90    the scan the hardware low-pass looks at: clipped at 4 m, 3-beam mean, maximum of
                                                    the 300 beams around the middle
  * ros_agent/agents/dreamer/src/agent.py:96-107   motor: target speed += a / 1000 (0.065) if action['motor'] >= 0.5 else -= b / 1000
                                                    (0.05) per decision, clipped to [1.7, 5] m/s
  * ros_agent/agents/dreamer/src/agent.py:111-119  steering = 0 - action['steering'] * k * 0.42 with k = 0.6 ("working better in
                                                    hardware") / 0.7 ("working better in simulation"); low-pass (20 - val) / 20 :
                                                    val / 20 with val = 18 - 3 forward_max (hardware), 1/6 : 5/6 (simulation)
  * ros_agent/agents/acme/src/agent.py:84-95, ros_agent/agents/sb3/src/agent.py:84-95
                                                    k = 0.4; low-pass (div - 1) / div : 1 / div with div = 1.5 speed + 0.5
The numbers themselves are the committed fixture tests/golden/deployment_mapping.json (made by
tests/golden/make_golden_deployment_mapping.py from the files above); `NodeFilter` reads them from there.

Sign: the drive message's steering angle is ROS's (positive = LEFT, counter-clockwise); the nodes NEGATE the policy's command,
so a positive command of the reference's simulator steers RIGHT - this env's STEER_GAIN < 0 (wheel angle counter-clockwise
positive).  Scale: the effective full lock on a bicycle model is k x 0.42 = 0.168 .. 0.294 rad; the spec's 0.19 lies inside."""
from __future__ import annotations

import json
import os

import numpy as np

FIXTURE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "deployment_mapping.json")


def mapping():
    with open(FIXTURE) as f:
        return json.load(f)


class NodeFilter:
    """Per decision and env: policy command in [-1, 1]^2 (the actor's output BEFORE ReduceActionSpace: motor, steering) ->
    the same kind of array for an env whose full lock is the NOMINAL 0.42 rad and whose actions are remapped (dream.py:138).
    kind: 'sim' (k 0.7, 1/6 : 5/6), 'hw' (k 0.6, scan-dependent low-pass), 'acme' (k 0.4, speed-dependent low-pass),
    'direct:<k>' (scale only).  motor_law: the dreamer node's bang-bang target speed drives this env's throttle, which settles
    at max_velocity x m (spec.MAX_VEL)."""

    def __init__(self, n, kind, motor_law=False, max_vel=5.0):
        m = mapping()
        self.kind, self.motor_law, self.max_vel = kind, bool(motor_law), float(max_vel)
        d = m["dreamer_node"]
        self.scale = {"sim": d["scale_simulation"], "hw": d["scale_hardware"], "acme": m["acme_node"]["scale"]}.get(kind)
        if self.scale is None:
            self.scale = float(kind.split(":")[1])
        self.sign = d["sign"]                                    # -1: the negation in front of action['steering']
        self.lp_sim = d["lowpass_simulation"]                    # [1/6, 5/6]
        self.up, self.down = d["speed_step_up"], d["speed_step_down"]
        # (the acme / sb3 nodes integrate the motor command with two ROS parameters that are not in the tree, agent.py:84; the
        # dreamer node's bang-bang stands in, inside THEIR speed clip)
        self.v_lo, self.v_hi = m["acme_node"]["speed_clip"] if kind == "acme" else d["speed_clip"]
        self.threshold = d["motor_threshold"]
        self.steer = np.zeros(n)                 # the node's self._steering as a fraction of 0.42 rad, in the ENV's sign (+ = right)
        self.speed = np.full(n, self.v_lo)       # the node's self._motor: a target speed [m/s]

    def __call__(self, raw, scan_m):
        motor = (raw[:, 0].astype(np.float64) + 1.0) / 2.0 * (1.0 - 0.005) + 0.005       # racing_dreamer.py:53-59
        cmd = raw[:, 1].astype(np.float64) * self.scale
        if self.kind == "sim":
            self.steer = self.steer * self.lp_sim[0] + cmd * self.lp_sim[1]
        elif self.kind == "hw":
            r = np.clip(np.asarray(scan_m, np.float64), None, 4.0)
            r[:, 3:-3] = (r[:, 3:-3] + r[:, 2:-4] + r[:, 4:-2]) / 3.0
            val = 18.0 - r[:, 540 - 150:540 + 150].max(1) * 3.0
            self.steer = self.steer * (20.0 - val) / 20.0 + cmd * val / 20.0
        elif self.kind == "acme":
            div = self.speed * 1.5 + 0.5
            self.steer = self.steer * (div - 1.0) / div + cmd / div
        else:
            self.steer = cmd
        out = np.asarray(raw, np.float32).copy()
        out[:, 1] = np.clip(self.steer, -1.0, 1.0)
        if self.motor_law or self.kind == "acme":
            self.speed = np.clip(np.where(motor < self.threshold, self.speed - self.down, self.speed + self.up), self.v_lo, self.v_hi)
        if self.motor_law:
            m = self.speed / self.max_vel
            out[:, 0] = np.clip((m - 0.005) / 0.995 * 2.0 - 1.0, -1.0, 1.0)
        return out
