"""Plain-Python restatement of the reference's env wrapper semantics - TEST INFRASTRUCTURE, NOT PRODUCT.

The HIP path folds these wrappers into the kernels (action remap, action repeat, time limit) or
into the output arena (speed, the Collect transition record).  This port is the checker for that
fused behaviour; it is PINNED against golden vectors captured from the reference's own wrapper
code (tests/golden/wrappers_golden.npz, made by tests/golden/make_golden.py):

  reduce_action ............ ReduceActionSpace._normalize      dreamer/wrappers.py:128-130
  action_repeat_dreamer .... ActionRepeat.step                 dreamer/wrappers.py:107-116
  action_repeat_baselines .. ActionRepeat.step                 baselines/racing/environment/single_agent.py:31-40
  TimeLimit ................ TimeLimit.step/reset              dreamer/wrappers.py:147-158
  speed .................... RaceCarWrapper.step               dreamer/wrappers.py:66
  flat_action_bounds ....... RaceCarWrapper.action_space       dreamer/wrappers.py:55-60
  Collect .................. Collect.step/reset/_convert       dreamer/wrappers.py:210-250
  max_speed_reward ......... MaximizeSpeed.reward              baselines/racing/environment/tasks.py:6-15
  normalize_obs ............ NormalizeObservations.observation baselines/racing/environment/single_agent.py:92-99
  flatten_clip_action ...... Flatten.step                      baselines/racing/environment/single_agent.py:55-58
  preprocess_lidar ......... tools.preprocess                  dreamer/tools.py:274
"""
from __future__ import annotations

import math

import numpy as np


def reduce_action(action, low=(0.005, -1.0), high=(1.0, 1.0)):
    low, high = np.array(low), np.array(high)
    return (action + 1) / 2 * (high - low) + low


def action_repeat_dreamer(step_fn, agent_ids, action, amount):
    """Up to `amount` inner steps; stops as soon as ANY agent is done; rewards summed per agent."""
    obs, info = None, None
    dones = {a: False for a in agent_ids}
    total = {a: 0.0 for a in agent_ids}
    n = 0
    while n < amount and not any(dones.values()):
        obs, rewards, dones, info = step_fn(action)
        total = {a: total[a] + rewards[a] for a in agent_ids}
        n += 1
    return obs, total, dones, info, n


def action_repeat_baselines(step_fn, action, n):
    """First step unconditional, then up to n-1 more; break after adding the reward when done."""
    obs, reward, done, info = step_fn(action)
    total, calls = reward, 1
    for _ in range(n - 1):
        obs, reward, done, info = step_fn(action)
        calls += 1
        total += reward
        if done:
            break
    return obs, total, done, info, calls


class TimeLimit:
    def __init__(self, duration):
        self.duration, self._step = duration, None

    def reset(self):
        self._step = 0

    def step(self, dones):
        assert self._step is not None, 'Must reset environment.'
        self._step += 1
        if self._step >= self.duration:
            dones = {a: True for a in dones}
            self._step = None
        return dones


def speed(velocity):
    return np.linalg.norm(np.asarray(velocity)[:3])


def flat_action_bounds(motor_low, motor_high, steering_low, steering_high):
    return np.append(motor_low, steering_low), np.append(motor_high, steering_high)


def convert(value, precision=32):
    value = np.array(value)
    if np.issubdtype(value.dtype, np.floating):
        return value.astype({16: np.float16, 32: np.float32, 64: np.float64}[precision])
    if np.issubdtype(value.dtype, np.signedinteger):
        return value.astype({16: np.int16, 32: np.int32, 64: np.int64}[precision])
    if np.issubdtype(value.dtype, np.uint8):
        return value.astype(np.uint8)
    raise NotImplementedError(value.dtype)


class Collect:
    """Per-agent episode assembly: reset row + one transition per step, cast at episode end."""

    def __init__(self, action_shape=(2,), precision=32):
        self.precision, self.action_shape, self.rows = precision, action_shape, []

    def reset(self, obs):
        row = dict(obs)
        row.update(action=np.zeros(self.action_shape), reward=0.0, discount=1.0, progress=-1.0, time=0.0)
        self.rows = [row]

    def step(self, obs, action, reward, done, info):
        row = {k: convert(v, self.precision) for k, v in obs.items()}
        row.update(action=action, reward=reward, discount=np.array(1 - float(done)),
                   progress=info['lap'] + info['progress'] - 1, time=info['time'])
        self.rows.append(row)
        if done:
            return {k: convert([r[k] for r in self.rows], self.precision) for k in self.rows[0]}
        return None


def max_speed_reward(steering, velocity, wall_collision):
    if wall_collision:
        return -1.0
    return -math.exp(math.fabs(steering) - velocity)


def normalize_obs(obs, low, high):
    return (obs - low) * (1.0 / (high - low))


def flatten_clip_action(action):
    """Flatten.step: clip to [-1, 1]; gym sorts Dict keys, so the flat order is [motor, steering]."""
    a = np.clip(action, -1.0, 1.0)
    return {"motor": a[0:1], "steering": a[1:2]}


def preprocess_lidar(lidar):
    return lidar / 15.0 - 0.5
