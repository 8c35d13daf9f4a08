"""Follow-the-gap prefill agent with the interface dreamer/dream.py:213-215 uses:
``GapFollower().action(obs) -> (motor, steering)`` from ``obs['lidar']`` (1080 beams, 270 deg).

The class the reference imports lives in the external racecar_gym repository (`agents/gap_follower.py`, not
under /root/reference), so this is the classic follow-the-gap law written from its description: clip far
ranges, smooth, zero a safety bubble around the closest return, steer at the centre of the widest run of
beams longer than `gap_range`, slow down when steering hard.  (The reference's own ROS node, ros_agent/agents/follow_the_gap/src/agent.py,
is a more elaborate disparity-extender + PID variant for the real car: `ReferenceGapFollower` below restates THAT law
and is pinned to the node's own outputs.)  Steering is returned normalised to [-1, 1] (max 0.42 rad).  The batched device
versions are BatchedRaceEnv.follow_the_gap() and .follow_the_gap_reference().
"""
import numpy as np


class GapFollower:
    def __init__(self, max_range: float = 6.0, bubble_radius: int = 60, smooth: int = 5, gap_range: float = 2.0,
                 fov_deg: float = 270.0,
                 wheel_max: float = 0.19, straights_speed: float = 0.6, corners_speed: float = 0.3):
        self.max_range, self.bubble, self.smooth, self.gap_range = max_range, bubble_radius, smooth, gap_range
        self.fov = np.radians(fov_deg)
        self.wheel_max = wheel_max             # front-wheel angle at full command; a positive command steers right (spec.STEER_GAIN)
        self.straights_speed, self.corners_speed = straights_speed, corners_speed

    def action(self, obs):
        lidar = np.asarray(obs["lidar"], np.float64).reshape(-1)[-1080:]
        n = lidar.size
        lo, hi = n // 8, n - n // 8                     # ignore the rear-most 1/8 on each side (agent.py:131-133)
        r = np.convolve(np.clip(lidar[lo:hi], 0, self.max_range), np.ones(self.smooth) / self.smooth, "same")
        closest = int(r.argmin())
        r[max(0, closest - self.bubble):closest + self.bubble + 1] = 0.0
        free = np.concatenate([[0], (r > self.gap_range).astype(np.int8), [0]])
        edges = np.diff(free)
        starts, ends = np.nonzero(edges == 1)[0], np.nonzero(edges == -1)[0]
        if len(starts) == 0:
            return 0.0, 0.0
        k = int((ends - starts).argmax())
        best = lo + (starts[k] + ends[k] - 1) / 2.0
        angle = self.fov / 2.0 - best * self.fov / (n - 1)          # beam 0 is at +135 deg, sweep is clockwise
        steering = float(np.clip(-angle / self.wheel_max, -1.0, 1.0))
        motor = self.corners_speed if abs(steering) > 0.35 else self.straights_speed
        return motor, steering


class ReferenceGapFollower:
    """The law of the reference's own follow-the-gap node (ros_agent/agents/follow_the_gap/src/agent.py:128-234) as a host
    agent with the same `action(obs) -> (motor, steering)` interface - for the single-env shim; the batched device form
    is `BatchedRaceEnv.follow_the_gap_reference()`.  Forward arc of +-90 deg clipped at the look-ahead distance,
    disparities (maximum of their 10-degree window, > 9 window medians, > 0.2 m) extended by the vehicle's half-width, heading
    = mean angle of the beams at or above the 83.3rd percentile, steering = 1.4 heading - 0.1 d(heading)/dt within +-24
    deg, the node's speed law.  `dt` = seconds between calls (0.01 s x action_repeat); `reset()` at an episode start drops
    the derivative term's memory.  tests/test_golden_ftg.py checks it against the node's own outputs (golden G9)."""
    LOOKAHEAD = 2.0 * (7.0 ** 2 / (2.0 * 8.26))
    WIDTH = 0.3302 * 1.2
    MAX_STEER = np.deg2rad(24.0)

    def __init__(self, dt: float = 0.04, wheel_max: float = 0.19, max_velocity: float = 5.0, range_max: float = 15.0):
        self.dt, self.wheel_max, self.max_velocity, self.range_max = dt, wheel_max, max_velocity, range_max
        self.previous = None
        self.last = {}

    def reset(self):
        self.previous = None

    @staticmethod
    def _filtered(x, width, pad_mode, reducer):
        half = width // 2
        win = np.lib.stride_tricks.sliding_window_view(np.pad(x, (half, width - 1 - half), mode=pad_mode), width)
        return reducer(win)

    def heading(self, lidar):
        """lidar: 1080 ranges in the env's order (beam 0 at +135 deg, clockwise).  Returns (heading, free distance)."""
        ros = np.asarray(lidar, np.float64).reshape(-1)[-1080:][::-1]          # ROS order: from -135 deg, counter-clockwise
        inc, amin = 1.5 * np.pi / 1079, -0.75 * np.pi
        first, last = int((-np.pi / 2 - amin) / inc), int((np.pi / 2 - amin) / inc)
        angles = np.arange(first, last + 1) * inc + amin
        r = np.clip(ros[first:last + 1], 0.0, self.LOOKAHEAD)
        jump = np.abs(np.diff(r))
        width = int(np.deg2rad(10.0) / inc)
        peak = self._filtered(jump, width, "symmetric", lambda w: w.max(axis=1))
        med = self._filtered(jump, width, "edge", lambda w: np.sort(w, axis=1)[:, width // 2])
        adjusted = r.copy()
        for i in np.nonzero((jump == peak) & (jump > 9.0 * med) & (jump > 0.2))[0]:
            near = r[i - 1:i + 2].min()
            c = (2.0 * near * near - self.WIDTH ** 2) / (2.0 * near * near) if near > 0 else np.nan
            if not -1.0 <= c <= 1.0:
                a = b = 0                                       # the car is wider than twice the range: beam 0 only
            else:
                half = np.arccos(c)
                a, b = (int(np.clip(int((angles[i] + s * half - angles[0]) / inc), 0, len(r) - 1)) for s in (-1.0, 1.0))
            adjusted[a:b + 1] = np.minimum(adjusted[a:b + 1], near)
        chosen = (adjusted >= np.percentile(adjusted, 100 * (1.0 - np.deg2rad(30.0) / np.pi))) & (adjusted < self.range_max)
        return float(angles[chosen].mean()), float(r[chosen].mean())

    def action(self, obs):
        h, dist = self.heading(obs["lidar"])
        d_term = 0.0 if self.previous is None else 0.1 * (self.previous - h) / self.dt
        self.previous = h
        steer = float(np.clip(1.4 * h - d_term, -self.MAX_STEER, self.MAX_STEER))
        speed = 6.0 - (abs(steer) / self.MAX_STEER) * 1.8 if abs(steer) > np.deg2rad(5.0) else 6.0
        if dist < 5:
            speed = min(speed, dist / 5 * 4)
        speed = max(speed, 1.5)
        self.last = dict(heading=h, heading_distance=dist, steering_angle=steer, speed=speed)
        return float(np.clip(speed / self.max_velocity, -1.0, 1.0)), float(np.clip(-steer / self.wheel_max, -1.0, 1.0))
