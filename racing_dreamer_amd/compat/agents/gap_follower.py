"""Follow-the-gap prefill agent with the interface dreamer/dream.py:213-215 uses:
``GapFollower().action(obs) -> (motor, steering)`` from ``obs['lidar']`` (1080 beams, 270 deg).

The class the reference imports lives in the external racecar_gym repository (`agents/gap_follower.py`, not
under /root/reference), so this is the classic follow-the-gap law written from its description: clip far
ranges, smooth, zero a safety bubble around the closest return, steer at the centre of the widest run of
beams longer than `gap_range`, slow down when steering hard.  (The reference's own ROS node, ros_agent/agents/follow_the_gap/src/agent.py,
is a more elaborate disparity-extender + PID variant for the real car.)  Steering is returned normalised to
[-1, 1] (max 0.42 rad).  The batched device version is BatchedRaceEnv.follow_the_gap().
"""
import numpy as np


class GapFollower:
    def __init__(self, max_range: float = 3.0, bubble_radius: int = 60, smooth: int = 5, gap_range: float = 1.0,
                 fov_deg: float = 270.0,
                 max_steering: float = 0.42, straights_speed: float = 0.6, corners_speed: float = 0.3):
        self.max_range, self.bubble, self.smooth, self.gap_range = max_range, bubble_radius, smooth, gap_range
        self.fov = np.radians(fov_deg)
        self.max_steering = max_steering
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
        steering = float(np.clip(angle / self.max_steering, -1.0, 1.0))
        motor = self.corners_speed if abs(steering) > 0.35 else self.straights_speed
        return motor, steering
