"""CPU restatement (float64 NumPy) of the reference's follow-the-gap law - TEST INFRASTRUCTURE, NOT PRODUCT.

Follows ros_agent/agents/follow_the_gap/src/agent.py of the reference:
  * `target_heading`  <- laserscan_callback, agent.py:128-193 (with get_lidar_scan_arc, agent.py:116-126): forward arc of
    +-90 deg, ranges clipped at the look-ahead distance, disparities = local maxima of |range differences| that stand out
    against the 10-degree median by a factor 9 and exceed 0.2 m, each extended by the vehicle's half-width seen at the
    nearer range, heading = mean angle of the beams at or above the 83.3rd percentile of the adjusted ranges;
  * `drive_command`   <- publish_drive_from_heading, agent.py:200-234 with PID.calculate, agent.py:45-55 (kp 1.4, ki 0,
    kd 0.1, target 0): steering = 1.4 heading - 0.1 (previous heading - heading) / dt, clipped to +-24 deg; speed 6 m/s
    less up to 30 % for steering beyond 5 deg, at most 4/5 of the heading distance below 5 m, at least 1.5 m/s.
Constants: agent.py:73-78 (max_speed 7, max_decel 8.26 -> look-ahead 2 x 7^2 / (2 x 8.26) m; vehicle width 1.2 x 0.3302 m),
:83-86 (arcs), :88-89 (0.2 m, factor 9), :92, :101, :104.

Pinned by tests/golden/ftg_golden.npz (tests/golden/make_golden_ftg.py runs the reference's own node, imported with
name-only ROS stubs, on scans of this build's oracle and on synthetic ones): tests/test_golden_ftg.py.
ROS plumbing that is not part of the law is left out: the two "first message only sets a timestamp" returns
(agent.py:132-134,206-208) and the 50 / 100 Hz throttles (:138,:212).  (The reference's slice `ranges[i - 1:i + 2]` would be
empty for a disparity at the arc's first beam, agent.py:167 - which cannot be one: the edge-repeating median window there
holds that very jump 20 times out of 39, so it never exceeds 9 medians.)
"""
from __future__ import annotations

import numpy as np

MAX_SPEED, MAX_DECEL = 7.0, 8.26                                    # agent.py:73-74
LOOKAHEAD = 2.0 * (MAX_SPEED ** 2 / (2.0 * MAX_DECEL))              # agent.py:75-76
VEHICLE_WIDTH = 0.3302 * 1.2                                        # agent.py:78
ARC = (np.deg2rad(-90.0), np.deg2rad(90.0))                         # agent.py:83
HEADING_ARC = np.deg2rad(30.0)                                      # agent.py:84
HEADING_PERCENTILE = 100 * (1.0 - HEADING_ARC / (ARC[1] - ARC[0]))  # agent.py:85-86
MIN_GAP, MEDIAN_FACTOR = 0.2, 9.0                                   # agent.py:88-89
KP, KD = 1.4, 0.1                                                   # agent.py:92
MAX_VEHICLE_SPEED = 6.0                                             # agent.py:101
MAX_STEER = np.deg2rad(24.0)                                        # agent.py:104
FILTER_ARC = np.deg2rad(10.0)                                       # agent.py:150


def _windows(x, width, mode):
    """All centred windows of `width` samples: [len(x), width]; the borders continue the signal like scipy.ndimage's
    modes 'nearest' (edge value repeated) and 'reflect' (mirrored, edge sample doubled)."""
    left = width // 2
    pad = np.pad(x, (left, width - 1 - left), mode={"nearest": "edge", "reflect": "symmetric"}[mode])
    return np.lib.stride_tricks.sliding_window_view(pad, width)


def forward_arc(ranges, angle_min, angle_increment):
    """agent.py:116-126: first / last beam index by truncation, angles rebuilt from the indices."""
    first, last = (int((a - angle_min) / angle_increment) for a in ARC)
    idx = np.arange(first, last + 1)
    return idx.astype(np.float64) * angle_increment + angle_min, np.asarray(ranges, np.float64)[first:last + 1], first


def disparities(clipped, angle_increment):
    """Indices i of the arc at which |clipped[i + 1] - clipped[i]| is a disparity (agent.py:148-159)."""
    jump = np.abs(np.diff(clipped))
    width = int(FILTER_ARC / angle_increment)
    med = np.median(_windows(jump, width, "nearest"), axis=1) if width % 2 else np.sort(_windows(jump, width, "nearest"), axis=1)[:, width // 2]
    peak = _windows(jump, width, "reflect").max(axis=1)
    keep = (jump == peak) & (jump > med * MEDIAN_FACTOR) & (jump > MIN_GAP)
    return np.nonzero(keep)[0]


def target_heading(ranges_ros, angle_min, angle_increment, range_max):
    """One scan in ROS order (beam 0 at angle_min, counter-clockwise) -> (heading [rad], heading distance [m])."""
    angles, ranges, _ = forward_arc(ranges_ros, angle_min, angle_increment)
    ranges = np.clip(ranges, 0.0, LOOKAHEAD)                                    # agent.py:145-146
    adjusted = ranges.copy()
    n = len(ranges)
    for i in disparities(ranges, angle_increment):                             # agent.py:165-176
        near = ranges[i - 1:i + 2].min()                                        # (i >= 1: see the module docstring)
        with np.errstate(all="ignore"):
            half = np.arccos((2.0 * near * near - VEHICLE_WIDTH ** 2) / (2.0 * near * near))     # law of cosines, two sides `near`
            span = (np.array([angles[i] - half, angles[i] + half]) - angles[0]) / angle_increment
        # truncation to integers as NumPy does it (a NaN half-angle - the car is wider than twice the range - gives the
        # most negative integer, which the clip turns into beam 0)
        a, b = (int(np.clip(np.int64(v) if np.isfinite(v) else np.iinfo(np.int64).min, 0, n - 1)) for v in span)
        adjusted[a:b + 1] = np.minimum(adjusted[a:b + 1], near)
    threshold = np.percentile(adjusted, HEADING_PERCENTILE)                    # agent.py:183
    chosen = (adjusted >= threshold) & (adjusted < range_max)                  # np.digitize(...) == 2
    with np.errstate(all="ignore"):
        return float(np.mean(angles[chosen])), float(np.mean(ranges[chosen]))  # agent.py:184-185


def drive_command(heading, heading_distance, previous_heading, dt):
    """(steering angle [rad], speed [m/s]) from a heading; previous_heading = NaN for the first one (agent.py:51-52)."""
    control = KP * (0.0 - heading) + (0.0 if np.isnan(previous_heading) else KD * (previous_heading - heading) / dt)
    steer = float(np.clip(-control, -MAX_STEER, MAX_STEER))                    # agent.py:219-223
    speed = MAX_VEHICLE_SPEED
    if abs(steer) > np.deg2rad(5):                                             # agent.py:228-230
        speed = MAX_VEHICLE_SPEED - (abs(steer) / MAX_STEER) * (MAX_VEHICLE_SPEED * 0.30)
    if heading_distance < 5:                                                   # agent.py:231-232
        speed = min(speed, heading_distance / 5 * 4)
    return steer, max(speed, 1.5)                                              # agent.py:233
