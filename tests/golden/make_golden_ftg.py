#!/usr/bin/env python3
"""Generate tests/golden/ftg_golden.npz from the REFERENCE's own follow-the-gap node.

Runs only in the build container (needs /root/reference).  The node
(ros_agent/agents/follow_the_gap/src/agent.py) is plain NumPy/SciPy behind ROS plumbing: it imports once `rospy` and
the four message packages are present in sys.modules.  The stubs below provide only the *names* the module uses
(rospy.Time / Subscriber / Publisher, Float32, LaserScan, Odometry, AckermannDriveStamped) and the `np.int` / `np.float`
aliases NumPy 1.18 still had (agent.py:121-123,179); every number stored in the fixture is computed by the reference's
code:

  G9  laserscan_callback ............ agent.py:128-193  scan -> target heading, heading distance
      publish_drive_from_heading .... agent.py:200-234  heading -> steering angle, speed (PID kp 1.4, kd 0.1)

Inputs: LiDAR scans of this build's CPU oracle (cars driven along three tracks, 16 consecutive agent steps each at
dt = 0.04 s, in ROS order: angle_min = -135 deg, counter-clockwise) plus synthetic scans with hand-placed disparities.
Each car's scans are fed to a FRESH node, one message per step, with a fake clock; the node publishes from its third
message on (the first two only initialise its two timestamps, agent.py:132-134,206-208).

Fixture = inputs and expected outputs only (data, no source).  Library versions are recorded inside (the reference
pins numpy 1.18.5 / scipy 1.5.4, dreamer/requirements.txt:3,16; this container has newer ones - np.percentile's
interpolation arithmetic differs in the last bit between them, which matters for the one beam that sits exactly on the
percentile: tests/test_golden_ftg.py says how that is handled).
"""
import importlib.util
import os
import sys
import types
import warnings

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

N_BEAMS = 1080
ANGLE_MIN = -0.75 * np.pi                       # beam 0 of a ROS scan; this build's beam i is ROS beam 1079 - i
ANGLE_INC = 1.5 * np.pi / (N_BEAMS - 1)
RANGE_MAX = 15.0
DT = 0.04                                       # agent step: 4 sub-steps of 0.01 s (dreamer/dream.py:55)


# ----------------------------------------------------------------------------- stubs (names only)
class _Time:
    now_value = 0.0

    def __init__(self, secs=0, nsecs=0):
        self.secs, self.nsecs = int(secs), int(nsecs)

    def is_zero(self):
        return self.secs == 0 and self.nsecs == 0

    def to_sec(self):
        return self.secs + self.nsecs * 1e-9

    @staticmethod
    def now():
        s = int(_Time.now_value)
        return _Time(s, int(round((_Time.now_value - s) * 1e9)))


class _Pub:
    def __init__(self, *a, **k):
        self.sent = []

    def publish(self, msg):
        self.sent.append(msg)


class _Obj:
    pass


def _msg(*fields):
    def make():
        o = _Obj()
        for f in fields:
            cur = o
            parts = f.split(".")
            for p in parts[:-1]:
                if not hasattr(cur, p):
                    setattr(cur, p, _Obj())
                cur = getattr(cur, p)
            setattr(cur, parts[-1], 0.0)
        return o
    return make


def install_ros_stubs():
    rospy = types.ModuleType("rospy")
    rospy.Time = _Time
    rospy.Subscriber = lambda *a, **k: None
    rospy.Publisher = _Pub
    sys.modules["rospy"] = rospy
    for pkg, names in (("std_msgs", {"Float32": _msg("data")}),
                       ("sensor_msgs", {"LaserScan": _msg("header.stamp.secs")}),
                       ("nav_msgs", {"Odometry": _msg("header.stamp.secs")}),
                       ("ackermann_msgs", {"AckermannDriveStamped": _msg("header.stamp", "drive.steering_angle", "drive.speed")})):
        m, mm = types.ModuleType(pkg), types.ModuleType(pkg + ".msg")
        for k, v in names.items():
            setattr(mm, k, v)
        m.msg = mm
        sys.modules[pkg], sys.modules[pkg + ".msg"] = m, mm
    if not hasattr(np, "int"):
        np.int = int            # noqa: NPY001  (NumPy < 1.24 names the reference uses)
    if not hasattr(np, "float"):
        np.float = float        # noqa: NPY001


def load_reference_node():
    install_ros_stubs()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")         # `from scipy.ndimage import filters` is a deprecated spelling
        spec = importlib.util.spec_from_file_location("ref_ftg_agent", os.path.join(REF, "ros_agent/agents/follow_the_gap/src/agent.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    return mod


def scan_msg(ranges_ros, t):
    m = _Obj()
    m.header = _Obj()
    m.header.stamp = _Obj()
    s = int(t)
    m.header.stamp.secs, m.header.stamp.nsecs = s, int(round((t - s) * 1e9))
    m.angle_min, m.angle_max, m.angle_increment = ANGLE_MIN, -ANGLE_MIN, ANGLE_INC
    m.range_max = RANGE_MAX
    m.ranges = [float(v) for v in ranges_ros]
    return m


def run_node(mod, scans_ros):
    """Feed one car's consecutive scans [T, 1080] (ROS order) to a fresh node.  Returns heading, heading_distance [T]
    (NaN where the node computed none) and steering_angle, speed [T] (NaN where it published nothing)."""
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        node = mod.AgentNode("golden", "A")
    T = len(scans_ros)
    heading, hdist = np.full(T, np.nan), np.full(T, np.nan)
    steer, speed = np.full(T, np.nan), np.full(T, np.nan)
    inner = node.publish_drive_from_heading
    cur = {"k": 0}

    def tap(data, heading_distance):
        heading[cur["k"]], hdist[cur["k"]] = data.data, heading_distance
        return inner(data, heading_distance)

    node.publish_drive_from_heading = tap
    for k in range(T):
        cur["k"] = k
        t = 100.0 + k * DT                      # (not zero: a zero stamp means "no previous message" to the node)
        _Time.now_value = t
        before = len(node.drive_pub.sent)
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            node.laserscan_callback(scan_msg(scans_ros[k], t))
        if len(node.drive_pub.sent) > before:
            d = node.drive_pub.sent[-1].drive
            steer[k], speed[k] = d.steering_angle, d.speed
    return heading, hdist, steer, speed


def oracle_scans(track_name, n_cars, steps, seed):
    """LiDAR rows of this build's oracle: cars driven by its own gap follower, one row per agent step (repeat 4)."""
    from oracle import racecar_oracle as ro
    from racing_dreamer_amd.track_assets import load_track
    t = load_track(track_name)
    env = ro.OracleRaceEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution,
                           ro.OracleConfig(num_envs=n_cars, auto_reset=False, terminate_on_collision=False))
    out = env.reset(mode=ro.RESET_RANDOM, seed=seed)
    rows = []
    for k in range(steps + 6):
        act = ro.follow_the_gap(out["lidar"])
        out = env.step(act, repeat=4)
        if k >= 6:                              # (under way: not the standing start)
            rows.append(np.asarray(out["lidar"], np.float32).reshape(n_cars, N_BEAMS).copy())
    return np.stack(rows, 0)                    # [steps, cars, 1080], this build's beam order (clockwise from +135 deg)


def synthetic_scans(rng, n):
    """Single scans with hand-placed features: open space, a wall ahead, step disparities of several sizes at several
    bearings (also at the arc's ends), a narrow spike, a noisy floor.  ROS order."""
    out = []
    ang = ANGLE_MIN + ANGLE_INC * np.arange(N_BEAMS)
    for k in range(n):
        kind = k % 8
        r = np.full(N_BEAMS, 12.0)
        if kind == 1:
            r = 2.0 / np.maximum(np.cos(ang), 0.05)                     # wall 2 m ahead
        elif kind == 2:
            r = 1.2 + 0.0 * ang
            r[ang > rng.uniform(-1.2, 1.2)] = rng.uniform(3.0, 9.0)     # one step disparity
        elif kind == 3:
            r = rng.uniform(2.5, 5.0) + 0.3 * np.sin(3 * ang)
            for _ in range(3):
                a0 = rng.uniform(-1.5, 1.3)
                r[(ang > a0) & (ang < a0 + rng.uniform(0.05, 0.4))] = rng.uniform(0.6, 2.0)     # boxes
        elif kind == 4:
            r = 3.0 + 0.0 * ang
            i0 = rng.integers(185, 890)
            r[i0:i0 + 2] = 0.9                                           # a spike two beams wide
        elif kind == 5:
            r = 4.0 + rng.normal(0, 0.02, N_BEAMS)                       # noisy floor, no real gap
        elif kind == 6:
            r = 1.5 + 0.0 * ang
            r[ang > 1.45] = 6.0                                          # disparity near the arc's end
            r[ang < -1.5] = 5.0
        elif kind == 7:
            r = np.clip(0.8 + 2.0 * np.abs(np.sin(2.0 * ang + rng.uniform(0, 3))) + rng.normal(0, 0.01, N_BEAMS), 0.3, 15.0)
            r[(ang > 0.2) & (ang < 0.5)] = 8.0
        out.append(np.clip(r, 0.0, RANGE_MAX).astype(np.float32))
    return np.stack(out, 0)


def main():
    import scipy
    mod = load_reference_node()
    rng = np.random.default_rng(7)
    data = {"numpy_version": np.array(np.__version__), "scipy_version": np.array(scipy.__version__),
            "angle_min": np.float64(ANGLE_MIN), "angle_increment": np.float64(ANGLE_INC), "range_max": np.float64(RANGE_MAX),
            "dt": np.float64(DT)}
    tracks = ("columbia", "austria", "treitlstrasse_v2")
    for ti, name in enumerate(tracks):
        rows = oracle_scans(name, n_cars=4, steps=12, seed=3 + ti)          # [12, 4, 1080], clockwise order
        ros = rows[:, :, ::-1]                                              # ROS order
        T, B, _ = ros.shape
        H, D, S, V = (np.zeros((T, B)) for _ in range(4))
        for b in range(B):
            H[:, b], D[:, b], S[:, b], V[:, b] = run_node(mod, ros[:, b])
        data[f"{name}_lidar"] = rows                                        # as the env emits them (float32)
        data[f"{name}_heading"], data[f"{name}_heading_distance"] = H, D
        data[f"{name}_steering_angle"], data[f"{name}_speed"] = S, V
    syn = synthetic_scans(rng, 32)                                          # ROS order
    H, D = np.zeros(len(syn)), np.zeros(len(syn))
    for k in range(len(syn)):
        h, d, _, _ = run_node(mod, np.stack([syn[k], syn[k]], 0))           # (the second message computes the heading)
        H[k], D[k] = h[1], d[1]
    data["synthetic_lidar"] = syn[:, ::-1].copy()                           # stored in the env's (clockwise) order
    data["synthetic_heading"], data["synthetic_heading_distance"] = H, D
    data["tracks"] = np.array(tracks)
    out = os.path.join(HERE, "ftg_golden.npz")
    np.savez_compressed(out, **data)
    print(out, os.path.getsize(out), "bytes")
    for name in tracks:
        s = data[f"{name}_steering_angle"]
        print(name, "published", int(np.isfinite(s).sum()), "of", s.size, "steer range", np.nanmin(s), np.nanmax(s),
              "speed range", np.nanmin(data[f"{name}_speed"]), np.nanmax(data[f"{name}_speed"]))


if __name__ == "__main__":
    main()
