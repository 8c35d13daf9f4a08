"""Frozen constants of the batched racecar environment spec.

Values with a reference citation are taken from the reference tree; the rest are this
build's free parameters (the reference's simulator, racecar_gym + PyBullet, is an
un-vendored dependency, so nothing in the reference fixes them - SURVEY.md §8a).
The device code (csrc/racecar_spec.h) and the CPU oracle (oracle/) restate the same
numbers independently; tests/test_spec_consistency.py checks the three agree.
"""
import math

import numpy as np

# --- simulation clock -------------------------------------------------------------
DT = 0.01                     # dreamer/callbacks.py:23, ros_agent/utils.py:10

# --- LiDAR ------------------------------------------------------------------------
N_BEAMS = 1080                # dreamer/dream.py:66
FOV_DEG = 270.0               # dreamer/tools.py:84-86
MAX_RANGE = 15.0              # dreamer/tools.py:274
LIDAR_X = 0.25                # sensor origin ahead of the rear axle [m]            (free)

# --- vehicle ----------------------------------------------------------------------
WHEELBASE = 0.3302            # ros_agent/agents/follow_the_gap/src/agent.py:78
MAX_STEER = 0.42              # ros_agent/models/dreamer/racing_dreamer.py:14 (nominal scale of the steering action)
# Front-wheel angle of the BICYCLE model at full command, and its sign.  Two reference-held pins:
#  (1) the reference's own deployment mapping from a simulator command to an Ackermann (bicycle) drive message:
#        ros_agent/agents/dreamer/src/agent.py:111   steering = 0 - action['steering'] * 0.6 * 0.42   "working better in hardware"
#        ros_agent/agents/dreamer/src/agent.py:112   ... * 0.7 * 0.42                                  "working better in simulation"
#        ros_agent/agents/acme/src/agent.py:90, ros_agent/agents/sb3/src/agent.py:90   ... * 0.4 * 0.42
#      the NEGATION against ROS's left-positive steering angle = a positive command steers RIGHT; the effective lock is
#      0.4 .. 0.7 x 0.42 = 0.168 .. 0.294 rad - WHEEL_MAX lies inside (tests/golden/deployment_mapping.json,
#      tests/test_golden_policy.py::test_the_references_deployment_mapping_*);
#  (2) the reference's trained agents: the shipped austria agent laps here for a lock of 0.15 .. 0.19 rad and turns into the
#      inner wall of the first hairpin from 0.21 rad on (so the upper half of the authors' band, 0.25 .. 0.29, is refuted for
#      THIS kinematic model: profiles/r06_b_deployment_mapping.txt), the treitlstrasse agent for 0.18 .. 0.21.
WHEEL_MAX = 0.19              # [rad]; a POSITIVE command steers RIGHT (clockwise)
STEER_GAIN = -WHEEL_MAX       # command -> wheel angle, counter-clockwise positive
MAX_FORCE = 0.5               # ros_agent/models/dreamer/racing_dreamer.py:15
MAX_VEL = 5.0                 # ros_agent/models/dreamer/racing_dreamer.py:16
FORCE_TO_ACCEL = 8.0          # m/s^2 per unit motor force                          (free)
ACCEL_MAX = MAX_FORCE * FORCE_TO_ACCEL
DRAG = ACCEL_MAX / MAX_VEL     # 1/s linear resistance: throttle m settles at m * MAX_VEL       (free)
STEER_RATE = 3.2              # rad/s steering slew limit                           (free)
X_REAR, X_FRONT, HALF_W = -0.10, 0.45, 0.15   # footprint in the body frame [m]    (free)
FOOTPRINT_LONG_PTS, FOOTPRINT_SHORT_PTS = 12, 5

# --- task (dreamer/scenarios/max_progress/columbia.yml:9-10) ------------------------
N_CHECKPOINTS = 20            # (free)
PROGRESS_REWARD = 100.0       # (free; racecar_gym default)
TASK_MAX_PROGRESS, TASK_MAX_SPEED, TASK_N_STEP_PROGRESS = 0, 1, 2
NSTEP_MAX = 16                   # longest n_step_progress window [sub-steps]

# --- lidar_occupancy patch (dreamer/wrappers.py:374-378,398-405) --------------------
PATCH = 64
PATCH_WINDOW_CELLS = 200      # 2 * neigh_size -> 3.125 cells per output pixel
PATCH_CROP_HALF = 110         # neigh_size + 10: taps outside this window read 0

# --- reset modes (dreamer/dream.py:105-108,120) --------------------------------------
RESET_GRID, RESET_RANDOM, RESET_RANDOM_BALL = 0, 1, 2
RESET_MODES = {"grid": RESET_GRID, "random": RESET_RANDOM, "random_ball": RESET_RANDOM_BALL}
BALL_GAP_BINS = 12            # centerline bins (0.1 m each) between cars           (free)
GRID_LEAD_BINS = 8            # grid reset: last car 0.8 m after the start line     (free)
# `random` / `random_ball`: a pose on the track with a minimum wall distance, heading along the track (SURVEY.md H6;
# "sample in random points close within a ball", dream.py:105-108): centre-line bin uniform over the lap, lateral offset
# uniform within the room the track leaves at that bin, heading within +- HEADING_JITTER of the track's direction
SPAWN_CLEAR_R = 40            # cells searched for the nearest non-drivable cell (2 m)       (free)
SPAWN_MARGIN = 0.60           # [m] footprint's farthest corner 0.474 + two half cell diagonals 0.071  (derived)
SPAWN_W_MAX = 1.5             # [m] cap of the lateral offset                                   (free)
HEADING_JITTER = 0.35         # [rad]                                                           (free)
# where a bin leaves no lateral room the heading turns only as far as the footprint's own clearance k (cells, 0 .. 5) allows:
# a point <= 0.474 m from the rear axle moves <= 0.474 |dtheta|; k cells between centres leave 0.05 k - 0.0707 m  (derived)
SPAWN_FOOT_R = 5
HEADING_ROOM = (0.0, 0.0, 0.05, 0.155, 0.26, 0.35)
SPAWN_SAFE_SEARCH = 256       # bins searched forward for a start at which four cars 1.2 m apart do not overlap   (free)

# --- action remap (dreamer/dream.py:138) ---------------------------------------------
ACTION_LOW = (0.005, -1.0)
ACTION_HIGH = (1.0, 1.0)

# trajectory record, bytes per car per agent step (dreamer/wrappers.py:213-219)
RECORD_FLOATS = N_BEAMS + 6 + 6 + 1 + 2 + 1 + 1 + 1 + 1      # 1099 floats = 4396 B


def beam_table() -> np.ndarray:
    """float32 [N_BEAMS, 2] = (cos, sin) of the beam angle in the sensor frame.

    Beam 0 points to +135 deg (left-rear), the sweep is clockwise to -135 deg
    (dreamer/tools.py:84-86).  Evaluated in float64, rounded once to float32.
    """
    half = math.radians(FOV_DEG) / 2.0
    ang = half - np.arange(N_BEAMS, dtype=np.float64) * (2.0 * half / (N_BEAMS - 1))
    return np.stack([np.cos(ang), np.sin(ang)], axis=1).astype(np.float32)


def footprint_table() -> np.ndarray:
    """float32 [34, 2] perimeter sample points of the car rectangle in the body frame."""
    xs = np.linspace(X_REAR, X_FRONT, FOOTPRINT_LONG_PTS)
    ys = np.linspace(-HALF_W, HALF_W, FOOTPRINT_SHORT_PTS + 2)[1:-1]
    pts = [(x, -HALF_W) for x in xs] + [(x, HALF_W) for x in xs]
    pts += [(X_REAR, y) for y in ys] + [(X_FRONT, y) for y in ys]
    return np.asarray(pts, np.float64).astype(np.float32)


N_FOOTPRINT = 2 * FOOTPRINT_LONG_PTS + 2 * FOOTPRINT_SHORT_PTS
