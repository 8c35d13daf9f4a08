"""The three statements of the env spec constants agree: racing_dreamer_amd/spec.py (host),
csrc/racecar_spec.h (device), oracle/ (checker, NumPy and C)."""
import os
import re

import numpy as np

from oracle import racecar_oracle as ro
from racing_dreamer_amd import spec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _defines(path, prefix=""):
    out = {}
    for m in re.finditer(r"#define\s+" + prefix + r"(\w+)\s+(-?[0-9.eE+-]+)f?\b", open(path).read()):
        out[m.group(1)] = float(m.group(2))
    return out


def test_constants_agree():
    dev = _defines(os.path.join(ROOT, "racing_dreamer_amd", "csrc", "racecar_spec.h"), "RCS_")
    c = _defines(os.path.join(ROOT, "oracle", "racecar_oracle.c"))
    f = np.float32
    triples = [
        ("DT", spec.DT, ro.DT, c["DT"]), ("MAX_RANGE", spec.MAX_RANGE, ro.MAX_RANGE, c["MAX_RANGE"]),
        ("LIDAR_X", spec.LIDAR_X, ro.LIDAR_X, c["LIDAR_X"]), ("WHEELBASE", spec.WHEELBASE, ro.WHEELBASE, c["WHEELBASE"]),
        ("MAX_STEER", spec.MAX_STEER, ro.MAX_STEER, c["MAX_STEER"]), ("MAX_VEL", spec.MAX_VEL, ro.MAX_VEL, c["MAX_VEL"]),
        ("ACCEL_MAX", spec.ACCEL_MAX, ro.ACCEL_MAX, c["ACCEL_MAX"]), ("DRAG", spec.DRAG, ro.DRAG, c["DRAG"]),
        ("WHEEL_MAX", spec.WHEEL_MAX, ro.WHEEL_MAX, c["WHEEL_MAX"]), ("STEER_GAIN", spec.STEER_GAIN, ro.STEER_GAIN, c["STEER_GAIN"]),
        ("STEER_STEP", spec.STEER_RATE * spec.DT, ro.STEER_STEP, c["STEER_STEP"]),
        ("BOX_CX", (spec.X_FRONT + spec.X_REAR) / 2, ro.BOX_CX, c["BOX_CX"]),
        ("BOX_HL", (spec.X_FRONT - spec.X_REAR) / 2, ro.BOX_HL, c["BOX_HL"]),
        ("BOX_HW", spec.HALF_W, ro.BOX_HW, c["BOX_HW"]),
        ("PROGRESS_REWARD", spec.PROGRESS_REWARD, ro.PROGRESS_REWARD, c["PROGRESS_REWARD"]),
        ("PATCH_CELLS", spec.PATCH_WINDOW_CELLS / spec.PATCH, ro.PATCH_CELLS, c["PATCH_CELLS"]),
        ("PATCH_WINDOW", spec.PATCH_CROP_HALF, ro.PATCH_WINDOW, c["PATCH_WINDOW"]),
        ("SPAWN_MARGIN", spec.SPAWN_MARGIN, ro.SPAWN_MARGIN, c["SPAWN_MARGIN"]),
        ("SPAWN_W_MAX", spec.SPAWN_W_MAX, ro.SPAWN_W_MAX, c["SPAWN_W_MAX"]),
        ("HEADING_JITTER", spec.HEADING_JITTER, ro.HEADING_JITTER, c["HEADING_JITTER"]),
    ]
    for name, host, ora, cval in triples:
        assert f(host) == f(ora) == f(cval) == f(dev[name]), name
    assert spec.N_CHECKPOINTS == ro.N_CHECKPOINTS == int(c["N_CP"]) == int(dev["N_CHECKPOINTS"])
    assert spec.BALL_GAP_BINS == ro.BALL_GAP_BINS == int(c["BALL_GAP"]) == int(dev["BALL_GAP_BINS"])
    assert spec.GRID_LEAD_BINS == ro.GRID_LEAD_BINS == int(c["GRID_LEAD"]) == int(dev["GRID_LEAD_BINS"])
    assert spec.SPAWN_CLEAR_R == ro.SPAWN_CLEAR_R == int(c["SPAWN_CLEAR_R"]) == int(dev["SPAWN_CLEAR_R"])
    # the heading room of a bin without lateral room: the same six levels everywhere, each inside what the geometry allows
    src = open(os.path.join(ROOT, "racing_dreamer_amd", "csrc", "racecar_spec.h")).read()
    dev_rooms = [float(v.rstrip("f")) for v in re.search(r"#define RCS_HEADING_ROOM_INIT \{([^}]*)\}", src).group(1).split(",")]
    c_rooms = [float(v.rstrip("f")) for v in re.search(r"HEADING_ROOM\[6\] = \{([^}]*)\}", open(os.path.join(ROOT, "oracle", "racecar_oracle.c")).read()).group(1).split(",")]
    assert [f(v) for v in spec.HEADING_ROOM] == [f(v) for v in dev_rooms] == [f(v) for v in c_rooms] == list(ro.HEADING_ROOM)
    assert spec.SPAWN_FOOT_R == ro.SPAWN_FOOT_R == int(c["SPAWN_FOOT_R"]) == int(dev["SPAWN_FOOT_R"]) == len(spec.HEADING_ROOM) - 1
    reach = np.hypot(spec.X_FRONT, spec.HALF_W)
    for k, room in enumerate(spec.HEADING_ROOM):
        assert room <= spec.HEADING_JITTER and room * reach <= max(0.05 * k - 2 * 0.05 * np.sqrt(0.5) - 0.004, 0.0) + 1e-9, (k, room)
    # the margin covers the footprint's farthest corner from the rear axle plus what a cell-centre distance cannot see
    assert spec.SPAWN_MARGIN >= np.hypot(spec.X_FRONT, spec.HALF_W) + 2 * 0.05 * np.sqrt(0.5) + 0.05
    assert spec.N_BEAMS == ro.N_BEAMS == 1080 and spec.N_FOOTPRINT == int(dev["N_FOOTPRINT"]) == 34
    assert f(dev["PI"]) == ro.PI and f(dev["TWO_PI"]) == ro.TWO_PI


def test_tables_agree():
    cb, sb = ro.beam_table()
    assert np.array_equal(np.stack([cb, sb], 1), spec.beam_table())
    assert np.array_equal(ro.footprint_table(), spec.footprint_table())
    # the fixed-point footprint walk visits the same 34 points: lattice node (i, j) = (X_REAR + 0.05 i, -HALF_W + 0.05 j)
    pts = np.array([(spec.X_REAR + 0.05 * i, -spec.HALF_W + 0.05 * j) for i, j in ro.FOOT_LATTICE])
    assert len(ro.FOOT_LATTICE) == 34 and np.allclose(pts, ro.footprint_table(), atol=1e-7)


def test_patch_and_footprint_fixed_point_constants_agree():
    dev = _defines(os.path.join(ROOT, "racing_dreamer_amd", "csrc", "racecar_spec.h"), "RCS_")
    c = _defines(os.path.join(ROOT, "oracle", "racecar_oracle.c"))
    f = np.float32
    assert f(dev["FOOT_STEP"]) == ro.FOOT_STEP == f(c["FOOT_STEP"]) == f(0.05)
    assert f(dev["PATCH_STEP_Q16"]) == ro.PATCH_STEP_Q16 == f(c["PATCH_STEP_Q16"]) == f(3.125 * 65536)
    assert int(dev["PATCH_WINDOW_I"]) == ro.PATCH_WINDOW_I == int(c["PATCH_WINDOW_I"]) == spec.PATCH_CROP_HALF
