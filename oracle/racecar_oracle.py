"""CPU oracle of the batched racecar environment - TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
file.  The product path (racing_dreamer_amd/) never does: it fails loudly when the HIP
library is missing.

PARITY UNPINNED for the simulator core.  The reference's env.step() arithmetic (vehicle
integrator, LiDAR, progress reward, collision) lives in the un-vendored third-party
packages racecar_gym (pinned @icra22 / @gym-api / e117432572bb / a9e6f5f440f3:
dreamer/requirements.txt:6, baselines/requirements.txt:13,
baselines/docker/requirements_sb3.txt:88, requirements_acme.txt:100) and pybullet
3.2.1 / 3.0.8 (requirements_sb3.txt:76, requirements_acme.txt:85); neither is present
under /root/reference nor installable here, and the reference holds no tests or golden
vectors for this path.  This file therefore restates the build's own
env spec whose *interface* is pinned by the reference's call sites:

* step()/reset() contract, info keys ............ dreamer/wrappers.py:62-77,210-236
* action dict {'motor','steering'} .............. dreamer/wrappers.py:63
* 1080 beams, 270 deg clockwise from +135 deg ... dreamer/tools.py:84-86, dream.py:66
* 15 m max range ................................ dreamer/tools.py:274
* vehicle limits ................................ ros_agent/models/dreamer/racing_dreamer.py:14-16
* wheelbase ..................................... ros_agent/agents/follow_the_gap/src/agent.py:78
* task params (laps, 180 s, collision -1) ....... dreamer/scenarios/max_progress/columbia.yml:10
* max_speed reward .............................. baselines/racing/environment/tasks.py:6-15
* progress grid ................................. docs/maps/costmaps/generate-costmap.py:196-222
* lidar_occupancy patch geometry ................ dreamer/wrappers.py:390-408
* action remap .................................. dreamer/wrappers.py:128-130

What IS pinned against the reference: the wrapper-layer semantics and the patch
geometry, through tests/golden/*.npz captured from the importable reference wrappers
(tests/golden/make_golden.py); the Philox4x32-10 generator through the published
Random123 known-answer vectors; and - behaviourally - the core's handedness (a positive
steering command turns towards higher beam indices = right) and steering lock (0.19 rad)
through the reference's own trained Dreamer agents, which lap in this env and crash within
seconds in its mirror image or with another lock (tests/test_golden_policy.py,
oracle/dreamer_policy_port.py).

Numerics contract: every float operation below is a single correctly-rounded IEEE
binary32 operation (+, -, *, /, compare, floor, rint) applied in the written order - no
fused multiply-add, no libm.  The HIP kernels are compiled with -ffp-contract=off and
follow the same order, so device results are expected to be bit-identical.
"""
from __future__ import annotations

import math

import numpy as np

f32 = np.float32
i32 = np.int32
u32 = np.uint32
u64 = np.uint64

# ----------------------------------------------------------------------------- constants
DT = f32(0.01)
N_BEAMS = 1080
FOV_DEG = 270.0
MAX_RANGE = f32(15.0)
LIDAR_X = f32(0.25)
WHEELBASE = f32(0.3302)
MAX_STEER = f32(0.42)
MAX_VEL = f32(5.0)
ACCEL_MAX = f32(4.0)             # max_force 0.5 * 8.0
DRAG = f32(0.8)                  # 1/s: ACCEL_MAX / MAX_VEL, so full throttle settles at max_velocity
STEER_STEP = f32(0.032)          # 3.2 rad/s * dt
# The steering command: +1 = full lock to the RIGHT (clockwise), front wheels at WHEEL_MAX.  Both are pinned by the
# reference's own trained agents (ros_agent/checkpoints, tests/test_golden_policy.py): beams run left -> right
# (dreamer/tools.py:84-86 draws index 0 on the left), and the Dreamer policies trained in the reference's simulator steer
# towards HIGHER beam indices with POSITIVE commands - they lap in this env and crash within seconds in its mirror image -
# and they lap only while full lock is 0.17-0.20 rad (0.12: understeer into the outer wall, 0.21: the inner one): the
# simulated car turns less than half as sharply as the 0.42 rad the action space is nominally scaled to
# (racing_dreamer.py:14).
WHEEL_MAX = f32(0.19)
STEER_GAIN = f32(-0.19)          # command -> wheel angle in the vehicle frame (counter-clockwise positive)
INV_DT = f32(100.0)
X_REAR, X_FRONT, HALF_W = -0.10, 0.45, 0.15
BOX_CX = f32(0.175)              # (X_FRONT + X_REAR) / 2
BOX_HL = f32(0.275)              # (X_FRONT - X_REAR) / 2
BOX_HW = f32(0.15)
FOOT_STEP = f32(0.05)             # pitch of the footprint lattice [m]: X_REAR + 0.05 i, -HALF_W + 0.05 j
Q16 = f32(65536.0)
# border of the 12 x 7 lattice: the two long sides (j = 0, 6), then the short sides without their corners
FOOT_LATTICE = ([(i, 0) for i in range(12)] + [(i, 6) for i in range(12)]
                + [(0, j) for j in range(1, 6)] + [(11, j) for j in range(1, 6)])
N_CHECKPOINTS = 20
PROGRESS_REWARD = f32(100.0)
PATCH = 64
PATCH_CELLS = f32(3.125)         # 200 cells / 64 px      (dreamer/wrappers.py:402-405)
PATCH_WINDOW = f32(110.0)        # neigh_size + 10 cells  (dreamer/wrappers.py:398-399)
PATCH_WINDOW_I = 110
PATCH_STEP_Q16 = f32(204800.0)   # 3.125 cells per pixel in 16.16 fixed point
BALL_GAP_BINS = 12               # 1.2 m between cars of one env at reset
GRID_LEAD_BINS = 8               # grid mode: the last car starts 0.8 m after the start line
# `random` / `random_ball` (dreamer/dream.py:105-108): a pose on the track with clearance, heading along it.
SPAWN_CLEAR_R = 40               # cells searched around a centre-line point for the nearest non-drivable cell (2 m)
SPAWN_MARGIN = f32(0.60)         # [m] kept between the rear-axle point and that cell's centre: the footprint's farthest corner
                                 # (0.474 m) + the two half cell diagonals (0.071 m) the cell-centre distance cannot see
SPAWN_W_MAX = f32(1.5)           # [m] cap of the lateral offset
HEADING_JITTER = f32(0.35)       # [rad] the heading is drawn within +- this of the track's direction
# ... where the track leaves lateral room.  Where it leaves none (w = 0: a corridor narrower than SPAWN_MARGIN either side), the
# heading may only turn as far as the footprint's own clearance allows: level k = the smallest, over the 34 footprint points of
# the centre-line pose, of isqrt(squared cell distance to the nearest non-drivable cell within +- SPAWN_FOOT_R), capped at 5;
# a point at most 0.474 m from the rear axle moves by at most 0.474 |dtheta|, and k cells between cell centres leave
# 0.05 k - 0.0707 m between the point and a blocked cell's area: dtheta < (0.05 k - 0.0707 - 0.005) / 0.474.
SPAWN_FOOT_R = 5
HEADING_ROOM = np.array([0.0, 0.0, 0.05, 0.155, 0.26, 0.35], f32)      # [rad] by level k = 0 .. 5
MAX_CARS = 4                     # cars per env at most (include/racecar_hip.h RC_MAX_CARS)
SPAWN_SAFE_SEARCH = 256          # several cars: bins searched forward for a start whose centre-line poses do not overlap
PI = f32(3.14159274101257324)
TWO_PI = f32(6.28318548202514648)
INF = f32(np.inf)

TASK_MAX_PROGRESS, TASK_MAX_SPEED, TASK_N_STEP_PROGRESS = 0, 1, 2
NSTEP_MAX = 16                   # longest n_step_progress window [sub-steps]
RESET_GRID, RESET_RANDOM, RESET_RANDOM_BALL = 0, 1, 2


def beam_table():
    half = math.radians(FOV_DEG) / 2.0
    ang = half - np.arange(N_BEAMS, dtype=np.float64) * (2.0 * half / (N_BEAMS - 1))
    return np.cos(ang).astype(f32), np.sin(ang).astype(f32)


def footprint_table():
    xs = np.linspace(X_REAR, X_FRONT, 12)
    ys = np.linspace(-HALF_W, HALF_W, 7)[1:-1]
    pts = [(x, -HALF_W) for x in xs] + [(x, HALF_W) for x in xs]
    pts += [(X_REAR, y) for y in ys] + [(X_FRONT, y) for y in ys]
    return np.asarray(pts, np.float64).astype(f32)


# ----------------------------------------------------------------------------- fp32 math
_TWO_OVER_PI = f32(0.636619772367581343)
_PIO2_1 = f32(1.5703125)
_PIO2_2 = f32(4.837512969970703125e-4)
_PIO2_3 = f32(7.54978995489188216e-8)
_S1, _S2, _S3 = f32(-1.6666654611e-1), f32(8.3321608736e-3), f32(-1.9515295891e-4)
_C1, _C2, _C3 = f32(4.166664568298827e-2), f32(-1.388731625493765e-3), f32(2.443315711809948e-5)


def clamp32(d, lo, hi):
    """lo if d < lo, hi if d > hi, else d - spelled with selects so signed zeros are defined."""
    return np.where(d < lo, lo, np.where(d > hi, hi, d)).astype(f32)


def sincos32(a):
    """(sin, cos) of float32 angles |a| <= ~2*pi, mul/add only (Cody-Waite + cephes sinf/cosf)."""
    a = np.asarray(a, f32)
    kf = np.rint(a * _TWO_OVER_PI)
    r = ((a - kf * _PIO2_1) - kf * _PIO2_2) - kf * _PIO2_3
    q = kf.astype(i32) & 3
    z = r * r
    s = r + (r * z) * (_S1 + z * (_S2 + z * _S3))
    c = (f32(1.0) - f32(0.5) * z) + (z * z) * (_C1 + z * (_C2 + z * _C3))
    sin = np.where(q == 0, s, np.where(q == 1, c, np.where(q == 2, -s, -c)))
    cos = np.where(q == 0, c, np.where(q == 1, -s, np.where(q == 2, -c, s)))
    return sin.astype(f32), cos.astype(f32)


_LOG2E = f32(1.44269504088896341)
_LN2_1 = f32(0.693359375)
_LN2_2 = f32(-2.12194440e-4)
_E = [f32(1.9875691500e-4), f32(1.3981999507e-3), f32(8.3334519073e-3),
      f32(4.1665795894e-2), f32(1.6666665459e-1), f32(5.0000001201e-1)]


def exp32(x):
    """float32 exp, mul/add only (cephes expf), argument clamped to [-80, 80]."""
    x = clamp32(np.asarray(x, f32), f32(-80.0), f32(80.0))
    kf = np.rint(x * _LOG2E)
    r = (x - kf * _LN2_1) - kf * _LN2_2
    z = r * r
    p = _E[0]
    for c in _E[1:]:
        p = p * r + c
    y = (p * z + r) + f32(1.0)
    scale = ((kf.astype(i32) + 127) << 23).astype(i32).view(f32)
    return (y * scale).astype(f32)


# ----------------------------------------------------------------------------- Philox4x32-10
_PM0, _PM1 = u64(0xD2511F53), u64(0xCD9E8D57)
_PW0, _PW1 = 0x9E3779B9, 0xBB67AE85


def philox4x32(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al., SC'11; Random123).  All inputs broadcastable uint32."""
    c0, c1, c2, c3 = (np.asarray(c, u32).astype(u64) for c in np.broadcast_arrays(c0, c1, c2, c3))
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    m32 = u64(0xFFFFFFFF)
    for _ in range(10):
        p0 = _PM0 * c0
        p1 = _PM1 * c2
        hi0, lo0 = p0 >> u64(32), p0 & m32
        hi1, lo1 = p1 >> u64(32), p1 & m32
        c0, c1, c2, c3 = hi1 ^ c1 ^ u64(k0), lo1, hi0 ^ c3 ^ u64(k1), lo0
        k0 = (k0 + _PW0) & 0xFFFFFFFF
        k1 = (k1 + _PW1) & 0xFFFFFFFF
    return c0.astype(u32), c1.astype(u32), c2.astype(u32), c3.astype(u32)


def random_actions(seed, step, n_cars, first_car=0):
    """U(-1,1)^2 actions keyed by (seed, step, global car id): float32 [n_cars, 2]."""
    car = np.arange(first_car, first_car + n_cars, dtype=np.uint64).astype(u32)
    r0, r1, _, _ = philox4x32(car, u32(step), u32(1), u32(0), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = np.stack([(r >> u32(8)).astype(f32) * f32(2.0 ** -24) for r in (r0, r1)], axis=1)
    return (u * f32(2.0) - f32(1.0)).astype(f32)


def quantise_lidar_u16(lidar, transform=0):
    """The uint16 copy of a LiDAR row in the half-size trajectory record (include/racecar_hip.h, rc_set_compact_slab):
    q = rne(fl(fl(v + off) * scale)), (off, scale) = (0, 65535/15) for metres, (0.5, 65535) for the Dreamer scaling
    (dreamer/tools.py:274), (0, 65535) for the unit scaling (single_agent.py:92-99)."""
    off, scale = {0: (f32(0.0), f32(65535.0) / f32(15.0)), 1: (f32(0.5), f32(65535.0)), 2: (f32(0.0), f32(65535.0))}[transform]
    t = (np.asarray(lidar, f32) + off) * scale
    return np.rint(t).astype(np.uint16)        # np.rint rounds half to even, like the kernel's 2^23 add


def follow_the_gap(lidar, motor_straight=0.6, motor_corner=0.3):
    """Batched follow-the-gap (fp32 spec of racing_dreamer_amd's rc_follow_the_gap; interface of
    agents.gap_follower.GapFollower used by dreamer/dream.py:211-216): float32 [n, 2] = (motor, steering)."""
    lidar = np.asarray(lidar, f32).reshape(-1, N_BEAMS)
    lo, n, bubble = 135, 810, 60
    r = np.where(lidar[:, lo:lo + n] > f32(6.0), f32(6.0), lidar[:, lo:lo + n]).astype(f32)
    pad = np.zeros((len(r), n + 4), f32)
    pad[:, 2:-2] = r
    sm = ((((pad[:, 0:n] + pad[:, 1:n + 1]) + pad[:, 2:n + 2]) + pad[:, 3:n + 3]) + pad[:, 4:n + 4]) * f32(0.2)
    closest = sm.argmin(axis=1)                              # first minimum
    idx = np.arange(n)[None, :]
    gap = (sm > f32(2.0)) & ((idx < closest[:, None] - bubble) | (idx > closest[:, None] + bubble))
    out = np.zeros((len(r), 2), f32)
    for c in range(len(r)):
        g = np.concatenate([[0], gap[c].astype(np.int8), [0]])
        edges = np.diff(g)
        starts, ends = np.nonzero(edges == 1)[0], np.nonzero(edges == -1)[0]
        if len(starts) == 0:
            continue
        k = int((ends - starts).argmax())                    # first longest run
        centre = f32(lo) + f32(2 * starts[k] + (ends[k] - starts[k]) - 1) * f32(0.5)
        angle = f32(2.35619449019234492885) - centre * f32(0.00436737625568553)
        steering = clamp32(angle / STEER_GAIN, f32(-1.0), f32(1.0))         # the command that points the wheels at the gap
        out[c, 0] = motor_corner if abs(steering) > f32(0.35) else motor_straight
        out[c, 1] = steering
    return out


# ----------------------------------------------------------------------------- the reference's follow-the-gap law, fp32 spec
# What rc_follow_the_gap_reference computes, operation for operation: the law of the reference's ROS node
# (ros_agent/agents/follow_the_gap/src/agent.py:128-234; restated in float64 and pinned to the node's own outputs in
# oracle/ftg_reference_port.py / tests/test_golden_ftg.py) in IEEE binary32 with a defined order of operations, so that the
# device kernel can be compared bit for bit.  Differences from the float64 law, all below 1e-5 rad except the last:
#  * the arc's beam angles and all range arithmetic in fp32; arccos by the polynomial below;
#  * the means are taken over integers: the mean beam INDEX of the chosen beams (the angle is affine in it) and the chosen
#    ranges rounded to 2^-19 m - sums of integers do not depend on the order in which a wave adds them up;
#  * the percentile is NumPy's, to the bit: the 601st smallest adjusted range a plus 2^-43 of the gap to the 602nd, b
#    ((n - 1) q / 100 = 600.0000000000001 for the node's q: np.percentile's `a + (b - a) * t`, t = 2^-43), evaluated in
#    binary64 - whether the beams that hold exactly a are at or above it depends on the size of that gap, and they can be a
#    whole plateau of equal ranges (the extension writes equal values).
FTGR_FIRST, FTGR_N, FTGR_WIDTH, FTGR_RANK = 179, 721, 39, 600
FTGR_INC = f32(1.5 * np.pi / 1079)
FTGR_AMIN = f32(-0.75 * np.pi)
FTGR_LOOKAHEAD = f32(2.0 * (7.0 ** 2 / (2.0 * 8.26)))
FTGR_W2 = f32((0.3302 * 1.2) ** 2)
FTGR_MAX_STEER = f32(np.deg2rad(24.0))
FTGR_DEG5 = f32(np.deg2rad(5.0))


def acos32(x):
    """arccos in fp32 by asin's minimax polynomial on [-0.5, 0.5] (cephes asinf) and the half-angle identities; one IEEE
    operation per operator, correctly rounded sqrt.  NaN for |x| > 1."""
    x = np.asarray(x, f32)

    def asin_small(t):                                   # |t| <= 0.5
        z = t * t
        pz = ((((f32(4.2163199048e-2) * z + f32(2.4181311049e-2)) * z + f32(4.5470025998e-2)) * z + f32(7.4953002686e-2)) * z
              + f32(1.6666752422e-1)) * z
        return pz * t + t

    with np.errstate(invalid="ignore"):
        big = np.abs(x) > f32(0.5)
        half = np.sqrt((f32(1.0) - np.abs(x)) * f32(0.5))
        a_big = f32(2.0) * asin_small(np.where(big, half, f32(0.0)))
        a_big = np.where(x < 0, f32(3.14159274101257324) - a_big, a_big)
        a_small = f32(1.57079637050628662) - asin_small(np.where(big, f32(0.0), x))
        out = np.where(big, a_big, a_small).astype(f32)
        return np.where(np.abs(x) <= f32(1.0), out, f32(np.nan)).astype(f32)


def ftgr_angles():
    return (np.arange(FTGR_FIRST, FTGR_FIRST + FTGR_N).astype(f32) * FTGR_INC + FTGR_AMIN).astype(f32)


def follow_the_gap_reference(lidar, prev_heading, dt, wheel_max=None, max_velocity=None):
    """lidar float32 [n, 1080] in metres (the env's beam order), prev_heading float32 [n] (NaN = none yet), dt seconds per
    agent step.  Returns dict: action float32 [n, 2] = (motor, steering) in [-1, 1] - the node's speed over the car's top
    speed, and the command that puts the front wheels at the node's steering angle (positive command = right, the node's
    angle is positive to the left; full lock beyond the car's WHEEL_MAX) -, heading, heading_distance, steering_angle, speed [n]."""
    lidar = np.asarray(lidar, f32).reshape(-1, N_BEAMS)
    n = len(lidar)
    steer_gain = STEER_GAIN if wheel_max is None else -f32(wheel_max)
    max_velocity = MAX_VEL if max_velocity is None else f32(max_velocity)
    ang = ftgr_angles()
    # arc element a = ROS beam 179 + a = env beam 900 - a
    arc = lidar[:, 900 - np.arange(FTGR_N)]
    with np.errstate(invalid="ignore"):
        r = np.where(arc > f32(0.0), arc, f32(0.0))                               # (a NaN range counts as 0, like a negative one)
        r = np.where(r < FTGR_LOOKAHEAD, r, FTGR_LOOKAHEAD).astype(f32)
    jump = np.abs(r[:, 1:] - r[:, :-1]).astype(f32)                               # [n, 720]
    half_w = FTGR_WIDTH // 2
    peak_pad = np.pad(jump, ((0, 0), (half_w, half_w)), mode="symmetric")
    med_pad = np.pad(jump, ((0, 0), (half_w, half_w)), mode="edge")
    win = np.lib.stride_tricks.sliding_window_view
    peak = win(peak_pad, FTGR_WIDTH, axis=1).max(axis=2)
    cand = (jump == peak) & (jump > f32(0.2))
    adjusted = r.copy()
    for c, a in zip(*np.nonzero(cand)):
        med = np.sort(med_pad[c, a:a + FTGR_WIDTH])[half_w]
        if not (jump[c, a] > med * f32(9.0)):
            continue
        near = min(r[c, max(a - 1, 0)], r[c, a], r[c, a + 1])                     # (a >= 1 for every disparity)
        two = f32(2.0) * (near * near)
        with np.errstate(all="ignore"):
            cosv = (two - FTGR_W2) / two
            half = acos32(cosv)
        if np.isnan(half):
            ia = ib = 0
        else:
            lo = ((ang[a] - half) - ang[0]) / FTGR_INC
            hi = ((ang[a] + half) - ang[0]) / FTGR_INC
            ia, ib = (int(np.clip(np.trunc(v), 0, FTGR_N - 1)) for v in (lo, hi))
        adjusted[c, ia:ib + 1] = np.minimum(adjusted[c, ia:ib + 1], near)
    srt = np.sort(adjusted, axis=1).astype(np.float64)
    thr = srt[:, FTGR_RANK] + (srt[:, FTGR_RANK + 1] - srt[:, FTGR_RANK]) * 2.0 ** -43        # binary64, one rounding (the add)
    chosen = (adjusted.astype(np.float64) >= thr[:, None]) & (adjusted < MAX_RANGE)
    count = chosen.sum(axis=1).astype(f32)
    sum_k = (chosen * np.arange(FTGR_N)[None, :]).sum(axis=1).astype(f32)
    heading = (((sum_k / count) + f32(FTGR_FIRST)) * FTGR_INC + FTGR_AMIN).astype(f32)
    q = np.rint(r * f32(524288.0)).astype(np.uint32)                              # 2^19 counts per metre
    sum_q = (q * chosen).sum(axis=1, dtype=np.uint64).astype(np.uint32).astype(f32)
    hd = ((sum_q / count) * f32(1.0 / 524288.0)).astype(f32)
    prev = np.asarray(prev_heading, f32).reshape(n)
    dt = f32(dt)
    with np.errstate(invalid="ignore"):
        d_term = np.where(np.isnan(prev), f32(0.0), (f32(0.1) * (prev - heading)) / dt).astype(f32)
    steer = np.minimum(np.maximum(f32(1.4) * heading - d_term, -FTGR_MAX_STEER), FTGR_MAX_STEER).astype(f32)
    speed = np.where(np.abs(steer) > FTGR_DEG5, f32(6.0) - (np.abs(steer) / FTGR_MAX_STEER) * f32(1.8), f32(6.0)).astype(f32)
    speed = np.where(hd < f32(5.0), np.minimum(speed, (hd / f32(5.0)) * f32(4.0)), speed).astype(f32)
    speed = np.maximum(speed, f32(1.5)).astype(f32)
    action = np.stack([clamp32(speed / max_velocity, f32(-1.0), f32(1.0)), clamp32(steer / steer_gain, f32(-1.0), f32(1.0))], 1).astype(f32)
    return dict(action=action, heading=heading, heading_distance=hd, steering_angle=steer, speed=speed)


# ----------------------------------------------------------------------------- the env
def _obb_overlap_poses(xa, ya, cta, sta, xb, yb, ctb, stb):
    """Oriented-rectangle overlap of two cars by separating axes (H5): rear-axle poses in, uint8 out."""
    ax = xa + BOX_CX * cta
    ay = ya + BOX_CX * sta
    bx = xb + BOX_CX * ctb
    by = yb + BOX_CX * stb
    dx, dy = bx - ax, by - ay
    c = np.abs(cta * ctb + sta * stb)
    s = np.abs(sta * ctb - cta * stb)
    ra = BOX_HL + (BOX_HL * c + BOX_HW * s)      # projections on A's long axis (and B's)
    rb = BOX_HW + (BOX_HL * s + BOX_HW * c)      # on the short axes
    sep = np.abs(dx * cta + dy * sta) > ra
    sep |= np.abs(dy * cta - dx * sta) > rb
    sep |= np.abs(dx * ctb + dy * stb) > ra
    sep |= np.abs(dy * ctb - dx * stb) > rb
    return (~sep).astype(np.uint8)


class OracleConfig:
    def __init__(self, num_envs=1, cars_per_env=1, laps=10, time_limit=180.0,
                 terminate_on_collision=True, collision_reward=-1.0, task=TASK_MAX_PROGRESS,
                 remap_actions=False, action_low=(0.005, -1.0), action_high=(1.0, 1.0),
                 time_limit_steps=0, auto_reset=False, first_env=0, render_occupancy=False,
                 car_tasks=None, n_steps=10):
        """car_tasks: task id per car slot (None / -1 entries = `task`); n_steps: window of TASK_N_STEP_PROGRESS, the
        task of the secondary agents B-D in baselines/scenarios/max_progress/columbia.yml:17-18,25-26,33-34 (its law
        lives in the un-vendored racecar_gym, so the spec fixes one: reward = PROGRESS_REWARD x the total progress
        (lap - 1 + progress) gained over the last n_steps sub-steps - since the reset while the episode is younger -
        no collision term, never done)."""
        self.__dict__.update(locals())
        del self.__dict__["self"]
        assert 1 <= n_steps <= NSTEP_MAX

    def task_of(self, a):
        t = -1 if self.car_tasks is None or a >= len(self.car_tasks) else int(self.car_tasks[a])
        return self.task if t < 0 else t


class OracleRaceEnv:
    """Vectorised NumPy restatement of the env spec.  One track per instance."""

    def __init__(self, occ, drivable, progress, centerline, origin, resolution, cfg: OracleConfig):
        self.cfg = cfg
        self.occ = np.asarray(occ, bool).copy()
        # sentinel ring: the outermost cells of the grid end every ray with "no return" and count
        # as wall for the footprint test, so the traversal needs no per-step bounds check
        self.ring = np.zeros_like(self.occ)
        self.ring[0, :] = self.ring[-1, :] = self.ring[:, 0] = self.ring[:, -1] = True
        self.occ |= self.ring
        self.drv = np.asarray(drivable, bool) & ~self.ring      # spec: the outermost ring of cells is not drivable
        self.progress_grid = np.asarray(progress, f32)
        self.centerline = np.asarray(centerline, f32)
        self.H, self.W = self.occ.shape
        self.org_x, self.org_y = f32(origin[0]), f32(origin[1])
        self.res = f32(resolution)
        self.inv_res = f32(1.0 / resolution)
        self.tmax = MAX_RANGE * self.inv_res
        self.B, self.A = cfg.num_envs, cfg.cars_per_env
        n = self.NC = self.B * self.A
        self.cb, self.sb = beam_table()
        self.fp = footprint_table()
        z = lambda dt: np.zeros(n, dt)
        self.x, self.y, self.theta, self.ct, self.st = z(f32), z(f32), z(f32), z(f32), z(f32)
        self.v, self.delta, self.omega, self.accel = z(f32), z(f32), z(f32), z(f32)
        self.progress, self.lap, self.cp = z(f32), z(i32), z(i32)
        self.wall, self.opp, self.wrong_way = z(np.uint8), z(np.uint8), z(np.uint8)
        self.done, self.truncated, self.fresh = z(np.uint8), z(np.uint8), z(np.uint8)
        self.reward = z(f32)
        self.nstep_hist = np.zeros((n, NSTEP_MAX), f32)      # total progress at sub-step s, slot s % n_steps
        self.action = np.zeros((n, 2), f32)
        self.steps = np.zeros(self.B, i32)         # sub-steps since reset
        self.agent_steps = np.zeros(self.B, i32)   # step() calls since reset
        self.episode = np.zeros(self.B, u32)       # resets so far (Philox counter word 1)
        self.needs_reset = np.ones(self.B, bool)
        self.lidar = np.zeros((n, N_BEAMS), f32)
        self.patch = np.zeros((n, PATCH, PATCH), np.uint8)
        self.seed = 0
        self.mode = RESET_GRID

    # ------------------------------------------------------------------ helpers
    def spawn_width(self):
        """float32 [n_centerline]: how far a `random` start may be moved sideways from centre-line point i - exact integer
        arithmetic: d2 = squared distance in cells from the point's cell to the nearest cell that is not drivable (the grid's
        outside included) in the window of +- SPAWN_CLEAR_R cells, at most (R + 1)^2; w = clamp(isqrt(d2) * res - SPAWN_MARGIN,
        0, SPAWN_W_MAX).  The Euclidean distance map is 1-Lipschitz, so a rear-axle point within w of the centre-line point
        keeps SPAWN_MARGIN to every wall whatever the heading: no start touches a wall."""
        if getattr(self, "_spawn_w", None) is not None:
            return self._spawn_w
        R = SPAWN_CLEAR_R
        ix, iy = self._cell(self.centerline[:, 0], self.centerline[:, 1])
        blocked = np.pad(~self.drv, R + 1, constant_values=True)
        off = np.arange(-R, R + 1)
        dist2 = (off[:, None] ** 2 + off[None, :] ** 2).astype(np.int64)
        d2 = np.full(len(ix), (R + 1) ** 2, np.int64)
        inside = (ix >= 0) & (ix < self.W) & (iy >= 0) & (iy < self.H)
        for k in np.nonzero(inside)[0]:
            win = blocked[iy[k] + 1:iy[k] + 2 * R + 2, ix[k] + 1:ix[k] + 2 * R + 2]
            if win.any():
                d2[k] = min(int(dist2[win].min()), (R + 1) ** 2)
        d2[~inside] = 0
        k = np.floor(np.sqrt(d2.astype(np.float64))).astype(np.int64)
        k = np.where((k + 1) ** 2 <= d2, k + 1, np.where(k * k > d2, k - 1, k))           # exact integer square root
        w = clamp32(k.astype(f32) * self.res - SPAWN_MARGIN, f32(0.0), SPAWN_W_MAX)
        self._spawn_w = w.astype(f32)
        return self._spawn_w

    def _wall_hit_poses(self, x, y, ct, st):
        """`_wall_hit` on poses instead of car slots (the same 34 fixed-point probes)."""
        k = FOOT_STEP * self.inv_res
        ex = np.rint((ct * k) * Q16).astype(np.int64)
        ey = np.rint((st * k) * Q16).astype(np.int64)
        x0 = np.rint(((x - self.org_x) * self.inv_res) * Q16).astype(np.int64)
        y0 = np.rint(((y - self.org_y) * self.inv_res) * Q16).astype(np.int64)
        hit = np.zeros(len(x0), bool)
        for i, j in FOOT_LATTICE:
            px = x0 + (i - 2) * ex - (j - 3) * ey
            py = y0 + (i - 2) * ey + (j - 3) * ex
            hit |= self._lookup(self.occ, (px >> 16).astype(i32), (py >> 16).astype(i32), True)
        return hit

    def spawn_usable(self):
        """bool [n_centerline]: the centre-line pose of bin i touches no wall (the footprint test of H5 on the table's own
        pose).  On hand-drawn maps with boxes on the track or one-cell corridors the most central cell of a BFS distance bin can
        lie where a car does not fit; such a bin is never a start (`spawn_rows`)."""
        if getattr(self, "_spawn_usable", None) is None:
            sn, cs = sincos32(self.centerline[:, 2])
            self._spawn_usable = ~self._wall_hit_poses(self.centerline[:, 0], self.centerline[:, 1], cs, sn)
        return self._spawn_usable

    def spawn_heading_room(self):
        """float32 [n_centerline]: how far a start at bin i may turn off the track's direction.  HEADING_JITTER where the bin has
        lateral room (spawn_width > 0: the margin argument holds for every heading); where it has none, HEADING_ROOM[k] with
        k = the smallest footprint-point clearance of the centre-line pose in cells (integers, see HEADING_ROOM)."""
        if getattr(self, "_spawn_h", None) is not None:
            return self._spawn_h
        R = SPAWN_FOOT_R
        n = len(self.centerline)
        sn, cs = sincos32(self.centerline[:, 2])
        k = FOOT_STEP * self.inv_res
        ex = np.rint((cs * k) * Q16).astype(np.int64)
        ey = np.rint((sn * k) * Q16).astype(np.int64)
        x0 = np.rint(((self.centerline[:, 0] - self.org_x) * self.inv_res) * Q16).astype(np.int64)
        y0 = np.rint(((self.centerline[:, 1] - self.org_y) * self.inv_res) * Q16).astype(np.int64)
        blocked = np.pad(~self.drv, R + 1, constant_values=True)
        off = np.arange(-R, R + 1)
        dist2 = (off[:, None] ** 2 + off[None, :] ** 2).astype(np.int64)[None]
        big = (R + 1) ** 2
        kmin = np.full(n, R, np.int64)
        for i, j in FOOT_LATTICE:
            ix = ((x0 + (i - 2) * ex - (j - 3) * ey) >> 16)
            iy = ((y0 + (i - 2) * ey + (j - 3) * ex) >> 16)
            inside = (ix >= 0) & (ix < self.W) & (iy >= 0) & (iy < self.H)
            jx, jy = np.clip(ix, 0, self.W - 1), np.clip(iy, 0, self.H - 1)
            win = blocked[(jy + 1)[:, None, None] + np.arange(2 * R + 1)[None, :, None], (jx + 1)[:, None, None] + np.arange(2 * R + 1)[None, None, :]]
            d2 = np.where(win, dist2, big).reshape(n, -1).min(1)
            d2 = np.where(inside, d2, 0)
            kk = np.floor(np.sqrt(d2.astype(np.float64))).astype(np.int64)
            kk = np.where((kk + 1) ** 2 <= d2, kk + 1, np.where(kk * kk > d2, kk - 1, kk))       # exact integer square root
            kmin = np.minimum(kmin, kk)
        h = np.where(self.spawn_width() > f32(0.0), HEADING_JITTER, HEADING_ROOM[np.clip(kmin, 0, R)])
        self._spawn_h = h.astype(f32)
        return self._spawn_h

    def spawn_rows(self):
        """The spawn table: (x, y, theta, lateral room w, heading room h) float32 [n_centerline] each.  Row i is bin u(i) = the
        first USABLE bin among i, i + 1, ..., i + SPAWN_SAFE_SEARCH - 1 around the lap (i itself if there is none): a start drawn
        at a bin whose centre-line pose touches a wall goes to the next bin where a car fits.  On every track a scenario of the
        reference names u(i) = i."""
        if getattr(self, "_spawn_rows", None) is not None:
            return self._spawn_rows
        n = len(self.centerline)
        usable = self.spawn_usable()
        i = np.arange(n)
        u, found = i.copy(), usable.copy()
        for k in range(1, min(SPAWN_SAFE_SEARCH, n)):
            j = (i + k) % n
            take = ~found & usable[j]
            u[take] = j[take]
            found |= take
        self._spawn_u = u
        self._spawn_rows = (self.centerline[u, 0], self.centerline[u, 1], self.centerline[u, 2], self.spawn_width()[u],
                            self.spawn_heading_room()[u])
        return self._spawn_rows

    def spawn_safe(self):
        """int64 [n_centerline]: where a multi-car start drawn at bin i really goes.  Bin j is SOUND if bins j, j - 12, j - 24,
        j - 36 are all usable (`spawn_usable`) and the centre-line poses of RC_MAX_CARS = 4 cars there do not overlap pairwise
        (the rectangle test of H5 on the table's
        own poses); safe[i] = the first sound bin among i, i + 1, ..., i + SPAWN_SAFE_SEARCH - 1 (around the lap), i itself if
        there is none.  A centre line is the most central cell per BFS distance bin, and where the BFS wavefronts of the
        progress grid fold - the raw columbia.pgm's (track `columbia_slam`) last bins run back along the bins before them, the start pixel lying in an open area
         - bins 1.2 m apart along the table are centimetres apart on the ground; such bins are never the
        anchor of a multi-car start.  One car: not used (every bin is a start)."""
        if getattr(self, "_spawn_safe", None) is not None:
            return self._spawn_safe
        n = len(self.centerline)
        x, y = self.centerline[:, 0], self.centerline[:, 1]
        sn, cs = sincos32(self.centerline[:, 2])
        i = np.arange(n)
        sound = np.ones(n, bool)
        usable = self.spawn_usable()
        for a in range(MAX_CARS):
            sound &= usable[(i - a * BALL_GAP_BINS) % n]
        for a in range(MAX_CARS):
            for b in range(a + 1, MAX_CARS):
                ia, ib = (i - a * BALL_GAP_BINS) % n, (i - b * BALL_GAP_BINS) % n
                sound &= _obb_overlap_poses(x[ia], y[ia], cs[ia], sn[ia], x[ib], y[ib], cs[ib], sn[ib]) == 0
        safe = i.copy()
        found = sound.copy()
        for k in range(1, min(SPAWN_SAFE_SEARCH, n)):
            j = (i + k) % n
            take = ~found & sound[j]
            safe[take] = j[take]
            found |= take
        self._spawn_safe = safe.astype(np.int64)
        return self._spawn_safe

    def _cell(self, wx, wy):
        gx = (wx - self.org_x) * self.inv_res
        gy = (wy - self.org_y) * self.inv_res
        return np.floor(gx).astype(i32), np.floor(gy).astype(i32)

    def _lookup(self, grid, ix, iy, oob):
        inb = (ix >= 0) & (ix < self.W) & (iy >= 0) & (iy < self.H)
        out = np.full(ix.shape, oob, grid.dtype)
        out[inb] = grid[iy[inb], ix[inb]]
        return out

    # ------------------------------------------------------------------ reset (H6)
    def reset(self, mask=None, mode=RESET_GRID, seed=0):
        self.seed, self.mode = int(seed), int(mode)
        envs = np.arange(self.B) if mask is None else np.nonzero(np.asarray(mask).astype(bool))[0]
        self._reset_envs(envs)
        self._observe()
        return self.outputs()

    def _reset_envs(self, envs):
        """Reset law (H6).  `grid`: the cars on the centre line behind the start, BALL_GAP_BINS apart.  `random` (one car) and
        `random_ball` (several): one Philox draw per env picks a centre-line bin uniformly over the lap; car a stands at bin
        idx0 - a * BALL_GAP_BINS (row idx of the spawn table: `spawn_rows`), moved SIDEWAYS by u * w (u uniform in [-1, 1), w = that
        row's lateral room: the clearance the track leaves there) and turned by v * h off the track's direction (v uniform in
        [-1, 1), h = the row's heading room: HEADING_JITTER wherever there is lateral room) - "a random pose
        on the track with a minimum wall distance, heading along the track" (SURVEY.md H6 and appendix A); the cars of an env
        lie within a ball of 1.2 (A - 1) m + the track's width around the drawn point ("sample in random points close within
        a ball", dreamer/dream.py:105-108).  Should two of the proposed cars overlap (rectangle test of H5), ALL cars of that
        env take the centre-line poses instead (u = v = 0), which never overlap.  Words: Philox(global env id, episode, k, 0):
        k = 0 -> (bin, u_0, v_0, -), k = 1 -> (u_1, v_1, u_2, v_2), k = 2 -> (u_3, v_3, -, -)."""
        if envs.size == 0:
            return
        cfg, n_cl = self.cfg, len(self.centerline)
        g = (envs + cfg.first_env).astype(np.uint64).astype(u32)
        k0, k1 = self.seed & 0xFFFFFFFF, (self.seed >> 32) & 0xFFFFFFFF
        words = [philox4x32(g, self.episode[envs], u32(0), u32(0), k0, k1)]
        for call in range(1, 1 + (self.A - 1 + 1) // 2):
            words.append(philox4x32(g, self.episode[envs], u32(call), u32(0), k0, k1))
        r0 = words[0][0]
        self.episode[envs] += u32(1)
        jitter = self.mode != RESET_GRID
        if not jitter:
            idx0 = np.full(envs.size, BALL_GAP_BINS * (self.A - 1) + GRID_LEAD_BINS, np.int64)
        else:
            idx0 = ((r0.astype(u64) * u64(n_cl)) >> u64(32)).astype(np.int64)
        if self.A > 1:
            idx0 = self.spawn_safe()[idx0]              # never anchor several cars where the centre line folds: spawn_safe (the grid too)
        unit = lambda w: (w >> u32(8)).astype(f32) * f32(5.9604644775390625e-8) * f32(2.0) - f32(1.0)     # [-1, 1), exact
        row_x, row_y, row_th, width, heading = self.spawn_rows()
        centre, proposed = [], []
        for a in range(self.A):
            idx = (idx0 - a * BALL_GAP_BINS) % n_cl
            cx, cy, th = row_x[idx], row_y[idx], row_th[idx]
            st0, ct0 = sincos32(th)
            centre.append((cx, cy, th, st0, ct0))
            if not jitter:
                proposed.append(centre[-1])
                continue
            wu, wv = (words[0][1], words[0][2]) if a == 0 else (words[1 + (a - 1) // 2][2 * ((a - 1) % 2)],
                                                                words[1 + (a - 1) // 2][2 * ((a - 1) % 2) + 1])
            off = unit(wu) * width[idx]
            x = cx - off * st0
            y = cy + off * ct0
            th2 = th + unit(wv) * heading[idx]
            th2 = np.where(th2 > PI, th2 - TWO_PI, th2)
            th2 = np.where(th2 < -PI, th2 + TWO_PI, th2)
            st2, ct2 = sincos32(th2)
            proposed.append((x, y, th2, st2, ct2))
        for a in range(self.A):
            cars = envs * self.A + a
            self.x[cars], self.y[cars], self.theta[cars], self.st[cars], self.ct[cars] = proposed[a]
        if jitter and self.A > 1:
            clash = np.zeros(envs.size, bool)
            for a in range(self.A):
                for b in range(a + 1, self.A):
                    clash |= self._obb_overlap(envs * self.A + a, envs * self.A + b) != 0
            for a in range(self.A):
                cars = (envs * self.A + a)[clash]
                for arr, val in zip((self.x, self.y, self.theta, self.st, self.ct), centre[a]):
                    arr[cars] = val[clash]
        for a in range(self.A):
            cars = envs * self.A + a
            ix, iy = self._cell(self.x[cars], self.y[cars])
            p = self._lookup(self.progress_grid, ix, iy, f32(-1.0))
            p = np.where(p < f32(0.0), f32(0.0), p)
            self.progress[cars] = p
            self.cp[cars] = np.minimum((p * f32(N_CHECKPOINTS)).astype(i32), N_CHECKPOINTS - 1)
            for arr in (self.v, self.delta, self.omega, self.accel, self.reward):
                arr[cars] = f32(0.0)
            for arr in (self.wall, self.opp, self.wrong_way, self.done, self.truncated):
                arr[cars] = 0
            self.lap[cars] = 1
            self.fresh[cars] = 1
            self.action[cars] = f32(0.0)
            self.nstep_hist[cars, :] = p[:, None]
        self.steps[envs] = 0
        self.agent_steps[envs] = 0
        self.needs_reset[envs] = False

    # ------------------------------------------------------------------ dynamics (H2)
    def _substep(self, envs, motor, steer):
        """One dt for every car of `envs`.  motor/steer: float32 [len(envs), A]."""
        A = self.A
        for a in range(A):
            c = envs * A + a
            m, s = motor[:, a], steer[:, a]
            v, delta, theta = self.v[c], self.delta[c], self.theta[c]
            force = np.abs(m) * ACCEL_MAX
            acc = np.where(m >= f32(0.0), force, -force) - DRAG * v
            v = clamp32(v + acc * DT, f32(0.0), MAX_VEL)
            dd = clamp32(s * STEER_GAIN - delta, -STEER_STEP, STEER_STEP)
            delta = delta + dd
            sd, cd = sincos32(delta)
            omega = (v / WHEELBASE) * (sd / cd)
            self.x[c] = self.x[c] + (v * self.ct[c]) * DT
            self.y[c] = self.y[c] + (v * self.st[c]) * DT
            theta = theta + omega * DT
            theta = np.where(theta > PI, theta - TWO_PI, theta)
            theta = np.where(theta < -PI, theta + TWO_PI, theta)
            self.theta[c] = theta
            self.st[c], self.ct[c] = sincos32(theta)
            self.v[c], self.delta[c], self.omega[c] = v, delta, omega
            self.accel[c] = acc.astype(f32)
        self.steps[envs] += 1
        # --- collisions (H5)
        for a in range(A):
            c = envs * A + a
            self.wall[c] = self._wall_hit(c)
            self.opp[c] = 0
        for a in range(A):
            for b in range(a + 1, A):
                ca, cb_ = envs * A + a, envs * A + b
                o = self._obb_overlap(ca, cb_)
                self.opp[ca] |= o
                self.opp[cb_] |= o
        # --- progress / reward / done (H4, H15)
        cfg = self.cfg
        tlim = f32(cfg.time_limit)
        time = self.steps[envs].astype(f32) * DT
        for a in range(A):
            c = envs * A + a
            ix, iy = self._cell(self.x[c], self.y[c])
            p_new = self._lookup(self.progress_grid, ix, iy, f32(-1.0))
            p_old, lap_old, cp_old = self.progress[c], self.lap[c], self.cp[c]
            valid = p_new >= f32(0.0)
            p_new = np.where(valid, p_new, p_old)
            cp_new = np.minimum((p_new * f32(N_CHECKPOINTS)).astype(i32), N_CHECKPOINTS - 1)
            d = np.mod(cp_new - cp_old, N_CHECKPOINTS)
            fwd = (d > 0) & (d <= N_CHECKPOINTS // 2)
            bwd = d > N_CHECKPOINTS // 2
            lap = lap_old + np.where(fwd & (cp_new < cp_old), 1, 0) - np.where(bwd & (cp_new > cp_old), 1, 0)
            lap = lap.astype(i32)
            self.wrong_way[c] = np.where(fwd, 0, np.where(bwd, 1, self.wrong_way[c]))
            self.cp[c] = np.where(fwd | bwd, cp_new, cp_old)
            self.lap[c], self.progress[c] = lap, p_new
            collided = (self.wall[c] | self.opp[c]).astype(bool)
            task = cfg.task_of(a)
            if task == TASK_N_STEP_PROGRESS:
                slot = self.steps[envs] % cfg.n_steps
                total = (lap - 1).astype(f32) + p_new
                r = (total - self.nstep_hist[c, slot]) * PROGRESS_REWARD
                self.nstep_hist[c, slot] = total
                done = np.zeros(envs.size, bool)
            elif task == TASK_MAX_PROGRESS:
                delta = (lap - lap_old).astype(f32) + (p_new - p_old)
                r = delta * PROGRESS_REWARD + np.where(collided, f32(cfg.collision_reward), f32(0.0))
                done = (collided & bool(cfg.terminate_on_collision)) | (lap > cfg.laps) | (time > tlim)
            else:  # baselines/racing/environment/tasks.py:6-18
                r = np.where(self.wall[c].astype(bool), f32(-1.0), -exp32(np.abs(steer[:, a]) - self.v[c]))
                done = np.zeros(envs.size, bool)
            self.reward[c] = self.reward[c] + r.astype(f32)
            self.done[c] = done

    def _wall_hit(self, c):
        """Wall contact of cars `c`: any of the 34 perimeter points of the 0.55 x 0.30 m body rectangle in an occupied
        (or ring, or off-grid) cell.  The points are the border of a 12 x 7 lattice of 0.05 m pitch in the body frame
        (`footprint_table`), evaluated in 16.16 FIXED-POINT cell coordinates so that the device walks them with integer
        adds and every implementation lands in exactly the same cells: rear axle P0 = rne(65536 g), lattice vectors
        e = rne(65536 k (cos, sin)), f = (-e.y, e.x) with k = 0.05 m / resolution, point (i, j) = P0 + (i - 2) e +
        (j - 3) f, cell = point >> 16 (arithmetic shift).  A car whose position is not a finite number within 8192
        cells of the grid origin counts as in contact."""
        k = FOOT_STEP * self.inv_res
        ex = np.rint((self.ct[c] * k) * Q16).astype(np.int64)
        ey = np.rint((self.st[c] * k) * Q16).astype(np.int64)
        gx = (self.x[c] - self.org_x) * self.inv_res
        gy = (self.y[c] - self.org_y) * self.inv_res
        with np.errstate(invalid="ignore"):
            bad = ~((np.abs(gx) <= f32(8192.0)) & (np.abs(gy) <= f32(8192.0)))
            x0 = np.rint(np.where(bad, f32(0.0), gx) * Q16).astype(np.int64)
            y0 = np.rint(np.where(bad, f32(0.0), gy) * Q16).astype(np.int64)
        hit = bad.copy()
        for i, j in FOOT_LATTICE:
            px = x0 + (i - 2) * ex - (j - 3) * ey
            py = y0 + (i - 2) * ey + (j - 3) * ex
            hit |= self._lookup(self.occ, (px >> 16).astype(i32), (py >> 16).astype(i32), True)
        return hit

    def _obb_overlap(self, ca, cb_):
        return _obb_overlap_poses(self.x[ca], self.y[ca], self.ct[ca], self.st[ca], self.x[cb_], self.y[cb_], self.ct[cb_], self.st[cb_])

    # ------------------------------------------------------------------ step (H7-H10)
    def step(self, actions, repeat=1):
        cfg, A = self.cfg, self.A
        actions = np.asarray(actions, f32).reshape(self.B, A, 2)
        assert not self.needs_reset.any(), "Must reset environment."   # dreamer/wrappers.py:148
        env_done = self.done.reshape(self.B, A).any(axis=1)
        live = np.nonzero(~env_done)[0]
        self.fresh[:] = 0
        self.reward[:] = f32(0.0)
        self.action[:] = actions.reshape(-1, 2)
        a = actions[live]
        if cfg.remap_actions:                                          # dreamer/wrappers.py:128-130
            lo, hi = np.asarray(cfg.action_low, f32), np.asarray(cfg.action_high, f32)
            a = ((a + f32(1.0)) * f32(0.5)) * (hi - lo) + lo
        a = clamp32(a, f32(-1.0), f32(1.0))
        motor, steer = a[..., 0], a[..., 1]
        active = np.ones(live.size, bool)
        for _ in range(repeat):                                        # dreamer/wrappers.py:107-116
            if not active.any():
                break
            self._substep(live[active], motor[active], steer[active])
            active &= ~self.done.reshape(self.B, A)[live].any(axis=1)
        self.agent_steps[live] += 1
        if cfg.time_limit_steps > 0:                                   # dreamer/wrappers.py:147-154
            over = live[self.agent_steps[live] >= cfg.time_limit_steps]
            oc = (over[:, None] * A + np.arange(A)[None, :]).ravel()
            self.done[oc] = 1
            self.truncated[oc] = 1
        out = self._scalar_outputs()
        if cfg.auto_reset:
            self._reset_envs(np.nonzero(self.done.reshape(self.B, A).any(axis=1))[0])
        self._observe()
        out.update(self._obs_outputs())
        return out

    # ------------------------------------------------------------------ observations
    def _observe(self):
        self.lidar = self.raycast()
        if self.cfg.render_occupancy == "reference":
            self.patch = self.render_patch_exact()
        elif self.cfg.render_occupancy:
            self.patch = self.render_patch()

    def render_patch_exact(self):
        """obs_type `lidar_occupancy_reference`: the reference's own crop -> spline rotation -> bicubic resize restated to the
        binary64 operation (oracle/patch_reference.py, the spec; dreamer/wrappers.py:396-406); zeros on the first observation
        of an episode (:413).  Needs `self.frame_track` = the Track (its crop of the source image places the north-up pixel
        frame the reference indexes)."""
        from . import patch_reference as px
        poses = np.stack([self.x, self.y, self.theta], 1).astype(np.float64)
        out = px.render_patch_exact(self.frame_track, poses)
        out[self.fresh != 0] = 0
        return out

    def raycast(self, chunk_cars=2048):
        out = np.empty((self.NC, N_BEAMS), f32)
        for c0 in range(0, self.NC, chunk_cars):
            out[c0:c0 + chunk_cars] = self._raycast_cars(np.arange(c0, min(c0 + chunk_cars, self.NC)))
        return out

    def _raycast_cars(self, cars):
        """LiDAR scan (H3): exact grid traversal, boundaries from integer cell indices."""
        A, n = self.A, cars.size
        ct, st = self.ct[cars][:, None], self.st[cars][:, None]
        lx = self.x[cars][:, None] + LIDAR_X * ct
        ly = self.y[cars][:, None] + LIDAR_X * st
        dx = np.broadcast_to(ct * self.cb[None, :] - st * self.sb[None, :], (n, N_BEAMS)).ravel()
        dy = np.broadcast_to(st * self.cb[None, :] + ct * self.sb[None, :], (n, N_BEAMS)).ravel()
        gx = np.broadcast_to((lx - self.org_x) * self.inv_res, (n, N_BEAMS)).ravel()
        gy = np.broadcast_to((ly - self.org_y) * self.inv_res, (n, N_BEAMS)).ravel()
        ix, iy = np.floor(gx).astype(i32), np.floor(gy).astype(i32)
        rng = np.full(n * N_BEAMS, MAX_RANGE, f32)
        inb = (ix >= 0) & (ix < self.W) & (iy >= 0) & (iy < self.H)
        start_hit = ~inb
        start_hit[inb] = self.occ[iy[inb], ix[inb]]    # includes the sentinel ring
        rng[start_hit] = f32(0.0)
        with np.errstate(divide="ignore"):
            idx = np.where(dx != 0, f32(1.0) / dx, f32(0.0)).astype(f32)
            idy = np.where(dy != 0, f32(1.0) / dy, f32(0.0)).astype(f32)
        sx = np.where(dx > 0, 1, -1).astype(i32)
        sy = np.where(dy > 0, 1, -1).astype(i32)
        bx = (ix + (dx > 0)).astype(f32)
        by = (iy + (dy > 0)).astype(f32)
        tx = np.where(dx != 0, (bx - gx) * idx, INF).astype(f32)
        ty = np.where(dy != 0, (by - gy) * idy, INF).astype(f32)
        act = np.nonzero(~start_hit)[0]
        while act.size:
            txa, tya = tx[act], ty[act]
            stepx = txa < tya
            t = np.where(stepx, txa, tya)
            over = t >= self.tmax                      # no return within 15 m
            keep = ~over
            act, stepx, t = act[keep], stepx[keep], t[keep]
            ax, ay = act[stepx], act[~stepx]
            ix[ax] += sx[ax]
            bx[ax] += sx[ax].astype(f32)
            tx[ax] = (bx[ax] - gx[ax]) * idx[ax]
            iy[ay] += sy[ay]
            by[ay] += sy[ay].astype(f32)
            ty[ay] = (by[ay] - gy[ay]) * idy[ay]
            cx, cy = ix[act], iy[act]
            stop = self.occ[cy, cx]                    # wall or sentinel ring (never out of bounds)
            hit = stop & ~self.ring[cy, cx]            # the ring itself gives no return
            rng[act[hit]] = t[hit] * self.res
            act = act[~stop]
        rng = rng.reshape(n, N_BEAMS)
        if A > 1:                                      # inter-car returns (H18)
            env = cars // A
            for other in range(A):
                oc = env * A + other
                tcar = self._ray_vs_car(lx, ly, dx.reshape(n, N_BEAMS), dy.reshape(n, N_BEAMS), oc)
                tcar = np.where((oc != cars)[:, None], tcar, INF)
                rng = np.where(tcar < rng, tcar, rng)
        return rng.astype(f32)

    def _ray_vs_car(self, lx, ly, dx, dy, oc):
        """Distance [m] along each ray to car `oc`'s rectangle (slab test in its body frame)."""
        ct2, st2 = self.ct[oc][:, None], self.st[oc][:, None]
        cx = self.x[oc][:, None] + BOX_CX * ct2
        cy = self.y[oc][:, None] + BOX_CX * st2
        rx, ry = lx - cx, ly - cy
        px = rx * ct2 + ry * st2
        py = ry * ct2 - rx * st2
        ex = dx * ct2 + dy * st2
        ey = dy * ct2 - dx * st2
        tn = np.full(dx.shape, -INF, f32)
        tf = np.full(dx.shape, INF, f32)
        miss = np.zeros(dx.shape, bool)
        with np.errstate(divide="ignore", invalid="ignore"):
            for p, e, h in ((px, ex, BOX_HL), (py, ey, BOX_HW)):
                p = np.broadcast_to(p, dx.shape)
                par = e == 0
                inv = np.where(par, f32(0.0), f32(1.0) / np.where(par, f32(1.0), e)).astype(f32)
                t1 = (-h - p) * inv
                t2 = (h - p) * inv
                lo, hi = np.where(t1 < t2, t1, t2), np.where(t1 < t2, t2, t1)
                tn = np.where(par, tn, np.where(lo > tn, lo, tn))
                tf = np.where(par, tf, np.where(hi < tf, hi, tf))
                miss |= par & (np.abs(p) > h)
        hit = ~miss & (tn <= tf) & (tf >= f32(0.0))
        t = np.where(tn > f32(0.0), tn, f32(0.0))
        return np.where(hit & (t < MAX_RANGE), t, INF).astype(f32)

    def render_patch(self, cars=None):
        """lidar_occupancy (H11, dreamer/wrappers.py:390-408): ego-aligned 64x64, heading = +col, 1 = drivable.
        Direct inverse map of the reference's crop -> rotate -> centre-crop -> resize chain: the patch is centred on
        the north-west corner of the car's cell (the centre of the reference's [pr-110, pr+110) x [pc-110, pc+110)
        crop), one nearest-cell tap per output pixel (3.125 cells per pixel), and taps outside that 220-cell window
        read 0 like the corners the reference's rotation leaves empty.

        The tap positions are an incremental FIXED-POINT walk (16 fractional bits), so that a row of the patch costs
        two integer adds per pixel on the device and the result is exactly reproducible: with
        (a, b) = rne(3.125 * 65536 * (cos, sin)) the tap of pixel (row r, column c) sits at cell offset
        (X >> 16, Y >> 16), X = X00 + c a + r b, Y = Y00 + c b - r a, X00 = (63 (-a - b)) >> 1, Y00 = (63 (a - b)) >> 1
        (the pixel centres (c - 31.5, r - 31.5) rotated by the heading; >> is the arithmetic shift = floor).  The
        step vector is off by at most 2^-17 cell, 5e-4 cell over the 63 steps of a row: far inside the 99 %
        agreement with the reference's own patches that pins this function (tests/test_golden_patch.py)."""
        cars = np.arange(self.NC) if cars is None else cars
        r = np.arange(PATCH, dtype=np.int64)[None, :, None]
        c = np.arange(PATCH, dtype=np.int64)[None, None, :]
        out = np.zeros((cars.size, PATCH, PATCH), np.uint8)
        for k0 in range(0, cars.size, 256):
            cc = cars[k0:k0 + 256]
            a = np.rint(self.ct[cc] * PATCH_STEP_Q16).astype(np.int64)[:, None, None]
            b = np.rint(self.st[cc] * PATCH_STEP_Q16).astype(np.int64)[:, None, None]
            icx, icy = self._cell(self.x[cc], self.y[cc])
            x00, y00 = (63 * (-a - b)) >> 1, (63 * (a - b)) >> 1
            fx = ((x00 + c * a + r * b) >> 16).astype(i32)
            fy = ((y00 + c * b - r * a) >> 16).astype(i32)
            inwin = (fx >= -PATCH_WINDOW_I) & (fx < PATCH_WINDOW_I) & (fy >= -PATCH_WINDOW_I) & (fy < PATCH_WINDOW_I)
            ix = icx[:, None, None] + fx
            iy = (icy[:, None, None] + 1) + fy
            img = (self._lookup(self.drv, ix, iy, False) & inwin).astype(np.uint8)
            img[self.fresh[cc].astype(bool)] = 0              # dreamer/wrappers.py:413
            out[k0:k0 + 256] = img
        return out

    # ------------------------------------------------------------------ outputs
    def _scalar_outputs(self):
        """Per-step results of the step that just ran (terminal values if the env finished)."""
        env_of = np.arange(self.NC) // self.A
        return dict(
            action=self.action.copy(), reward=self.reward.copy(),
            discount=(f32(1.0) - self.done.astype(f32)),
            progress_total=((self.lap - 1).astype(f32) + self.progress),
            time=self.steps[env_of].astype(f32) * DT,
            progress=self.progress.copy(), lap=self.lap.copy(), checkpoint=self.cp.copy(),
            done=self.done.copy(), truncated=self.truncated.copy(), wall_collision=self.wall.copy(),
            opponent_collision=self.opp.copy(), wrong_way=self.wrong_way.copy(),
        )

    def _obs_outputs(self):
        """Observation of the current state (after the auto-reset, if one happened)."""
        n = self.NC
        pose = np.zeros((n, 6), f32)
        vel = np.zeros((n, 6), f32)
        pose[:, 0], pose[:, 1], pose[:, 5] = self.x, self.y, self.theta
        vel[:, 0], vel[:, 5] = self.v, self.omega
        d = dict(lidar=self.lidar.copy(), pose=pose, velocity=vel, speed=np.abs(self.v),
                 acceleration=self.accel.copy(), steering_angle=self.delta.copy(), fresh=self.fresh.copy())
        if self.cfg.render_occupancy:
            d["lidar_occupancy"] = self.patch.copy()
        return d

    def outputs(self):
        out = self._scalar_outputs()
        out.update(self._obs_outputs())
        return out
