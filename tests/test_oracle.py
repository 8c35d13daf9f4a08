"""The CPU oracle itself: published known answers, hand-derived known answers, properties, and the
C restatement pinned bit-for-bit to the NumPy one."""
import os
import numpy as np
import pytest

from helpers import make_oracle
from oracle import c_oracle
from oracle import racecar_oracle as ro
from racing_dreamer_amd.track_assets import load_track, synthetic_track


def test_philox4x32_10_random123_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = tuple(int(v) for v in ro.philox4x32(*ctr, *key))
        assert got == want


def test_sincos_and_exp_accuracy():
    a = np.linspace(-6.6, 6.6, 400001).astype(np.float32)
    s, c = ro.sincos32(a)
    assert np.abs(s - np.sin(a.astype(np.float64))).max() < 2e-7
    assert np.abs(c - np.cos(a.astype(np.float64))).max() < 2e-7
    x = np.linspace(-8, 3, 20001).astype(np.float32)
    assert np.max(np.abs(ro.exp32(x) / np.exp(x.astype(np.float64)) - 1)) < 2e-7


def _box_env(n=1, size=200):
    """Empty square room: walls only on the border, car anywhere."""
    occ = np.zeros((size, size), bool)
    occ[:3, :] = occ[-3:, :] = occ[:, :3] = occ[:, -3:] = True
    drv = ~occ
    prog = np.where(drv, 0.5, -1.0).astype(np.float32)
    cl = np.array([[5.0, 5.0, 0.0, 0.5]], np.float32)
    return ro.OracleRaceEnv(occ, drv, prog, cl, (0.0, 0.0), 0.05, ro.OracleConfig(num_envs=n))


def test_raycast_known_answer_axis_aligned_walls():
    env = _box_env()
    env.reset()
    # car at (5, 5) heading +x: sensor at x = 5.25.  Walls: cells 0..2 and 197..199 -> x in [9.85, 10).
    lid = env.lidar[0]
    beam_fwd = (ro.N_BEAMS - 1) / 2            # 539.5: no beam is exactly forward; use the side beams
    i_left = int(round((135 - 90) / (270 / 1079)))      # beam at +90 deg
    ang = np.deg2rad(135 - i_left * 270 / 1079)
    want_left = (9.85 - 5.0) / np.sin(ang)               # wall face at y = 9.85
    assert abs(lid[i_left] - want_left) < 1e-4
    i_right = ro.N_BEAMS - 1 - i_left
    assert abs(lid[i_right] - want_left) < 1e-4          # symmetric room -> symmetric scan
    assert np.allclose(lid, lid[::-1], atol=1e-4)
    # the most forward beams see the front wall at x = 9.85
    k = 539
    ang = np.deg2rad(135 - k * 270 / 1079)
    assert abs(lid[k] - (9.85 - 5.25) / np.cos(ang)) < 1e-4
    assert lid.min() > 0 and lid.max() <= 15.0


def test_raycast_max_range_and_start_inside_wall():
    occ = np.zeros((700, 700), bool)
    occ[:3, :] = occ[-3:, :] = occ[:, :3] = occ[:, -3:] = True
    env = ro.OracleRaceEnv(occ, ~occ, np.where(~occ, 0.5, -1).astype(np.float32),
                           np.array([[17.5, 17.5, 0.0, 0.5], [0.05, 17.5, 0.0, 0.5]], np.float32),
                           (0.0, 0.0), 0.05, ro.OracleConfig(num_envs=1))
    env.reset()
    assert np.all(env.lidar[0] == np.float32(15.0))      # nearest wall 17.35 m away: no return
    env.x[:], env.y[:] = -0.2, 17.5                      # sensor at x = 0.05: inside the wall cells
    assert np.all(env.raycast()[0] == 0.0)


def test_bicycle_known_answer_circle_arc():
    """Constant speed and steering -> circle of radius L / tan(delta) (explicit Euler, dt = 0.01); a NEGATIVE command
    turns the wheels to the left (counter-clockwise, delta > 0): delta = command x STEER_GAIN."""
    env = _box_env(size=600)
    env.centerline[0, :2] = 15.0
    env.reset()
    env.v[:] = 2.0
    env.delta[:] = 0.15
    act = np.array([[0.4, -0.15 / 0.19]], np.float32)    # throttle 0.4 holds 2 m/s, steering holds delta
    env.v[:] = 2.0
    r = 0.3302 / np.tan(0.15)
    x0, y0, th0 = float(env.x[0]), float(env.y[0]), float(env.theta[0])
    n = 200
    for _ in range(n):
        env.v[:] = 2.0                                    # hold speed (motor = 0 would brake)
        env.step(act)
    th = th0 + 2.0 / r * 0.01 * n
    cx, cy = x0 - r * np.sin(th0), y0 + r * np.cos(th0)
    want = (cx + r * np.sin(th), cy - r * np.cos(th))
    assert abs(float(env.x[0]) - want[0]) < 0.02 and abs(float(env.y[0]) - want[1]) < 0.02
    assert abs(((float(env.theta[0]) - th + np.pi) % (2 * np.pi)) - np.pi) < 1e-3


def test_longitudinal_model_limits():
    """dv/dt = |m| * 4 - 0.8 v (throttle) resp. -|m| * 4 - 0.8 v (brake): throttle m settles at 5 m m/s."""
    env = _box_env(size=2000)
    env.centerline[0, :2] = (5.0, 50.0)
    env.raycast = lambda *a, **k: np.zeros((env.NC, ro.N_BEAMS), np.float32)      # (1 350 steps: the scans are not what is tested)
    env.reset()
    out = None
    for k in range(200):
        out = env.step(np.array([[1.0, 0.0]], np.float32))
    assert abs(out["speed"][0] - 5.0 * (1 - np.exp(-0.8 * 2.0))) < 0.02        # 2 s of full throttle
    assert abs(out["pose"][0, 0] - (5.0 + 5.0 * (2.0 - (1 - np.exp(-1.6)) / 0.8))) < 0.06
    for k in range(1000):
        out = env.step(np.array([[0.5, 0.0]], np.float32))
    assert abs(out["speed"][0] - 2.5) < 0.02 and out["speed"][0] <= 5.0             # half throttle -> 2.5 m/s
    for k in range(150):
        out = env.step(np.array([[-1.0, 0.0]], np.float32))
    assert out["speed"][0] == 0.0                          # negative motor = brake to standstill, never reverse


def test_progress_lap_and_reward_over_a_scripted_lap():
    t = synthetic_track()
    env = make_oracle(t, num_envs=1, laps=1)
    p0 = float(env.reset()["progress_total"][0])
    total, laps_seen, prog = 0.0, [], []
    for k in range(len(t.centerline) + 5):                # teleport along the centerline, one bin per step
        i = (k + ro.GRID_LEAD_BINS + 1) % len(t.centerline)
        env.x[:], env.y[:], env.theta[:] = t.centerline[i, 0], t.centerline[i, 1], t.centerline[i, 2]
        env.st[:], env.ct[:] = ro.sincos32(env.theta)
        env.v[:] = 0.0
        out = env.step(np.zeros((1, 2), np.float32))
        total += float(out["reward"][0])
        laps_seen.append(int(out["lap"][0]))
        prog.append(float(out["progress_total"][0]))
        if out["done"][0]:
            break
    assert laps_seen[0] == 1 and laps_seen[-1] == 2 and out["done"][0] == 1      # lap > laps ends the episode
    assert not out["wrong_way"][0]
    assert np.all(np.diff(prog) > -1e-6)                                          # monotone through the line
    assert abs(total - 100.0 * (prog[-1] - p0)) < 0.2          # reward = 100 * delta progress


def test_wrong_way_flag():
    t = synthetic_track()
    env = make_oracle(t, num_envs=1)
    env.reset()
    n = len(t.centerline)
    for k in range(1, n // 3):
        i = (-k * 3) % n
        env.x[:], env.y[:] = t.centerline[i, 0], t.centerline[i, 1]
        out = env.step(np.zeros((1, 2), np.float32))
    assert out["wrong_way"][0] == 1 and out["lap"][0] <= 1


def test_collision_terminates_with_collision_reward():
    t = synthetic_track()
    env = make_oracle(t, num_envs=1)
    env.reset()
    out = None
    for _ in range(400):
        out = env.step(np.array([[1.0, 1.0]], np.float32))    # full throttle, full left lock
        if out["done"][0]:
            break
    assert out["done"][0] == 1 and out["wall_collision"][0] == 1 and out["discount"][0] == 0.0
    assert out["reward"][0] < -0.9                             # collision_reward -1 dominates one step's progress
    frozen = env.step(np.array([[1.0, 0.0]], np.float32))      # finished env without auto-reset is frozen
    assert frozen["reward"][0] == 0.0 and frozen["done"][0] == 1
    assert np.array_equal(frozen["pose"], out["pose"])


def test_step_before_reset_raises():
    env = make_oracle(synthetic_track(), num_envs=2)
    with pytest.raises(AssertionError, match="Must reset environment."):
        env.step(np.zeros((2, 2), np.float32))


@pytest.mark.parametrize("cars,occ,mode", [(1, True, ro.RESET_RANDOM), (2, False, ro.RESET_RANDOM_BALL),
                                           (4, False, ro.RESET_GRID)])
def test_c_oracle_is_bit_identical_to_numpy_oracle(cars, occ, mode):
    t = load_track("treitlstrasse_v2")
    cfg = ro.OracleConfig(num_envs=40, cars_per_env=cars, auto_reset=True, render_occupancy=occ,
                          time_limit_steps=17, remap_actions=(cars == 1))
    a = ro.OracleRaceEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg)
    b = c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg, threads=3)
    oa, ob = a.reset(mode=mode, seed=5), b.reset(mode=mode, seed=5)
    dones = 0
    for k in range(30):
        act = ro.random_actions(3, k, 40 * cars)
        assert np.array_equal(act, b.random_actions(3, k))
        oa, ob = a.step(act, repeat=3), b.step(act, repeat=3)
        for key in oa:
            assert np.array_equal(oa[key], ob[key]), (key, k)
        dones += int(oa["done"].sum())
    assert dones > 0


def test_c_oracle_max_speed_task_matches():
    t = load_track("columbia")
    cfg = ro.OracleConfig(num_envs=16, task=ro.TASK_MAX_SPEED)
    a = ro.OracleRaceEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg)
    b = c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg)
    a.reset(mode=1, seed=2), b.reset(mode=1, seed=2)
    for k in range(10):
        act = ro.random_actions(9, k, 16)
        oa, ob = a.step(act, repeat=2), b.step(act, repeat=2)
        assert np.array_equal(oa["reward"], ob["reward"]) and np.all(oa["done"] == 0)


def test_n_step_progress_task_of_the_secondary_agents():
    """baselines/scenarios/max_progress/columbia.yml:17-18: agents B-D run `n_step_progress {n_steps: 10}`.  Spec:
    reward = 100 x total progress gained over the last n sub-steps (since the reset while younger), never done, no
    collision term; car A keeps maximize_progress.  Known answer from the recorded progress, and C == NumPy."""
    t = load_track("columbia")
    cfg = ro.OracleConfig(num_envs=6, cars_per_env=3, car_tasks=[-1, ro.TASK_N_STEP_PROGRESS, ro.TASK_N_STEP_PROGRESS],
                          n_steps=10, terminate_on_collision=False)
    a = ro.OracleRaceEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg)
    b = c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg)
    oa, ob = a.reset(mode=0), b.reset(mode=0)
    totals = [oa["progress_total"].astype(np.float32).copy()]
    for k in range(25):
        act = np.tile(np.array([[0.7, 0.02 * (k % 3)]], np.float32), (18, 1))
        oa, ob = a.step(act), b.step(act)
        for name in ("reward", "done", "progress_total", "pose"):
            assert np.array_equal(oa[name], ob[name]), (name, k)
        totals.append(oa["progress_total"].copy())
        back = totals[max(k + 1 - 10, 0)]
        want = (totals[-1] - back) * np.float32(100.0)
        sec = np.arange(18) % 3 != 0
        assert np.array_equal(oa["reward"][sec], want[sec]) and not oa["done"][sec].any()
        first = oa["reward"][~sec]
        assert np.allclose(first, (totals[-1] - totals[-2])[~sec] * 100.0, atol=1e-4)
    assert (oa["reward"][sec] >= 0).all() and (oa["reward"][sec] > 0).any()              # the cars did move


def test_two_cars_see_and_hit_each_other():
    t = synthetic_track()
    env = make_oracle(t, num_envs=1, cars_per_env=2)
    env.reset(mode=ro.RESET_GRID)
    # car 1 starts 1.2 m behind car 0 on the straight: its forward beams return car 0's rear (< wall distance)
    solo = make_oracle(t, num_envs=1, cars_per_env=1)
    solo.reset(mode=ro.RESET_GRID)
    solo.x[:], solo.y[:], solo.theta[:] = env.x[1], env.y[1], env.theta[1]
    solo.st[:], solo.ct[:] = ro.sincos32(solo.theta)
    alone = solo.raycast()[0]
    both = env.raycast()[1]
    assert np.all(both <= alone) and (both < alone - 0.5).sum() > 10
    gap = float(np.hypot(env.x[0] - env.x[1], env.y[0] - env.y[1]))
    assert abs(both[538:542].min() - (gap - 0.25 - 0.10)) < 0.06       # sensor 0.25 ahead, rear overhang 0.10
    # drive car 1 into car 0
    hit = False
    for _ in range(300):
        out = env.step(np.array([[-1.0, 0.0], [1.0, 0.0]], np.float32))
        if out["opponent_collision"].any():
            hit = True
            break
    assert hit and out["opponent_collision"].tolist() == [1, 1] and out["done"].all()


def test_auto_reset_uses_fresh_philox_counter_per_episode():
    t = load_track("austria")
    env = make_oracle(t, num_envs=8, auto_reset=True)
    env.reset(mode=ro.RESET_RANDOM, seed=1)
    starts = [env.x.copy()]
    for k in range(3):
        env.done[:] = 1                       # force every env to finish
        env._reset_envs(np.arange(8))
        starts.append(env.x.copy())
    assert all(not np.array_equal(starts[0], s) for s in starts[1:])
    assert np.array_equal(env.episode, np.full(8, 4, np.uint32))


def test_sharding_independence_of_rng_streams():
    t = load_track("austria")
    full = make_oracle(t, num_envs=12, auto_reset=True)
    part = make_oracle(t, num_envs=4, auto_reset=True, first_env=8)
    a, b = full.reset(mode=1, seed=9), part.reset(mode=1, seed=9)
    assert np.array_equal(a["pose"][8:], b["pose"])
    assert np.array_equal(ro.random_actions(5, 3, 12)[8:], ro.random_actions(5, 3, 4, first_car=8))


def test_scan_equals_brute_force_ray_box_intersection():
    """A second opinion on the LiDAR semantics that shares no code and no algorithm with the oracle's grid traversal:
    every ray against EVERY stop cell as an axis-aligned box (slab method, float64), nearest entry wins; a ring cell in
    front of the nearest wall means no return.  Real track, arbitrary poses."""
    t = load_track("columbia")
    rng = np.random.default_rng(5)
    n = 24
    idx = rng.integers(0, len(t.centerline), n)
    poses = t.centerline[idx, :3].astype(np.float64)
    poses[:, :2] += rng.uniform(-0.3, 0.3, (n, 2))
    poses[:, 2] = rng.uniform(-np.pi, np.pi, n)
    env = ro.OracleRaceEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, ro.OracleConfig(num_envs=n))
    env.reset()
    env.x[:], env.y[:], env.theta[:] = poses[:, 0], poses[:, 1], poses[:, 2]
    env.st[:], env.ct[:] = ro.sincos32(env.theta)
    got = env.raycast()                                              # [n, 1080] float32
    beams = np.arange(0, 1080, 9)
    cb, sb = ro.beam_table()
    ct, st = env.ct.astype(np.float64)[:, None], env.st.astype(np.float64)[:, None]
    dx = ct * cb[beams][None, :] - st * sb[beams][None, :]           # [n, nb]
    dy = st * cb[beams][None, :] + ct * sb[beams][None, :]
    res = t.resolution
    gx = ((env.x.astype(np.float64) + ro.LIDAR_X * ct[:, 0]) - t.origin[0]) / res
    gy = ((env.y.astype(np.float64) + ro.LIDAR_X * st[:, 0]) - t.origin[1]) / res
    stop = env.occ                                                   # walls + sentinel ring
    cy, cx = np.nonzero(stop)
    is_ring = env.ring[cy, cx]
    worst = 0.0
    for k in range(n):
        ox, oy = gx[k], gy[k]
        for b in range(len(beams)):
            ddx, ddy = dx[k, b], dy[k, b]
            with np.errstate(divide="ignore", invalid="ignore"):
                tx1, tx2 = (cx - ox) / ddx, (cx + 1 - ox) / ddx
                ty1, ty2 = (cy - oy) / ddy, (cy + 1 - oy) / ddy
            tn = np.maximum(np.minimum(tx1, tx2), np.minimum(ty1, ty2))
            tf = np.minimum(np.maximum(tx1, tx2), np.maximum(ty1, ty2))
            hit = (tn <= tf) & (tf >= 0)
            tn = np.where(hit, np.maximum(tn, 0.0), np.inf)
            j = int(np.argmin(tn))
            want = 15.0 if (not np.isfinite(tn[j]) or is_ring[j] or tn[j] * res >= 15.0) else tn[j] * res
            have = float(got[k, beams[b]])
            if abs(have - want) > 1e-3:
                # the only legitimate difference: a ray that grazes a cell corner within rounding (the traversal's
                # tie rule decides there) - then the two candidates' entry distances are a hair apart
                second = np.partition(tn, 1)[1]
                assert abs(second - tn[j]) * res < 2e-3 or abs(have - second * res) < 1e-3, (k, beams[b], have, want)
            else:
                worst = max(worst, abs(have - want))
    # fp32 traversal against float64 geometry: the sensor position carries ~1.5e-6 m of fp32 rounding, which a ray at a
    # shallow angle to the wall face it enters through divides by that angle's sine
    assert worst < 2e-4


@pytest.mark.parametrize("track_name,cars", [("columbia", 1), ("austria", 1), ("treitlstrasse_v2", 2), ("barcelona", 4), ("columbia_slam", 1),
                                             ("columbia_slam", 3), ("columbia", 3)])
def test_random_starts_follow_the_law_of_survey_h6(track_name, cars):
    """H6 (dreamer/dream.py:105-108): `random` = a pose on the track with a minimum wall distance, heading along the
    track; `random_ball` = the cars of an env close together around one random point.  Checked on the C port (bit-identical
    to the NumPy statement, test_c_oracle_is_bit_identical...): every start lies within the lateral room of its centre-line
    bin and within +- HEADING_JITTER of its direction, no start touches a wall, the starts of 20 000 envs are all different,
    the bins are hit uniformly, a second episode draws afresh, and `grid` is unchanged (the centre line itself)."""
    from oracle import c_oracle
    from racing_dreamer_amd.track_assets import load_track
    t = load_track(track_name)
    n = 20000
    cfg = ro.OracleConfig(num_envs=n, cars_per_env=cars, auto_reset=True)
    env = c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg, threads=8)
    ref = ro.OracleRaceEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, ro.OracleConfig(num_envs=4, cars_per_env=cars))
    width = ref.spawn_width()
    assert np.array_equal(width, env._keep["spawn_w"]) and width.max() > 0.3 and width.min() >= 0.0
    mode = ro.RESET_RANDOM if cars == 1 else ro.RESET_RANDOM_BALL
    out = env.reset(mode=mode, seed=5)
    pose = np.asarray(out["pose"]).reshape(n, cars, 6)
    xy, yaw = pose[..., :2].astype(np.float64), pose[..., 5].astype(np.float64)
    cl = t.centerline.astype(np.float64)
    g = (np.arange(n) + cfg.first_env).astype(np.uint32)
    r0 = ro.philox4x32(g, np.zeros(n, np.uint32), np.uint32(0), np.uint32(0), 5, 0)[0]
    idx0 = ((r0.astype(np.uint64) * np.uint64(len(cl))) >> np.uint64(32)).astype(np.int64)
    safe = ref.spawn_safe()
    assert np.array_equal(safe, env._keep["spawn_safe"])
    drawn = idx0
    if cars > 1:                            # several cars are anchored at the first bin from the drawn one on whose poses do not overlap
        idx0 = safe[idx0]
        assert np.mean(idx0 != drawn) < 0.1
    for a in range(cars):
        idx = (idx0 - a * ro.BALL_GAP_BINS) % len(cl)
        off = xy[:, a] - cl[idx, :2]
        lateral = -off[:, 0] * np.sin(cl[idx, 2]) + off[:, 1] * np.cos(cl[idx, 2])
        along = off[:, 0] * np.cos(cl[idx, 2]) + off[:, 1] * np.sin(cl[idx, 2])
        assert np.all(np.abs(lateral) <= width[idx] + 1e-5) and np.all(np.abs(along) < 1e-5), a
        dyaw = (yaw[:, a] - cl[idx, 2] + np.pi) % (2 * np.pi) - np.pi
        assert np.all(np.abs(dyaw) <= float(ro.HEADING_JITTER) + 1e-5), a
        moved = width[idx] > 0.2
        if a == 0 or track_name != "columbia_slam":     # (the raw columbia map: the cars behind the first stand where the room is small)
            assert np.std(lateral[moved] / width[idx][moved]) > 0.5 and np.std(dyaw) > 0.15      # uniform: sigma = 0.577 / 0.2
    hist = np.bincount(drawn * 40 // len(cl), minlength=40)           # the lap in 40 stretches: 500 starts each
    assert hist.min() > 0.8 * n / 40 and hist.max() < 1.2 * n / 40
    distinct = len(np.unique(pose.reshape(n, -1), axis=0))
    # (the raw columbia map with several cars: next to the fold sound bins lie close on the ground, proposals clash there and the
    # env takes the centre-line poses - the law's fallback -, which envs share)
    assert distinct == n or (cars > 1 and distinct > 0.99 * n)
    # nothing touches anything at the start: one step with the brakes on leaves every car where it is and evaluates the contacts
    act = np.zeros((n * cars, 2), np.float32)
    act[:, 0] = -1.0
    out = env.step(act)
    wall, opp = np.asarray(out["wall_collision"]).reshape(n, cars), np.asarray(out["opponent_collision"]).reshape(n, cars)
    assert int(wall.sum()) == 0 and int(opp.sum()) == 0      # (columbia_slam's folded bins included since round 5: spawn_safe)
    # a finished env draws a NEW pose (episode counter in the Philox counter)
    env2 = c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, ro.OracleConfig(num_envs=64, cars_per_env=cars))
    a0 = np.asarray(env2.reset(mode=mode, seed=5)["pose"]).copy()
    a1 = np.asarray(env2.reset(mode=mode, seed=5)["pose"]).copy()
    assert np.array_equal(a0.reshape(64, cars, 6), pose[:64]) and not np.array_equal(a0, a1)
    grid = np.asarray(env2.reset(mode=ro.RESET_GRID, seed=5)["pose"]).reshape(64, cars, 6)
    for a in range(cars):
        i = (ro.BALL_GAP_BINS * (cars - 1) + ro.GRID_LEAD_BINS - a * ro.BALL_GAP_BINS) % len(cl)
        assert np.array_equal(grid[:, a, :2], np.broadcast_to(t.centerline[i, :2], (64, 2))) and np.all(grid[:, a, 5] == t.centerline[i, 2])


def test_no_multi_car_start_overlaps_on_any_compiled_map():
    """20 000 `random` (A = 1) / `random_ball` (A = 2, 3, 4) starts per track on EVERY compiled map - no two
    cars of an env overlap, none touches a wall (the spec's own tests, `_obb_overlap` and `_wall_hit`, on the poses the reset law produced;
    no scan is run: 60 000 x 1080 rays per case would make this a test of minutes).  columbia_slam's (the raw columbia.pgm's) last four centre-line bins run back
    along the four before them (the BFS wavefronts of its progress grid fold at the finish line):
    `spawn_safe` moves a start drawn there to the next bin whose four poses are clear, and leaves every bin of an ordinary
    track where it is."""
    import json
    from racing_dreamer_amd.track_assets import TRACK_DIR
    with open(os.path.join(TRACK_DIR, "index.json")) as f:
        names = sorted(k for k, v in json.load(f).items() if v["status"] == "ok")
    assert len(names) >= 29
    n, moved, unusable, tight = 20000, {}, {}, {}
    for name in names:
        t = load_track(name)
        for cars in (1, 2, 3, 4):
            env = ro.OracleRaceEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution,
                                   ro.OracleConfig(num_envs=n, cars_per_env=cars, auto_reset=True))
            env.seed, env.mode = 11, (ro.RESET_RANDOM if cars == 1 else ro.RESET_RANDOM_BALL)
            env._reset_envs(np.arange(n))
            e = np.arange(n)
            for a in range(cars):
                # NO start touches a wall, on ANY compiled map (round 5 exempted nine hand-drawn maps here)
                assert int(env._wall_hit(e * cars + a).sum()) == 0, (name, cars, a)
                for b in range(a + 1, cars):
                    assert int(env._obb_overlap(e * cars + a, e * cars + b).sum()) == 0, (name, cars, a, b)
        safe = env.spawn_safe()
        moved[name] = int((safe != np.arange(len(safe))).sum())
        unusable[name] = int((~env.spawn_usable()).sum())
        tight[name] = int((env.spawn_heading_room() < ro.HEADING_JITTER).sum())
        # a redirected row is a usable bin's; the grid start (A = 1 .. 4) touches nothing and no two of its cars overlap, on every map
        env.spawn_rows()
        assert env.spawn_usable()[env._spawn_u].all(), name
        for cars in (1, 2, 3, 4):
            g = ro.OracleRaceEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, ro.OracleConfig(num_envs=1, cars_per_env=cars))
            g.seed, g.mode = 0, ro.RESET_GRID
            g._reset_envs(np.arange(1))
            assert not any(int(g._wall_hit(np.array([a]))[0]) for a in range(cars)), (name, cars)
            assert not any(int(g._obb_overlap(np.array([a]), np.array([b]))[0]) for a in range(cars) for b in range(a + 1, cars)), (name, cars)
    # the scenario tracks of the reference are untouched by the law for narrow places: every bin usable, (nearly) every bin with
    # the full heading jitter; the hand-drawn maps with boxes on the line are where bins are skipped and headings held
    for name in ("austria", "barcelona", "columbia", "treitlstrasse_v2", "gbr"):
        assert unusable[name] == 0 and tight[name] <= 1, (name, unusable[name], tight[name])
    assert unusable["levinelobby"] >= 20 and unusable["torino"] >= 5 and tight["torino_redraw_small_with_obstacles"] >= 10, (unusable, tight)
    # (columbia: the smoothing of the centre line across the finish-line seam bunches four bins there: 12 anchors move by <= 4 bins)
    assert moved["columbia_slam"] >= 4 and moved["columbia"] <= 16 and moved["austria"] == 0 and moved["barcelona"] == 0, moved
    # the C port builds the same table (its resets are compared with these bit for bit elsewhere)
    from oracle import c_oracle
    t = load_track("columbia_slam")
    cenv = c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, ro.OracleConfig(num_envs=2, cars_per_env=2))
    ref = ro.OracleRaceEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, ro.OracleConfig(num_envs=2, cars_per_env=2))
    assert np.array_equal(cenv._keep["spawn_safe"], ref.spawn_safe())
