"""G10 - the simulator CORE against the reference's own trained agents.

The reference ships the Dreamer policies its ROS node deploys (ros_agent/checkpoints/{austria,treitlstrasse}_dreamer: RSSM +
actor weights, trained in the reference's simulator on 1 080-beam scans with the action space of ReduceActionSpace).  They see
nothing but the scan and emit (motor, steering), so whether they DRIVE here tests what no wrapper-level fixture can: the
handedness of the env (beam order against steering sign), how sharply the car turns, how it accelerates.  The weights are
committed fixtures (tests/golden/make_golden_dreamer_policy.py); the network is restated in oracle/dreamer_policy_port.py.

What these tests pin: a positive steering command turns RIGHT (towards higher beam indices); full lock is
0.19 rad at the front wheels - not the nominal 0.42 -; and with both, the austria agent laps austria at ~3.5 m/s without a
single wall contact, from the grid and from random poses, and carries over to barcelona, which it has never seen.
"""
import os

import numpy as np
import pytest

from helpers import make_oracle
from oracle import c_oracle
from oracle import racecar_oracle as ro
from oracle.dreamer_policy_port import DreamerPolicy
from racing_dreamer_amd.track_assets import load_track

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def weights(name):
    return np.load(os.path.join(GOLDEN, f"dreamer_policy_{name}.npz"))


def drive(env, policy, n, steps, mode=ro.RESET_GRID, mirror=False, repeat=4, seed=1):
    """The reference's control loop (racing_dreamer.py:61-76, dreamer/dream.py:55 action_repeat 4) on any backend with the
    oracle's reset / step interface; `mirror` negates the steering command = the same policy in the env's mirror image."""
    out = env.reset(mode=mode, seed=seed)
    state = policy.initial(n)
    crashes, speeds = 0, []
    for k in range(steps):
        scan = np.asarray(out["lidar"]).reshape(n, ro.N_BEAMS)
        fresh = np.asarray(out["fresh"]).reshape(n) != 0
        action, state = policy.act(scan, state, reset=fresh if k else None)
        sent = action.copy()
        if mirror:
            sent[:, 1] = -sent[:, 1]
        out = env.step(sent, repeat=repeat)
        crashes += int(np.count_nonzero(np.asarray(out["wall_collision"])))
        speeds.append(float(np.asarray(out["speed"]).mean()))
    laps = np.asarray(out["lap"]).reshape(n) + np.asarray(out["progress"]).reshape(n)
    return crashes, float(np.mean(speeds[50:])), laps


def c_env(track_name, n):
    t = load_track(track_name)
    cfg = ro.OracleConfig(num_envs=n, auto_reset=True, remap_actions=True)      # ReduceActionSpace, as the agent was trained
    return c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg, threads=8)


def test_the_fixture_is_the_reference_checkpoint():
    for name in ("austria", "treitlstrasse"):
        w = weights(name)
        assert "ros_agent/checkpoints" in str(w["source"]) and "sha256" in str(w["source"])
        assert w["obs1_w"].shape == (1280, 200) and w["gru_bias"].shape == (2, 600) and w["hout_w"].shape == (400, 4)
        assert all(w[k].dtype == np.float32 for k in ("gru_kernel", "img1_w", "h3_w"))


def test_reference_austria_agent_laps_austria_and_crashes_in_the_mirror_image():
    n = 12
    policy = DreamerPolicy(weights("austria"), sample=True, seed=0)
    crashes, speed, laps = drive(c_env("austria", n), policy, n, 760)
    # 30 s of driving: every car past the finish line (lap 2), none touched a wall
    assert crashes == 0 and speed > 3.0 and laps.min() >= 2.0, (crashes, speed, laps)
    # the mirror image (steering sign flipped = beam order flipped): a wall within a second or two, again and again
    policy = DreamerPolicy(weights("austria"), sample=True, seed=0)
    crashes_m, speed_m, laps_m = drive(c_env("austria", n), policy, n, 250, mirror=True)
    assert crashes_m >= 50 and laps_m.max() < 1.2, (crashes_m, laps_m)


def test_reference_austria_agent_from_random_poses_and_on_a_track_it_never_saw():
    n = 16
    for track, mode, sample in (("austria", ro.RESET_RANDOM, True), ("austria", ro.RESET_GRID, False), ("barcelona", ro.RESET_GRID, True)):
        policy = DreamerPolicy(weights("austria"), sample=sample, seed=3)
        crashes, speed, laps = drive(c_env(track, n), policy, n, 400, mode=mode)
        assert crashes == 0 and speed > 3.0, (track, mode, sample, crashes, speed)


def test_reference_treitlstrasse_agent_drives_its_track():
    n = 12
    policy = DreamerPolicy(weights("treitlstrasse"), sample=True, seed=0)
    crashes, speed, laps = drive(c_env("treitlstrasse_v2", n), policy, n, 700)
    # (trained for transfer to the real car: slow and careful; 3 contacts in 32 000 agent steps when measured at length)
    assert crashes <= 1 and laps.mean() > 1.4, (crashes, speed, laps)
    policy = DreamerPolicy(weights("treitlstrasse"), sample=True, seed=0)
    crashes_m, _, laps_m = drive(c_env("treitlstrasse_v2", n), policy, n, 250, mirror=True)
    assert crashes_m >= 50 and laps_m.max() < 1.2, (crashes_m, laps_m)


@pytest.mark.parametrize("wheel_max", [0.10, 0.42])
def test_the_agent_refutes_other_steering_locks(wheel_max, monkeypatch):
    """The calibration is not slack: with full lock at 0.10 rad the austria agent understeers into a wall within 300 agent
    steps, with the nominal 0.42 rad it turns into the inner one (NumPy oracle, constants patched)."""
    monkeypatch.setattr(ro, "STEER_GAIN", np.float32(-wheel_max))
    n = 4
    env = make_oracle(load_track("austria"), num_envs=n, auto_reset=True, remap_actions=True)
    policy = DreamerPolicy(weights("austria"), sample=True, seed=0)
    crashes, _, laps = drive(env, policy, n, 330)
    assert crashes >= n, (wheel_max, crashes, laps)


def test_reference_reward_model_follows_this_envs_reward():
    """The reference's REWARD head (reward.pkl: trained on the reference simulator's rewards) evaluated on the agent's latent
    state while it laps here: its prediction rises and falls with this env's progress reward (correlation > 0.7 over 7 000
    agent steps).  Its scale does not match - 0.25 per agent step predicted where the env pays 0.15 at 3.5 m/s, a factor the
    build cannot explain - so only the proportionality is asserted."""
    n = 16
    env = c_env("austria", n)
    policy = DreamerPolicy(weights("austria"), sample=True, seed=0)
    out = env.reset(mode=ro.RESET_GRID, seed=1)
    state = policy.initial(n)
    predicted, paid = [], []
    for k in range(480):
        action, state = policy.act(np.asarray(out["lidar"]).reshape(n, ro.N_BEAMS), state)
        if k >= 40:
            predicted.append(policy.predicted_reward(state))
            paid.append(np.asarray(out["reward"]).reshape(n).copy())
        out = env.step(action, repeat=4)
    predicted, paid = np.concatenate(predicted), np.concatenate(paid)
    corr = float(np.corrcoef(predicted, paid)[0, 1])
    assert corr > 0.7 and predicted.mean() > 0.1 and paid.mean() > 0.1, (corr, predicted.mean(), paid.mean())


def _image_correlation(a, b):
    a, b = a.reshape(len(a), -1), b.reshape(len(b), -1)
    a, b = a - a.mean(1, keepdims=True), b - b.mean(1, keepdims=True)
    return (a * b).sum(1) / np.sqrt((a * a).sum(1) * (b * b).sum(1) + 1e-9)


def test_reference_occupancy_decoder_reconstructs_this_envs_patches_the_right_way_round():
    """G11.  treitlstrasse_dreamer_20210224 was trained to reconstruct the `lidar_occupancy` image from the latent state
    (LidarOccupancyDecoder), i.e. on (scan -> image) pairs of the REFERENCE simulator.  Fed with this env's scans while its own
    actor drives, what it reconstructs matches this env's render of the same pose as it stands - not mirrored, not rotated:
    the render's registration to the scan (which side of the image the low beam indices are on) is the reference's."""
    n = 16
    t = load_track("treitlstrasse_v2")
    cfg = ro.OracleConfig(num_envs=n, auto_reset=True, remap_actions=True, render_occupancy=True)
    env = c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg, threads=8)
    policy = DreamerPolicy(weights("treitlstrasse_occupancy"), sample=False)
    assert policy.normalized
    out = env.reset(mode=ro.RESET_RANDOM, seed=1)
    state = policy.initial(n)
    scores = {k: [] for k in ("identity", "mirrored", "upside_down", "rotated")}
    crashes = 0
    for k in range(260):
        action, state = policy.act(np.asarray(out["lidar"]).reshape(n, ro.N_BEAMS), state,
                                   reset=(np.asarray(out["fresh"]).reshape(n) != 0) if k else None)
        if k >= 20 and k % 4 == 0:
            seen = policy.decoded_occupancy(state)
            mine = np.asarray(out["lidar_occupancy"]).reshape(n, 64, 64).astype(np.float32)
            scores["identity"].append(_image_correlation(seen, mine))
            scores["mirrored"].append(_image_correlation(seen, mine[:, :, ::-1]))
            scores["upside_down"].append(_image_correlation(seen, mine[:, ::-1, :]))
            scores["rotated"].append(_image_correlation(seen, np.rot90(mine, 1, (1, 2))))
        out = env.step(action, repeat=4)
        crashes += int(np.count_nonzero(np.asarray(out["wall_collision"])))
    s = {k: np.concatenate(v) for k, v in scores.items()}
    assert crashes <= n, crashes                                       # (its own actor drives: 1 m/s, ~ 10 contacts in 4 160 agent steps from random poses)
    assert s["identity"].mean() > 0.5 and s["identity"].mean() > s["mirrored"].mean() + 0.08, {k: v.mean() for k, v in s.items()}
    assert (s["identity"] > s["mirrored"]).mean() > 0.7                # frame by frame
    assert s["upside_down"].mean() < 0.4 and abs(s["rotated"].mean()) < 0.15
