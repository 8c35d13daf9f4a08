"""oracle/wrappers_port.py (and the oracle's fused semantics) against golden vectors captured from the
reference's own wrapper code (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

from oracle import racecar_oracle as ro
from oracle import wrappers_port as wp

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "wrappers_golden.npz"))


def test_g1_reduce_action_space():
    out = np.stack([wp.reduce_action(a) for a in G["g1_in"]])
    assert np.array_equal(out, G["g1_out"])                       # float64, exact
    # the fused fp32 remap of the env spec agrees to fp32 rounding
    a = G["g1_in"]
    lo, hi = np.asarray([0.005, -1.0], np.float32), np.asarray([1.0, 1.0], np.float32)
    fused = ((a + np.float32(1.0)) * np.float32(0.5)) * (hi - lo) + lo
    assert np.abs(fused - G["g1_out"]).max() <= 2e-7


class _Scripted:
    def __init__(self, ids, rew, don):
        self.ids, self.rew, self.don, self.t = ids, rew, don, 0

    def step(self, action):
        k = self.t
        self.t += 1
        r = {a: float(self.rew[k][i]) for i, a in enumerate(self.ids)}
        d = {a: bool(self.don[k][i]) for i, a in enumerate(self.ids)}
        return {a: {"lidar": np.full(1, float(self.t))} for a in self.ids}, r, d, {}


def test_g2_action_repeat_dreamer():
    for (amount, done_at), row, dones, calls, last in zip(G["g2_cases"], G["g2_rew"], G["g2_done"], G["g2_calls"],
                                                          G["g2_last_lidar"]):
        rew = row[:32].reshape(16, 2)
        don = np.zeros((16, 2), bool)
        if done_at >= 0:
            don[done_at, done_at % 2] = True
        env = _Scripted(["A", "B"], rew, don)
        obs, tot, d, info, n = wp.action_repeat_dreamer(env.step, ["A", "B"], None, int(amount))
        assert n == calls
        assert [tot["A"], tot["B"]] == list(row[32:])
        assert [d["A"], d["B"]] == list(dones)
        assert obs["A"]["lidar"][0] == last


def test_g2_action_repeat_baselines():
    for (n, done_at), row in zip(G["g2b_cases"], G["g2b"]):
        rew = row[:16]
        don = np.zeros(16, bool)
        if done_at >= 0:
            don[done_at] = True
        state = {"t": 0}

        def step(action):
            k = state["t"]
            state["t"] += 1
            return {"lidar": np.full(4, float(state["t"]))}, float(rew[k]), bool(don[k]), {}
        obs, tot, done, info, calls = wp.action_repeat_baselines(step, None, int(n))
        assert [tot, float(done), calls, obs["lidar"][0]] == list(row[16:])


def test_g2_single_agent_variants_agree_when_done_is_terminal():
    """For one agent both reference variants stop at the first done and sum the same rewards -
    except that the baselines variant ignores a done raised by its unconditional FIRST step.  The env
    spec folds the Dreamer rule into the kernel."""
    rng = np.random.default_rng(0)
    for done_at in (None, 1, 2, 3):
        rew = rng.uniform(-1, 1, (8, 1))
        don = np.zeros((8, 1), bool)
        if done_at is not None:
            don[done_at, 0] = True
        a = _Scripted(["A"], rew, don)
        _, tot_d, d_d, _, n_d = wp.action_repeat_dreamer(a.step, ["A"], None, 4)
        st = {"t": 0}

        def step(action):
            k = st["t"]
            st["t"] += 1
            return None, float(rew[k, 0]), bool(don[k, 0]), {}
        _, tot_b, d_b, _, n_b = wp.action_repeat_baselines(step, None, 4)
        assert (n_d, tot_d["A"], d_d["A"]) == (n_b, tot_b, d_b)


def test_g3_time_limit():
    tl = wp.TimeLimit(7)
    with pytest.raises(AssertionError) as e:
        tl.step({"A": False})
    assert str(e.value) == str(G["g3_assert_before_reset"]) == "Must reset environment."
    tl.reset()
    got = [tl.step({"A": False})["A"] for _ in range(7)]
    assert got == list(G["g3_dones"])
    with pytest.raises(AssertionError) as e:
        tl.step({"A": False})
    assert str(e.value) == str(G["g3_assert_after_limit"])


def test_g4_speed_and_action_space():
    got = np.array([wp.speed(v) for v in G["g4_vel"]])
    assert np.array_equal(got, G["g4_speed"])
    lo, hi = wp.flat_action_bounds(np.array([-1.0], np.float32), np.array([1.0], np.float32),
                                   np.array([-1.0], np.float32), np.array([1.0], np.float32))
    assert np.array_equal(lo, G["g4_low"]) and np.array_equal(hi, G["g4_high"])
    assert float(G["g4_reset_speed"]) == 0.0
    # flat action [motor, steering] (dreamer/wrappers.py:63)
    assert np.isclose(float(G["g4_sent_motor"]), 0.3) and np.isclose(float(G["g4_sent_steering"]), -0.2)
    # the env spec's vehicle moves along its heading only, so ||velocity[:3]|| == |v|
    v = np.array([3.25, 0, 0, 0, 0, 0.7])
    assert wp.speed(v) == abs(v[0])


def test_g5_collect_episode_record():
    T = len(G["g5_actions"])
    col = wp.Collect()
    lidar0 = G["g5_ep_lidar"][0]
    col.reset({"lidar": lidar0, "pose": G["g5_ep_pose"][0], "velocity": G["g5_ep_velocity"][0],
               "speed": G["g5_ep_speed"][0]})
    ep = None
    for k in range(T):
        t = k + 1
        obs = {"lidar": G["g5_ep_lidar"][t], "pose": G["g5_ep_pose"][t], "velocity": G["g5_ep_velocity"][t],
               "speed": G["g5_ep_speed"][t]}
        info = {"lap": 1 + (t // 5), "progress": 0.1 * (t % 5), "time": 0.01 * t}
        ep = col.step(obs, G["g5_actions"][k], float(G["g5_rewards"][k]), k == T - 1, info)
    assert ep is not None
    want_keys = sorted(k[6:] for k in G.files if k.startswith("g5_ep_"))
    assert sorted(ep) == want_keys
    for k in want_keys:
        assert ep[k].dtype == G["g5_ep_" + k].dtype, k
        assert np.array_equal(ep[k], G["g5_ep_" + k]), k
    assert len(ep["reward"]) == T + 1
    assert ep["progress"][0] == -1.0 and ep["discount"][0] == 1.0 and ep["discount"][-1] == 0.0
    assert str(G["g5_dtypes"]).count("float32") == len(want_keys)
    # the obs dict returned to the caller also carries the transition keys (shallow copy, wrappers.py:213)
    assert set(str(G["g5_returned_keys"]).split(",")) >= {"action", "reward", "discount", "progress", "time"}


def test_g7_max_speed_reward():
    got = np.array([wp.max_speed_reward(s, v, w) for s, v, w in zip(G["g7_steering"], G["g7_velocity"], G["g7_wall"])])
    assert np.array_equal(got, G["g7_reward"])
    # the env spec evaluates the same law in fp32 with its own exp (oracle exp32)
    fused = np.where(G["g7_wall"], np.float32(-1.0),
                     -ro.exp32(np.abs(G["g7_steering"].astype(np.float32)) - G["g7_velocity"].astype(np.float32)))
    assert np.max(np.abs(fused - G["g7_reward"]) / np.abs(G["g7_reward"])) <= 5e-7


def test_g8_normalize_and_flatten():
    low, high = np.full(4, 0.25, np.float32), np.full(4, 15.0, np.float32)
    for k, (a, obs, sent) in enumerate(zip(G["g8_actions"], G["g8_obs"], G["g8_sent"])):
        d = wp.flatten_clip_action(a)
        assert [float(d["motor"][0]), float(d["steering"][0])] == list(sent)
        raw = np.full(4, float(k + 1), np.float32)
        assert np.allclose(wp.normalize_obs(raw, low, high), obs, rtol=0, atol=1e-7)
