"""SURVEY.md §8a H13 / §8f N1 from the caller's side: the rollout driver of the reference (`tools.simulate`,
dreamer/tools.py:154-206) and its dataset reader (`tools.load_episodes`, tools.py:235-264), restated in
oracle/caller_port.py because tools.py itself needs TensorFlow, run over

  * the REFERENCE's own wrapper stack (dreamer/wrappers.py, order of dreamer/dream.py:103-140) on this repo's
    racecar_gym shim - CPU oracle backend, needs /root/reference, build container only;
  * `EpisodeRecorder` files, consumed exactly as `load_episodes` consumes the reference's episode files.
The `-m gpu` counterpart (HIP backend) is tests/test_gpu_shim.py::test_caller_loop_on_the_hip_shim."""
import os

import numpy as np
import pytest
import torch

from conftest import REF
from helpers import make_oracle
from oracle import caller_port as cp
from oracle import racecar_oracle as ro
from racing_dreamer_amd.track_assets import synthetic_track
from racing_dreamer_amd.trajectory import EpisodeRecorder, count_steps, save_episodes

needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout")


class ScriptedPolicy:
    """Stub of the agent call `agents[i](obs, done, state)` (tools.py:186-189): records what it was handed."""

    def __init__(self, motor=0.6):
        self.calls, self.motor = [], motor

    def __call__(self, obs, done, state):
        self.calls.append(({k: np.asarray(v).shape for k, v in obs.items()}, np.asarray(done).copy(), state))
        scan = obs["lidar"][0]
        steer = float(np.clip((scan[700:900].mean() - scan[180:380].mean()) * 0.4, -1, 1))   # towards the open side
        return np.array([[self.motor, steer]]), (0 if state is None else state + 1)      # (+ = right)


def _dreamer_env(W, episodes, duration=25):
    env = W.RaceCarBaseEnv(track="columbia", task="max_progress")
    env = W.RaceCarWrapper(env, agent_id="A")
    env = W.ActionRepeat(env, 4)
    env = W.ReduceActionSpace(env, low=[0.005, -1.0], high=[1.0, 1.0])
    env = W.OccupancyMapObs(env)
    env = W.FixedResetMode(env, mode="random")
    env = W.TimeLimit(env, duration)
    return W.Collect(env, [lambda eps: episodes.extend(eps)], 32)


class CountingEnv:
    def __init__(self, env):
        self.env, self.resets, self.steps, self.done_before_reset = env, 0, 0, []
        self._last_done = True

    def reset(self):
        self.resets += 1
        self.done_before_reset.append(self._last_done)
        return self.env.reset()

    def step(self, actions):
        self.steps += 1
        out = self.env.step(actions)
        self._last_done = any(out[2].values())
        return out


@needs_ref
def test_simulate_loop_over_the_reference_wrapper_stack_on_the_shim(ref_wrappers, tmp_path):
    episodes = []
    env = CountingEnv(_dreamer_env(ref_wrappers, episodes))
    pol = ScriptedPolicy()
    # --- episodes mode: exactly 3 episodes, reset once per episode and only after a done
    state, stats = cp.rollout([pol], env, ["A"], episodes=3)
    assert len(episodes) == 3 and env.resets == 3 and all(env.done_before_reset)
    assert stats["env_steps"] == env.steps == sum(len(e["reward"]) - 1 for e in episodes)
    step_over, ep_over, dones, length, obs, pstate = state
    assert ep_over == 0 and step_over == env.steps          # steps=0: `step - steps` is the number of steps taken
    assert any(dones.values()) and int(length.sum()) == 0   # an episode boundary: the running length was cleared
    # batch dimension of 1 on every observation value, the done flag and the carried policy state (tools.py:184-188)
    shapes, done0, st0 = pol.calls[0]
    assert shapes["lidar"] == (1, 1080) and shapes["lidar_occupancy"] == (1, 64, 64, 1) and shapes["speed"] == (1,)
    assert done0.shape == (1,) and bool(done0[0]) and st0 is None
    assert [c[2] for c in pol.calls[1:4]] == [0, 1, 2]
    first_of_episode = [i for i, c in enumerate(pol.calls) if c[1][0]]
    assert len(first_of_episode) == 3                       # done=True is seen exactly at the first step of each episode
    # statistics of the first agent are collected at the NEXT reset: 2 of the 3 episodes so far (tools.py:179-182)
    assert len(stats["progress"]) == 2 and len(stats["return"]) == 2
    for k in range(2):
        assert stats["progress"][k] == pytest.approx(float(episodes[k]["progress"][1:].max()), abs=1e-6)
        assert stats["return"][k] == pytest.approx(float(episodes[k]["reward"].sum()), abs=1e-4)
    # --- steps mode, resumed from the returned state: runs until the finished episodes hold >= 40 agent steps
    before = env.steps
    state2, stats2 = cp.rollout([pol], env, ["A"], steps=40, state=(0, 0, dones, length, obs, pstate))
    taken = env.steps - before
    done_steps = sum(len(e["reward"]) - 1 for e in episodes[3:])
    assert done_steps >= 40 and state2[0] == done_steps - 40 and taken == done_steps
    assert env.resets == len(episodes) and all(env.done_before_reset)
    # --- the episode files the collector's callback would write, counted the way the driver counts them
    save_episodes(tmp_path, episodes)
    n_eps, n_steps = cp.count_episode_files(tmp_path)
    assert n_eps == len(episodes) and n_steps == env.steps == count_steps(tmp_path)


def _views(out, B, A):
    v = {}
    for k, a in out.items():
        a = np.asarray(a)
        t = torch.from_numpy(a.reshape(B, A, *a.shape[1:]).copy())
        v[k] = t.unsqueeze(-1) if k == "lidar_occupancy" else t
    return v


def test_load_episodes_windows_over_recorder_files(tmp_path):
    """N1 consumer side: EpisodeRecorder output read back as dreamer/tools.py:235-264 reads the reference's files."""
    B = 5
    env = make_oracle(synthetic_track(), num_envs=B, auto_reset=True, render_occupancy=True, time_limit_steps=12)
    rec = EpisodeRecorder(B, 1, env_indices=[0, 1, 2, 3, 4], directory=str(tmp_path))
    rec.on_reset(_views(env.reset(mode=ro.RESET_RANDOM, seed=3), B, 1))
    episodes = []
    for k in range(40):
        act = ro.random_actions(9, k, B)
        act[:, 0] = 0.8
        episodes += rec.on_step(_views(env.step(act, repeat=4), B, 1))
    assert len(episodes) >= 10
    # plus one episode that is too short for the window below (the reader must skip it, tools.py:253-255)
    save_episodes(tmp_path, [{k: v[:8] for k, v in episodes[0].items()}])
    L = 8
    assert min(len(e["reward"]) for e in episodes) > L
    by_len = {}
    for e in episodes:
        by_len.setdefault(len(e["reward"]), []).append(e)
    for balance in (False, True):
        wins = list(cp.episode_windows(tmp_path, rescan=64, length=L, balance=balance, seed=1, rounds=2))
        assert 0 < len(wins) <= 128
        saw_start = saw_end = False
        for w in wins:
            assert sorted(w) == sorted(episodes[0])
            assert all(len(v) == L for v in w.values())                     # EVERY key sliced by the same window
            assert w["lidar"].dtype == np.float32 and w["lidar_occupancy"].dtype == np.uint8
            # the window is a contiguous piece of one recorded episode
            match = False
            for e in (x for n, xs in by_len.items() if n > L for x in xs):
                t0 = np.nonzero(np.all(e["lidar"] == w["lidar"][0], axis=1))[0]
                for s in t0:
                    if s + L <= len(e["reward"]) and all(np.array_equal(e[k][s:s + L], w[k]) for k in w):
                        match = True
                        saw_start |= s == 0
                        saw_end |= s + L == len(e["reward"])
            assert match
        assert saw_end                                       # `available + 1`: the last step of an episode is reachable
        if not balance:
            assert saw_start
    # reset row and terminal row survive the round trip through the files
    full = list(cp.episode_windows(tmp_path, rescan=32, length=None, seed=0))
    full = [w for w in full if len(w["reward"]) > L]          # (not the truncated copy made above)
    assert full and all(w["progress"][0] == -1.0 and w["discount"][0] == 1.0 and w["discount"][-1] == 0.0 for w in full)
