"""The racecar_gym shim on the real HIP backend (B = 1 view) against the CPU oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scenario(tmp_path, track, agents=("A",), task="maximize_progress", params=None):
    import yaml
    params = params if params is not None else dict(laps=10, time_limit=180.0, terminate_on_collision=True,
                                                      collision_reward=-1.0)
    spec = {"world": {"name": track},
            "agents": [{"id": a, "vehicle": {"name": "racecar", "sensors": ["lidar", "pose", "velocity"]},
                        "task": {"task_name": task, "params": params}} for a in agents]}
    p = tmp_path / f"{track}.yml"
    p.write_text(yaml.safe_dump(spec))
    return str(p)


def test_shim_step_matches_oracle(tmp_path):
    from racing_dreamer_amd import compat
    compat.install()
    from racecar_gym.envs.multi_agent_race import MultiAgentRaceEnv, MultiAgentScenario
    from helpers import make_oracle
    from oracle import racecar_oracle as ro
    from racing_dreamer_amd.track_assets import load_track
    env = MultiAgentRaceEnv(MultiAgentScenario.from_spec(_scenario(tmp_path, "austria", ("A", "B"))))
    ora = make_oracle(load_track("austria"), num_envs=1, cars_per_env=2)
    obs = env.reset(mode="random_ball")
    oo = ora.reset(mode=ro.RESET_RANDOM_BALL, seed=0)
    assert np.array_equal(obs["A"]["lidar"], oo["lidar"][0].astype(np.float64))
    assert set(obs["A"]) == {"lidar", "pose", "velocity"} and obs["B"]["pose"].shape == (6,)
    for k in range(40):
        a = ro.random_actions(2, k, 2)
        a[:, 0] = np.abs(a[:, 0])
        obs, rew, done, info = env.step({"A": {"motor": a[0, 0], "steering": a[0, 1]},
                                         "B": {"motor": np.array([a[1, 0]]), "steering": np.array([a[1, 1]])}})
        oo = ora.step(a)
        for i, aid in enumerate("AB"):
            assert np.array_equal(obs[aid]["lidar"], oo["lidar"][i].astype(np.float64))
            assert np.array_equal(info[aid]["pose"], oo["pose"][i].astype(np.float64))
            assert rew[aid] == float(oo["reward"][i]) and done[aid] == bool(oo["done"][i])
            assert info[aid]["lap"] == int(oo["lap"][i]) and info[aid]["progress"] == float(oo["progress"][i])
            assert info[aid]["wall_collision"] == bool(oo["wall_collision"][i])
            assert isinstance(info[aid]["wrong_way"], bool) and info[aid]["time"] == pytest.approx(0.01 * (k + 1))
        if any(done.values()):
            break
    frame = env.render(mode="birds_eye", agent="A")
    assert frame.ndim == 3 and frame.shape[2] == 3 and frame.dtype == np.uint8
    env.close()


def test_registered_host_task_and_changing_track(tmp_path):
    """baselines/racing/environment/tasks.py registers `max_speed`; the shim evaluates it on the host."""
    import math
    from racing_dreamer_amd import compat
    compat.install()
    from racecar_gym import SingleAgentScenario, Task, register_task
    from racecar_gym.envs import ChangingTrackSingleAgentRaceEnv

    class MaximizeSpeed(Task):
        def reward(self, agent_id, state, action):
            s = state[agent_id]
            return -1.0 if s["wall_collision"] else -math.exp(math.fabs(action["steering"][0]) - s["velocity"][0])

        def done(self, agent_id, state):
            return False

    register_task(name="max_speed", task=MaximizeSpeed)
    scen = [SingleAgentScenario.from_spec(_scenario(tmp_path, t, task="max_speed", params={})) for t in
            ("columbia", "treitlstrasse_v2")]
    env = ChangingTrackSingleAgentRaceEnv(scenarios=scen, order="sequential")
    assert env.action_space["motor"].shape == (1,) and np.isfinite(env.observation_space["lidar"].high).all()
    obs = env.reset(mode="grid")
    assert env.scenario.world._config.name == "columbia"
    for k in range(5):
        obs, r, d, info = env.step({"motor": np.array([1.0]), "steering": np.array([0.2])})
        assert r == -math.exp(0.2 - info["velocity"][0]) and d is False
        assert {"wrong_way", "progress", "lap"} <= set(info)
    env.reset(mode="grid")
    assert env.scenario.world._config.name == "treitlstrasse_v2"
    env.close()


def test_shim_single_env_step_rate(tmp_path):
    """The B = 1 compatibility path is for drop-in use, not throughput - but it should still be far above the
    reference's ~270 sim-steps/s per worker."""
    import time
    from racing_dreamer_amd import compat
    compat.install()
    from racecar_gym.envs.multi_agent_race import MultiAgentRaceEnv, MultiAgentScenario
    env = MultiAgentRaceEnv(MultiAgentScenario.from_spec(_scenario(
        tmp_path, "columbia", params=dict(laps=10, time_limit=180.0, terminate_on_collision=False, collision_reward=0.0))))
    env.reset(mode="grid")
    act = {"A": {"motor": 0.3, "steering": 0.05}}
    for _ in range(20):
        env.step(act)
    t0 = time.perf_counter()
    n = 300
    for _ in range(n):
        obs, rew, done, info = env.step(act)
    rate = n / (time.perf_counter() - t0)
    print(f"shim single-env rate: {rate:.0f} steps/s")
    assert rate > 1000
    env.close()


class _PortStack:
    """The Dreamer wrapper order (dreamer/dream.py:134-140, 103-115) assembled from the pinned restatements in
    oracle/wrappers_port.py - the reference's own classes cannot travel to the GPU box - over a shim env:
    RaceCarWrapper (flat action -> dict, speed) -> ActionRepeat -> ReduceActionSpace -> FixedResetMode -> TimeLimit
    -> Collect."""

    def __init__(self, env, repeat, duration, mode, sink):
        from oracle import wrappers_port as wp
        self.wp, self.env, self.repeat, self.mode, self.sink = wp, env, repeat, mode, sink
        self.limit, self.collect = wp.TimeLimit(duration), wp.Collect()

    def _obs(self, obs, velocity):
        o = dict(obs["A"])
        o["speed"] = self.wp.speed(velocity)
        return {"A": o}

    def reset(self):
        obs = self.env.reset(mode=self.mode)
        self.limit.reset()
        obs = self._obs(obs, np.zeros(6))
        obs["A"]["speed"] = 0.0                                       # wrappers.py:74
        self.collect.reset(obs["A"])
        return obs

    def step(self, actions):
        a = self.wp.reduce_action(np.asarray(actions["A"], np.float64))
        inner = lambda act: self.env.step({"A": {"motor": act[0], "steering": act[1]}})
        obs, total, dones, info, _ = self.wp.action_repeat_dreamer(inner, ["A"], a, self.repeat)
        dones = self.limit.step(dones)
        obs = self._obs(obs, info["A"]["velocity"])
        ep = self.collect.step(obs["A"], actions["A"], total["A"], dones["A"], info["A"])
        if ep is not None:
            self.sink.append(ep)
        return obs, total, dones, info


def test_caller_loop_on_the_hip_shim(tmp_path, monkeypatch):
    """SURVEY.md H13 on the device: the rollout driver (oracle/caller_port.py = dreamer/tools.py:154-206) over the
    wrapper order of dream.py on the shim, once with the HIP backend and once with the CPU oracle as backend: the
    episodes the collector hands to its callbacks must be identical, value for value."""
    from racing_dreamer_amd import compat
    compat.install()
    import racecar_gym.envs.multi_agent_race as mar
    from oracle import caller_port as cp
    from oracle_backend import OracleBackend

    def policy(obs, done, state):
        assert obs["lidar"].shape == (1, 1080) and done.shape == (1,)
        scan = obs["lidar"][0]
        steer = float(np.clip((scan[700:900].mean() - scan[180:380].mean()) * 0.4, -1, 1))
        k = 0 if state is None else state + 1
        return np.array([[0.3 + 0.1 * (k % 3), steer]]), k                     # (+ = right)

    path = _scenario(tmp_path, "columbia")
    runs = []
    for backend in (None, OracleBackend):
        if backend is not None:
            monkeypatch.setattr(mar, "_BACKEND", backend)
        eps = []
        env = _PortStack(mar.MultiAgentRaceEnv(mar.MultiAgentScenario.from_spec(path)), repeat=4, duration=30,
                         mode="random", sink=eps)
        state, stats = cp.rollout([policy], env, ["A"], episodes=3)
        state, stats2 = cp.rollout([policy], env, ["A"], steps=20, state=(0, 0) + state[2:])
        runs.append((eps, stats, stats2, state[0]))
        env.env.close()
    (hip_eps, hip_stats, hip_stats2, hip_over), (ora_eps, ora_stats, ora_stats2, ora_over) = runs
    assert len(hip_eps) == len(ora_eps) >= 4 and hip_over == ora_over
    assert hip_stats["env_steps"] == ora_stats["env_steps"] and hip_stats["resets"] == 3
    for a, b in zip(hip_eps, ora_eps):
        assert sorted(a) == sorted(b)
        for k in a:
            assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), k
    assert hip_stats["progress"] == ora_stats["progress"] and hip_stats["return"] == ora_stats["return"]


def test_episode_recorder_on_the_hip_env_equals_the_oracle_recording(tmp_path):
    """N1 on the device: EpisodeRecorder over BatchedRaceEnv views against the same recorder over the CPU oracle."""
    import torch
    from helpers import make_oracle
    from oracle import racecar_oracle as ro
    from racing_dreamer_amd.batched_env import BatchedRaceEnv
    from racing_dreamer_amd.track_assets import load_track
    from racing_dreamer_amd.trajectory import EpisodeRecorder, count_steps
    track, B = load_track("treitlstrasse_v2"), 48
    picks = [0, 7, 31, 47]
    env = BatchedRaceEnv(track, B, 1, obs_type="lidar_occupancy", auto_reset=True, time_limit_steps=15)
    ora = make_oracle(track, num_envs=B, auto_reset=True, render_occupancy=True, time_limit_steps=15)
    rec_d = EpisodeRecorder(B, 1, picks, directory=str(tmp_path / "hip"))
    rec_o = EpisodeRecorder(B, 1, picks)

    def oviews(out):
        v = {}
        for k, a in out.items():
            a = np.asarray(a)
            t = torch.from_numpy(a.reshape(B, 1, *a.shape[1:]).copy())
            v[k] = t.unsqueeze(-1) if k == "lidar_occupancy" else t
        return v

    rec_d.on_reset(env.reset(mode="random", seed=6))
    rec_o.on_reset(oviews(ora.reset(mode=ro.RESET_RANDOM, seed=6)))
    eps_d, eps_o = [], []
    for k in range(40):
        act = ro.random_actions(3, k, B)
        eps_d += rec_d.on_step(env.step(torch.from_numpy(act).cuda(), repeat=4))
        eps_o += rec_o.on_step(oviews(ora.step(act, repeat=4)))
    assert len(eps_d) == len(eps_o) >= 8
    for a, b in zip(eps_d, eps_o):
        assert sorted(a) == sorted(b)
        for k in a:
            assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), k
    assert count_steps(tmp_path / "hip") == sum(len(e["reward"]) - 1 for e in eps_d)
    env.close()


def test_vector_env_steps_eight_envs_with_one_launch(tmp_path):
    """VectorizedMultiAgentRaceEnv / VectorizedSingleAgentRaceEnv on the HIP backend: eight same-track envs are ONE handle
    (one dynamics + one scan launch and one device-to-host copy per step), equal to the oracle's B = 8 env; a two-track
    list is two handles."""
    from racing_dreamer_amd import compat
    compat.install()
    from racecar_gym import SingleAgentScenario
    from racecar_gym.envs import VectorizedMultiAgentRaceEnv, VectorizedSingleAgentRaceEnv
    from racecar_gym.envs.multi_agent_race import MultiAgentScenario
    from helpers import make_oracle
    from oracle import racecar_oracle as ro
    from racing_dreamer_amd.track_assets import load_track
    path = _scenario(tmp_path, "austria", ("A", "B"))
    vec = VectorizedMultiAgentRaceEnv([MultiAgentScenario.from_spec(path) for _ in range(8)])
    assert vec.num_device_handles == 1
    core = vec._cores[0][0]
    core.env.set_profiling(True)
    ora = make_oracle(load_track("austria"), num_envs=8, cars_per_env=2)
    obs = vec.reset(mode="random_ball")
    oo = ora.reset(mode=ro.RESET_RANDOM_BALL, seed=0)
    steps = 25
    for k in range(steps):
        for e in range(8):
            for i, aid in enumerate("AB"):
                assert np.array_equal(obs[e][aid]["lidar"], oo["lidar"][2 * e + i].astype(np.float64)), (k, e, aid)
        a = ro.random_actions(3, k, 16)
        obs, rew, done, info = vec.step([{aid: {"motor": a[2 * e + i, 0], "steering": a[2 * e + i, 1]} for i, aid in enumerate("AB")}
                                         for e in range(8)])
        oo = ora.step(a)
        for e in range(8):
            for i, aid in enumerate("AB"):
                c = 2 * e + i
                assert rew[e][aid] == float(oo["reward"][c]) and done[e][aid] == bool(oo["done"][c]), (k, e, aid)
                assert info[e][aid]["progress"] == float(oo["progress"][c]) and info[e][aid]["wall_collision"] == bool(oo["wall_collision"][c])
    kt = core.env.kernel_times()
    assert kt["rc_dynamics_kernel"]["launches"] == steps and kt["rc_raycast_kernel"]["launches"] == steps + 1   # (+ the reset's scan)
    vec.close()
    single = [SingleAgentScenario.from_spec(_scenario(tmp_path, t)) for t in ("columbia", "austria", "columbia")]
    vec = VectorizedSingleAgentRaceEnv(single)
    assert vec.num_device_handles == 2
    obs = vec.reset(mode="grid")
    assert np.array_equal(obs[0]["lidar"], obs[2]["lidar"]) and not np.array_equal(obs[0]["lidar"], obs[1]["lidar"])
    obs, rew, done, info = vec.step([{"motor": np.array([0.4]), "steering": np.array([0.1])}] * 3)
    assert rew[0] == rew[2] and info[0]["progress"] == info[2]["progress"] and len(obs) == 3
    vec.close()
