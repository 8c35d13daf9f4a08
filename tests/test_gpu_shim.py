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
