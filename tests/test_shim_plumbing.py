"""BASELINE.json configs[0]: 1 env, columbia, obs_type=lidar, CPU step() - plumbing, no GPU.

The REFERENCE's own wrapper stack (dreamer/wrappers.py, order of dreamer/dream.py:103-140) runs unchanged
on top of this repo's racecar_gym shim; the device backend is replaced by the CPU oracle because the build
container has no GPU (tests/oracle_backend.py).  Needs /root/reference, so it only runs in the build
container; the same shim is exercised against the real HIP backend by tests/test_gpu_shim.py."""
import os
import sys

import numpy as np
import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout")


def test_reference_dreamer_wrapper_stack_runs_on_the_shim(ref_wrappers):
    W = ref_wrappers
    from agents.gap_follower import GapFollower
    episodes = []
    env = W.RaceCarBaseEnv(track="columbia", task="max_progress")
    env = W.RaceCarWrapper(env, agent_id="A")
    env = W.ActionRepeat(env, 4)
    env = W.ReduceActionSpace(env, low=[0.005, -1.0], high=[1.0, 1.0])
    env = W.OccupancyMapObs(env)
    env = W.FixedResetMode(env, mode="random")
    env = W.TimeLimit(env, 60)
    env = W.Collect(env, [lambda eps: episodes.append(eps)], 32)
    assert env.agent_ids == ["A"] and env.n_agents == 1
    with pytest.raises(AssertionError, match="Must reset environment."):
        env.step({"A": np.zeros(2)})
    ftg = GapFollower()
    obs = env.reset()
    assert obs["A"]["lidar"].shape == (1080,) and obs["A"]["speed"] == 0.0
    assert obs["A"]["lidar_occupancy"].shape == (64, 64, 1) and not obs["A"]["lidar_occupancy"].any()
    done, steps, total = False, 0, 0.0
    while not done:
        act = np.clip(np.array([-0.9, ftg.action(obs["A"])[-1]]), -1, 1)      # dream.py:213-215 prefill agent
        obs, rew, dones, info = env.step({"A": act})
        done = any(dones.values())
        total += rew["A"]
        steps += 1
        assert set(info["A"]) >= {"pose", "velocity", "lap", "progress", "time", "wrong_way", "wall_collision"}
    assert 1 <= steps <= 60
    ep = episodes[0][0]
    assert sorted(ep) == sorted(["lidar", "pose", "velocity", "speed", "lidar_occupancy", "action", "reward",
                                 "discount", "progress", "time"])
    assert ep["lidar"].shape == (steps + 1, 1080) and ep["lidar"].dtype == np.float32
    assert ep["lidar_occupancy"].shape == (steps + 1, 64, 64, 1) and ep["lidar_occupancy"].dtype == np.uint8
    assert ep["progress"][0] == -1.0 and ep["discount"][-1] == 0.0 and np.all(ep["discount"][:-1] == 1.0)
    assert abs(ep["reward"].sum() - total) < 1e-4
    assert ep["lidar"].max() <= 15.0 and ep["lidar"][1:].min() >= 0.0
    assert 0 < ep["lidar_occupancy"][1:].mean() < 1
    assert ep["time"][1] == pytest.approx(0.04) and ep["speed"][1] > 0      # 4 sub-steps of dt = 0.01


def test_reference_multi_agent_scenario_and_grid_reset(ref_wrappers):
    W = ref_wrappers
    os.chdir(os.path.join(REF, "baselines"))                                # 4-agent scenario A..D
    env = W.RaceCarWrapper(W.RaceCarBaseEnv(track="columbia", task="max_progress"), agent_id="A")
    assert env.agent_ids == ["A", "B", "C", "D"]
    assert "acceleration" in env.observation_space["A"].spaces and "speed" in env.observation_space["A"].spaces
    low, high = env.action_space["A"].low, env.action_space["A"].high
    assert low.tolist() == [-1.0, -1.0] and high.tolist() == [1.0, 1.0]
    obs = env.reset(mode="grid")
    xs = [obs[a]["pose"][:2] for a in env.agent_ids]
    gaps = [np.hypot(*(xs[i] - xs[i + 1])) for i in range(3)]
    assert all(0.8 < g < 2.5 for g in gaps)                                 # ~1.2 m of arc apart, never overlapping
    obs, rew, done, info = env.step({a: np.array([0.5, 0.0]) for a in env.agent_ids})
    assert set(rew) == set(done) == {"A", "B", "C", "D"}


def test_every_scenario_file_of_the_reference_loads_on_the_shim(ref_wrappers):
    """All 29 scenario files the reference ships (dreamer/scenarios/{eval,max_progress}, baselines/scenarios/{max_progress,max_speed})
    load through `from_spec`: the five scene names that are asset names, and the four whose scenes exist upstream only and resolve
    through `SCENE_ASSETS` (scenarios.py; each entry says what it rests on).  The scenario keeps the name the file gave
    (dreamer/evaluations/run_evaluation.py:48 compares it), the track under it is the resolved asset."""
    import glob
    from racecar_gym.envs import scenarios as sc
    from racing_dreamer_amd.track_assets import load_track
    files = sorted(glob.glob(os.path.join(REF, "dreamer", "scenarios", "*", "*.yml")) +
                   glob.glob(os.path.join(REF, "baselines", "scenarios", "*", "*.yml")))
    assert len(files) >= 29
    seen = set()
    for f in files:
        multi = sc.MultiAgentScenario.from_spec(f)
        single = sc.SingleAgentScenario.from_spec(f)
        name = multi.world._config.name
        seen.add(name)
        assert single.world._config.name == name and 1 <= len(multi.agents) <= 4
        asset = load_track(sc.SCENE_ASSETS.get(name, name))
        track = multi.world.track
        assert track.name == name and track.map_name == asset.map_name and np.array_equal(track.occ_words, asset.occ_words)
    assert seen == {"austria", "barcelona", "circle_cw", "columbia", "gbr", "plechaty", "torino", "treitlstrasse", "treitlstrasse_v2"}
    c = np.asarray(load_track("circle").centerline, np.float64)             # "circle_cw": the compiled circle is driven clockwise
    assert 0.5 * np.sum(c[:, 0] * np.roll(c[:, 1], -1) - np.roll(c[:, 0], -1) * c[:, 1]) < -100.0
    W = ref_wrappers                                                        # and one of the resolved scenes steps
    env = W.RaceCarWrapper(W.RaceCarBaseEnv(track="plechaty", task="max_progress"), agent_id="A")
    obs = env.reset(mode="grid")
    obs, rew, done, info = env.step({a: np.array([0.5, 0.0]) for a in env.agent_ids})
    assert obs["A"]["lidar"].shape == (1080,) and np.isfinite(obs["A"]["lidar"]).all() and not done["A"]


# ---------------------------------------------------------------------------------------------- baselines side
class _FilterObservation:                                   # gym.wrappers.FilterObservation (gym 0.17), not reference code
    def __init__(self, env, filter_keys):
        import gym
        self.env, self._keys = env, list(filter_keys)
        self.action_space = env.action_space
        self.observation_space = gym.spaces.Dict({k: v for k, v in env.observation_space.spaces.items() if k in self._keys})

    def __getattr__(self, name):
        return getattr(self.env, name)

    def _f(self, obs):
        return {k: v for k, v in obs.items() if k in self._keys}

    def reset(self, **kw):
        return self._f(self.env.reset(**kw))

    def step(self, action):
        obs, r, d, info = self.env.step(action)
        return self._f(obs), r, d, info


class _TimeLimit:                                           # gym.wrappers.TimeLimit (gym 0.17), not reference code
    def __init__(self, env, max_episode_steps):
        self.env, self._max, self._t = env, max_episode_steps, None
        self.action_space, self.observation_space = env.action_space, env.observation_space

    def __getattr__(self, name):
        return getattr(self.env, name)

    def reset(self, **kw):
        self._t = 0
        return self.env.reset(**kw)

    def step(self, action):
        assert self._t is not None, "Cannot call env.step() before calling reset()"
        obs, r, d, info = self.env.step(action)
        self._t += 1
        if self._t >= self._max:
            info["TimeLimit.truncated"] = not d
            d = True
        return obs, r, d, info


def test_reference_baselines_wrapper_stack_runs_on_the_shim(ref_wrappers, monkeypatch):
    """The env construction and wrapper order of baselines/racing/experiments/sb3/sb_experiment.py:42-63 with the
    reference's own wrapper classes (single_agent.py, common.py) on the shim's single-agent env."""
    import importlib.util

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    SA = load("ref_single_agent", os.path.join(REF, "baselines/racing/environment/single_agent.py"))
    CM = load("ref_common", os.path.join(REF, "baselines/racing/environment/common.py"))
    import racecar_gym.envs.single_agent_race as sar
    from racecar_gym import SingleAgentScenario
    from racecar_gym.envs import ChangingTrackSingleAgentRaceEnv
    monkeypatch.chdir(os.path.join(REF, "baselines"))
    scenarios = [SingleAgentScenario.from_spec(f"scenarios/max_progress/{t}.yml", rendering=False)
                 for t in ("columbia", "austria")]                                              # sb_experiment.py:61-63
    env = ChangingTrackSingleAgentRaceEnv(scenarios=scenarios, order="sequential")
    env = _FilterObservation(env, filter_keys=["lidar"])                                        # :43-48
    env = SA.Flatten(env, flatten_obs=True, flatten_actions=True)
    env = SA.NormalizeObservations(env)
    env = CM.FixedResetMode(env, mode="random")
    env = _TimeLimit(env, max_episode_steps=40)
    env = SA.ActionRepeat(env, n=4)
    assert env.observation_space.shape == (1080,) and env.action_space.shape == (2,)
    for episode, track in enumerate(("columbia", "austria")):                                   # sequential track order
        obs = env.reset()
        assert obs.shape == (1080,) and obs.min() >= 0.0 and obs.max() <= 1.0                   # ranges / 15
        done, steps, total = False, 0, 0.0
        while not done:
            obs, r, done, info = env.step(np.array([0.4, 0.1 if steps % 2 else -0.1]))          # flat (motor, steering)
            total += r
            steps += 1
            assert obs.shape == (1080,) and 0.0 <= obs.min() and obs.max() <= 1.0
        assert 1 <= steps <= 10                                                                 # 40 sub-steps / repeat 4
        assert {"wrong_way", "progress", "lap", "wall_collision", "time"} <= set(info)
        assert np.isfinite(total)


def test_vector_env_is_one_batched_handle_per_track(ref_wrappers, monkeypatch):
    """VectorizedSingleAgentRaceEnv (baselines/racing/environment/environment.py:5,40): eight envs of one scenario sit on ONE
    backend handle with B = 8 - and behave like eight independent envs, each with its own reset stream: env i of the list
    equals a lone oracle env whose global index is i, step for step.  A list over two tracks becomes two handles; results
    come back in list order."""
    from oracle import c_oracle
    from oracle import racecar_oracle as ro
    from racecar_gym import SingleAgentScenario
    from racecar_gym.envs import VectorizedSingleAgentRaceEnv
    from racing_dreamer_amd.track_assets import load_track
    monkeypatch.chdir(os.path.join(REF, "baselines"))
    scen = [SingleAgentScenario.from_spec("scenarios/max_progress/columbia.yml", rendering=False) for _ in range(8)]
    vec = VectorizedSingleAgentRaceEnv(scenarios=scen)
    assert vec.num_device_handles == 1 and vec._cores[0][0].env.B == 8 and len(vec.action_space) == 8
    t = load_track("columbia")
    agent = scen[0].agent
    p = agent.task_params
    lone = []
    for i in range(8):
        cfg = ro.OracleConfig(num_envs=1, first_env=i, laps=int(p.get("laps", 10)), time_limit=float(p.get("time_limit", 180.0)),
                              terminate_on_collision=bool(p.get("terminate_on_collision", True)),
                              collision_reward=float(p.get("collision_reward", 0.0)))
        lone.append(c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg))
    obs = vec.reset(mode="random")
    want = [e.reset(mode=ro.RESET_RANDOM, seed=0) for e in lone]
    poses = np.stack([o["pose"] for o in obs])
    assert len({tuple(np.round(q, 6)) for q in poses}) == 8              # eight different spawn poses, not one eight times
    rng = np.random.default_rng(0)
    for k in range(12):
        for i in range(8):
            assert np.array_equal(obs[i]["lidar"], want[i]["lidar"][0].astype(np.float64)), (k, i)
            assert np.array_equal(obs[i]["pose"], want[i]["pose"][0].astype(np.float64)), (k, i)
        a = rng.uniform(-1, 1, (8, 2)).astype(np.float32)
        obs, rew, done, info = vec.step([{"motor": np.array([a[i, 0]]), "steering": np.array([a[i, 1]])} for i in range(8)])
        want = [lone[i].step(a[i:i + 1]) for i in range(8)]
        for i in range(8):
            assert rew[i] == float(want[i]["reward"][0]) and done[i] == bool(want[i]["done"][0]), (k, i)
            assert info[i]["progress"] == float(want[i]["progress"][0]) and info[i]["lap"] == int(want[i]["lap"][0])
    vec.close()
    mixed = [SingleAgentScenario.from_spec(f"scenarios/max_progress/{name}.yml", rendering=False)
             for name in ("columbia", "austria", "columbia", "austria", "austria")]
    vec = VectorizedSingleAgentRaceEnv(scenarios=mixed)
    assert vec.num_device_handles == 2 and sorted(c.env.B for c, _ in vec._cores) == [2, 3]
    obs = vec.reset(mode="grid")
    assert len(obs) == 5 and np.array_equal(obs[0]["pose"], obs[2]["pose"]) and np.array_equal(obs[1]["pose"], obs[4]["pose"])
    obs, rew, done, info = vec.step([{"motor": np.array([0.5]), "steering": np.array([0.0])}] * 5)
    assert len(obs) == len(rew) == len(done) == len(info) == 5 and all(isinstance(d, bool) for d in done)
    frames = vec.render(mode="birds_eye")
    assert len(frames) == 5 and frames[0].shape == frames[1].shape and frames[0].ndim == 3
    vec.close()


def test_host_progress_task_equals_the_device_task(ref_wrappers, monkeypatch):
    """racecar_gym.tasks.progress_based.MaximizeProgressTask (ros_agent/helpers/wrappers.py:17-20) evaluated on the state dicts
    the shim returns == the reward and done flag the backend computed itself, step for step, over episodes that end in walls."""
    from racecar_gym import SingleAgentScenario
    from racecar_gym.envs import SingleAgentRaceEnv
    from racecar_gym.tasks.progress_based import MaximizeProgressTask
    monkeypatch.chdir(os.path.join(REF, "dreamer"))
    scen = SingleAgentScenario.from_spec("scenarios/max_progress/columbia.yml", rendering=False)
    p = scen.agent.task_params
    env = SingleAgentRaceEnv(scen)
    task = MaximizeProgressTask(**p)
    rng = np.random.default_rng(1)
    ended = 0
    for episode in range(4):
        env.reset(mode="random" if episode else "grid")
        task.reset()
        for k in range(400):
            a = {"motor": np.array([rng.uniform(0.2, 1.0)]), "steering": np.array([rng.uniform(-1, 1)])}
            obs, rew, done, info = env.step(a)
            state = {scen.agent.id: info}
            assert task.reward(scen.agent.id, state, a) == pytest.approx(rew, abs=2e-4), (episode, k)
            assert task.done(scen.agent.id, state) == done
            if done:
                ended += 1
                break
    assert ended >= 2 and float(p["collision_reward"]) != 0.0
    env.close()


# ---------------------------------------------------------------------------------------------- with a gym that checks
def _install_strict_gym():
    """A `gym` whose Env is a real class (gym 0.17's attributes) and whose Wrapper, TimeLimit, FilterObservation and a
    `check_env`-style helper all insist on `isinstance(env, gym.Env)` - what SB3's check_env and vec-envs do with the env
    baselines/racing/experiments/sb3/sb_experiment.py:42-64,97-112 builds.  Not reference code."""
    import types
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden as mg

    class Env:
        metadata = {"render.modes": []}
        reward_range = (-float("inf"), float("inf"))
        spec = None
        action_space = None
        observation_space = None

        @property
        def unwrapped(self):
            return self

        def seed(self, seed=None):
            return

        def close(self):
            pass

    class Wrapper(Env):
        def __init__(self, env):
            assert isinstance(env, Env), f"{type(env).__name__} is not a gym.Env"
            self.env = env
            self.action_space, self.observation_space = env.action_space, env.observation_space
            self.reward_range, self.metadata = env.reward_range, env.metadata

        def __getattr__(self, name):
            if name.startswith("_"):
                raise AttributeError(name)
            return getattr(self.env, name)

        @property
        def unwrapped(self):
            return self.env.unwrapped

        def step(self, action):
            return self.env.step(action)

        def reset(self, **kw):
            return self.env.reset(**kw)

    class ObservationWrapper(Wrapper):
        def reset(self, **kw):
            return self.observation(self.env.reset(**kw))

        def step(self, action):
            obs, r, d, info = self.env.step(action)
            return self.observation(obs), r, d, info

    class FilterObservation(ObservationWrapper):            # gym.wrappers.FilterObservation
        def __init__(self, env, filter_keys=None):
            super().__init__(env)
            self._keys = list(filter_keys)
            self.observation_space = spaces.Dict({k: v for k, v in env.observation_space.spaces.items() if k in self._keys})

        def observation(self, obs):
            return {k: v for k, v in obs.items() if k in self._keys}

    class TimeLimit(Wrapper):                               # gym.wrappers.TimeLimit
        def __init__(self, env, max_episode_steps=None):
            super().__init__(env)
            self._max, self._t = max_episode_steps, None

        def reset(self, **kw):
            self._t = 0
            return self.env.reset(**kw)

        def step(self, action):
            assert self._t is not None, "Cannot call env.step() before calling reset()"
            obs, r, d, info = self.env.step(action)
            self._t += 1
            if self._t >= self._max:
                info["TimeLimit.truncated"] = not d
                d = True
            return obs, r, d, info

    gym = types.ModuleType("gym")
    spaces = types.ModuleType("gym.spaces")
    spaces.Box, spaces.Dict = mg.Box, mg.Dict
    spaces.flatten_space, spaces.flatten, spaces.unflatten = mg._flatten_space, mg._flatten, mg._unflatten
    wrappers = types.ModuleType("gym.wrappers")
    wrappers.TimeLimit, wrappers.FilterObservation = TimeLimit, FilterObservation
    gym.spaces, gym.wrappers, gym.Env, gym.Wrapper, gym.ObservationWrapper = spaces, wrappers, Env, Wrapper, ObservationWrapper
    sys.modules.update({"gym": gym, "gym.spaces": spaces, "gym.wrappers": wrappers})
    return gym


def _check_env(env, gym):
    """The structural part of stable_baselines3.common.env_checker.check_env."""
    assert isinstance(env, gym.Env), "Your environment must inherit from the gym.Env class"
    assert env.observation_space is not None and env.action_space is not None
    assert isinstance(env.reward_range, tuple) and len(env.reward_range) == 2
    assert isinstance(env.metadata, dict) and env.unwrapped is not None


def test_shim_envs_are_gym_envs_when_gym_is_present(monkeypatch):
    """with gym installed every env class of the shim IS a `gym.Env` - so the env construction of
    baselines/racing/experiments/sb3/sb_experiment.py:42-64 works with wrappers that insist on it - and without gym they are the
    stand-in base (every other test of this file)."""
    import importlib.util
    saved = {k: v for k, v in sys.modules.items() if k == "gym" or k.startswith("gym.") or k == "racecar_gym" or k.startswith("racecar_gym.")}
    for k in saved:
        del sys.modules[k]
    try:
        gym = _install_strict_gym()
        from racing_dreamer_amd import compat
        compat.install()
        import racecar_gym
        import racecar_gym._spaces as sp
        import racecar_gym.envs.multi_agent_race as mar
        from oracle_backend import OracleBackend
        monkeypatch.setattr(mar, "_BACKEND", OracleBackend)
        assert sp.HAVE_GYM and sp.EnvBase is gym.Env
        from racecar_gym import SingleAgentScenario
        from racecar_gym.envs import (ChangingTrackMultiAgentRaceEnv, ChangingTrackSingleAgentRaceEnv, MultiAgentRaceEnv, MultiAgentScenario,
                                      SingleAgentRaceEnv, VectorizedMultiAgentRaceEnv, VectorizedSingleAgentRaceEnv)
        for cls in (SingleAgentRaceEnv, MultiAgentRaceEnv, ChangingTrackSingleAgentRaceEnv, ChangingTrackMultiAgentRaceEnv,
                    VectorizedSingleAgentRaceEnv, VectorizedMultiAgentRaceEnv):
            assert issubclass(cls, gym.Env), cls

        def load(name, path):
            spec = importlib.util.spec_from_file_location(name, path)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            return mod
        SA = load("ref_single_agent_strict", os.path.join(REF, "baselines/racing/environment/single_agent.py"))
        CM = load("ref_common_strict", os.path.join(REF, "baselines/racing/environment/common.py"))
        monkeypatch.chdir(os.path.join(REF, "baselines"))
        scenarios = [SingleAgentScenario.from_spec(f"scenarios/max_progress/{t}.yml", rendering=False) for t in ("columbia", "austria")]
        env = ChangingTrackSingleAgentRaceEnv(scenarios=scenarios, order="sequential")            # sb_experiment.py:61-63
        _check_env(env, gym)
        env = gym.wrappers.FilterObservation(env, filter_keys=["lidar"])                          # :43
        env = SA.Flatten(env, flatten_obs=True, flatten_actions=True)                             # :44
        env = SA.NormalizeObservations(env)                                                       # :45
        env = CM.FixedResetMode(env, mode="random")                                               # :46
        env = gym.wrappers.TimeLimit(env, max_episode_steps=40)                                   # :47
        env = SA.ActionRepeat(env, n=4)                                                           # :48
        _check_env(env, gym)
        assert isinstance(env.unwrapped, ChangingTrackSingleAgentRaceEnv)
        obs = env.reset()
        assert obs.shape == (1080,) and 0.0 <= obs.min() and obs.max() <= 1.0
        done, steps = False, 0
        while not done:
            obs, r, done, info = env.step(np.array([0.4, 0.05]))
            steps += 1
        assert 1 <= steps <= 10 and {"wrong_way", "progress", "lap"} <= set(info)
        env.close()
        multi = MultiAgentRaceEnv(MultiAgentScenario.from_spec("scenarios/max_progress/columbia.yml", rendering=False))
        _check_env(multi, gym)
        multi.close()
    finally:
        for k in [m for m in sys.modules if m == "gym" or m.startswith("gym.") or m == "racecar_gym" or m.startswith("racecar_gym.")]:
            del sys.modules[k]
        sys.modules.update(saved)
