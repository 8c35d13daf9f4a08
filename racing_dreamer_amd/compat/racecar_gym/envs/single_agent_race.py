"""Single-agent and changing-track envs of racecar_gym's surface, over MultiAgentRaceEnv.

  * ChangingTrackSingleAgentRaceEnv(scenarios, order='sequential'|'random'|'manual')
        baselines/racing/experiments/sb3/sb_experiment.py:61-63, acme/experiment.py:91-92
  * ChangingTrackMultiAgentRaceEnv(scenarios, order='manual') + set_next_env()
        dreamer/evaluations/make_env.py:9-10, run_evaluation.py:48-49
"""
from __future__ import annotations

import random
from typing import List

from .._spaces import EnvBase
from .multi_agent_race import MultiAgentRaceEnv
from .scenarios import MultiAgentScenario, SingleAgentScenario


class SingleAgentRaceEnv(EnvBase):
    """(a `gym.Env` when gym is installed - what `gym.wrappers.TimeLimit`, `FilterObservation` and SB3's `check_env` are
    handed in baselines/racing/experiments/sb3/sb_experiment.py:42-64,97-112)"""
    metadata = {"render.modes": ["follow", "birds_eye"]}
    reward_range = (-float("inf"), float("inf"))
    spec = None

    def __init__(self, scenario: SingleAgentScenario, device: int = 0, seed: int = 0):
        self._scenario = scenario
        self._id = scenario.agent.id
        self._env = MultiAgentRaceEnv(scenario.as_multi(), device=device, seed=seed)
        self.observation_space = self._env.observation_space[self._id]
        self.action_space = self._env.action_space[self._id]

    @property
    def scenario(self):
        return self._scenario

    def step(self, action):
        obs, rew, done, state = self._env.step({self._id: action})
        return obs[self._id], rew[self._id], done[self._id], state[self._id]

    def reset(self, mode: str = "grid"):
        return self._env.reset(mode=mode)[self._id]

    def render(self, mode: str = "follow", **kwargs):
        kwargs.pop("agent", None)
        return self._env.render(mode=mode, agent=self._id, **kwargs)

    def seed(self, seed=None):
        self._env.seed(seed)

    def close(self):
        self._env.close()


class _ChangingTrack(EnvBase):
    """Holds one env per scenario and moves to the next on reset (order 'sequential' / 'random') or on
    set_next_env() (order 'manual')."""
    reward_range = (-float("inf"), float("inf"))
    spec = None

    def __init__(self, envs: List, order: str):
        if order not in ("sequential", "random", "manual"):
            raise ValueError(f"unknown order {order!r}")
        self._envs, self._order, self._index = envs, order, 0
        self._started = False

    @property
    def _current(self):
        return self._envs[self._index]

    @property
    def scenario(self):
        return self._current.scenario

    @property
    def observation_space(self):
        return self._current.observation_space

    @property
    def action_space(self):
        return self._current.action_space

    def set_next_env(self):
        self._index = (self._index + 1) % len(self._envs)

    def step(self, action):
        return self._current.step(action)

    def reset(self, mode: str = "grid"):
        if self._started:
            if self._order == "sequential":
                self.set_next_env()
            elif self._order == "random":
                self._index = random.randrange(len(self._envs))
        self._started = True
        return self._current.reset(mode=mode)

    def render(self, mode: str = "follow", **kwargs):
        return self._current.render(mode=mode, **kwargs)

    def seed(self, seed=None):
        for e in self._envs:
            e.seed(seed)

    def close(self):
        for e in self._envs:
            e.close()


class ChangingTrackSingleAgentRaceEnv(_ChangingTrack):
    metadata = SingleAgentRaceEnv.metadata

    def __init__(self, scenarios: List[SingleAgentScenario], order: str = "sequential", device: int = 0):
        super().__init__([SingleAgentRaceEnv(s, device=device) for s in scenarios], order)


class ChangingTrackMultiAgentRaceEnv(_ChangingTrack):
    metadata = MultiAgentRaceEnv.metadata

    def __init__(self, scenarios: List[MultiAgentScenario], order: str = "sequential", device: int = 0):
        super().__init__([MultiAgentRaceEnv(s, device=device) for s in scenarios], order)


class _Vectorized(EnvBase):
    """Synchronous vector env over a list of scenarios (baselines/racing/environment/environment.py:5,40).  Envs whose
    scenarios agree (track, agents, tasks: `scenario_key`) share ONE device handle - a BatchedRaceEnv with B = their
    number: one launch and one device-to-host copy per step for the whole group - and a list over several tracks becomes
    one handle per track.  Env i of the list is env i of the job: its random resets are drawn from its own stream (the
    Philox key holds the env's index in its group), so a vector env of n equal scenarios is n DIFFERENT envs, where n
    separately built B = 1 envs with one seed would all be the same one."""
    reward_range = (-float("inf"), float("inf"))
    spec = None

    def __init__(self, scenarios, device: int = 0, single: bool = False):
        from .multi_agent_race import RaceCore, scenario_key
        self._single = single
        multi = [s.as_multi() if single else s for s in scenarios]
        groups = {}
        for i, sc in enumerate(multi):
            groups.setdefault(scenario_key(sc), []).append(i)
        self._cores, self._where = [], [None] * len(multi)
        for members in groups.values():
            core = RaceCore(multi[members[0]], len(members), device=device)
            for slot, i in enumerate(members):
                self._where[i] = (len(self._cores), slot)
            self._cores.append((core, members))
        self._scenarios = list(scenarios)
        self._ids = [[a.id for a in sc.agents] for sc in multi]
        pick = (lambda space, ids: space[ids[0]]) if single else (lambda space, ids: space)
        self.observation_space = [pick(self._cores[c][0].observation_space, self._ids[i]) for i, (c, _) in enumerate(self._where)]
        self.action_space = [pick(self._cores[c][0].action_space, self._ids[i]) for i, (c, _) in enumerate(self._where)]

    @property
    def num_device_handles(self) -> int:
        return len(self._cores)

    def _scatter(self, per_core):
        """per_core[c] = list over that core's slots -> list over the envs of the vector env."""
        out = [None] * len(self._where)
        for i, (c, slot) in enumerate(self._where):
            v = per_core[c][slot]
            out[i] = v[self._ids[i][0]] if self._single else v
        return out

    def step(self, actions):
        res = []
        for core, members in self._cores:
            acts = [({self._ids[i][0]: actions[i]} if self._single else actions[i]) for i in members]
            res.append(core.step(acts))
        return tuple(self._scatter([r[k] for r in res]) for k in range(4))

    def reset(self, mode: str = "grid"):
        return self._scatter([core.reset(mode=mode) for core, _ in self._cores])

    def render(self, mode: str = "follow", agents=None, **kw):
        from .rendering import render_view
        agents = agents or [None] * len(self._where)
        states = [core.fetch()[2] for core, _ in self._cores]
        frames = []
        for i, (c, slot) in enumerate(self._where):
            track = self._cores[c][0].scenario.world.track
            frames.append(render_view(track, states[c][slot], focus=agents[i] or self._ids[i][0], mode=mode))
        return frames

    def close(self):
        for core, _ in self._cores:
            core.close()


class VectorizedSingleAgentRaceEnv(_Vectorized):
    def __init__(self, scenarios: List[SingleAgentScenario], device: int = 0):
        super().__init__(scenarios, device=device, single=True)


class VectorizedMultiAgentRaceEnv(_Vectorized):
    def __init__(self, scenarios: List[MultiAgentScenario], device: int = 0):
        super().__init__(scenarios, device=device, single=False)
