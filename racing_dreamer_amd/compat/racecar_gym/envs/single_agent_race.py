"""Single-agent and changing-track envs of racecar_gym's surface, over MultiAgentRaceEnv.

  * ChangingTrackSingleAgentRaceEnv(scenarios, order='sequential'|'random'|'manual')
        baselines/racing/experiments/sb3/sb_experiment.py:61-63, acme/experiment.py:91-92
  * ChangingTrackMultiAgentRaceEnv(scenarios, order='manual') + set_next_env()
        dreamer/evaluations/make_env.py:9-10, run_evaluation.py:48-49
"""
from __future__ import annotations

import random
from typing import List

from .multi_agent_race import MultiAgentRaceEnv
from .scenarios import MultiAgentScenario, SingleAgentScenario


class SingleAgentRaceEnv:
    metadata = {"render.modes": ["follow", "birds_eye"]}

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


class _ChangingTrack:
    """Holds one env per scenario and moves to the next on reset (order 'sequential' / 'random') or on
    set_next_env() (order 'manual')."""

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


class _Vectorized:
    """Synchronous list-of-envs vector env (baselines/racing/environment/environment.py:5).  For large
    batches use racing_dreamer_amd.BatchedRaceEnv directly: it steps every env in one kernel launch."""

    def __init__(self, envs):
        self._envs = envs
        self.observation_space = [e.observation_space for e in envs]
        self.action_space = [e.action_space for e in envs]

    def step(self, actions):
        res = [e.step(a) for e, a in zip(self._envs, actions)]
        return tuple(list(x) for x in zip(*res))

    def reset(self, mode: str = "grid"):
        return [e.reset(mode=mode) for e in self._envs]

    def render(self, mode: str = "follow", agents=None, **kw):
        agents = agents or [None] * len(self._envs)
        return [e.render(mode=mode, agent=a, **kw) for e, a in zip(self._envs, agents)]

    def close(self):
        for e in self._envs:
            e.close()


class VectorizedSingleAgentRaceEnv(_Vectorized):
    def __init__(self, scenarios: List[SingleAgentScenario], device: int = 0):
        super().__init__([SingleAgentRaceEnv(s, device=device) for s in scenarios])


class VectorizedMultiAgentRaceEnv(_Vectorized):
    def __init__(self, scenarios: List[MultiAgentScenario], device: int = 0):
        super().__init__([MultiAgentRaceEnv(s, device=device) for s in scenarios])
