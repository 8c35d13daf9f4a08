"""MultiAgentRaceEnv with racecar_gym's dict API, one env on the MI355X (B = 1 view of BatchedRaceEnv).

Contract taken from the reference's call sites:
  * ``step({id: {'motor': f, 'steering': f}}) -> (obs, reward, done, state)``, each keyed by agent id;
    the 4th value is the per-agent state with ``pose`` (6, yaw last), ``velocity`` (6), ``lap`` (first lap 1),
    ``progress``, ``time``, ``wrong_way``, ``wall_collision``   (dreamer/wrappers.py:62-69,217-219,395)
  * ``reset(mode='grid'|'random'|'random_ball') -> obs``        (dreamer/wrappers.py:72,91-92)
  * ``observation_space`` / ``action_space`` as nested Dict spaces, no 'speed' key  (dreamer/wrappers.py:42-60)
  * ``scenario`` attribute, ``render(mode=..., agent=...)``, ``close()``           (dreamer/wrappers.py:30-32,182)
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np

from racing_dreamer_amd import spec
from racing_dreamer_amd.batched_env import BatchedRaceEnv

from .. import tasks as task_registry
from .._spaces import Box, EnvBase, Dict as DictSpace
from .scenarios import MultiAgentScenario

# The device backend.  Always the HIP env; tests substitute a recording/oracle double through this name.
_BACKEND = BatchedRaceEnv

_SENSOR_SPACES = {
    "lidar": lambda: Box(0.0, spec.MAX_RANGE, (spec.N_BEAMS,), np.float64),          # finite bounds: single_agent.py:78-81
    "pose": lambda: Box(-100.0, 100.0, (6,), np.float64),
    "velocity": lambda: Box(-10.0, 10.0, (6,), np.float64),
    "acceleration": lambda: Box(-1000.0, 1000.0, (6,), np.float64),
}


def scenario_key(scenario: MultiAgentScenario):
    """Scenarios with equal keys can share ONE device handle: same track, same agents (ids, sensors), same tasks."""
    return (scenario.world.track.name,
            tuple((a.id, tuple(a.sensors), a.task_name, tuple(sorted((a.task_params or {}).items()))) for a in scenario.agents))


class RaceCore:
    """`n` envs of one scenario on ONE device handle (BatchedRaceEnv with B = n): one launch and one device-to-host copy
    per step whatever n is.  MultiAgentRaceEnv is its n = 1 view; the vector envs (single_agent_race._Vectorized) put
    every group of same-scenario envs on one core."""

    def __init__(self, scenario: MultiAgentScenario, n: int = 1, device: int = 0, seed: int = 0):
        self.scenario, self.n = scenario, int(n)
        self.ids: List[str] = [a.id for a in scenario.agents]
        if len(self.ids) > 4:
            raise ValueError("at most 4 agents per env")
        self.seed = seed
        main = scenario.agents[0]
        self.host_tasks = [dict() for _ in range(self.n)]          # per env: agent id -> registered Task object
        params = dict(main.task_params)
        builtin = main.task_name in task_registry.BUILTIN_TASKS
        for a in scenario.agents:
            if a.task_name not in task_registry.BUILTIN_TASKS:
                cls = task_registry.get_task(a.task_name)
                if cls is None:
                    raise KeyError(f"task {a.task_name!r} is neither built in nor registered (racecar_gym.register_task)")
                for e in range(self.n):
                    self.host_tasks[e][a.id] = cls(**a.task_params) if a.task_params else cls()
        kw = dict(laps=int(params.get("laps", 10)), time_limit=float(params.get("time_limit", 180.0)),
                  terminate_on_collision=bool(params.get("terminate_on_collision", True)),
                  collision_reward=float(params.get("collision_reward", 0.0)))
        if not builtin:   # a host-side Task object decides reward/done: the device must not end the episode
            kw = dict(laps=2 ** 30, time_limit=3.0e38, terminate_on_collision=False, collision_reward=0.0)
        # per-agent device tasks: e.g. A maximize_progress, B-D n_step_progress {n_steps: 10}
        # (baselines/scenarios/max_progress/columbia.yml:9-10,17-18)
        names = {0: "maximize_progress", 2: "n_step_progress"}
        kw["car_tasks"] = [names[task_registry.BUILTIN_TASKS[a.task_name]] if a.task_name in task_registry.BUILTIN_TASKS
                           else None for a in scenario.agents]
        nsteps = [int(a.task_params["n_steps"]) for a in scenario.agents if "n_steps" in (a.task_params or {})]
        if nsteps:
            kw["n_steps"] = nsteps[0]
        self.env = _BACKEND(scenario.world.track, self.n, len(self.ids), obs_type="lidar", device=device, seed=seed, **kw)
        self.observation_space = DictSpace({a.id: DictSpace({s: _SENSOR_SPACES[s]() for s in a.sensors})
                                            for a in scenario.agents})
        self.action_space = DictSpace({a.id: DictSpace({"motor": Box(-1.0, 1.0, (1,), np.float32),
                                                        "steering": Box(-1.0, 1.0, (1,), np.float32)})
                                       for a in scenario.agents})
        self._act = np.zeros((self.n, len(self.ids), 2), np.float32)

    # ------------------------------------------------------------------ helpers
    def fetch(self):
        """One device-to-host copy; returns (raw host arrays [n, agents, ...], per-env obs dicts, per-env state dicts)."""
        env = self.env
        if hasattr(env, "host_snapshot"):            # one device-to-host copy of the whole (4.5 KB per car) arena
            h = env.host_snapshot()
        else:
            env.sync()
            h = {k: env.views[k].cpu().numpy() for k in
                 ("lidar", "pose", "velocity", "acceleration", "reward", "done", "progress", "lap", "time",
                  "wrong_way", "wall_collision", "opponent_collision", "checkpoint")}
        obs_all, state_all = [], []
        for e in range(self.n):
            obs, state = {}, {}
            for i, a in enumerate(self.scenario.agents):
                acc = np.zeros(6)
                acc[0] = h["acceleration"][e, i]
                sensors = {"lidar": h["lidar"][e, i].astype(np.float64), "pose": h["pose"][e, i].astype(np.float64),
                           "velocity": h["velocity"][e, i].astype(np.float64), "acceleration": acc}
                obs[a.id] = {s: sensors[s] for s in a.sensors}
                state[a.id] = {
                    "pose": sensors["pose"], "velocity": sensors["velocity"], "acceleration": acc,
                    "lap": int(h["lap"][e, i]), "progress": float(h["progress"][e, i]), "time": float(h["time"][e, i]),
                    "wrong_way": bool(h["wrong_way"][e, i]), "wall_collision": bool(h["wall_collision"][e, i]),
                    "opponent_collisions": [b for j, b in enumerate(self.ids)
                                            if j != i and h["opponent_collision"][e, i] and h["opponent_collision"][e, j]],
                    "checkpoint": int(h["checkpoint"][e, i]),
                }
            obs_all.append(obs)
            state_all.append(state)
        return h, obs_all, state_all

    def step(self, actions: List[Dict]):
        """actions: one {agent id: {'motor', 'steering'}} dict per env.  Returns per-env lists (obs, rewards, dones, states)."""
        import torch
        for e, action in enumerate(actions):
            for i, aid in enumerate(self.ids):
                self._act[e, i, 0] = float(np.asarray(action[aid]["motor"]).reshape(-1)[0])
                self._act[e, i, 1] = float(np.asarray(action[aid]["steering"]).reshape(-1)[0])
        self.env.step(torch.from_numpy(self._act).to(self.env.device), repeat=1)
        h, obs, state = self.fetch()
        rewards = [{aid: float(h["reward"][e, i]) for i, aid in enumerate(self.ids)} for e in range(self.n)]
        dones = [{aid: bool(h["done"][e, i]) for i, aid in enumerate(self.ids)} for e in range(self.n)]
        for e in range(self.n):
            for aid, task in self.host_tasks[e].items():       # e.g. MaximizeSpeed, baselines/.../tasks.py:4-22
                rewards[e][aid] = task.reward(aid, state[e], actions[e][aid])
                dones[e][aid] = task.done(aid, state[e])
        return obs, rewards, dones, state

    def reset(self, mode: str = "grid"):
        for tasks in self.host_tasks:
            for task in tasks.values():
                task.reset()
        self.env.reset(mode=mode, seed=self.seed)
        return self.fetch()[1]

    def close(self):
        self.env.close()


class MultiAgentRaceEnv(EnvBase):
    """(a `gym.Env` when gym is installed: `_spaces.EnvBase`)"""
    metadata = {"render.modes": ["follow", "birds_eye"]}
    reward_range = (-float("inf"), float("inf"))
    spec = None

    def __init__(self, scenario: MultiAgentScenario, device: int = 0, seed: int = 0, _core: RaceCore = None, _slot: int = 0):
        self._scenario = scenario
        self._core = _core if _core is not None else RaceCore(scenario, 1, device=device, seed=seed)
        self._slot = _slot
        self._ids = self._core.ids
        self.observation_space = self._core.observation_space
        self.action_space = self._core.action_space
        self._episode = 0

    @property
    def scenario(self) -> MultiAgentScenario:
        return self._scenario

    @property
    def _env(self):                       # the device handle (tests and tools reach for it)
        return self._core.env

    # ------------------------------------------------------------------ gym-style API
    def step(self, action: Dict):
        obs, rewards, dones, state = self._core.step([action])
        return obs[0], rewards[0], dones[0], state[0]

    def reset(self, mode: str = "grid"):
        self._episode += 1
        return self._core.reset(mode=mode)[0]

    def render(self, mode: str = "follow", agent: str = None, **kwargs):
        """HxWx3 uint8 frame of the view `mode` ('birds_eye' | 'follow') on agent `agent`, as the Render wrapper asks
        for once per video after every step (dreamer/wrappers.py:178-183)."""
        from .rendering import render_view
        state = self._core.fetch()[2][self._slot]
        return render_view(self._scenario.world.track, state, focus=agent or self._ids[0], mode=mode)

    def seed(self, seed=None):
        self._core.seed = 0 if seed is None else int(seed)

    def close(self):
        self._core.close()
