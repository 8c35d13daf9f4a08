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
from .._spaces import Box, Dict as DictSpace
from .scenarios import MultiAgentScenario

# The device backend.  Always the HIP env; tests substitute a recording/oracle double through this name.
_BACKEND = BatchedRaceEnv

_SENSOR_SPACES = {
    "lidar": lambda: Box(0.0, spec.MAX_RANGE, (spec.N_BEAMS,), np.float64),          # finite bounds: single_agent.py:78-81
    "pose": lambda: Box(-100.0, 100.0, (6,), np.float64),
    "velocity": lambda: Box(-10.0, 10.0, (6,), np.float64),
    "acceleration": lambda: Box(-1000.0, 1000.0, (6,), np.float64),
}


class MultiAgentRaceEnv:
    metadata = {"render.modes": ["follow", "birds_eye"]}

    def __init__(self, scenario: MultiAgentScenario, device: int = 0, seed: int = 0):
        self._scenario = scenario
        self._ids: List[str] = [a.id for a in scenario.agents]
        if len(self._ids) > 4:
            raise ValueError("at most 4 agents per env")
        self._seed = seed
        main = scenario.agents[0]
        self._host_tasks = {}
        params = dict(main.task_params)
        builtin = main.task_name in task_registry.BUILTIN_TASKS
        for a in scenario.agents:
            if a.task_name not in task_registry.BUILTIN_TASKS:
                cls = task_registry.get_task(a.task_name)
                if cls is None:
                    raise KeyError(f"task {a.task_name!r} is neither built in nor registered (racecar_gym.register_task)")
                self._host_tasks[a.id] = cls(**a.task_params) if a.task_params else cls()
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
        self._env = _BACKEND(scenario.world.track, 1, len(self._ids), obs_type="lidar", device=device,
                                   seed=seed, **kw)
        self.observation_space = DictSpace({a.id: DictSpace({s: _SENSOR_SPACES[s]() for s in a.sensors})
                                            for a in scenario.agents})
        self.action_space = DictSpace({a.id: DictSpace({"motor": Box(-1.0, 1.0, (1,), np.float32),
                                                        "steering": Box(-1.0, 1.0, (1,), np.float32)})
                                       for a in scenario.agents})
        self._episode = 0
        self._act = np.zeros((1, len(self._ids), 2), np.float32)

    @property
    def scenario(self) -> MultiAgentScenario:
        return self._scenario

    # ------------------------------------------------------------------ helpers
    def _fetch(self):
        env = self._env
        if hasattr(env, "host_snapshot"):            # one device-to-host copy of the whole (4.5 KB per car) arena
            h = {k: v[0] for k, v in env.host_snapshot().items()}
        else:
            env.sync()
            h = {k: env.views[k][0].cpu().numpy() for k in
                 ("lidar", "pose", "velocity", "acceleration", "reward", "done", "progress", "lap", "time",
                  "wrong_way", "wall_collision", "opponent_collision", "checkpoint")}
        obs, state = {}, {}
        for i, a in enumerate(self._scenario.agents):
            acc = np.zeros(6)
            acc[0] = h["acceleration"][i]
            sensors = {"lidar": h["lidar"][i].astype(np.float64), "pose": h["pose"][i].astype(np.float64),
                       "velocity": h["velocity"][i].astype(np.float64), "acceleration": acc}
            obs[a.id] = {s: sensors[s] for s in a.sensors}
            state[a.id] = {
                "pose": sensors["pose"], "velocity": sensors["velocity"], "acceleration": acc,
                "lap": int(h["lap"][i]), "progress": float(h["progress"][i]), "time": float(h["time"][i]),
                "wrong_way": bool(h["wrong_way"][i]), "wall_collision": bool(h["wall_collision"][i]),
                "opponent_collisions": [b for j, b in enumerate(self._ids)
                                        if j != i and h["opponent_collision"][i] and h["opponent_collision"][j]],
                "checkpoint": int(h["checkpoint"][i]),
            }
        return h, obs, state

    # ------------------------------------------------------------------ gym-style API
    def step(self, action: Dict):
        import torch
        for i, aid in enumerate(self._ids):
            self._act[0, i, 0] = float(np.asarray(action[aid]["motor"]).reshape(-1)[0])
            self._act[0, i, 1] = float(np.asarray(action[aid]["steering"]).reshape(-1)[0])
        self._env.step(torch.from_numpy(self._act).to(self._env.device), repeat=1)
        h, obs, state = self._fetch()
        rewards = {aid: float(h["reward"][i]) for i, aid in enumerate(self._ids)}
        dones = {aid: bool(h["done"][i]) for i, aid in enumerate(self._ids)}
        for aid, task in self._host_tasks.items():       # e.g. MaximizeSpeed, baselines/.../tasks.py:4-22
            rewards[aid] = task.reward(aid, state, action[aid])
            dones[aid] = task.done(aid, state)
        return obs, rewards, dones, state

    def reset(self, mode: str = "grid"):
        for task in self._host_tasks.values():
            task.reset()
        self._env.reset(mode=mode, seed=self._seed)
        self._episode += 1
        return self._fetch()[1]

    def render(self, mode: str = "follow", agent: str = None, **kwargs):
        """HxWx3 uint8 frame of the view `mode` ('birds_eye' | 'follow') on agent `agent`, as the Render wrapper asks
        for once per video after every step (dreamer/wrappers.py:178-183)."""
        from .rendering import render_view
        _, _, state = self._fetch()
        return render_view(self._scenario.world.track, state, focus=agent or self._ids[0], mode=mode)

    def seed(self, seed=None):
        self._seed = 0 if seed is None else int(seed)

    def close(self):
        self._env.close()
