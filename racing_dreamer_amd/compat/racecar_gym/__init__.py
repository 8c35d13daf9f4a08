"""`racecar_gym`-compatible surface backed by the MI355X batched env (racing_dreamer_amd).

Exports the names the reference imports: racecar_gym.{SingleAgentScenario, Task, register_task},
racecar_gym.envs.*, racecar_gym.envs.multi_agent_race.*, racecar_gym.tasks.*, racecar_gym.core.gridmaps.GridMap,
racecar_gym.bullet.{configs.SceneConfig, providers.resolve_path}.
"""
from .tasks import Task, register_task
from .envs import (ChangingTrackMultiAgentRaceEnv, ChangingTrackSingleAgentRaceEnv, MultiAgentRaceEnv,
                   MultiAgentScenario, SingleAgentRaceEnv, SingleAgentScenario, VectorizedMultiAgentRaceEnv,
                   VectorizedSingleAgentRaceEnv)

__all__ = ["Task", "register_task", "MultiAgentRaceEnv", "MultiAgentScenario", "SingleAgentScenario",
           "SingleAgentRaceEnv", "ChangingTrackSingleAgentRaceEnv", "ChangingTrackMultiAgentRaceEnv",
           "VectorizedSingleAgentRaceEnv", "VectorizedMultiAgentRaceEnv"]
