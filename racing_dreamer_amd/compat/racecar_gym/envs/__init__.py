from .multi_agent_race import MultiAgentRaceEnv
from .scenarios import MultiAgentScenario, SingleAgentScenario
from .single_agent_race import (ChangingTrackMultiAgentRaceEnv, ChangingTrackSingleAgentRaceEnv, SingleAgentRaceEnv,
                                VectorizedMultiAgentRaceEnv, VectorizedSingleAgentRaceEnv)

# dreamer/wrappers.py:5 imports the scenario class from the env module
from . import multi_agent_race as _mar
_mar.MultiAgentScenario = MultiAgentScenario

__all__ = ["MultiAgentRaceEnv", "MultiAgentScenario", "SingleAgentScenario", "SingleAgentRaceEnv",
           "ChangingTrackSingleAgentRaceEnv", "ChangingTrackMultiAgentRaceEnv", "VectorizedSingleAgentRaceEnv",
           "VectorizedMultiAgentRaceEnv"]
