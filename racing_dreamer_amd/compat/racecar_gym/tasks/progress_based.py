"""racecar_gym.tasks.progress_based.MaximizeProgressTask (ros_agent/helpers/wrappers.py:17-20 registers it under
"maximize_progress"; parameters: dreamer/scenarios/max_progress/columbia.yml:9-10).

The built-in progress task runs on the device (rc_dynamics_kernel, env spec 5 in DESIGN.md §2) - the shim never needs this
class for it.  It is the same law as a host `Task` over the per-agent state dicts the env returns (`lap`, `progress`,
`time`, `wall_collision`, `opponent_collisions`), for callers that evaluate tasks themselves or register a variant:
reward = progress_reward x delta(lap + progress) + collision_reward on contact + frame_reward, done = contact (if
terminate_on_collision) or lap > laps or time > time_limit."""
from . import Task


class MaximizeProgressTask(Task):
    def __init__(self, laps: int = 10, time_limit: float = 180.0, terminate_on_collision: bool = True,
                 delta_progress: float = 0.0, collision_reward: float = 0.0, frame_reward: float = 0.0,
                 progress_reward: float = 100.0, n_min_rays_termination: int = 1080):
        self.laps, self.time_limit = laps, time_limit
        self.terminate_on_collision, self.collision_reward = terminate_on_collision, collision_reward
        self.delta_progress, self.frame_reward, self.progress_reward = delta_progress, frame_reward, progress_reward
        self._last = {}

    @staticmethod
    def _collided(agent_state) -> bool:
        return bool(agent_state["wall_collision"]) or len(agent_state.get("opponent_collisions", ())) > 0

    def reward(self, agent_id, state, action) -> float:
        s = state[agent_id]
        total = float(s["lap"]) + float(s["progress"])
        last = self._last.get(agent_id, total)            # an episode's first step: measured from where the car stands
        self._last[agent_id] = total
        reward = self.frame_reward + (total - last) * self.progress_reward
        if self._collided(s):
            reward += self.collision_reward
        return reward

    def done(self, agent_id, state) -> bool:
        s = state[agent_id]
        return bool((self.terminate_on_collision and self._collided(s)) or s["lap"] > self.laps or s["time"] > self.time_limit)

    def reset(self):
        self._last = {}
