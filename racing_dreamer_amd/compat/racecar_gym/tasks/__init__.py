"""Task registry (racecar_gym.tasks.register_task, ros_agent/helpers/wrappers.py:17;
racecar_gym.register_task, baselines/racing/environment/tasks.py:1,22)."""


class Task:
    """Host-side task interface of racecar_gym: reward(agent_id, state, action), done(agent_id, state), reset()."""

    def reward(self, agent_id, state, action) -> float:
        raise NotImplementedError

    def done(self, agent_id, state) -> bool:
        raise NotImplementedError

    def reset(self):
        pass


# tasks evaluated on the device by the HIP kernels (name -> rc task id)
BUILTIN_TASKS = {"maximize_progress": 0, "max_progress": 0, "n_step_progress": 2}
_registry = {}


def register_task(name: str, task) -> None:
    _registry[name] = task


def get_task(name: str):
    return _registry.get(name)
