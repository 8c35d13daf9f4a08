"""racecar_gym.tasks.progress_based.MaximizeProgressTask (ros_agent/helpers/wrappers.py:18): the
parameter holder of the built-in progress task.  Its arithmetic runs on the device
(csrc/racecar_kernels.hip, rc_dynamics_kernel); this class only carries the parameters."""
from . import Task


class MaximizeProgressTask(Task):
    def __init__(self, laps: int = 10, time_limit: float = 180.0, terminate_on_collision: bool = True,
                 delta_progress: float = 0.0, collision_reward: float = 0.0, frame_reward: float = 0.0,
                 progress_reward: float = 100.0, n_min_rays_termination: int = 1080):
        self.laps, self.time_limit = laps, time_limit
        self.terminate_on_collision, self.collision_reward = terminate_on_collision, collision_reward
        self.delta_progress, self.frame_reward, self.progress_reward = delta_progress, frame_reward, progress_reward
