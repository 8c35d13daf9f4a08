"""Test double: the surface of BatchedRaceEnv that the racecar_gym shim uses, served by the CPU oracle.
Lets the shim (and the reference's own wrapper stack on top of it) run in the GPU-less build container."""
import numpy as np
import torch

from oracle import c_oracle
from oracle import racecar_oracle as ro
from racing_dreamer_amd import spec


class OracleBackend:
    def __init__(self, track, num_envs, cars_per_env=1, obs_type="lidar", device=0, seed=0, laps=10,
                 time_limit=180.0, terminate_on_collision=True, collision_reward=-1.0, car_tasks=None, n_steps=10, **kw):
        cfg = ro.OracleConfig(num_envs=num_envs, cars_per_env=cars_per_env, laps=laps, time_limit=time_limit,
                              terminate_on_collision=terminate_on_collision, collision_reward=collision_reward,
                              render_occupancy=(obs_type == "lidar_occupancy"), n_steps=n_steps,
                              car_tasks=None if car_tasks is None else [
                                  -1 if t is None else {"maximize_progress": 0, "max_speed": 1, "n_step_progress": 2}[t]
                                  for t in car_tasks])
        self.env = c_oracle.COracleEnv(track.occ, track.drivable, track.progress, track.centerline, track.origin,
                                       track.resolution, cfg)
        self.B, self.A, self.seed = num_envs, cars_per_env, seed
        self.device = torch.device("cpu")
        self.views = {}

    def _publish(self, out):
        for k, v in out.items():
            v = np.asarray(v)
            self.views[k] = torch.from_numpy(v.reshape(self.B, self.A, *v.shape[1:]).copy())

    def reset(self, mask=None, mode="grid", seed=None):
        if seed is not None:
            self.seed = seed
        self._publish(self.env.reset(mask=mask, mode=spec.RESET_MODES[mode], seed=self.seed))
        return self.views

    def step(self, actions, repeat=1):
        self._publish(self.env.step(actions.cpu().numpy().reshape(-1, 2), repeat=repeat))
        return self.views

    def sync(self):
        pass

    def close(self):
        pass
