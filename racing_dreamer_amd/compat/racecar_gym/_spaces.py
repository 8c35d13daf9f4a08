"""gym.spaces and gym.Env if gym is installed, else minimal Box/Dict/Env stand-ins with the attributes the reference's
wrappers read (.low, .high, .shape, .dtype, .spaces, [], .sample(); .unwrapped, .reward_range, .spec, .metadata).

With gym present the shim's env classes ARE `gym.Env`s (`EnvBase`): the reference hands them to `gym.wrappers.TimeLimit` /
`FilterObservation` and to SB3 (baselines/racing/experiments/sb3/sb_experiment.py:42-64,97-112), whose `check_env` / vec-env
code tests `isinstance(env, gym.Env)`."""
import numpy as np

try:                                    # pragma: no cover - depends on the host environment
    import gym                           # type: ignore
    EnvBase = gym.Env                    # (first: a `gym` without Env - a names-only stub - takes the stand-ins below whole)
    from gym.spaces import Box, Dict     # type: ignore
    HAVE_GYM = True
except Exception:                        # gym is not installed in the build container
    HAVE_GYM = False

    class EnvBase:
        """What old gym's `gym.Env` gives a subclass besides the abstract methods."""
        metadata = {"render.modes": []}
        reward_range = (-float("inf"), float("inf"))
        spec = None
        action_space = None
        observation_space = None

        @property
        def unwrapped(self):
            return self

        def seed(self, seed=None):
            return

        def close(self):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *args):
            self.close()
            return False

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            if shape is None:
                shape = np.asarray(low).shape
            self.low = np.full(shape, low, dtype=dtype) if np.isscalar(low) else np.asarray(low, dtype=dtype).reshape(shape)
            self.high = np.full(shape, high, dtype=dtype) if np.isscalar(high) else np.asarray(high, dtype=dtype).reshape(shape)
            self.shape, self.dtype = tuple(shape), np.dtype(dtype)
            self._rng = np.random.default_rng()

        def sample(self):
            return self._rng.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high)

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    class Dict:
        def __init__(self, spaces):
            self.spaces = dict(sorted(spaces.items())) if isinstance(spaces, dict) else dict(spaces)

        def __getitem__(self, k):
            return self.spaces[k]

        def keys(self):
            return self.spaces.keys()

        def contains(self, x):
            return isinstance(x, dict) and set(x) == set(self.spaces) and all(
                self.spaces[k].contains(v) for k, v in x.items())

        __contains__ = contains

        def sample(self):
            return {k: s.sample() for k, s in self.spaces.items()}

        def __repr__(self):
            return f"Dict({self.spaces})"
