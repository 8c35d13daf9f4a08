"""gym.spaces if gym is installed, else minimal Box/Dict stand-ins with the attributes the reference's
wrappers read (.low, .high, .shape, .dtype, .spaces, [], .sample())."""
import numpy as np

try:                                    # pragma: no cover - depends on the host environment
    from gym.spaces import Box, Dict     # type: ignore
    HAVE_GYM = True
except Exception:                        # gym is not installed in the build container
    HAVE_GYM = False

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
