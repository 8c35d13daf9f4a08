"""Name-compatibility for dreamer/plotting/plot_trajectories.py:6 (plotting only, out of the hot path)."""


class SceneConfig:
    def __init__(self, name: str = "", **kw):
        self.name = name
        self.__dict__.update(kw)
