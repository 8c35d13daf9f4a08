"""racecar_gym.bullet.configs.SceneConfig as dreamer/plotting/plot_trajectories.py:6,20-25 uses it: an attribute tree loaded from
a scene YAML - `config.load(file)`, then `config.sdf`, `config.map.maps`, `config.map.starting_grid` (paths relative to the
file until passed through `resolve_path`), `config.map.origin`, `config.map.resolution`.  The scene files this build can
write for its compiled tracks (`racing_dreamer_amd.track_compiler.export_scene`) have exactly those keys:

    name: austria
    sdf: austria.sdf                     # (a name only: this build has no 3-D scene)
    map:
      maps: maps/maps.npz                # generate-costmap.py's keys (track_compiler.export_maps_npz)
      starting_grid: maps/starting_grid.npz
      resolution: 0.05
      origin: [-50.0, -50.0, 0.0]
"""
import yaml


class _Node:
    """Nested attribute access over a mapping (unknown attributes read None, as a config class with defaults would)."""

    def __init__(self, data=None):
        for k, v in (data or {}).items():
            setattr(self, k, _Node(v) if isinstance(v, dict) else v)

    def __getattr__(self, name):              # only called for attributes that are not set
        if name.startswith("__"):
            raise AttributeError(name)
        return None

    def as_dict(self):
        return {k: (v.as_dict() if isinstance(v, _Node) else v) for k, v in vars(self).items()}


class MapConfig(_Node):
    pass


class SceneConfig(_Node):
    def __init__(self, name: str = "", **kw):
        super().__init__(dict(name=name, **kw))
        if not isinstance(getattr(self, "map", None), _Node):
            self.map = MapConfig()

    def load(self, file: str) -> "SceneConfig":
        with open(file) as f:
            data = yaml.safe_load(f) or {}
        for k, v in data.items():
            setattr(self, k, MapConfig(v) if k == "map" and isinstance(v, dict) else (_Node(v) if isinstance(v, dict) else v))
        return self
