"""racecar_gym.bullet.providers.resolve_path (dreamer/plotting/plot_trajectories.py:7,22-24): a path given relative to a
scene file, resolved against that file's directory."""
import os


def resolve_path(file: str, relative_path: str) -> str:
    if relative_path is None:
        return None
    if os.path.isabs(relative_path):
        return relative_path
    return os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(file)), relative_path))
