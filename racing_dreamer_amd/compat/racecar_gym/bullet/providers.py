"""Name-compatibility for dreamer/plotting/plot_trajectories.py:7 (plotting only)."""
import os


def resolve_path(file: str, relative_path: str) -> str:
    return os.path.normpath(os.path.join(os.path.dirname(file), relative_path))
