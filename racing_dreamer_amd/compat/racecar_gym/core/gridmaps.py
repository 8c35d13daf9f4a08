"""GridMap as the reference's callers use it: `to_pixel(pose) -> (row, col)` and `_map[...]`
(dreamer/wrappers.py:376,396-399), `.map` (dreamer/plotting/plot_trajectories.py:43).  Full source-image
frame, north-up, so the reference's own OccupancyMapObs wrapper works on it unchanged."""
import numpy as np


class GridMap:
    def __init__(self, grid_map: np.ndarray, origin, resolution: float):
        self._map = grid_map
        self._origin = tuple(origin)
        self._resolution = float(resolution)
        self._height, self._width = grid_map.shape[:2]

    @property
    def map(self):
        return self._map

    def to_pixel(self, position):
        x, y = position[0], position[1]
        row = int(self._height - (y - self._origin[1]) / self._resolution)
        col = int((x - self._origin[0]) / self._resolution)
        return row, col

    def to_meter(self, px, py):
        return ((px + 0.5) * self._resolution + self._origin[0],
                (self._height - py - 0.5) * self._resolution + self._origin[1])

    def get_value(self, position):
        r, c = self.to_pixel(position)
        if 0 <= r < self._height and 0 <= c < self._width:
            return self._map[r, c]
        return 0


def full_frame(track, cropped: np.ndarray, fill=0) -> np.ndarray:
    """Embed a cropped south-up grid of `track` into the source image frame (north-up)."""
    r0, c0, fh, fw = track.crop
    full = np.full((fh, fw), fill, dtype=cropped.dtype)
    full[r0:r0 + track.height, c0:c0 + track.width] = cropped[::-1]
    return full


def full_frame_origin(track):
    r0, c0, fh, fw = track.crop
    res = track.resolution
    return (track.origin[0] - c0 * res, track.origin[1] - (fh - (r0 + track.height)) * res)
