"""Software bird's-eye renderer for eval videos (dreamer/wrappers.py:161-195 Render wrapper,
baselines/racing/experiments/sb3/callbacks.py:111-112).  Host NumPy; not on the hot path."""
import numpy as np

from racing_dreamer_amd import spec

_COLORS = [(66, 135, 245), (220, 50, 47), (238, 200, 40), (200, 60, 200)]


def render_birds_eye(track, state, focus, follow=False, size=(240, 320)):
    h, w = size
    occ, drv = track.occ, track.drivable
    base = np.full((track.height, track.width, 3), 235, np.uint8)
    base[drv] = (255, 255, 255)
    base[occ] = (40, 40, 40)
    for k, (aid, st) in enumerate(state.items()):
        x, y, yaw = st["pose"][0], st["pose"][1], st["pose"][5]
        c, s = np.cos(yaw), np.sin(yaw)
        for fx in np.linspace(spec.X_REAR, spec.X_FRONT, 12):
            for fy in np.linspace(-spec.HALF_W, spec.HALF_W, 7):
                ix = int((x + fx * c - fy * s - track.origin[0]) / track.resolution)
                iy = int((y + fx * s + fy * c - track.origin[1]) / track.resolution)
                if 0 <= ix < track.width and 0 <= iy < track.height:
                    base[iy, ix] = _COLORS[k % len(_COLORS)]
    img = base[::-1]                                           # north-up
    if follow and focus in state:
        x, y = state[focus]["pose"][0], state[focus]["pose"][1]
        col = int((x - track.origin[0]) / track.resolution)
        row = track.height - 1 - int((y - track.origin[1]) / track.resolution)
        pad = np.pad(img, ((100, 100), (100, 100), (0, 0)), constant_values=235)
        img = pad[row:row + 200, col:col + 200]
    ys = (np.arange(h) * img.shape[0] / h).astype(int)
    xs = (np.arange(w) * img.shape[1] / w).astype(int)
    return np.ascontiguousarray(img[ys][:, xs])
