"""Software renderer for the eval videos of the reference: `env.render(mode='birds_eye' | 'follow', agent=id)` as the
`Render` wrapper calls it once per mode and agent after every step (dreamer/wrappers.py:161-195), and
`eval_env.render(mode='birds_eye')` of the baselines' evaluation callback
(baselines/racing/experiments/sb3/callbacks.py:111-112, acme/experiment.py:151).  Host NumPy, not on the hot path.

Upstream draws these views with PyBullet cameras (a camera above the agent / a chase camera behind it); here both are
top-down rasterisations of the same 2-D world the simulator steps, by inverse mapping (one nearest-cell tap of the
track grids per output pixel, then the cars as filled oriented rectangles):

  birds_eye   north-up window of 16 m x 12 m centred on the agent
  follow      ego-aligned window of 8 m x 6 m: the agent's heading points up, the agent sits a quarter of the way up
              from the bottom edge (what a chase camera shows: mostly the road ahead)

Frames are uint8 [240, 320, 3].  Colours: drivable area white, walls dark, outside light grey; agent k in colour k
(blue, red, yellow, magenta: the vehicle colours of baselines/scenarios/max_progress/columbia.yml), the focus agent
with a darker heading mark on its front third.
"""
import numpy as np

from racing_dreamer_amd import spec

FRAME = (240, 320)
COLORS = np.array([(66, 135, 245), (220, 50, 47), (238, 200, 40), (200, 60, 200)], np.uint8)
_VIEWS = {"birds_eye": dict(height_m=12.0, ahead=0.5), "follow": dict(height_m=6.0, ahead=0.75)}


def render_view(track, state, focus, mode="birds_eye", size=FRAME):
    """`state`: {agent id: {'pose': (x, y, z, roll, pitch, yaw), ...}} as returned by the shim's step()."""
    if mode not in _VIEWS:
        raise ValueError(f"render mode must be one of {sorted(_VIEWS)}, got {mode!r}")
    h, w = size
    ids = list(state)
    if focus not in state:
        focus = ids[0]
    fx, fy, fyaw = (float(state[focus]["pose"][k]) for k in (0, 1, 5))
    v = _VIEWS[mode]
    mpp = v["height_m"] / h                                        # metres per pixel
    # view frame: u to the right, w up on the screen; the agent sits at screen (w / 2, ahead * h from the top)
    up = np.array([0.0, 1.0]) if mode == "birds_eye" else np.array([np.cos(fyaw), np.sin(fyaw)])
    right = np.array([up[1], -up[0]])
    cols = (np.arange(w) + 0.5 - w / 2) * mpp
    rows = (v["ahead"] * h - (np.arange(h) + 0.5)) * mpp
    wx = fx + cols[None, :] * right[0] + rows[:, None] * up[0]
    wy = fy + cols[None, :] * right[1] + rows[:, None] * up[1]
    ix = np.floor((wx - track.origin[0]) / track.resolution).astype(np.int64)
    iy = np.floor((wy - track.origin[1]) / track.resolution).astype(np.int64)
    inside = (ix >= 0) & (ix < track.width) & (iy >= 0) & (iy < track.height)
    ixc, iyc = np.clip(ix, 0, track.width - 1), np.clip(iy, 0, track.height - 1)
    img = np.full((h, w, 3), 235, np.uint8)
    img[inside & track.drivable[iyc, ixc]] = (255, 255, 255)
    img[inside & track.occ[iyc, ixc]] = (40, 40, 40)
    for k, aid in enumerate(ids):
        x, y, yaw = (float(state[aid]["pose"][j]) for j in (0, 1, 5))
        c, s = np.cos(yaw), np.sin(yaw)
        bx = (wx - x) * c + (wy - y) * s                           # body frame: +x forward from the rear axle
        by = (wy - y) * c - (wx - x) * s
        body = (bx >= spec.X_REAR) & (bx <= spec.X_FRONT) & (np.abs(by) <= spec.HALF_W)
        img[body] = COLORS[k % len(COLORS)]
        if aid == focus:
            nose = body & (bx >= spec.X_FRONT - (spec.X_FRONT - spec.X_REAR) / 3)
            img[nose] = (COLORS[k % len(COLORS)].astype(np.int32) * 6 // 10).astype(np.uint8)
    return img


def render_birds_eye(track, state, focus, follow=False, size=FRAME):
    """Kept for callers of the first version of this module."""
    return render_view(track, state, focus, "follow" if follow else "birds_eye", size)
