#!/usr/bin/env python3
"""Find a start position for a map whose default one (world (0, 0), generate-costmap.py:463-464) does not lie on the track:
python tools/find_start.py MAP [MAP ...]  (build container only: reads /root/reference/docs/maps/maps).

The reference's generator takes the start on its command line (--start_x / --start_y, generate-costmap.py:460-475) and leaves the
choice to whoever runs it.  This picks one reproducibly: among the free cells whose distance to the nearest wall is 0.3 .. 3 m
(cells a car can stand in, in something corridor-like), sampled on a coarse lattice, the first candidate in scan order whose compiled track CLOSES - the cells just behind
the finish line are reached last, at >= 0.9 of the longest BFS distance: the wave went round -, else the longest open one.  Prints `"name": [x, y]` lines for racing_dreamer_amd/tracks/start_positions.json."""
import os
import sys

import numpy as np
from scipy import ndimage

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from racing_dreamer_amd import track_compiler as tc      # noqa: E402

MAPS = "/root/reference/docs/maps/maps"


def world_of(shape, origin, res, col, row):
    """Inverse of track_compiler.start_pixel (with the generator's width-flip quirk, generate-costmap.py:51)."""
    return origin[0] + (col + 0.5) * res, origin[1] + (shape[1] - 1 - row - 0.5) * res


def closes(steps, finish, sc, sr, max_steps):
    behind = steps[max(sr - 3, 0):sr + 4, max(sc - 3, 0):max(sc - 1, 0)]
    behind = behind[behind >= 0]
    return behind.size > 0 and behind.max() >= 0.9 * max_steps


def find(name, stride=24):
    map_name, props, res, gray, free, _, _ = _load(name)
    edt = ndimage.distance_transform_edt(free) * res
    h, w = free.shape
    best = None
    for row in range(stride // 2, h, stride):
        for col in range(stride // 2, w - 1, stride):
            if not (0.3 <= edt[row, col] <= 3.0):
                continue
            x, y = world_of(gray.shape, props["origin"], res, col, row)
            sc, sr = tc.start_pixel(gray.shape, props["origin"], res, (x, y))
            if (sc, sr) != (col, row) or not (1 <= sc < w - 1 and 1 <= sr < h) or not free[sr, sc]:
                continue
            steps, finish, drivable, max_steps = tc.bfs_from_start(free, sc, sr)
            if max_steps >= 0xFFFF or max_steps < 200:
                continue
            e2 = ndimage.distance_transform_edt(drivable)
            if (e2 * e2).max() >= 0xFFFF:
                continue                                    # an open area wider than 12.8 m: not a track
            closed = closes(steps, finish, sc, sr, max_steps)
            key = (closed, max_steps, -row, -col)
            if best is None or key > best[0]:
                best = (key, (round(float(x), 3), round(float(y), 3)), (sc, sr))
            if closed:
                return best                                 # every start on a closed loop compiles the same loop: the first in scan order
    return best


def _load(name):
    import yaml
    with open(os.path.join(MAPS, name + ".yaml")) as f:
        props = yaml.safe_load(f)
    res = float(props["resolution"])
    gray = tc.load_gray(os.path.join(MAPS, props["image"]))
    free = gray / np.amax(gray) > props["occupied_thresh"]
    return name, props, res, gray, free, 0, 0


if __name__ == "__main__":
    for name in sys.argv[1:]:
        b = find(name)
        if b is None:
            print(f"{name}: no candidate compiles")
        else:
            (closed, length, _, _), xy, px = b
            print(f'"{name}": [{xy[0]}, {xy[1]}],    # pixel {px}, {length} BFS steps, {"closed loop" if closed else "OPEN: the wave does not come round"}', flush=True)
