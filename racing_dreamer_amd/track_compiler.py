"""Offline track compiler: ROS map_server image + yaml  ->  packed track asset (.npz).

This is the "step before" the hot path.  It restates, with
NumPy/SciPy only, what the reference's costmap generator computes
(``docs/maps/costmaps/generate-costmap.py``):

* binarisation ``gray / max > occupied_thresh``               (generate-costmap.py:39-43)
* start pixel from the world start position, including the reference's quirk of
  flipping y with ``image.shape[1]`` (the *width*)            (generate-costmap.py:49-52)
* finish-line blocking one column behind the start pixel      (generate-costmap.py:151-163)
* 8-connected breadth-first distance from the start pixel, one unit per 3x3
  dilation; finish-line pixels get the final counter value    (generate-costmap.py:196-220)
* ``drivable_area = reached | finish_line``                    (generate-costmap.py:223)
* progress = distance * resolution / max                      (generate-costmap.py:221-222)
* obstacle distance = EDT(drivable_area) * resolution / max   (generate-costmap.py:380-382)

Deliberately not restated: the hard-coded ``binary_image[987, 1294] = 0`` edit
(generate-costmap.py:46, a Treitlstrasse_3-U_v3 tweak that raises IndexError on the
350x435 columbia map), the smoothed/eroded "distance to target" layers and the
race-line spline (not consumed by the simulator path).

The asset stores *integers* (occupancy bits, BFS step counts, squared EDT) so the
fp32 grids the device consumes are re-derived bit-identically at load time
(`racing_dreamer_amd.track_assets`).  The grid is cropped to the drivable area's
bounding box plus a margin and stored south-up: cell (ix, iy) covers
``x in [ox + ix*res, ox + (ix+1)*res)``, ``y in [oy + iy*res, ...)``.

It needs the map images, which live in the reference checkout; the compiled
assets under ``racing_dreamer_amd/tracks/`` are committed, so nothing here runs
on the GPU box.
"""
from __future__ import annotations

import argparse
import os
from dataclasses import dataclass

import numpy as np
import yaml
from PIL import Image
from scipy import ndimage

# track name (as used by the reference's scenario files) -> map yaml basename
# (docs/maps/README.md:23,27-30; dreamer/scenarios/max_progress/*.yml world.name)
# columbia (round 5): the authors' own drawing of the F1TENTH Columbia track, `columbia_small` - NOT `columbia.pgm`, the raw
# f1tenth_simulator map of the same name (an open blob of free space around a thin divider, 24.3 m round its inner edge, on which no
# shipped agent drives).  Evidence: the published progress figures for columbia (2.0 - 2.2 laps in 40 s for every
# method, dreamer/plotting/structs.py:23-28) are 3.0 - 3.4 m/s on columbia_small's 61.2 m and an absurd 1.3 m/s on the blob; both
# shipped Dreamer agents lap columbia_small zero-shot (2.4 - 2.5 laps) and hit a wall in every episode on the blob; and
# columbia_small carries the start-line markings (chequered flags, arrow at world (0, 0)) of the authors' other scene maps
# (f1_aut, f1_esp, Treitlstrasse_3-U_*).  The raw map stays available as `columbia_slam`.
TRACK_TO_MAP = {
    "columbia": "columbia_small",
    "austria": "f1_aut",
    "barcelona": "f1_esp",
    "gbr": "f1_gbr",
    "treitlstrasse_v2": "Treitlstrasse_3-U_v2",
}

# maps whose yaml basename is now another track's name: stored under these asset names
MAP_ASSET_NAME = {"columbia": "columbia_slam"}

CROP_MARGIN = 16          # cells of context kept around the drivable bbox (0.8 m)
CENTERLINE_BIN = 2        # BFS steps per centerline bin (0.1 m of arc)
CENTERLINE_SMOOTH = 9     # circular moving-average window (bins)
CENTERLINE_TANGENT = 5    # heading from bins k-5 .. k+5
SPAWN_MIN_CLEARANCE = 0.25  # metres


def load_gray(image_path: str) -> np.ndarray:
    """Grey image as skimage.io.imread(as_gray=True) yields it (generate-costmap.py:38).

    2-D images pass through unchanged; RGB(A) goes rgba2rgb (white background) then
    rgb2gray with the ITU-R 709 weights skimage uses.
    """
    a = np.asarray(Image.open(image_path))
    if a.ndim == 2:
        return a.astype(np.float64)
    a = a.astype(np.float64) / 255.0
    if a.shape[2] == 4:
        alpha = a[..., 3:4]
        rgb = a[..., :3] * alpha + (1.0 - alpha)
    else:
        rgb = a[..., :3]
    return rgb @ np.array([0.2125, 0.7154, 0.0721])


def start_pixel(shape, origin, resolution, world_start=(0.0, 0.0)):
    """(col, row) of the start position, with the reference's width-flip quirk."""
    g = (np.asarray(world_start, np.float64) - np.asarray(origin[:2], np.float64)) / resolution
    g[1] = shape[1] - g[1] - 1          # sic: shape[1], generate-costmap.py:51
    g = g.astype(int)
    return int(g[0]), int(g[1])


def bfs_from_start(free: np.ndarray, start_col: int, start_row: int, forward: bool = True):
    """Finish-line blocking + 8-connected BFS (generate-costmap.py:131-224; `forward=False` blocks the column in
    front of the start pixel instead of the one behind it, :159-161, which yields the distance TO the line).

    Returns (steps int32 [-1 where unreached], finish_line bool, drivable bool, n_iter).
    ``steps`` of finish-line pixels is the final counter value, as in the reference.
    """
    free = free.copy()
    h, w = free.shape
    finish = np.zeros_like(free)

    def block(row, col, step):
        while 0 <= row < h and free[row, col]:
            free[row, col] = False
            finish[row, col] = True
            row += step

    col = start_col - 1 if forward else start_col + 1
    block(start_row, col, +1)
    block(start_row - 1, col, -1)

    steps = np.full(free.shape, -1, np.int32)
    reached = np.zeros_like(free)
    reached[start_row, start_col] = True
    steps[start_row, start_col] = 0
    frontier = np.array([[start_row, start_col]], np.int64)
    offs = np.array([(dr, dc) for dr in (-1, 0, 1) for dc in (-1, 0, 1) if dr or dc], np.int64)
    cur = 0
    while True:
        cur += 1
        cand = (frontier[:, None, :] + offs[None, :, :]).reshape(-1, 2)
        ok = (cand[:, 0] >= 0) & (cand[:, 0] < h) & (cand[:, 1] >= 0) & (cand[:, 1] < w)
        cand = cand[ok]
        keep = free[cand[:, 0], cand[:, 1]] & ~reached[cand[:, 0], cand[:, 1]]
        cand = cand[keep]
        if len(cand) == 0:
            break
        flat = np.unique(cand[:, 0] * w + cand[:, 1])
        frontier = np.stack([flat // w, flat % w], axis=1)
        reached[frontier[:, 0], frontier[:, 1]] = True
        steps[frontier[:, 0], frontier[:, 1]] = cur
    steps[finish] = cur
    drivable = reached | finish
    return steps, finish, drivable, cur


@dataclass
class CompiledTrack:
    name: str
    map_name: str
    resolution: float
    origin: np.ndarray        # world (x, y) of the corner of cropped cell (0, 0)
    occ: np.ndarray           # bool [H, W], south-up, True = occupied
    drivable: np.ndarray      # bool [H, W]
    steps: np.ndarray         # int32 [H, W], -1 outside drivable
    max_steps: int
    edt_sq: np.ndarray        # int32 [H, W] squared EDT in cells (0 outside drivable)
    crop: np.ndarray          # (row0, col0, full_h, full_w) of the crop in the source image
    start_px: np.ndarray      # (col, row) in the source image
    centerline: np.ndarray    # float32 [n, 4] = x, y, heading, progress


def _circular_mean(a: np.ndarray, win: int) -> np.ndarray:
    k = win // 2
    acc = np.zeros_like(a, dtype=np.float64)
    for s in range(-k, k + 1):
        acc += np.roll(a, s, axis=0)
    return acc / (2 * k + 1)


def build_centerline(steps, drivable, edt_sq, max_steps, resolution, origin):
    """Spawn/centerline table: one pose per 0.1 m of BFS arc (build's own).

    Per bin the most central drivable cell (max EDT, lowest flat index on ties), positions
    smoothed circularly, heading = direction of increasing progress.
    """
    h, w = steps.shape
    nb = max_steps // CENTERLINE_BIN
    flat_steps = steps.ravel()
    sel = np.nonzero((flat_steps >= 0) & (flat_steps < nb * CENTERLINE_BIN))[0]
    bins = flat_steps[sel] // CENTERLINE_BIN
    e = edt_sq.ravel()[sel]
    order = np.lexsort((sel, -e, bins))          # per bin: max edt first, then lowest index
    first = np.ones(len(order), bool)
    first[1:] = bins[order][1:] != bins[order][:-1]
    best = sel[order[first]]
    present = bins[order[first]]
    assert len(present) == nb and np.all(present == np.arange(nb)), "empty centerline bin"
    cy, cx = best // w, best % w
    raw = np.stack([cx + 0.5, cy + 0.5], axis=1)          # cell units, cell centres
    sm = _circular_mean(raw, CENTERLINE_SMOOTH)
    # keep the smoothed point only where it stays clear of walls
    ci = np.clip(np.floor(sm).astype(int), 0, [w - 1, h - 1])
    clear = np.sqrt(edt_sq[ci[:, 1], ci[:, 0]]) * resolution
    pts = np.where((clear >= SPAWN_MIN_CLEARANCE)[:, None] & drivable[ci[:, 1], ci[:, 0]][:, None], sm, raw)
    fwd = np.roll(pts, -CENTERLINE_TANGENT, axis=0) - np.roll(pts, CENTERLINE_TANGENT, axis=0)
    heading = np.arctan2(fwd[:, 1], fwd[:, 0])
    xy = np.asarray(origin, np.float64)[None, :] + pts * resolution
    prog = (np.arange(nb) * CENTERLINE_BIN + CENTERLINE_BIN * 0.5) / max_steps
    return np.concatenate([xy, heading[:, None], prog[:, None]], axis=1).astype(np.float32)


class TrackCompileError(ValueError):
    """The map cannot be compiled with this start position - the reference's generator fails on it too."""


def _load_map(name: str, maps_dir: str, world_start):
    by_asset = {v: k for k, v in MAP_ASSET_NAME.items()}
    map_name = TRACK_TO_MAP.get(name, by_asset.get(name, name))
    with open(os.path.join(maps_dir, map_name + ".yaml")) as f:
        props = yaml.safe_load(f)
    res = float(props["resolution"])
    image_path = os.path.join(maps_dir, props["image"])
    if not os.path.exists(image_path):
        raise TrackCompileError(f"{name}: image {props['image']} is not in the checkout (.MISSING_LARGE_BLOBS)")
    gray = load_gray(image_path)
    norm = gray / np.amax(gray)
    free = norm > props["occupied_thresh"]
    sc, sr = start_pixel(gray.shape, props["origin"], res, world_start)
    h, w = free.shape
    if not (1 <= sc < w - 1 and 1 <= sr < h):
        raise TrackCompileError(f"{name}: start position {tuple(world_start)} maps to pixel ({sc}, {sr}) outside the "
                                f"{h}x{w} image (generate-costmap.py:49-52 indexes out of range there too); pass a start position")
    if not free[sr, sc]:
        raise TrackCompileError(f"{name}: start pixel ({sc}, {sr}) of world {tuple(world_start)} is occupied; pass a start position")
    return map_name, props, res, gray, free, sc, sr


def compile_track(name: str, maps_dir: str, world_start=(0.0, 0.0)) -> CompiledTrack:
    map_name, props, res, gray, free, sc, sr = _load_map(name, maps_dir, world_start)
    steps, finish, drivable, max_steps = bfs_from_start(free, sc, sr)
    if max_steps < 4 * CENTERLINE_BIN * CENTERLINE_TANGENT:
        raise TrackCompileError(f"{name}: only {max_steps} cells reachable from the start pixel - not a closed track")
    edt = ndimage.distance_transform_edt(drivable)
    edt_sq = np.rint(edt * edt).astype(np.int32)

    rows, cols = np.nonzero(drivable)
    fh, fw = free.shape
    r0, r1 = max(rows.min() - CROP_MARGIN, 0), min(rows.max() + CROP_MARGIN + 1, fh)
    c0, c1 = max(cols.min() - CROP_MARGIN, 0), min(cols.max() + CROP_MARGIN + 1, fw)

    def south_up(a):
        return np.ascontiguousarray(a[r0:r1, c0:c1][::-1])

    occ = south_up(~free)
    drv = south_up(drivable)
    st = south_up(steps)
    esq = south_up(edt_sq)
    # image row r maps to world y = oy + (fh - 1 - r) * res  (ROS map_server: origin = lower-left pixel)
    origin = np.array([props["origin"][0] + c0 * res, props["origin"][1] + (fh - r1) * res], np.float64)
    cl = build_centerline(st, drv, esq, max_steps, res, origin)
    return CompiledTrack(name, map_name, res, origin, occ, drv, st, max_steps, esq,
                         np.array([r0, c0, fh, fw], np.int32), np.array([sc, sr], np.int32), cl)


def export_scene(name: str, maps_dir: str, out_dir: str, world_start=(0.0, 0.0)) -> str:
    """Write a racecar_gym-style scene directory for a track - `<out_dir>/<name>/<name>.yml` with the keys
    `SceneConfig.load` consumers read (dreamer/plotting/plot_trajectories.py:17-37: sdf, map.maps, map.starting_grid,
    map.origin, map.resolution), `maps/maps.npz` (export_maps_npz) and `maps/starting_grid.npz` (the grid-mode start poses
    of this build: x, y, yaw per slot).  Returns the yml path."""
    import yaml
    from .track_assets import load_track
    scene = os.path.join(out_dir, name)
    os.makedirs(os.path.join(scene, "maps"), exist_ok=True)
    export_maps_npz(name, maps_dir, os.path.join(scene, "maps", "maps.npz"), world_start)
    map_name, props, res, gray, free, sc, sr = _load_map(name, maps_dir, world_start)
    t = load_track(name)
    cl = np.asarray(t.centerline, np.float64)
    slots = [(8 + 12 * k) % len(cl) for k in range(4)]                  # grid mode: 0.8 m after the line, 1.2 m apart
    np.savez(os.path.join(scene, "maps", "starting_grid.npz"), data=cl[slots][:, :3])
    path = os.path.join(scene, f"{name}.yml")
    with open(path, "w") as f:
        yaml.safe_dump({"name": name, "sdf": f"{name}.sdf",
                        "map": {"maps": "maps/maps.npz", "starting_grid": "maps/starting_grid.npz",
                                "resolution": float(res), "origin": [float(v) for v in props["origin"]]}}, f)
    return path


def export_maps_npz(name: str, maps_dir: str, out_path: str, world_start=(0.0, 0.0)) -> dict:
    """Write the racecar_gym-style `maps.npz` of a track: the keys and value conventions of the reference's costmap
    generator (generate-costmap.py:410-425), full source-image frame, row 0 = top of the image - what
    `GridMap(np.load(maps)[key], origin, resolution)` consumers read (dreamer/plotting/plot_trajectories.py:26-37:
    'norm_distance_from_start', 'norm_distance_to_obstacle', 'drivable_area').

      properties ................. [world start x, y, grid start col, row, image centre 0, 1, occupied_thresh,
                                   min(image), max(image), image rows, cols, resolution]      (:410-421)
      drivable_area .............. bool: reached from the start pixel, finish line included    (:223)
      norm_distance_from_start ... BFS distance x resolution / its maximum, 0 outside          (:198-222)
      norm_distance_to_obstacle .. EDT(drivable_area) x resolution / its maximum               (:380-382)
      norm_distance_to ........... the same BFS run against the driving direction (:374-375 forward_direction=False);
                                   the reference adds Gaussian-blurred copies of it when `use_blurred_factor` is set
                                   (:262-263), an attribute its __init__ never defines - the plain distance is written.
    Returns the dict that was saved."""
    map_name, props, res, gray, free, sc, sr = _load_map(name, maps_dir, world_start)
    steps, finish, drivable, n_fwd = bfs_from_start(free, sc, sr, forward=True)
    back, _, _, n_back = bfs_from_start(free, sc, sr, forward=False)
    dist_from = np.where(steps >= 0, steps, 0).astype(np.float64) * res
    dist_to = np.where(back >= 0, back, 0).astype(np.float64) * res
    edt = ndimage.distance_transform_edt(drivable).astype(np.float64) * res
    center = (np.asarray(gray.shape) + np.asarray(props["origin"][:2], np.float64) / res).astype(int)
    data = dict(
        properties=np.array([world_start[0], world_start[1], sc, sr, center[0], center[1], props["occupied_thresh"],
                             np.amin(gray), np.amax(gray), gray.shape[0], gray.shape[1], res], np.float64),
        drivable_area=drivable,
        norm_distance_from_start=dist_from / np.amax(dist_from),
        norm_distance_to=dist_to / np.amax(dist_to),
        norm_distance_to_obstacle=edt / np.amax(edt))
    np.savez(out_path, **data)
    return data


def save_track(t: CompiledTrack, out_path: str) -> None:
    h, w = t.occ.shape
    steps16 = np.where(t.steps < 0, 0xFFFF, t.steps).astype(np.uint16)
    assert t.max_steps < 0xFFFF and t.edt_sq.max() < 0xFFFF
    np.savez_compressed(
        out_path,
        name=np.array(t.name), map_name=np.array(t.map_name),
        resolution=np.float64(t.resolution), origin=t.origin,
        shape=np.array([h, w], np.int32),
        occ=np.packbits(t.occ, axis=1, bitorder="little"),
        drivable=np.packbits(t.drivable, axis=1, bitorder="little"),
        steps=steps16, max_steps=np.int32(t.max_steps),
        edt_sq=t.edt_sq.astype(np.uint16),
        crop=t.crop, start_px=t.start_px, centerline=t.centerline,
    )


START_POSITIONS = os.path.join(os.path.dirname(__file__), "tracks", "start_positions.json")


def start_positions() -> dict:
    """Per-map start positions (world x, y) for the maps whose default start - world (0, 0), generate-costmap.py:463-464 - is not
    on the track: what the reference's generator takes as --start_x / --start_y (generate-costmap.py:460-475).  Chosen by
    tools/find_start.py (the first lattice cell in scan order from which the compiled track closes), committed as data."""
    import json
    if not os.path.exists(START_POSITIONS):
        return {}
    with open(START_POSITIONS) as f:
        return {k: tuple(v["start"]) for k, v in json.load(f).items() if "start" in v}


def compile_all(maps_dir: str, out_dir: str) -> dict:
    """Every map yaml of `maps_dir` with the generator's default start position (world (0, 0),
    generate-costmap.py:463-464) - or the one `start_positions()` holds for it: assets for those that compile, the reason for
    those that do not, `index.json` with both.  Maps that already have a track name (TRACK_TO_MAP) are stored under that name."""
    import glob
    import json
    by_map = {v: k for k, v in TRACK_TO_MAP.items()}
    starts = start_positions()
    refused, open_maps = {}, set()
    if os.path.exists(START_POSITIONS):
        with open(START_POSITIONS) as f:
            table = json.load(f)
            refused = {k: v["refused"] for k, v in table.items() if "refused" in v}
            open_maps = {k for k, v in table.items() if v.get("open")}
    index = {}
    for path in sorted(glob.glob(os.path.join(maps_dir, "*.yaml"))):
        map_name = os.path.basename(path)[:-5]
        name = by_map.get(map_name, MAP_ASSET_NAME.get(map_name, map_name))
        if name in refused:
            index[name] = {"map": map_name, "status": "not compiled", "reason": refused[name]}
            continue
        try:
            t = compile_track(name, maps_dir, starts.get(name, (0.0, 0.0)))
        except (TrackCompileError, AssertionError) as e:
            index[name] = {"map": map_name, "status": "not compiled", "reason": str(e) or "empty centre-line bin (open track)"}
            continue
        if t.max_steps >= 0xFFFF or t.edt_sq.max() >= 0xFFFF:
            index[name] = {"map": map_name, "status": "not compiled",
                           "reason": "the start pixel lies in an open area wider than 12.8 m: not a track"}
            continue
        save_track(t, os.path.join(out_dir, name + ".npz"))
        h, w = t.occ.shape
        index[name] = {"map": map_name, "status": "ok", "cells": [int(h), int(w)], "max_steps": int(t.max_steps),
                       "track_length_m": round(t.max_steps * t.resolution, 2)}
        if name in starts:
            index[name]["start_position"] = [float(v) for v in starts[name]]
        if name in open_maps:
            index[name]["open"] = True          # not a loop: the BFS wave does not come round, a lap is never completed
    with open(os.path.join(out_dir, "index.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)
    return index


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--maps", default="/root/reference/docs/maps/maps")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "tracks"))
    ap.add_argument("--all", action="store_true", help="compile every map yaml of --maps (default start position)")
    ap.add_argument("--maps-npz", metavar="PATH", help="also write the racecar_gym-style maps.npz of the (single) track given")
    ap.add_argument("tracks", nargs="*", default=list(TRACK_TO_MAP))
    args = ap.parse_args(argv)
    os.makedirs(args.out, exist_ok=True)
    if args.all:
        for name, e in compile_all(args.maps, args.out).items():
            print(f"{name:38s} {e['status']:13s} {e.get('cells', e.get('reason', ''))}")
        return
    for name in args.tracks:
        t = compile_track(name, args.maps)
        path = os.path.join(args.out, name + ".npz")
        save_track(t, path)
        h, w = t.occ.shape
        print(f"{name:18s} {h}x{w} cells  max_steps={t.max_steps}  centerline={len(t.centerline)}"
              f"  bits={h * ((w + 31) // 32) * 4} B  file={os.path.getsize(path)} B")
    if args.maps_npz:
        export_maps_npz(args.tracks[0], args.maps, args.maps_npz)


if __name__ == "__main__":
    main()
