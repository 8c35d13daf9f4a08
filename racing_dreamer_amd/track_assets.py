"""Track assets: load a compiled track (.npz) and derive the arrays the device consumes.

A track is:

* ``occ_words``  uint32 [H, pitch]  - occupancy, 1 bit per 0.05 m cell, bit i of word j is
  cell ix = 32*j + i; ``pitch`` is odd so consecutive rows start on different LDS banks
* ``drv_words``  uint32 [H, pitch]  - drivable area (what `lidar_occupancy` renders)
* ``progress``   float32 [H, W]     - ``norm_distance_from_start`` (generate-costmap.py:220-222),
  -1 outside the drivable area
* ``centerline`` float32 [n, 4]     - x, y, heading, progress per 0.1 m of arc (spawn table)

Everything is re-derived from the integers stored in the asset with float64 NumPy, so
host, oracle and device all see the same float32 bits.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from functools import lru_cache

import numpy as np

TRACK_DIR = os.path.join(os.path.dirname(__file__), "tracks")


@dataclass(frozen=True)
class Track:
    name: str
    map_name: str
    resolution: float
    inv_resolution: float
    origin: tuple            # world (x, y) of the corner of cell (0, 0)
    height: int
    width: int
    pitch: int               # uint32 words per bitmap row
    occ_words: np.ndarray
    drv_words: np.ndarray
    progress: np.ndarray
    edt_m: np.ndarray        # float32 [H, W] distance to the nearest non-drivable cell [m]
    centerline: np.ndarray
    max_steps: int
    crop: tuple              # (row0, col0, full_h, full_w) in the source image
    start_px: tuple

    @property
    def occ(self) -> np.ndarray:
        return unpack_words(self.occ_words, self.width)

    @property
    def drivable(self) -> np.ndarray:
        return unpack_words(self.drv_words, self.width)

    @property
    def bitmap_bytes(self) -> int:
        return self.height * self.pitch * 4

    @property
    def open(self) -> bool:
        """True for a compiled map that is NOT a loop (tracks/index.json "open": a building lobby compiled from a start of its
        own): scans, steps and collisions are as on any map, but the progress grid does not come round - a lap is never completed,
        `lap > laps` never ends an episode, and the centre-line table's two ends are not neighbours."""
        return self.name in open_tracks()


def pack_words(mask: np.ndarray, pitch: int) -> np.ndarray:
    h, w = mask.shape
    padded = np.zeros((h, pitch * 32), bool)
    padded[:, :w] = mask
    b = np.packbits(padded, axis=1, bitorder="little")
    return np.ascontiguousarray(b).view("<u4").reshape(h, pitch)


def unpack_words(words: np.ndarray, width: int) -> np.ndarray:
    b = np.ascontiguousarray(words).view(np.uint8)
    return np.unpackbits(b, axis=1, bitorder="little")[:, :width].astype(bool)


def track_from_npz(path: str) -> Track:
    d = np.load(path)
    h, w = (int(v) for v in d["shape"])
    res = float(d["resolution"])
    occ = np.unpackbits(d["occ"], axis=1, bitorder="little")[:, :w].astype(bool)
    drv = np.unpackbits(d["drivable"], axis=1, bitorder="little")[:, :w].astype(bool)
    steps = d["steps"].astype(np.int64)
    max_steps = int(d["max_steps"])
    # generate-costmap.py:221-222: distances * resolution, then / amax, in float64
    dist = np.where(steps == 0xFFFF, 0, steps).astype(np.float64) * res
    prog = dist / (float(max_steps) * res)
    progress = np.where(drv, prog, -1.0).astype(np.float32)
    edt_m = (np.sqrt(d["edt_sq"].astype(np.float64)) * res).astype(np.float32)
    pitch = ((w + 31) // 32) | 1
    return Track(
        name=str(d["name"]), map_name=str(d["map_name"]), resolution=res, inv_resolution=1.0 / res,
        origin=(float(d["origin"][0]), float(d["origin"][1])), height=h, width=w, pitch=pitch,
        occ_words=pack_words(occ, pitch), drv_words=pack_words(drv, pitch),
        progress=np.ascontiguousarray(progress), edt_m=edt_m,
        centerline=np.ascontiguousarray(d["centerline"].astype(np.float32)),
        max_steps=max_steps, crop=tuple(int(v) for v in d["crop"]),
        start_px=tuple(int(v) for v in d["start_px"]),
    )


@lru_cache(maxsize=None)
def load_track(name: str) -> Track:
    path = name if name.endswith(".npz") else os.path.join(TRACK_DIR, name + ".npz")
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"track asset {path!r} not found; compile it with "
            f"`python -m racing_dreamer_amd.track_compiler {name}` (needs the map images)")
    return track_from_npz(path)


@lru_cache(maxsize=None)
def open_tracks() -> frozenset:
    import json
    path = os.path.join(TRACK_DIR, "index.json")
    if not os.path.exists(path):
        return frozenset()
    with open(path) as f:
        return frozenset(k for k, v in json.load(f).items() if v.get("open"))


def available_tracks():
    return sorted(f[:-4] for f in os.listdir(TRACK_DIR) if f.endswith(".npz"))


def synthetic_track(height=96, width=160, wall=6, name="synthetic_oval") -> Track:
    """A small rectangular ring track built in memory (unit tests, no asset file needed)."""
    from .track_compiler import bfs_from_start, build_centerline
    from scipy import ndimage

    free = np.zeros((height, width), bool)
    free[wall:height - wall, wall:width - wall] = True
    inner = 3 * wall
    free[wall + inner:height - wall - inner, wall + inner:width - wall - inner] = False
    # image rows are north-up here; pick a start in the bottom straight heading +x
    sr, sc = height - wall - inner // 2 - 1, width // 2
    steps, finish, drivable, max_steps = bfs_from_start(free, sc, sr)
    edt_sq = np.rint(ndimage.distance_transform_edt(drivable) ** 2).astype(np.int32)
    res = 0.05
    occ, drv, st, esq = (np.ascontiguousarray(a[::-1]) for a in (~free, drivable, steps, edt_sq))
    origin = (-sc * res, -(height - 1 - sr) * res)
    cl = build_centerline(st, drv, esq, max_steps, res, origin)
    dist = np.where(st < 0, 0, st).astype(np.float64) * res
    progress = np.where(drv, dist / (max_steps * res), -1.0).astype(np.float32)
    pitch = ((width + 31) // 32) | 1
    return Track(name, name, res, 1.0 / res, origin, height, width, pitch,
                 pack_words(occ, pitch), pack_words(drv, pitch), progress,
                 (np.sqrt(esq.astype(np.float64)) * res).astype(np.float32), cl, int(max_steps),
                 (0, 0, height, width), (sc, sr))
