"""Track assets and the offline track compiler (restating docs/maps/costmaps/generate-costmap.py)."""
import os

import numpy as np
import pytest

from racing_dreamer_amd import track_assets as ta
from racing_dreamer_amd import track_compiler as tc

REF_MAPS = "/root/reference/docs/maps/maps"


BASELINE_TRACKS = ("columbia", "austria", "barcelona", "treitlstrasse_v2", "gbr")


def test_assets_present_and_sane():
    """Every compiled map of docs/maps/maps: 29 of the 36 yaml files compile with the generator's default start position, 3 more
    with a start position of their own (tracks/start_positions.json: levinelobby, unreal, vegas - what the generator takes as
    --start_x / --start_y, generate-costmap.py:460-475); tracks/index.json lists the other 4 with the reason (two images are
    not in the checkout; porto and stata_basement flood under the generator's own BFS from ANY start)."""
    import json
    names = ta.available_tracks()
    index = json.load(open(os.path.join(ta.TRACK_DIR, "index.json")))
    assert set(names) == {n for n, e in index.items() if e["status"] == "ok"} and len(names) >= 29
    assert sum(e["status"] != "ok" for e in index.values()) == 4 and all(e.get("reason") for e in index.values() if e["status"] != "ok")
    assert len(names) == 32 and {"levinelobby", "unreal", "vegas"} <= set(names)
    assert all("start_position" in index[n] for n in ("levinelobby", "unreal", "vegas"))
    for n in BASELINE_TRACKS:                                                  # the tracks BASELINE.json / the scenarios name
        assert n in names
    for n in names:
        t = ta.load_track(n)
        assert (t.resolution == 0.05 or n == "unreal") and t.pitch % 2 == 1 and t.pitch * 32 >= t.width       # (unreal: 1 / 35 m per cell)
        assert max(t.height, t.width) <= 4096                                  # rc_load_track's limit
        assert t.progress.shape == (t.height, t.width) and t.progress.max() == 1.0
        drv, occ = t.drivable, t.occ
        assert not (drv & occ).any()
        assert np.all(t.progress[drv] >= 0) and np.all(t.progress[~drv] == -1)
        # every spawn pose is drivable and ordered by progress
        cl = t.centerline
        ix = np.floor((cl[:, 0] - t.origin[0]) / t.resolution).astype(int)
        iy = np.floor((cl[:, 1] - t.origin[1]) / t.resolution).astype(int)
        assert drv[iy, ix].all() and np.all(np.diff(cl[:, 3]) > 0)
        if n in BASELINE_TRACKS:
            assert t.bitmap_bytes <= 160 * 1024                                # lidar_occupancy keeps the bitmap in one CU's LDS
            assert t.edt_m[iy, ix].min() >= 0.2                                # spawn poses clear of the walls
            gaps = np.hypot(*(np.roll(cl[:, :2], -1, axis=0) - cl[:, :2]).T)
            assert gaps.max() < 0.5                                            # a closed loop, evenly spaced


@pytest.mark.skipif(not os.path.isdir(REF_MAPS), reason="reference maps only exist in the build container")
def test_maps_npz_export_has_the_generators_keys_and_values(tmp_path):
    """generate-costmap.py:410-425: key names, value conventions and frame of the racecar_gym-style maps.npz."""
    out = tmp_path / "maps.npz"
    tc.export_maps_npz("columbia_slam", REF_MAPS, str(out))      # docs/maps/maps/columbia.pgm: not square, the raw f1tenth_simulator map
    d = np.load(out)
    assert sorted(d.files) == ["drivable_area", "norm_distance_from_start", "norm_distance_to",
                               "norm_distance_to_obstacle", "properties"]
    full = (350, 435)                                                           # docs/maps/maps/columbia.pgm
    for k in (f for f in d.files if f != "properties"):
        assert d[k].shape == full, k
    pr = d["properties"]
    assert len(pr) == 12 and tuple(pr[:2]) == (0.0, 0.0) and tuple(pr[2:4]) == (214.0, 153.0)     # start pixel (col, row)
    assert pr[6] == 0.65 and tuple(pr[9:11]) == full and pr[11] == 0.05
    a = ta.load_track("columbia_slam")
    r0, c0, fh, fw = a.crop
    drv_full = np.zeros(full, bool)
    drv_full[r0:r0 + a.height, c0:c0 + a.width] = a.drivable[::-1]                                  # asset rows are south-up
    assert np.array_equal(d["drivable_area"], drv_full)
    prog = np.zeros(full)
    prog[r0:r0 + a.height, c0:c0 + a.width] = np.where(a.drivable, a.progress.astype(np.float64), 0.0)[::-1]
    assert np.allclose(d["norm_distance_from_start"], prog, atol=1e-7) and d["norm_distance_from_start"].max() == 1.0
    edt = np.zeros(full, np.float32)
    edt[r0:r0 + a.height, c0:c0 + a.width] = a.edt_m[::-1]
    assert np.allclose(d["norm_distance_to_obstacle"], edt / edt.max(), atol=1e-6)
    # the backward distance: roughly complementary to the forward one along the track (both are 8-connected shortest
    # paths, which cut the corners on opposite sides)
    to = d["norm_distance_to"]
    assert to.max() == 1.0 and np.all(to[~drv_full] == 0)
    inside = drv_full & (d["norm_distance_from_start"] > 0.05) & (d["norm_distance_from_start"] < 0.95)
    s = d["norm_distance_from_start"][inside] + to[inside]
    assert abs(np.median(s) - 1.0) < 0.08 and np.percentile(np.abs(s - 1.0), 95) < 0.12


def test_pack_unpack_roundtrip():
    rng = np.random.default_rng(0)
    m = rng.uniform(size=(37, 101)) < 0.3
    w = ta.pack_words(m, 5)
    assert w.shape == (37, 5) and w.dtype == np.uint32
    assert np.array_equal(ta.unpack_words(w, 101), m)
    assert ((w[3, 70 >> 5] >> (70 & 31)) & 1) == int(m[3, 70])              # bit i of word j = cell 32 j + i


def test_bfs_matches_naive_dilation_restatement():
    """The frontier BFS equals the reference's formulation: one 3x3 binary dilation per distance unit
    (generate-costmap.py:196-208), checked on a small maze with scipy's dilation."""
    from scipy import ndimage
    rng = np.random.default_rng(1)
    free = np.ones((40, 60), bool)
    free[0, :] = free[-1, :] = free[:, 0] = free[:, -1] = False
    free[10:30, 20:40] = False
    free[rng.uniform(size=free.shape) < 0.04] = False
    sc, sr = 30, 35
    free[sr, sc - 3:sc + 3] = True
    steps, finish, drivable, n = tc.bfs_from_start(free, sc, sr)
    # naive restatement
    binary = free.copy()
    fl = np.zeros_like(free)
    for row, st in ((sr, +1), (sr - 1, -1)):
        while binary[row, sc - 1]:
            binary[row, sc - 1] = False
            fl[row, sc - 1] = True
            row += st
    mask = np.zeros_like(free)
    mask[sr, sc] = True
    dist = np.zeros(free.shape)
    cur = 0.0
    while True:
        cur += 1.0
        dil = ndimage.binary_dilation(mask, structure=np.ones((3, 3), bool))
        new = binary & (mask ^ dil)
        if not new.any():
            break
        dist[new] = cur
        mask |= new
    dist[fl] = cur
    assert n == int(cur) and np.array_equal(finish, fl) and np.array_equal(drivable, mask | fl)
    assert np.array_equal(np.where(steps < 0, 0, steps), dist.astype(int) * (mask | fl))


@pytest.mark.skipif(not os.path.isdir(REF_MAPS), reason="reference maps only exist in the build container")
def test_committed_assets_reproduce_from_the_reference_maps():
    t = tc.compile_track("treitlstrasse_v2", REF_MAPS)
    a = ta.load_track("treitlstrasse_v2")
    assert np.array_equal(t.occ, a.occ) and np.array_equal(t.drivable, a.drivable)
    assert t.max_steps == a.max_steps and np.array_equal(t.centerline, a.centerline)
    # start pixel: world (0, 0) with the reference's width-flip quirk (generate-costmap.py:49-52)
    assert tuple(t.start_px) == (1000, 999)
    assert tc.start_pixel((350, 435), [-10.70654, -14.020793, 0.0], 0.05) == (214, 153)


def test_synthetic_track_progress_is_a_loop():
    t = ta.synthetic_track()
    cl = t.centerline
    assert len(cl) > 100
    d = np.hypot(*(np.roll(cl[:, :2], -1, axis=0) - cl[:, :2]).T)
    assert d.max() < 0.3                                                     # closed, evenly spaced (0.1 m bins)


def test_scene_export_is_what_the_plotting_code_loads(tmp_path):
    """`track_compiler.export_scene` writes the scene directory racecar_gym keeps per track; the sequence of
    dreamer/plotting/plot_trajectories.py:17-38 (`load_map`: SceneConfig().load, resolve_path, GridMap over maps.npz) run on it
    through the shim's classes gives the occupancy map the env itself uses, and `to_pixel` puts the starting grid on the
    drivable area."""
    import os
    from racing_dreamer_amd import compat
    compat.install()
    from racecar_gym.bullet.configs import SceneConfig
    from racecar_gym.bullet.providers import resolve_path
    from racecar_gym.core.gridmaps import GridMap
    from racing_dreamer_amd.compat.racecar_gym.envs.scenarios import World
    from racing_dreamer_amd.track_assets import load_track
    from racing_dreamer_amd.track_compiler import export_scene
    maps_dir = "/root/reference/docs/maps/maps"
    if not os.path.isdir(maps_dir):
        import pytest
        pytest.skip("needs the reference's map images (build container only)")
    track = "columbia_slam"
    config_file = export_scene(track, maps_dir, str(tmp_path))
    assert config_file.endswith(f"{track}/{track}.yml")
    config = SceneConfig()                                                  # plot_trajectories.py:20-25
    config.load(config_file)
    config.sdf = resolve_path(file=config_file, relative_path=config.sdf)
    config.map.maps = resolve_path(file=config_file, relative_path=config.map.maps)
    config.map.starting_grid = resolve_path(file=config_file, relative_path=config.map.starting_grid)
    maps = dict([(name, GridMap(grid_map=np.load(config.map.maps)[data], origin=config.map.origin,
                                resolution=config.map.resolution))
                 for name, data in [("progress", "norm_distance_from_start"), ("obstacle", "norm_distance_to_obstacle"),
                                    ("occupancy", "drivable_area")]])             # :26-37
    occ = maps["occupancy"]
    world = World(load_track(track))                                        # what the env's scenario exposes (wrappers.py:376)
    assert occ.map.shape == world._maps["occupancy"].map.shape and np.array_equal(occ.map > 0, world._maps["occupancy"].map > 0)
    # (columbia's image is not square: the generator's start pixel is flipped with the image WIDTH, generate-costmap.py:51,
    # so world (0, 0) is not where GridMap.to_pixel puts it - the env's own poses are the ones to check)
    grid = np.load(config.map.starting_grid)["data"]
    assert grid.shape == (4, 3)
    for x, y, yaw in grid:
        rr, cc = occ.to_pixel((x, y))
        assert occ.map[rr, cc] and maps["progress"].map[rr, cc] >= 0.0 and maps["obstacle"].map[rr, cc] > 0.0, (x, y)
    assert os.path.isabs(config.sdf) and config.name == track and config.map.no_such_key is None


def test_the_one_open_map_is_flagged_open():
    """levinelobby - a building lobby compiled from a start of its own - is not a loop (tracks/start_positions.json
    says so; the BFS wave does not come round).  index.json and Track carry the flag, every other compiled map is a loop, and
    the track compiler writes the flag from the start-position table."""
    import json
    from racing_dreamer_amd.track_assets import TRACK_DIR, load_track, open_tracks
    with open(os.path.join(TRACK_DIR, "index.json")) as f:
        index = json.load(f)
    assert open_tracks() == {"levinelobby"} == {k for k, v in index.items() if v.get("open")}
    assert load_track("levinelobby").open and not load_track("austria").open and not load_track("columbia").open
    with open(os.path.join(TRACK_DIR, "start_positions.json")) as f:
        assert {k for k, v in json.load(f).items() if v.get("open")} == {"levinelobby"}
