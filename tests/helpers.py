"""Shared test helpers: build an oracle env and a device env over the same track + config."""
import numpy as np

from oracle import racecar_oracle as ro

EXACT_FLOAT = ["lidar", "pose", "velocity", "speed", "action", "reward", "discount", "progress_total", "time",
               "progress", "acceleration", "steering_angle"]
EXACT_INT = ["lap", "checkpoint", "done", "truncated", "wall_collision", "opponent_collision", "wrong_way", "fresh"]
TOL = 1e-5   # BASELINE.json north_star: within 1e-5 fp32 on pose and LiDAR ranges; flags bit-exact


def make_oracle(track, **kw):
    env = ro.OracleRaceEnv(track.occ, track.drivable, track.progress, track.centerline, track.origin,
                           track.resolution, ro.OracleConfig(**kw))
    env.frame_track = track            # (render_occupancy='reference' indexes the source image's pixel frame)
    return env


def compare_outputs(dev_views, ora_out, n_envs, n_cars_per_env, context=""):
    """Device views (torch) vs oracle dict (numpy).  Flags bit-exact, floats <= 1e-5 AND bit-identical."""
    import torch
    torch.cuda.synchronize()
    for name in EXACT_INT:
        d = dev_views[name].cpu().numpy().reshape(-1)
        o = np.asarray(ora_out[name]).reshape(-1)
        bad = np.nonzero(d != o)[0]
        assert bad.size == 0, f"{context}: flag {name} differs at {bad[:8]} dev={d[bad[:8]]} oracle={o[bad[:8]]}"
    for name in EXACT_FLOAT:
        d = dev_views[name].cpu().numpy().reshape(-1)
        o = np.asarray(ora_out[name], np.float32).reshape(-1)
        err = np.abs(d.astype(np.float64) - o.astype(np.float64))
        assert err.max() <= TOL, f"{context}: {name} max abs err {err.max()} at {err.argmax()}"
        nbad = int((d != o).sum())
        assert nbad == 0, f"{context}: {name} within 1e-5 but not bit-identical in {nbad} elements (max err {err.max()})"
    if "lidar_occupancy" in ora_out:
        d = dev_views["lidar_occupancy"].cpu().numpy().reshape(-1)
        o = np.asarray(ora_out["lidar_occupancy"]).reshape(-1)
        assert np.array_equal(d, o), f"{context}: lidar_occupancy differs in {(d != o).sum()} pixels"
