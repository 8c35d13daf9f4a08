"""Soak run (GPU box): long random-action rollouts on every track, every output finite, ranges within [0, 15], and a
spot check of the last scan - and, where the track's bitmap fits the LDS, the last lidar_occupancy patches - against the CPU
oracle.  Also prints the time rc_load_track takes per track (table builds + the bounded validation scan from every free cell)."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import torch
from racing_dreamer_amd.batched_env import BatchedRaceEnv
from racing_dreamer_amd.track_assets import load_track
from oracle import racecar_oracle as ro, c_oracle

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
names = ("austria", "columbia", "barcelona", "gbr", "treitlstrasse_v2")
if len(sys.argv) > 2 and sys.argv[2] == "all":           # python tools/soak.py 1000 all: every compiled track
    from racing_dreamer_amd.track_assets import available_tracks
    names = tuple(available_tracks())
for name in names:
    t = load_track(name)
    t0 = time.perf_counter()
    try:
        env = BatchedRaceEnv(t, 16384, 1, auto_reset=True, obs_type="lidar_occupancy")
        occ = True
    except Exception:                                   # the render needs the bitmap in the 160 KB LDS
        env = BatchedRaceEnv(t, 16384, 1, auto_reset=True)
        occ = False
    env.sync()
    t_load = time.perf_counter() - t0
    env.reset(mode="random", seed=1)
    torch.cuda.set_stream(env.stream)
    t0 = time.perf_counter()
    for k in range(steps):
        out = env.step_random(7, k, repeat=1 + (k % 4 == 0) * 3)
    env.sync()
    dt = time.perf_counter() - t0
    lid = out["lidar"].reshape(-1, 1080)
    ok = bool(torch.isfinite(lid).all()) and float(lid.min()) >= 0.0 and float(lid.max()) <= 15.0
    pose = out["pose"].reshape(-1, 6).cpu().numpy()
    n = 512
    cfg = ro.OracleConfig(num_envs=n)
    o = c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg, threads=8)
    o.reset()
    o.arr["x"][:], o.arr["y"][:], o.arr["theta"][:] = pose[:n, 0], pose[:n, 1], pose[:n, 5]
    o.arr["st"][:], o.arr["ct"][:] = ro.sincos32(pose[:n, 5].astype(np.float32))
    o.arr["fresh"][:] = out["fresh"].reshape(-1)[:n].cpu().numpy()
    o.cfg.render_occupancy = occ
    o._observe()
    same = np.array_equal(o.lidar, lid[:n].cpu().numpy())
    same_patch = (not occ) or np.array_equal(o.patch, out["lidar_occupancy"].reshape(-1, 64, 64)[:n].cpu().numpy())
    print(f"{name:18s} load {t_load:5.2f} s  {steps} steps in {dt:5.2f} s  finite/in-range {ok}  last scan == oracle {same}  "
          f"{'last patches == oracle ' + str(same_patch) if occ else '(bitmap too large for the render)'}  overruns {env.scan_overruns()}", flush=True)
    assert ok and same and same_patch
    env.close()
print("soak ok")
