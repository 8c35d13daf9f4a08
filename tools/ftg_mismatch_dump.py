"""Find scans on which rc_follow_the_gap_reference and its binary32 spec disagree (GPU box); dumps them to gpurun_out/ftg_mismatch.npz."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import racecar_oracle as ro
from racing_dreamer_amd.batched_env import BatchedRaceEnv
n = 1024
found = []
for track in ("columbia", "austria", "barcelona", "columbia_slam"):
    env = BatchedRaceEnv(track, n, 1, auto_reset=True, remap_actions=False, action_repeat=4)
    out = env.reset(mode="random", seed=5)
    prev = np.full(n, np.nan, np.float32)
    for k in range(150):
        act, det = env.follow_the_gap_reference(detail=True)
        torch.cuda.synchronize()
        fresh = out["fresh"].cpu().numpy().reshape(n) != 0
        scan = out["lidar"].cpu().numpy().reshape(n, 1080)
        pv = np.where(fresh, np.float32(np.nan), prev)
        want = ro.follow_the_gap_reference(scan, pv, 0.04)
        d = det.cpu().numpy()
        bad = np.zeros(n, bool)
        for j, name in enumerate(("heading", "heading_distance", "steering_angle", "speed")):
            bad |= ~((d[:, j] == want[name]) | (np.isnan(d[:, j]) & np.isnan(want[name])))
        for i in np.nonzero(bad)[0]:
            found.append((track, k, int(i), scan[i].copy(), pv[i], d[i].copy(), np.array([want[m][i] for m in ("heading", "heading_distance", "steering_angle", "speed")])))
        prev = d[:, 0].copy()
        out = env.step(None)
    env.close()
    print(track, "mismatching (car, step) pairs so far:", len(found), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/ftg_mismatch.npz", track=np.array([f[0] for f in found]), step=np.array([f[1] for f in found]), car=np.array([f[2] for f in found]),
         scan=np.array([f[3] for f in found], np.float32).reshape(-1, 1080), prev=np.array([f[4] for f in found], np.float32),
         device=np.array([f[5] for f in found], np.float32).reshape(-1, 4), spec=np.array([f[6] for f in found], np.float32).reshape(-1, 4))
for f in found[:10]:
    print(f[0], f[1], f[2], "prev", f[4], "device", f[5], "spec", f[6])
