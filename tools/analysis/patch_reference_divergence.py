#!/usr/bin/env python3
"""Round 6: `lidar_occupancy` three ways over 10^4 poses per track - not only the 96 golden ones.

    library   oracle.patch_reference.render_patch_reference: scipy.ndimage.rotate + PIL resize, the reference's own calls
              (dreamer/wrappers.py:396-406); == the G6 goldens 379 / 379
    exact     oracle.patch_reference.render_patch_exact: the restatement without a library (the spec of obs_type
              `lidar_occupancy_reference`, what the C oracle and rc_patch_exact_kernel follow)
    fast      the shipped obs_type `lidar_occupancy`: one nearest-cell tap per output pixel (oracle render_patch)

Poses: along the centre line with +- 0.4 m offsets and any heading (a car anywhere on the track), plus a tenth pushed up to
1.5 m off the line (past the border).  Prints per track: patches on which exact != library (expected 0: they may differ only
where cos / sin or the BLAS behind scipy's 2 x 2 products differ in the last bit AND a spline value lies within 1e-13 of 0.5);
pixel agreement of fast with library (mean / min over patches), how many patches agree completely, and where the differing
pixels lie (share within one output pixel of an edge of the library patch).
    python tools/analysis/patch_reference_divergence.py [--poses 10000] [--workers 6] > profiles/r06_e_patch_reference_divergence.txt"""
import argparse
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def work(job):
    name, poses = job
    from scipy import ndimage
    from helpers import make_oracle
    from oracle import patch_reference as pr
    from oracle import racecar_oracle as ro
    from racing_dreamer_amd.track_assets import load_track
    t = load_track(name)
    lib = pr.render_patch_reference(t, poses)
    exact = pr.render_patch_exact(t, poses)
    p32 = poses.astype(np.float32)
    env = make_oracle(t, num_envs=len(poses), render_occupancy=True)
    env.x[:], env.y[:], env.theta[:] = p32[:, 0], p32[:, 1], p32[:, 2]
    env.st[:], env.ct[:] = ro.sincos32(env.theta)
    env.fresh[:] = 0
    fast = env.render_patch()
    k = 3
    edge = ndimage.maximum_filter(lib, size=(1, k, k), mode="nearest") != ndimage.minimum_filter(lib, size=(1, k, k), mode="nearest")
    bad = fast != lib
    return dict(n=len(poses), exact_bad=int((exact != lib).any(axis=(1, 2)).sum()), exact_bad_pixels=int((exact != lib).sum()),
                agree=(~bad).mean(axis=(1, 2)), bad=int(bad.sum()), bad_on_edge=int((bad & edge).sum()), edge=int(edge.sum()))


def poses_for(track, n, rng):
    cl = track.centerline.astype(np.float64)
    idx = rng.integers(0, len(cl), n)
    p = cl[idx, :3].copy()
    p[:, :2] += rng.uniform(-0.4, 0.4, (n, 2))
    far = rng.random(n) < 0.1
    side = rng.choice([-1.0, 1.0], n) * rng.uniform(0.5, 1.5, n)
    p[far, 0] += (-np.sin(p[far, 2]) * side[far])
    p[far, 1] += (np.cos(p[far, 2]) * side[far])
    p[:, 2] = rng.uniform(-np.pi, np.pi, n)
    # the env's state is float32: the poses ARE float32 values (widened), as the device will see them
    return p.astype(np.float32).astype(np.float64)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--poses", type=int, default=10000)
    ap.add_argument("--workers", type=int, default=6)
    ap.add_argument("--tracks", default="austria,columbia,treitlstrasse_v2,barcelona")
    a = ap.parse_args()
    from racing_dreamer_amd.track_assets import load_track
    print(f"{a.poses} poses per track; library = scipy.ndimage.rotate + PIL resize (the reference's calls), exact = the restatement, fast = the shipped sampler")
    for name in a.tracks.split(","):
        t = load_track(name)
        poses = poses_for(t, a.poses, np.random.default_rng(12345))
        jobs = [(name, poses[i:i + 250]) for i in range(0, len(poses), 250)]
        with ProcessPoolExecutor(a.workers) as ex:
            res = list(ex.map(work, jobs))
        agree = np.concatenate([r["agree"] for r in res])
        bad, on_edge = sum(r["bad"] for r in res), sum(r["bad_on_edge"] for r in res)
        print(f"{name:18s} exact != library on {sum(r['exact_bad'] for r in res)} of {a.poses} patches ({sum(r['exact_bad_pixels'] for r in res)} pixels) | "
              f"fast vs library: pixel agreement mean {agree.mean() * 100:.3f} % min {agree.min() * 100:.2f} %, "
              f"{int((agree == 1.0).sum())} patches identical, {bad} differing pixels of {a.poses * 4096}, {on_edge / max(bad, 1) * 100:.1f} % of them within one pixel of an edge "
              f"(edge band = {sum(r['edge'] for r in res) / (a.poses * 4096) * 100:.1f} % of all pixels)", flush=True)
