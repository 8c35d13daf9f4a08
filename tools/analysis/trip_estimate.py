#!/usr/bin/env python3
"""How many table lookups ("trips") would a ray need with free rectangles chosen per DIRECTION BIN instead of one per quadrant?
(CPU, no GPU: python tools/analysis/trip_estimate.py [track] [poses])

The scan's time is its wave-level trips: a wave's round lasts as long as the ray with the most lookups among its 64.
Today every cell holds ONE free rectangle per quadrant - the one with the best geometric mean of the exit distances of four sample
directions (rc_build_quad_kernel) - and only the first trip is direction-specific (the 64-bin first-trip table).  This script counts
lookups per ray and per 64-ray round on random on-track poses for (a) that scheme, (b) rectangles chosen separately for 2 / 4 / 8
direction bins per quadrant, each maximising the travel along its own bin's direction.  Rectangles, whatever their choice, leave
the returned ranges untouched (any free rectangle is a valid skip): this is about table size against trips only.  The first trip is
modelled alike in all schemes (best rectangle among 16 direction bins per quadrant).  An estimate: the device's first-trip shapes
are wedge-following staircases, not rectangles, so absolute counts differ from the stamps; the ratio between schemes is the result."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import racecar_oracle as ro  # noqa: E402
from racing_dreamer_amd.track_assets import load_track  # noqa: E402

CAP = 255


def runs_pos(free):
    """Free run length from every cell towards +x (0 on a stop cell), capped."""
    h, w = free.shape
    run = np.zeros((h, w), np.int32)
    r = np.zeros(h, np.int32)
    for x in range(w - 1, -1, -1):
        r = np.where(free[:, x], np.minimum(r + 1, CAP), 0)
        run[:, x] = r
    return run


def build(free, dirs):
    """For every cell and every direction (angle in (0, 90) deg inside the quadrant, rays heading +x +y): the free rectangle anchored
    at the cell that maximises the travel of a ray of that direction from the cell's origin; plus the compromise rectangle
    (geometric mean over the four sample directions of rc_build_quad_kernel).  Returns (bw, bh) [K, H, W] and (cw, ch) [H, W]."""
    h, w = free.shape
    run = runs_pos(free)
    ka, kb = 1.0 / np.cos(np.radians(dirs)), 1.0 / np.sin(np.radians(dirs))
    sa = np.array([1.0195911, 1.2026898, 1.7999525, 5.1258309]); sb = sa[::-1]
    K = len(dirs)
    best = np.full((K, h, w), -1.0); bw = np.ones((K, h, w), np.int16); bh = np.ones((K, h, w), np.int16)
    cbest = np.full((h, w), -1e30); cw = np.ones((h, w), np.int16); ch = np.ones((h, w), np.int16)
    cur = np.full((h, w), CAP, np.int32)
    for n in range(1, CAP + 1):
        shifted = np.zeros((h, w), np.int32)
        if n - 1 < h:
            shifted[:h - (n - 1)] = run[n - 1:]
        cur = np.minimum(cur, shifted)
        live = cur > 0
        if not live.any():
            break
        cf = cur.astype(np.float64)
        for k in range(K):
            sc = np.where(live, np.minimum(cf * ka[k], n * kb[k]), -1.0)
            up = sc > best[k]
            best[k][up] = sc[up]; bw[k][up] = cur[up]; bh[k][up] = n
        with np.errstate(divide="ignore"):
            sc = np.where(live, sum(np.log(np.minimum(cf * sa[i], n * sb[i])) for i in range(4)), -1e30)
        up = sc > cbest
        cbest[up] = sc[up]; cw[up] = cur[up]; ch[up] = n
    return bw, bh, cw, ch


def main():
    track = sys.argv[1] if len(sys.argv) > 1 else "austria"
    n_pose = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    t = load_track(track)
    occ = np.asarray(t.occ, bool).copy()
    occ[0, :] = occ[-1, :] = occ[:, 0] = occ[:, -1] = True
    H, W = occ.shape
    env = ro.OracleRaceEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, ro.OracleConfig(num_envs=n_pose, auto_reset=True))
    env.seed, env.mode = 3, ro.RESET_RANDOM
    env._reset_envs(np.arange(n_pose))
    cb, sb = ro.beam_table()
    ct, st = env.ct.astype(np.float64), env.st.astype(np.float64)
    lx = env.x + ro.LIDAR_X * ct if hasattr(ro, "LIDAR_X") else env.x
    ly = env.y + ro.LIDAR_X * st if hasattr(ro, "LIDAR_X") else env.y
    gx0 = np.repeat((lx - t.origin[0]) / t.resolution, 1080); gy0 = np.repeat((ly - t.origin[1]) / t.resolution, 1080)
    dx = (ct[:, None] * cb[None, :] - st[:, None] * sb[None, :]).reshape(-1).astype(np.float64)
    dy = (ct[:, None] * sb[None, :] + st[:, None] * cb[None, :]).reshape(-1).astype(np.float64)
    n_ray = dx.size
    schemes = {"one rectangle per quadrant (today)": None, "2 direction bins per quadrant": 2, "4 direction bins per quadrant": 4, "8 direction bins per quadrant": 8}
    lookups = {k: np.zeros(n_ray, np.int32) for k in schemes}
    for q in range(4):
        sx, sy = (-1 if q & 1 else 1), (-1 if q & 2 else 1)
        sel = np.nonzero(((dx < 0) == (sx < 0)) & ((dy < 0) == (sy < 0)))[0]
        if sel.size == 0:
            continue
        M = occ[::sy, ::sx]
        free = ~M
        first_dirs = (np.arange(16) + 0.5) * (90.0 / 16)
        tables = {}
        fb_w, fb_h, cw, ch = build(free, first_dirs)
        tables[None] = (cw[None], ch[None])
        for k in (2, 4, 8):
            d = (np.arange(k) + 0.5) * (90.0 / k)
            w_, h_, _, _ = build(free, d)
            tables[k] = (w_, h_)
        px0 = np.where(sx < 0, W - gx0[sel], gx0[sel]); py0 = np.where(sy < 0, H - gy0[sel], gy0[sel])
        ax, ay = np.abs(dx[sel]), np.abs(dy[sel])
        ang = np.degrees(np.arctan2(ay, ax))
        for name, k in schemes.items():
            tw, th = tables[k]
            nb = tw.shape[0]
            b_later = np.minimum((ang / (90.0 / nb)).astype(np.int64), nb - 1)
            b_first = np.minimum((ang / (90.0 / 16)).astype(np.int64), 15)
            px, py = px0.copy(), py0.copy()
            active = np.ones(sel.size, bool)
            count = np.zeros(sel.size, np.int32)
            trav = np.zeros(sel.size)
            trip = 0
            while active.any() and trip < 400:
                idx = np.nonzero(active)[0]
                ix = np.clip(np.floor(px[idx]).astype(np.int64), 0, W - 1); iy = np.clip(np.floor(py[idx]).astype(np.int64), 0, H - 1)
                count[idx] += 1                                            # one lookup
                wall = M[iy, ix] | (trav[idx] >= ro.MAX_RANGE / t.resolution)
                if trip == 0:
                    w_ = fb_w[b_first[idx], iy, ix].astype(np.float64); h_ = fb_h[b_first[idx], iy, ix].astype(np.float64)
                else:
                    w_ = tw[b_later[idx], iy, ix].astype(np.float64); h_ = th[b_later[idx], iy, ix].astype(np.float64)
                with np.errstate(divide="ignore", invalid="ignore"):
                    tx = np.where(ax[idx] > 0, (ix + w_ - px[idx]) / ax[idx], np.inf)
                    ty = np.where(ay[idx] > 0, (iy + h_ - py[idx]) / ay[idx], np.inf)
                tt = np.minimum(tx, ty)
                nx = np.where(tx <= ty, ix + w_ + 1e-9, px[idx] + tt * ax[idx])
                ny = np.where(ty < tx, iy + h_ + 1e-9, py[idx] + tt * ay[idx])
                go = ~wall
                px[idx[go]] = nx[go]; py[idx[go]] = ny[go]; trav[idx[go]] += tt[go]
                active[idx[wall]] = False
                trip += 1
            lookups[name][sel] = count
    print(f"{track}: {n_pose} random on-track poses x 1080 beams; lookups per ray (the last one finds the wall) and per 64-ray round (the most of its lanes)")
    base = None
    for name in schemes:
        c = lookups[name].reshape(n_pose, 1080)
        pad = np.concatenate([c, np.zeros((n_pose, 8), np.int32)], 1).reshape(n_pose, 17, 64)
        per_round = pad.max(2)
        m_ray, m_round = c.mean(), per_round.mean()
        if base is None:
            base = m_round
        hist = " ".join(f"{(per_round == v).mean() * 100:.0f}%" for v in range(2, 9))
        print(f"  {name:38s} per ray {m_ray:5.2f}   per round {m_round:5.2f} ({(m_round / base - 1) * 100:+5.1f} %)   lane use {m_ray / m_round * 100:4.0f} %   rounds with 2..8 lookups: {hist}")
    cells = H * W
    print(f"  table bytes (2 B per cell and rectangle): today {4 * 2 * cells / 1e6:.1f} MB, 2 bins {8 * 2 * cells / 1e6:.1f}, 4 bins {16 * 2 * cells / 1e6:.1f}, 8 bins {32 * 2 * cells / 1e6:.1f} (one XCD's L2: 4 MB)")


if __name__ == "__main__":
    main()
