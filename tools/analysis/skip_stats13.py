"""Analysis helper: first-trip rectangles that only need to be free INSIDE THE SECTOR the bin's rays can touch
(start point anywhere in the start cell, slope anywhere in the bin) - the rectangle's corners may lie in walls.
Later trips use the 4 quadrant planes of fully free rectangles."""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tools/analysis')
from oracle import racecar_oracle as ro, c_oracle
from racing_dreamer_amd.track_assets import load_track
from skip_stats9 import best_rect
from skip_stats10 import expd

CAP = 127


def sector_rect(occ, ix, iy, sx, sy, s1, s2, samples):
    """Best (w, h) for slopes |dy/dx| in [s1, s2], quadrant (sx, sy), start cell (ix, iy)."""
    H, W = occ.shape
    transpose = s1 >= 1.0
    if transpose:                       # y-dominant: swap the roles of the axes, slopes become 1/s
        s1, s2 = 1.0 / s2, 1.0 / s1
    def blocked(c, r):                  # cell at primary offset c, secondary offset r
        x, y = (ix + sx * r, iy + sy * c) if transpose else (ix + sx * c, iy + sy * r)
        return not (0 <= x < W and 0 <= y < H) or occ[y, x]
    best, bw, bh = -1.0, 1, 1
    hmax = CAP
    for c in range(CAP):
        lo = int(np.floor(s1 * max(0, c - 1) - 0.01)); hi = int(np.floor(1 + s2 * (c + 1) + 0.01))
        fb = None
        for r in range(max(lo, 0), min(hi, hmax - 1) + 1):
            if blocked(c, r): fb = r; break
        if fb is not None: hmax = min(hmax, fb)
        if hmax <= 0: break
        w, h = c + 1, hmax
        pw, ph = (h, w) if transpose else (w, h)     # back to (x extent, y extent)
        sc = sum(min(pw / ca, ph / sa) for ca, sa in samples)
        if sc > best: best, bw, bh = sc, pw, ph
    return bw, bh


def emulate(track, cars, m, kmin, kmax, tabs4, sector=True):
    occ = track.occ.copy(); occ[0,:]=occ[-1,:]=occ[:,0]=occ[:,-1]=True
    cb, sb = ro.beam_table()
    x, y, th = cars.T; ct, st = np.cos(th), np.sin(th)
    lx, ly = x + 0.25*ct, y + 0.25*st
    dx = (ct[:,None]*cb - st[:,None]*sb).ravel(); dy = (st[:,None]*cb + ct[:,None]*sb).ravel()
    gx = np.repeat((lx - track.origin[0])/0.05, 1080); gy = np.repeat((ly - track.origin[1])/0.05, 1080)
    ix = np.floor(gx).astype(int); iy = np.floor(gy).astype(int)
    n = len(ix); it = np.zeros(n, int); act = ~occ[iy, ix]
    px, py = dx > 0, dy > 0
    q = py.astype(int)*2 + px.astype(int)
    idx, idy = 1/np.where(dx==0,1e-30,dx), 1/np.where(dy==0,1e-30,dy)
    k = np.clip(np.floor(m * np.log2(np.maximum(np.abs(dy), 1e-30) / np.maximum(np.abs(dx), 1e-30))).astype(int), kmin, kmax)
    cache = {}
    frx = np.zeros(n, int); fry = np.zeros(n, int)
    for i in np.nonzero(act)[0]:
        key = (ix[i], iy[i], q[i], k[i])
        if key not in cache:
            lo, hi = k[i] / m, (k[i] + 1) / m
            s1 = 0.0 if k[i] == kmin else 2.0 ** lo * (1 - 1e-6)
            s2 = 1e9 if k[i] == kmax else 2.0 ** hi * (1 + 1e-6)
            a = [np.arctan(2.0 ** (lo + (hi - lo) * f)) for f in (0.25, 0.75)]
            cache[key] = sector_rect(occ, ix[i], iy[i], 1 if px[i] else -1, 1 if py[i] else -1, s1, s2, [(np.cos(v), np.sin(v)) for v in a])
        frx[i], fry[i] = cache[key]
    sw, sh = tabs4
    trip = 0
    while act.any():
        a_ = np.nonzero(act)[0]; it[a_] += 1
        if trip == 0:
            rx = frx[a_] - 1; ry = fry[a_] - 1
        else:
            rx = sw[q[a_], iy[a_], ix[a_]] - 1; ry = sh[q[a_], iy[a_], ix[a_]] - 1
        trip += 1
        xe = np.where(px[a_], ix[a_] + 1 + rx, ix[a_] - rx); ye = np.where(py[a_], iy[a_] + 1 + ry, iy[a_] - ry)
        txe = (xe - gx[a_])*idx[a_]; tye = (ye - gy[a_])*idy[a_]
        xexit = txe < tye; tt = np.where(xexit, txe, tye)
        nx = np.where(xexit, np.where(px[a_], xe, xe - 1), np.floor(gx[a_] + tt*dx[a_] + 1e-9*np.sign(dx[a_])).astype(int))
        ny = np.where(xexit, np.floor(gy[a_] + tt*dy[a_] + 1e-9*np.sign(dy[a_])).astype(int), np.where(py[a_], ye, ye - 1))
        nx = np.clip(nx, 0, track.width-1); ny = np.clip(ny, 0, track.height-1)
        ix[a_], iy[a_] = nx, ny
        act[a_[occ[ny, nx]]] = False
    return it


if __name__ == '__main__':
    t = load_track(sys.argv[1] if len(sys.argv) > 1 else 'austria')
    cfg = ro.OracleConfig(num_envs=128, auto_reset=True)
    b = c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg, threads=8)
    b.reset(mode=1, seed=0)
    for k in range(30): b.step(b.random_actions(1, k))
    cars = np.stack([b.arr['x'], b.arr['y'], b.arr['theta']], 1).astype(np.float64)
    occ = t.occ.copy(); occ[0,:]=occ[-1,:]=occ[:,0]=occ[:,-1]=True
    t4 = [best_rect(occ, sx, sy, expd([22.5, 67.5])) for sy in (-1, 1) for sx in (-1, 1)]
    tabs4 = (np.stack([a[0] for a in t4]), np.stack([a[1] for a in t4]))
    for m, kmin, kmax in ((2, -8, 7), (4, -16, 15), (8, -32, 31)):
        it = emulate(t, cars, m, kmin, kmax, tabs4)
        w = it.reshape(-1, 1080)[:, :1024].reshape(-1, 64)
        print(f'sector-free first trip, m {m} bins/quadrant {kmax-kmin+1:3d}: trips/ray {it.mean():.2f}  per-wave max {w.max(1).mean():.2f}  p99 {np.percentile(it, 99):.0f} max {it.max()}')
