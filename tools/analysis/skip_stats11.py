"""Analysis helper: direction-specialised rectangles for the FIRST trip only (the start cell is shared by a car's
1080 rays and read once per car) against specialising every trip.  nfirst / nrest = planes per quadrant."""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tools/analysis')
from oracle import racecar_oracle as ro, c_oracle
from racing_dreamer_amd.track_assets import load_track
from skip_stats9 import best_rect
from skip_stats10 import expd

_cache = {}


def tables(occ, nclass):
    if nclass not in _cache:
        tabs = []
        for sy in (-1, 1):
            for sx in (-1, 1):
                for k in range(nclass):
                    lo, hi = 90.0 / nclass * k, 90.0 / nclass * (k + 1)
                    tabs.append(best_rect(occ, sx, sy, expd([lo + (hi - lo) * 0.25, lo + (hi - lo) * 0.75])))
        _cache[nclass] = (np.stack([t[0] for t in tabs]), np.stack([t[1] for t in tabs]))
    return _cache[nclass]


def emulate(track, cars, nfirst, nrest):
    occ = track.occ.copy(); occ[0,:]=occ[-1,:]=occ[:,0]=occ[:,-1]=True
    cb, sb = ro.beam_table()
    x, y, th = cars.T; ct, st = np.cos(th), np.sin(th)
    lx, ly = x + 0.25*ct, y + 0.25*st
    dx = (ct[:,None]*cb - st[:,None]*sb).ravel(); dy = (st[:,None]*cb + ct[:,None]*sb).ravel()
    gx = np.repeat((lx - track.origin[0])/0.05, 1080); gy = np.repeat((ly - track.origin[1])/0.05, 1080)
    ix = np.floor(gx).astype(int); iy = np.floor(gy).astype(int)
    n = len(ix); it = np.zeros(n, int); act = ~occ[iy, ix]
    px, py = dx > 0, dy > 0
    ang = np.degrees(np.arctan2(np.abs(dy), np.abs(dx)))
    q = py.astype(int)*2 + px.astype(int)
    idx, idy = 1/np.where(dx==0,1e-30,dx), 1/np.where(dy==0,1e-30,dy)
    first = True
    while act.any():
        nclass = nfirst if first else nrest
        first = False
        sw, sh = tables(occ, nclass)
        cls = q * nclass + np.minimum((ang / (90.0 / nclass)).astype(int), nclass - 1)
        a_ = np.nonzero(act)[0]; it[a_] += 1
        rx = sw[cls[a_], iy[a_], ix[a_]] - 1; ry = sh[cls[a_], iy[a_], ix[a_]] - 1
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
    cfg = ro.OracleConfig(num_envs=256, auto_reset=True)
    b = c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg, threads=8)
    b.reset(mode=1, seed=0)
    for k in range(30): b.step(b.random_actions(1, k))
    cars = np.stack([b.arr['x'], b.arr['y'], b.arr['theta']], 1).astype(np.float64)
    for nfirst, nrest in ((1, 1), (2, 1), (4, 1), (8, 1), (16, 1), (2, 2), (8, 2)):
        it = emulate(t, cars, nfirst, nrest)
        w = it.reshape(-1, 1080)[:, :1024].reshape(-1, 64)
        print(f'first {4*nfirst:3d} planes, rest {4*nrest:2d}: trips/ray {it.mean():.2f}  per-wave max {w.max(1).mean():.2f}')
