"""Analysis helper: first-trip rectangles specialised by LOG-SLOPE bins (bin = clamp(floor(m log2 |dy/dx|))), which the
kernel can derive from the float bits of |dy * (1/dx)| in two instructions; later trips use the 4 quadrant planes."""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tools/analysis')
from oracle import racecar_oracle as ro, c_oracle
from racing_dreamer_amd.track_assets import load_track
from skip_stats9 import best_rect
from skip_stats10 import expd


def emulate(track, cars, m, kmin, kmax, tabs4):
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
    nb = kmax - kmin + 1
    if m > 0:
        k = np.clip(np.floor(m * np.log2(np.maximum(np.abs(dy), 1e-30) / np.maximum(np.abs(dx), 1e-30))).astype(int), kmin, kmax) - kmin
        first = []
        for sy in (-1, 1):
            for sx in (-1, 1):
                for b in range(nb):
                    lo, hi = (b + kmin) / m, (b + kmin + 1) / m                     # log2 slope range of the bin
                    a = [np.degrees(np.arctan(2.0 ** (lo + (hi - lo) * f))) for f in (0.25, 0.75)]
                    first.append(best_rect(occ, sx, sy, expd(a)))
        fw = np.stack([t[0] for t in first]); fh = np.stack([t[1] for t in first])
        fcls = q * nb + k
    sw, sh = tabs4
    trip = 0
    while act.any():
        a_ = np.nonzero(act)[0]; it[a_] += 1
        if trip == 0 and m > 0:
            rx = fw[fcls[a_], iy[a_], ix[a_]] - 1; ry = fh[fcls[a_], iy[a_], ix[a_]] - 1
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
    cfg = ro.OracleConfig(num_envs=256, auto_reset=True)
    b = c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg, threads=8)
    b.reset(mode=1, seed=0)
    for k in range(30): b.step(b.random_actions(1, k))
    cars = np.stack([b.arr['x'], b.arr['y'], b.arr['theta']], 1).astype(np.float64)
    occ = t.occ.copy(); occ[0,:]=occ[-1,:]=occ[:,0]=occ[:,-1]=True
    t4 = [best_rect(occ, sx, sy, expd([22.5, 67.5])) for sy in (-1, 1) for sx in (-1, 1)]
    tabs4 = (np.stack([a[0] for a in t4]), np.stack([a[1] for a in t4]))
    for m, kmin, kmax in ((0, 0, 0), (1, -4, 3), (1, -8, 7), (2, -8, 7), (2, -16, 15), (4, -16, 15), (4, -32, 31)):
        it = emulate(t, cars, m, kmin, kmax, tabs4)
        w = it.reshape(-1, 1080)[:, :1024].reshape(-1, 64)
        print(f'm {m} bins/quadrant {kmax-kmin+1 if m else 1:3d} (log2 slope {kmin/max(m,1):+.1f}..{(kmax+1)/max(m,1):+.1f}): trips/ray {it.mean():.2f}  per-wave max {w.max(1).mean():.2f}')
