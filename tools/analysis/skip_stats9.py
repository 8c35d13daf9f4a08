"""Analysis helper: per (cell, quadrant) anchored free RECTANGLE (w, h stored separately), chosen by a score."""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tools/analysis')
from oracle import racecar_oracle as ro, c_oracle
from racing_dreamer_amd.track_assets import load_track


def best_rect(occ, sx, sy, score, cap=255):
    H, W = occ.shape
    o = occ[::-1] if sy < 0 else occ
    o = o[:, ::-1] if sx < 0 else o
    free = ~o
    runx = np.zeros((H, W + 1), np.int32)
    for x in range(W - 1, -1, -1):
        runx[:, x] = np.where(free[:, x], np.minimum(runx[:, x + 1] + 1, cap), 0)
    runx = runx[:, :W]
    bw = np.zeros((H, W), np.int32); bh = np.zeros((H, W), np.int32); bs = np.full((H, W), -1.0)
    cur = np.full((H, W), cap, np.int32)
    pad = np.zeros((cap, W), np.int32)
    rp = np.concatenate([runx, pad], 0)
    for h in range(1, cap + 1):
        cur = np.minimum(cur, rp[h - 1:h - 1 + H])
        if not cur.any(): break
        s = score(cur.astype(np.float64), float(h))
        better = (s > bs) & (cur > 0)
        bw[better] = cur[better]; bh[better] = h; bs[better] = s[better]
    def back(k):
        k = k[:, ::-1] if sx < 0 else k
        return k[::-1] if sy < 0 else k
    return back(bw), back(bh)


def emulate(track, cars, score):
    occ = track.occ.copy(); occ[0,:]=occ[-1,:]=occ[:,0]=occ[:,-1]=True
    cb, sb = ro.beam_table()
    x, y, th = cars.T; ct, st = np.cos(th), np.sin(th)
    lx, ly = x + 0.25*ct, y + 0.25*st
    dx = (ct[:,None]*cb - st[:,None]*sb).ravel(); dy = (st[:,None]*cb + ct[:,None]*sb).ravel()
    gx = np.repeat((lx - track.origin[0])/0.05, 1080); gy = np.repeat((ly - track.origin[1])/0.05, 1080)
    ix = np.floor(gx).astype(int); iy = np.floor(gy).astype(int)
    n = len(ix); it = np.zeros(n, int); act = ~occ[iy, ix]
    px, py = dx > 0, dy > 0
    cls = py.astype(int)*2 + px.astype(int)
    tabs = [best_rect(occ, sx, sy, score) for sy in (-1, 1) for sx in (-1, 1)]
    sw = np.stack([t[0] for t in tabs]); sh = np.stack([t[1] for t in tabs])
    idx, idy = 1/np.where(dx==0,1e-30,dx), 1/np.where(dy==0,1e-30,dy)
    while act.any():
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
    def expd(angles):
        c = np.cos(np.radians(angles)); s_ = np.sin(np.radians(angles))
        return lambda w, h: sum(np.minimum(w / ci, h / si) for ci, si in zip(c, s_))
    scores = {'exp4': expd([11.25, 33.75, 56.25, 78.75]), 'exp8': expd(np.arange(8) * 11.25 + 5.6),
              'exp2': expd([22.5, 67.5]),               'square': lambda w, h: np.minimum(w, h), 'area': lambda w, h: w*h,
              'min+0.25max': lambda w, h: np.minimum(w, h) + 0.25*np.maximum(w, h),
              'perimeter': lambda w, h: w + h,
              'min+0.5max': lambda w, h: np.minimum(w, h) + 0.5*np.maximum(w, h)}
    for name, sc in scores.items():
        it = emulate(t, cars, sc)
        w = it.reshape(-1, 1080)[:, :1024].reshape(-1, 64)
        print(f'{name:12s}: trips/ray {it.mean():.2f}  per-wave max {w.max(1).mean():.2f}  p99 {np.percentile(it,99):.0f} max {it.max()}')
