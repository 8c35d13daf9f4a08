"""Analysis helper: anchored free rectangles per direction class (quadrant squares, octant a:1 rectangles):
trips per ray / per-wave maximum."""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
from oracle import racecar_oracle as ro, c_oracle
from racing_dreamer_amd.track_assets import load_track


def anchored(occ, sx, sy, ax, ay, cap=255):
    """k[y, x] = largest k such that the rectangle of ax*k by ay*k cells with corner cell (x, y), extending
    towards (sx, sy), is free."""
    H, W = occ.shape
    o = occ[::-1] if sy < 0 else occ
    o = o[:, ::-1] if sx < 0 else o          # now the rectangle extends towards +x, +y
    # run lengths of free cells towards +x and +y, then k = largest n with min over the n*ay rows of runx >= n*ax:
    # computed by brute force with an integral image on slices
    cap = min(cap, max(H, W))
    big = cap * max(ax, ay) + 1
    p = np.ones((H + big, W + big), np.int32); p[:H, :W] = o
    S = np.zeros((H + big + 1, W + big + 1), np.int32); S[1:, 1:] = p.cumsum(0).cumsum(1)
    k = np.zeros((H, W), np.int32)
    alive = ~o.copy()
    for n in range(1, cap + 1):
        w, h = ax * n, ay * n
        s = S[h:h + H, w:w + W] - S[0:H, w:w + W] - S[h:h + H, 0:W] + S[0:H, 0:W]
        alive &= (s == 0)
        if not alive.any(): break
        k[alive] = n
    k = k[:, ::-1] if sx < 0 else k
    k = k[::-1] if sy < 0 else k
    return k


def emulate(track, cars, mode):
    occ = track.occ.copy(); occ[0,:]=occ[-1,:]=occ[:,0]=occ[:,-1]=True
    cb, sb = ro.beam_table()
    x, y, th = cars.T; ct, st = np.cos(th), np.sin(th)
    lx, ly = x + 0.25*ct, y + 0.25*st
    dx = (ct[:,None]*cb - st[:,None]*sb).ravel(); dy = (st[:,None]*cb + ct[:,None]*sb).ravel()
    gx = np.repeat((lx - track.origin[0])/0.05, 1080); gy = np.repeat((ly - track.origin[1])/0.05, 1080)
    ix = np.floor(gx).astype(int); iy = np.floor(gy).astype(int)
    n = len(ix); it = np.zeros(n, int); act = ~occ[iy, ix]
    px, py = dx > 0, dy > 0
    a = mode
    if a == 1:
        cls = py.astype(int)*2 + px.astype(int)
        tabs = [anchored(occ, sx, sy, 1, 1) for sy in (-1, 1) for sx in (-1, 1)]
        mulx = np.ones(n, int); muly = np.ones(n, int)
    else:
        xmaj = np.abs(dx) >= np.abs(dy)
        cls = xmaj.astype(int)*4 + py.astype(int)*2 + px.astype(int)
        tabs = [anchored(occ, sx, sy, (a if m else 1), (1 if m else a)) for m in (0, 1) for sy in (-1, 1) for sx in (-1, 1)]
        mulx = np.where(xmaj, a, 1); muly = np.where(xmaj, 1, a)
    stack = np.stack(tabs)
    idx, idy = 1/np.where(dx==0,1e-30,dx), 1/np.where(dy==0,1e-30,dy)
    while act.any():
        a_ = np.nonzero(act)[0]; it[a_] += 1
        k = stack[cls[a_], iy[a_], ix[a_]]
        rx = np.maximum(k*mulx[a_] - 1, 0); ry = np.maximum(k*muly[a_] - 1, 0)
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
    for mode in (1, 2, 3, 4):
        it = emulate(t, cars, mode)
        w = it.reshape(-1, 1080)[:, :1024].reshape(-1, 64)
        print(f'aspect {mode}: trips/ray {it.mean():.2f}  per-wave max {w.max(1).mean():.2f}  p99 {np.percentile(it,99):.0f} max {it.max()}')
