"""Analysis helper: per-cell certificates with the distance capped (4-bit / 5-bit storage): trips per ray / wave."""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
from scipy import ndimage
from oracle import racecar_oracle as ro, c_oracle
from racing_dreamer_amd.track_assets import load_track

def emulate(track, cars, cap):
    occ = track.occ.copy(); occ[0,:]=occ[-1,:]=occ[:,0]=occ[:,-1]=True
    d = np.minimum(ndimage.distance_transform_cdt(~occ, metric='chessboard').astype(np.int32), cap)
    cb, sb = ro.beam_table()
    x, y, th = cars.T; ct, st = np.cos(th), np.sin(th)
    lx, ly = x + 0.25*ct, y + 0.25*st
    dx = (ct[:,None]*cb - st[:,None]*sb).ravel(); dy = (st[:,None]*cb + ct[:,None]*sb).ravel()
    gx = np.repeat((lx - track.origin[0])/0.05, 1080); gy = np.repeat((ly - track.origin[1])/0.05, 1080)
    ix = np.floor(gx).astype(int); iy = np.floor(gy).astype(int)
    n = len(ix); it = np.zeros(n, int); act = ~occ[iy, ix]
    px, py = dx > 0, dy > 0
    idx, idy = 1/np.where(dx==0,1e-30,dx), 1/np.where(dy==0,1e-30,dy)
    while act.any():
        a = np.nonzero(act)[0]; it[a] += 1
        r = d[iy[a], ix[a]] - 1
        xe = np.where(px[a], ix[a] + 1 + r, ix[a] - r); ye = np.where(py[a], iy[a] + 1 + r, iy[a] - r)
        txe = (xe - gx[a])*idx[a]; tye = (ye - gy[a])*idy[a]
        xexit = txe < tye; tt = np.where(xexit, txe, tye)
        nx = np.where(xexit, np.where(px[a], xe, xe - 1), np.floor(gx[a] + tt*dx[a] + 1e-9*np.sign(dx[a])).astype(int))
        ny = np.where(xexit, np.floor(gy[a] + tt*dy[a] + 1e-9*np.sign(dy[a])).astype(int), np.where(py[a], ye, ye - 1))
        nx = np.clip(nx, 0, track.width-1); ny = np.clip(ny, 0, track.height-1)
        ix[a], iy[a] = nx, ny
        act[a[occ[ny, nx]]] = False
    return it

if __name__ == '__main__':
    t = load_track(sys.argv[1] if len(sys.argv) > 1 else 'austria')
    cfg = ro.OracleConfig(num_envs=256, auto_reset=True)
    b = c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg, threads=8)
    b.reset(mode=1, seed=0)
    for k in range(30): b.step(b.random_actions(1, k))
    cars = np.stack([b.arr['x'], b.arr['y'], b.arr['theta']], 1).astype(np.float64)
    for cap in (255, 31, 15, 7):
        it = emulate(t, cars, cap)
        w = it.reshape(-1, 1080)[:, :1024].reshape(-1, 64)
        print(f'cap {cap:3d}: trips/ray {it.mean():.2f}  per-wave max {w.max(1).mean():.2f}')
