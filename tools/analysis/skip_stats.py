"""Analysis helper (not product, not oracle): emulate the free-rectangle skipping traversal in NumPy on a
sample of rays and report iterations per ray / per 64-lane wave for different block sizes."""
import sys
import numpy as np
sys.path.insert(0, '.')
from scipy import ndimage
from oracle import racecar_oracle as ro, c_oracle
from racing_dreamer_amd.track_assets import load_track

def block_table(stop, shift):
    d = ndimage.distance_transform_cdt(~stop, metric='chessboard').astype(np.int32)
    h, w = stop.shape; bs = 1 << shift
    H, W = -(-h // bs) * bs, -(-w // bs) * bs
    pad = np.zeros((H, W), np.int32); pad[:h, :w] = d
    return pad.reshape(H // bs, bs, W // bs, bs).min(axis=(1, 3)).clip(0, 255)

def emulate(track, cars_xyth, shift, percell=False):
    occ = track.occ.copy(); occ[0,:]=occ[-1,:]=occ[:,0]=occ[:,-1]=True
    blk = block_table(occ, shift); bs = 1 << shift
    dcell = ndimage.distance_transform_cdt(~occ, metric='chessboard').astype(np.int32)
    cb, sb = ro.beam_table()
    x, y, th = cars_xyth.T
    ct, st = np.cos(th), np.sin(th)
    lx, ly = x + 0.25*ct, y + 0.25*st
    dx = (ct[:,None]*cb - st[:,None]*sb).ravel(); dy = (st[:,None]*cb + ct[:,None]*sb).ravel()
    gx = np.repeat((lx - track.origin[0])/0.05, 1080); gy = np.repeat((ly - track.origin[1])/0.05, 1080)
    ix = np.floor(gx).astype(int); iy = np.floor(gy).astype(int)
    n = len(ix); iters = np.zeros(n, int); act = np.ones(n, bool)
    act &= ~occ[iy, ix]
    px, py = dx > 0, dy > 0
    idx, idy = 1/np.where(dx==0,1e-30,dx), 1/np.where(dy==0,1e-30,dy)
    while act.any():
        a = np.nonzero(act)[0]
        iters[a] += 1
        if percell:
            v = dcell[iy[a], ix[a]]; r = v - 1
            x0 = ix[a] - r; x1 = ix[a] + 1 + r; y0 = iy[a] - r; y1 = iy[a] + 1 + r
        else:
            v = blk[iy[a] >> shift, ix[a] >> shift]; r = v - 1
            bx, by = ix[a] & ~(bs-1), iy[a] & ~(bs-1)
            x0 = np.where(v > 0, bx - r, ix[a]); x1 = np.where(v > 0, bx + bs + r, ix[a] + 1)
            y0 = np.where(v > 0, by - r, iy[a]); y1 = np.where(v > 0, by + bs + r, iy[a] + 1)
        xe = np.where(px[a], x1, x0); ye = np.where(py[a], y1, y0)
        txe = (xe - gx[a]) * idx[a]; tye = (ye - gy[a]) * idy[a]
        xexit = txe < tye
        tt = np.where(xexit, txe, tye)
        over = tt >= 300
        nx = np.where(xexit, np.where(px[a], x1, x0 - 1), np.floor(gx[a] + tt*dx[a] + 1e-9*np.sign(dx[a])).astype(int))
        ny = np.where(xexit, np.floor(gy[a] + tt*dy[a] + 1e-9*np.sign(dy[a])).astype(int), np.where(py[a], y1, y0 - 1))
        nx = np.clip(nx, 0, track.width-1); ny = np.clip(ny, 0, track.height-1)
        ix[a], iy[a] = nx, ny
        stop = over | occ[ny, nx]
        act[a[stop]] = False
    return iters

if __name__ == '__main__':
    name = sys.argv[1] if len(sys.argv) > 1 else 'austria'
    t = load_track(name)
    cfg = ro.OracleConfig(num_envs=256, auto_reset=True)
    b = c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg, threads=8)
    b.reset(mode=1, seed=0)
    for k in range(30): o = b.step(b.random_actions(1, k))
    cars = np.stack([b.arr['x'], b.arr['y'], b.arr['theta']], 1).astype(np.float64)
    plain = (o['lidar']/0.05*1.27).ravel()
    print(name, 'plain DDA approx cells/ray', plain.mean())
    for label, kw in [('block 4', dict(shift=2)), ('block 8', dict(shift=3)), ('block 2', dict(shift=1)), ('per-cell field', dict(shift=2, percell=True))]:
        it = emulate(t, cars, **kw)
        w = it.reshape(-1, 1080)[:, :1024].reshape(-1, 64)
        print(f'  {label:16s} iters/ray mean {it.mean():.2f}  p50 {np.median(it):.0f}  p90 {np.percentile(it,90):.0f}  max {it.max()}   per-wave max mean {w.max(1).mean():.2f}  (wave max/mean {w.max(1).mean()/w.mean():.2f})')
