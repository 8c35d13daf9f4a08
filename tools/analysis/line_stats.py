"""Analysis helper: distinct L1 lines (64 table entries of 2 B) touched per wave-level table load, for row-major
and tiled layouts of the per-quadrant rectangle table (one wave = 64 consecutive beams of one car)."""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tools/analysis')
import skip_stats as s9
from oracle import racecar_oracle as ro, c_oracle
from racing_dreamer_amd.track_assets import load_track


def expd(angles):
    c = np.cos(np.radians(angles)); s_ = np.sin(np.radians(angles))
    return lambda w, h: sum(np.minimum(w / ci, h / si) for ci, si in zip(c, s_))


def run(track, cars):
    occ = track.occ.copy(); occ[0,:]=occ[-1,:]=occ[:,0]=occ[:,-1]=True
    cb, sb = ro.beam_table()
    x, y, th = cars.T; ct, st = np.cos(th), np.sin(th)
    lx, ly = x + 0.25*ct, y + 0.25*st
    dx = (ct[:,None]*cb - st[:,None]*sb).ravel(); dy = (st[:,None]*cb + ct[:,None]*sb).ravel()
    gx = np.repeat((lx - track.origin[0])/0.05, 1080); gy = np.repeat((ly - track.origin[1])/0.05, 1080)
    ix = np.floor(gx).astype(int); iy = np.floor(gy).astype(int)
    n = len(ix); act = ~occ[iy, ix]
    px, py = dx > 0, dy > 0
    cls = py.astype(int)*2 + px.astype(int)
    tabs = [s9.best_rect(occ, sx, sy, expd([22.5, 67.5]), 127) for sy in (-1, 1) for sx in (-1, 1)]
    sw = np.stack([t[0] for t in tabs]); sh = np.stack([t[1] for t in tabs])
    idx, idy = 1/np.where(dx==0,1e-30,dx), 1/np.where(dy==0,1e-30,dy)
    beam = np.tile(np.arange(1080), len(cars)); car = np.repeat(np.arange(len(cars)), 1080)
    wave = car * 17 + beam // 64
    layouts = {'row-major 64x1': lambda X, Y: (Y * 4096 + X) // 64 * 1,
               'tiles 8x8': lambda X, Y: (Y // 8) * 4096 + X // 8,
               'tiles 16x4': lambda X, Y: (Y // 4) * 4096 + X // 16,
               'tiles 4x16': lambda X, Y: (Y // 16) * 4096 + X // 4,
               'tiles 32x2': lambda X, Y: (Y // 2) * 4096 + X // 32}
    tot = {k: 0 for k in layouts}; loads = 0
    while act.any():
        a_ = np.nonzero(act)[0]
        rx = sw[cls[a_], iy[a_], ix[a_]] - 1; ry = sh[cls[a_], iy[a_], ix[a_]] - 1
        xe = np.where(px[a_], ix[a_] + 1 + rx, ix[a_] - rx); ye = np.where(py[a_], iy[a_] + 1 + ry, iy[a_] - ry)
        txe = (xe - gx[a_])*idx[a_]; tye = (ye - gy[a_])*idy[a_]
        xexit = txe < tye; tt = np.where(xexit, txe, tye)
        nx = np.where(xexit, np.where(px[a_], xe, xe - 1), np.floor(gx[a_] + tt*dx[a_] + 1e-9*np.sign(dx[a_])).astype(int))
        ny = np.where(xexit, np.floor(gy[a_] + tt*dy[a_] + 1e-9*np.sign(dy[a_])).astype(int), np.where(py[a_], ye, ye - 1))
        nx = np.clip(nx, 0, track.width-1); ny = np.clip(ny, 0, track.height-1)
        ix[a_], iy[a_] = nx, ny
        # the load at the end of this trip: lanes a_ read cell (nx, ny) of plane cls
        w = wave[a_]
        loads += len(np.unique(w))
        for k, f in layouts.items():
            line = f(nx, ny) * 4 + cls[a_]
            tot[k] += len(np.unique(w.astype(np.int64) * (1 << 40) + line))
        act[a_[occ[ny, nx]]] = False
    return loads, tot

if __name__ == '__main__':
    t = load_track(sys.argv[1] if len(sys.argv) > 1 else 'austria')
    cfg = ro.OracleConfig(num_envs=256, auto_reset=True)
    b = c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg, threads=8)
    b.reset(mode=1, seed=0)
    for k in range(30): b.step(b.random_actions(1, k))
    cars = np.stack([b.arr['x'], b.arr['y'], b.arr['theta']], 1).astype(np.float64)
    loads, tot = run(t, cars)
    print(f'wave-level loads per car {loads / len(cars):.1f}')
    for k, v in tot.items(): print(f'{k:16s}: {v / loads:.2f} lines per load, {v / len(cars):.0f} per car')
