"""Analysis helper (not product, not oracle): NumPy emulations of the scan's candidate traversal schemes on oracle rollouts -
iterations per ray and per 64-lane wave.  One script, one sub-command per scheme (the numbers in EXPERIMENTS.md / the kernel
comments cite them by these names; rounds 1-2 had them as skip_stats{,2,4,9,10,11,12,13}.py):

    python tools/analysis/skip_stats.py blocks           [track]   block tables: iterations per ray / per 64-lane wave for block sizes and per-cell certificates
    python tools/analysis/skip_stats.py lanes            [track]   per-lane trip traces and wave cost model of the rectangle / single-cell loop
    python tools/analysis/skip_stats.py slowest          [track]   composition of the slowest lane of a wave
    python tools/analysis/skip_stats.py quadrant         [track]   per-cell, per-quadrant free rectangles chosen by a score (best_rect)
    python tools/analysis/skip_stats.py octant           [track]   octant planes against quadrant planes
    python tools/analysis/skip_stats.py firsttrip-angle  [track]   first-trip rectangles by angle bins, for the first trip only or for every trip
    python tools/analysis/skip_stats.py firsttrip-slope  [track]   first-trip rectangles by the log-slope bins the kernel takes from float bits
    python tools/analysis/skip_stats.py sector           [track]   first-trip rectangles certified only inside the bin's sector
"""
import sys
import numpy as np
sys.path.insert(0, '.')
from scipy import ndimage
from oracle import racecar_oracle as ro, c_oracle
from racing_dreamer_amd.track_assets import load_track


# ---------------------------------------------------------------- blocks: block tables: iterations per ray / per 64-lane wave for block sizes and per-cell certificates
def block_table(stop, shift):
    d = ndimage.distance_transform_cdt(~stop, metric='chessboard').astype(np.int32)
    h, w = stop.shape; bs = 1 << shift
    H, W = -(-h // bs) * bs, -(-w // bs) * bs
    pad = np.zeros((H, W), np.int32); pad[:h, :w] = d
    return pad.reshape(H // bs, bs, W // bs, bs).min(axis=(1, 3)).clip(0, 255)

def emulate_blocks(track, cars_xyth, shift, percell=False):
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

def main_blocks(argv):
    name = argv[1] if len(argv) > 1 else 'austria'
    t = load_track(name)
    cfg = ro.OracleConfig(num_envs=256, auto_reset=True)
    b = c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg, threads=8)
    b.reset(mode=1, seed=0)
    for k in range(30): o = b.step(b.random_actions(1, k))
    cars = np.stack([b.arr['x'], b.arr['y'], b.arr['theta']], 1).astype(np.float64)
    plain = (o['lidar']/0.05*1.27).ravel()
    print(name, 'plain DDA approx cells/ray', plain.mean())
    for label, kw in [('block 4', dict(shift=2)), ('block 8', dict(shift=3)), ('block 2', dict(shift=1)), ('per-cell field', dict(shift=2, percell=True))]:
        it = emulate_blocks(t, cars, **kw)
        w = it.reshape(-1, 1080)[:, :1024].reshape(-1, 64)
        print(f'  {label:16s} iters/ray mean {it.mean():.2f}  p50 {np.median(it):.0f}  p90 {np.percentile(it,90):.0f}  max {it.max()}   per-wave max mean {w.max(1).mean():.2f}  (wave max/mean {w.max(1).mean()/w.mean():.2f})')

# ---------------------------------------------------------------- lanes: per-lane trip traces and wave cost model of the rectangle / single-cell loop
def trace(track, cars, shift, scheme, minv=1):
    occ = track.occ.copy(); occ[0,:]=occ[-1,:]=occ[:,0]=occ[:,-1]=True
    d = ndimage.distance_transform_cdt(~occ, metric='chessboard').astype(np.int32)
    h, w = occ.shape; bs = 1 << shift
    H, W = -(-h // bs) * bs, -(-w // bs) * bs
    pad = np.zeros((H, W), np.int32); pad[:h, :w] = d
    blk = pad.reshape(H // bs, bs, W // bs, bs).min(axis=(1, 3))
    cb, sb = ro.beam_table()
    x, y, th = cars.T; ct, st = np.cos(th), np.sin(th)
    lx, ly = x + 0.25*ct, y + 0.25*st
    dx = (ct[:,None]*cb - st[:,None]*sb).ravel(); dy = (st[:,None]*cb + ct[:,None]*sb).ravel()
    gx = np.repeat((lx - track.origin[0])/0.05, 1080); gy = np.repeat((ly - track.origin[1])/0.05, 1080)
    n = len(gx); seqs = [[] for _ in range(n)]
    for i in range(n):
        px, py = dx[i] > 0, dy[i] > 0
        ix, iy = int(np.floor(gx[i])), int(np.floor(gy[i]))
        if occ[iy, ix]: continue
        idx = 1/dx[i] if dx[i] else 1e30; idy = 1/dy[i] if dy[i] else 1e30
        xmaj = abs(dx[i]) >= abs(dy[i])
        while True:
            v = blk[iy >> shift, ix >> shift]; r = v - 1
            if v >= minv + (1 if scheme == 'major' else 0) and v >= 1:
                seqs[i].append('A')
                if scheme == 'rect':
                    bx, by = ix & ~(bs-1), iy & ~(bs-1)
                    x0, x1, y0, y1 = bx - r, bx + bs + r, by - r, by + bs + r
                    xe = x1 if px else x0; ye = y1 if py else y0
                    txe = (xe - gx[i])*idx; tye = (ye - gy[i])*idy
                    if txe < tye:
                        tt = txe; ix = x1 if px else x0 - 1; iy = int(np.floor(gy[i] + tt*dy[i]))
                    else:
                        tt = tye; iy = y1 if py else y0 - 1; ix = int(np.floor(gx[i] + tt*dx[i]))
                else:   # advance r cells along the major axis
                    if xmaj:
                        ixn = ix + (r if px else -r); xb = ixn if px else ixn + 1
                        tt = (xb - gx[i])*idx; ix = ixn; iy = int(np.floor(gy[i] + tt*dy[i] + 1e-9*np.sign(dy[i])))
                    else:
                        iyn = iy + (r if py else -r); yb = iyn if py else iyn + 1
                        tt = (yb - gy[i])*idy; iy = iyn; ix = int(np.floor(gx[i] + tt*dx[i] + 1e-9*np.sign(dx[i])))
            else:
                seqs[i].append('B')
                bxn = ix + 1 if px else ix; byn = iy + 1 if py else iy
                tx = (bxn - gx[i])*idx; ty = (byn - gy[i])*idy
                if tx < ty: tt = tx; ix += 1 if px else -1
                else: tt = ty; iy += 1 if py else -1
            if tt >= 300: break
            ix = min(max(ix, 0), w-1); iy = min(max(iy, 0), h-1)
            if occ[iy, ix]: break
    return seqs

def wave_costs(seqs, cA, cB, cU):
    tot_u = tot_2 = 0; nw = 0
    for c in range(0, len(seqs), 1080):
        for w0 in range(0, 1024, 64):
            ws = [''.join(s) for s in seqs[c + w0: c + w0 + 64]]
            tot_u += max(len(s) for s in ws) * cU
            ptr = [0]*64; cost = 0
            while any(p < len(s) for p, s in zip(ptr, ws)):
                for ph, cc in (('A', cA), ('B', cB)):
                    runs = []
                    for k, s in enumerate(ws):
                        j = ptr[k]
                        while j < len(s) and s[j] == ph: j += 1
                        runs.append(j - ptr[k]); ptr[k] = j
                    cost += max(runs) * cc
            tot_2 += cost; nw += 1
    return tot_u / nw, tot_2 / nw

def main_lanes(argv):
    name = argv[1] if len(argv) > 1 else 'austria'
    t = load_track(name)
    cfg = ro.OracleConfig(num_envs=24, auto_reset=True)
    b = c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg, threads=8)
    b.reset(mode=1, seed=0)
    for k in range(30): b.step(b.random_actions(1, k))
    cars = np.stack([b.arr['x'], b.arr['y'], b.arr['theta']], 1).astype(np.float64)
    for shift in (2, 3):
        for scheme, minv in (('rect', 1), ('rect', 2), ('major', 1)):
            s = trace(t, cars, shift, scheme, minv)
            nA = np.mean([q.count('A') for q in s]); nB = np.mean([q.count('B') for q in s])
            for cA, cB, cU in ((60, 20, 65), (45, 20, 50)):
                u, two = wave_costs(s, cA, cB, cU)
                print(f'{name} block {1<<shift} {scheme:5s} minv {minv}: per ray A {nA:.2f} B {nB:.2f} | costs cA={cA} cB={cB}: unified {u:.0f}  two-phase {two:.0f}')

# ---------------------------------------------------------------- slowest: composition of the slowest lane of a wave
def main_slowest(argv):
    t = load_track(argv[1] if len(argv) > 1 else 'austria')
    cfg = ro.OracleConfig(num_envs=48, auto_reset=True)
    b = c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg, threads=8)
    b.reset(mode=1, seed=0)
    for k in range(30): b.step(b.random_actions(1, k))
    cars = np.stack([b.arr['x'], b.arr['y'], b.arr['theta']], 1).astype(np.float64)
    seqs = trace(t, cars, 2, 'rect', 1)
    A = np.array([s.count('A') for s in seqs]); B = np.array([s.count('B') for s in seqs]); T = A + B
    tot = []; 
    for c in range(0, len(seqs), 1080):
        for w0 in range(0, 1024, 64):
            sl = slice(c + w0, c + w0 + 64)
            j = np.argmax(T[sl]); tot.append((T[sl][j], A[sl][j], B[sl][j], T[sl].mean(), A[sl].max(), B[sl].max()))
    tot = np.array(tot, float)
    print('per-wave slowest lane: total %.2f = A %.2f + B %.2f ; wave mean %.2f ; max A over lanes %.2f ; max B over lanes %.2f' % tuple(tot.mean(0)))
    print('rays: mean A %.2f B %.2f; share of rays with B>=8: %.3f, with A>=8: %.3f' % (A.mean(), B.mean(), (B>=8).mean(), (A>=8).mean()))

# ---------------------------------------------------------------- quadrant: per-cell, per-quadrant free rectangles chosen by a score (best_rect)
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


def emulate_quadrant(track, cars, score):
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

def main_quadrant(argv):
    t = load_track(argv[1] if len(argv) > 1 else 'austria')
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
        it = emulate_quadrant(t, cars, sc)
        w = it.reshape(-1, 1080)[:, :1024].reshape(-1, 64)
        print(f'{name:12s}: trips/ray {it.mean():.2f}  per-wave max {w.max(1).mean():.2f}  p99 {np.percentile(it,99):.0f} max {it.max()}')

# ---------------------------------------------------------------- octant: octant planes against quadrant planes
def expd(angles):
    c = np.cos(np.radians(angles)); s_ = np.sin(np.radians(angles))
    return lambda w, h: sum(np.minimum(w / ci, h / si) for ci, si in zip(c, s_))


def emulate_octant(track, cars, nclass):
    occ = track.occ.copy(); occ[0,:]=occ[-1,:]=occ[:,0]=occ[:,-1]=True
    cb, sb = ro.beam_table()
    x, y, th = cars.T; ct, st = np.cos(th), np.sin(th)
    lx, ly = x + 0.25*ct, y + 0.25*st
    dx = (ct[:,None]*cb - st[:,None]*sb).ravel(); dy = (st[:,None]*cb + ct[:,None]*sb).ravel()
    gx = np.repeat((lx - track.origin[0])/0.05, 1080); gy = np.repeat((ly - track.origin[1])/0.05, 1080)
    ix = np.floor(gx).astype(int); iy = np.floor(gy).astype(int)
    n = len(ix); it = np.zeros(n, int); act = ~occ[iy, ix]
    px, py = dx > 0, dy > 0
    ang = np.degrees(np.arctan2(np.abs(dy), np.abs(dx)))          # 0..90 inside the quadrant
    sub = np.minimum((ang / (90.0 / nclass)).astype(int), nclass - 1)
    cls = (py.astype(int)*2 + px.astype(int)) * nclass + sub
    tabs = []
    for sy in (-1, 1):
        for sx in (-1, 1):
            for k in range(nclass):
                lo, hi = 90.0 / nclass * k, 90.0 / nclass * (k + 1)
                a = [lo + (hi - lo) * 0.25, lo + (hi - lo) * 0.75]
                tabs.append(best_rect(occ, sx, sy, expd(a)))
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

def main_octant(argv):
    t = load_track(argv[1] if len(argv) > 1 else 'austria')
    cfg = ro.OracleConfig(num_envs=256, auto_reset=True)
    b = c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg, threads=8)
    b.reset(mode=1, seed=0)
    for k in range(30): b.step(b.random_actions(1, k))
    cars = np.stack([b.arr['x'], b.arr['y'], b.arr['theta']], 1).astype(np.float64)
    for nclass in (1, 2, 4):
        it = emulate_octant(t, cars, nclass)
        w = it.reshape(-1, 1080)[:, :1024].reshape(-1, 64)
        print(f'{4*nclass:2d} planes: trips/ray {it.mean():.2f}  per-wave max {w.max(1).mean():.2f}  p99 {np.percentile(it,99):.0f} max {it.max()}')

# ---------------------------------------------------------------- firsttrip-angle: first-trip rectangles by angle bins, for the first trip only or for every trip
_cache_firsttrip_angle = {}


def tables_firsttrip_angle(occ, nclass):
    if nclass not in _cache_firsttrip_angle:
        tabs = []
        for sy in (-1, 1):
            for sx in (-1, 1):
                for k in range(nclass):
                    lo, hi = 90.0 / nclass * k, 90.0 / nclass * (k + 1)
                    tabs.append(best_rect(occ, sx, sy, expd([lo + (hi - lo) * 0.25, lo + (hi - lo) * 0.75])))
        _cache_firsttrip_angle[nclass] = (np.stack([t[0] for t in tabs]), np.stack([t[1] for t in tabs]))
    return _cache_firsttrip_angle[nclass]


def emulate_firsttrip_angle(track, cars, nfirst, nrest):
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
        sw, sh = tables_firsttrip_angle(occ, nclass)
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

def main_firsttrip_angle(argv):
    t = load_track(argv[1] if len(argv) > 1 else 'austria')
    cfg = ro.OracleConfig(num_envs=256, auto_reset=True)
    b = c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg, threads=8)
    b.reset(mode=1, seed=0)
    for k in range(30): b.step(b.random_actions(1, k))
    cars = np.stack([b.arr['x'], b.arr['y'], b.arr['theta']], 1).astype(np.float64)
    for nfirst, nrest in ((1, 1), (2, 1), (4, 1), (8, 1), (16, 1), (2, 2), (8, 2)):
        it = emulate_firsttrip_angle(t, cars, nfirst, nrest)
        w = it.reshape(-1, 1080)[:, :1024].reshape(-1, 64)
        print(f'first {4*nfirst:3d} planes, rest {4*nrest:2d}: trips/ray {it.mean():.2f}  per-wave max {w.max(1).mean():.2f}')

# ---------------------------------------------------------------- firsttrip-slope: first-trip rectangles by the log-slope bins the kernel takes from float bits
def emulate_firsttrip_slope(track, cars, m, kmin, kmax, tabs4):
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

def main_firsttrip_slope(argv):
    t = load_track(argv[1] if len(argv) > 1 else 'austria')
    cfg = ro.OracleConfig(num_envs=256, auto_reset=True)
    b = c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg, threads=8)
    b.reset(mode=1, seed=0)
    for k in range(30): b.step(b.random_actions(1, k))
    cars = np.stack([b.arr['x'], b.arr['y'], b.arr['theta']], 1).astype(np.float64)
    occ = t.occ.copy(); occ[0,:]=occ[-1,:]=occ[:,0]=occ[:,-1]=True
    t4 = [best_rect(occ, sx, sy, expd([22.5, 67.5])) for sy in (-1, 1) for sx in (-1, 1)]
    tabs4 = (np.stack([a[0] for a in t4]), np.stack([a[1] for a in t4]))
    for m, kmin, kmax in ((0, 0, 0), (1, -4, 3), (1, -8, 7), (2, -8, 7), (2, -16, 15), (4, -16, 15), (4, -32, 31)):
        it = emulate_firsttrip_slope(t, cars, m, kmin, kmax, tabs4)
        w = it.reshape(-1, 1080)[:, :1024].reshape(-1, 64)
        print(f'm {m} bins/quadrant {kmax-kmin+1 if m else 1:3d} (log2 slope {kmin/max(m,1):+.1f}..{(kmax+1)/max(m,1):+.1f}): trips/ray {it.mean():.2f}  per-wave max {w.max(1).mean():.2f}')

# ---------------------------------------------------------------- sector: first-trip rectangles certified only inside the bin's sector
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


def emulate_sector(track, cars, m, kmin, kmax, tabs4, sector=True):
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

def main_sector(argv):
    t = load_track(argv[1] if len(argv) > 1 else 'austria')
    cfg = ro.OracleConfig(num_envs=128, auto_reset=True)
    b = c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg, threads=8)
    b.reset(mode=1, seed=0)
    for k in range(30): b.step(b.random_actions(1, k))
    cars = np.stack([b.arr['x'], b.arr['y'], b.arr['theta']], 1).astype(np.float64)
    occ = t.occ.copy(); occ[0,:]=occ[-1,:]=occ[:,0]=occ[:,-1]=True
    t4 = [best_rect(occ, sx, sy, expd([22.5, 67.5])) for sy in (-1, 1) for sx in (-1, 1)]
    tabs4 = (np.stack([a[0] for a in t4]), np.stack([a[1] for a in t4]))
    for m, kmin, kmax in ((2, -8, 7), (4, -16, 15), (8, -32, 31)):
        it = emulate_sector(t, cars, m, kmin, kmax, tabs4)
        w = it.reshape(-1, 1080)[:, :1024].reshape(-1, 64)
        print(f'sector-free first trip, m {m} bins/quadrant {kmax-kmin+1:3d}: trips/ray {it.mean():.2f}  per-wave max {w.max(1).mean():.2f}  p99 {np.percentile(it, 99):.0f} max {it.max()}')

COMMANDS = {'blocks': main_blocks, 'lanes': main_lanes, 'slowest': main_slowest, 'quadrant': main_quadrant, 'octant': main_octant, 'firsttrip-angle': main_firsttrip_angle, 'firsttrip-slope': main_firsttrip_slope, 'sector': main_sector}

if __name__ == '__main__':
    if len(sys.argv) < 2 or sys.argv[1] not in COMMANDS:
        sys.exit(__doc__)
    COMMANDS[sys.argv[1]]([sys.argv[0]] + sys.argv[2:])
