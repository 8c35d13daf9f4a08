"""Analysis helper: per-ray phase strings (A = certified skip, B = single-cell step) for candidate traversal
schemes, and the cost of a wave of 64 rays under (i) a unified loop, (ii) a two-phase loop."""
import sys
import numpy as np
sys.path.insert(0, '.')
from scipy import ndimage
from oracle import racecar_oracle as ro, c_oracle
from racing_dreamer_amd.track_assets import load_track

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

if __name__ == '__main__':
    name = sys.argv[1] if len(sys.argv) > 1 else 'austria'
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
