"""Analysis helper: composition (rectangle skips A vs single-cell steps B) of the slowest lane of each wave."""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tools/analysis')
from skip_stats2 import trace
from oracle import racecar_oracle as ro, c_oracle
from racing_dreamer_amd.track_assets import load_track
t = load_track(sys.argv[1] if len(sys.argv) > 1 else 'austria')
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
