"""Analysis helper: wave-per-car scan with lane refill (a lane takes the car's next beam as soon as enough lanes are
idle) against fixed rounds of 64 beams.  Costs in VALU wave-instructions: SETUP per (re)fill pass, TRIP per trip."""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import skip_stats9 as s9
from oracle import racecar_oracle as ro, c_oracle
from racing_dreamer_amd.track_assets import load_track

SETUP, TRIP = 41, 23


def expd(angles):
    c = np.cos(np.radians(angles)); s_ = np.sin(np.radians(angles))
    return lambda w, h: sum(np.minimum(w / ci, h / si) for ci, si in zip(c, s_))


def fixed_rounds(trips):
    cost = 0
    for r in range(0, 1080, 64):
        cost += SETUP + TRIP * trips[r:r + 64].max()
    return cost


def refill(trips, threshold):
    lanes = np.zeros(64, int)        # remaining trips per lane (0 = idle)
    nxt, cost = 0, 0
    while True:
        idle = lanes == 0
        if nxt < 1080 and (idle.sum() >= threshold or idle.all()):
            k = min(int(idle.sum()), 1080 - nxt)
            sel = np.nonzero(idle)[0][:k]
            lanes[sel] = trips[nxt:nxt + k]      # rays that start in a stop cell have 0 trips: stay idle (fine)
            nxt += k
            cost += SETUP
            continue
        if (lanes == 0).all():
            if nxt >= 1080: break
            continue
        lanes[lanes > 0] -= 1
        cost += TRIP
    return cost


if __name__ == '__main__':
    t = load_track(sys.argv[1] if len(sys.argv) > 1 else 'austria')
    cfg = ro.OracleConfig(num_envs=128, auto_reset=True)
    b = c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg, threads=8)
    b.reset(mode=1, seed=0)
    for k in range(30): b.step(b.random_actions(1, k))
    cars = np.stack([b.arr['x'], b.arr['y'], b.arr['theta']], 1).astype(np.float64)
    orig = s9.best_rect
    s9.best_rect = lambda occ, sx, sy, score, cap=127: orig(occ, sx, sy, score, cap)
    it = s9.emulate(t, cars, expd([22.5, 67.5])).reshape(len(cars), 1080)
    base = np.mean([fixed_rounds(r) for r in it])
    ideal = np.mean([17 * SETUP + TRIP * r.sum() / 64 for r in it])
    print(f'fixed rounds: {base:.0f} VALU per car; perfect packing bound {ideal:.0f}')
    for th in (1, 8, 16, 24, 32, 48, 64):
        c = np.mean([refill(r, th) for r in it])
        print(f'refill when >= {th:2d} lanes idle: {c:.0f} ({base / c:.2f}x)')
