"""Analysis helper: utilisation of a 64-lane wave that works through a car's 1080 rays with batched lane refill
(persistent-lane scheduling) versus the static 64-consecutive-beams mapping."""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tools/analysis')
from skip_stats import emulate
from oracle import racecar_oracle as ro, c_oracle
from racing_dreamer_amd.track_assets import load_track

def static_cost(it, c_iter, c_setup):
    tot = 0
    for car in it.reshape(-1, 1080):
        for w in range(0, 1080, 64):
            tot += car[w:w+64].max() * c_iter + c_setup
    return tot

def refill_cost(it, c_iter, c_setup, thresh, order='interleave'):
    tot = 0
    for car in it.reshape(-1, 1080):
        queue = list(car)            # rays in beam order; lanes take the next unassigned ray
        rem = np.zeros(64, int)      # remaining iterations per lane
        qi = 0
        while True:
            idle = np.nonzero(rem == 0)[0]
            if qi < len(queue) and (len(idle) >= thresh or len(idle) == 64):
                k = min(len(idle), len(queue) - qi)
                rem[idle[:k]] = queue[qi:qi + k]; qi += k
                tot += c_setup
                continue
            if rem.max() == 0:
                break
            # run until enough lanes are idle (or everything finished)
            active = np.sort(rem[rem > 0])
            if qi < len(queue):
                need = thresh - (64 - len(active))
                steps = active[min(max(need, 1), len(active)) - 1]
            else:
                steps = active[-1]
            rem = np.maximum(rem - steps, 0)
            tot += steps * c_iter
    return tot

if __name__ == '__main__':
    t = load_track('austria')
    cfg = ro.OracleConfig(num_envs=128, auto_reset=True)
    b = c_oracle.COracleEnv(t.occ, t.drivable, t.progress, t.centerline, t.origin, t.resolution, cfg, threads=8)
    b.reset(mode=1, seed=0)
    for k in range(30): b.step(b.random_actions(1, k))
    cars = np.stack([b.arr['x'], b.arr['y'], b.arr['theta']], 1).astype(np.float64)
    it = emulate(t, cars, 2)
    it = np.maximum(it, 1)
    for c_iter, c_setup in ((70, 120), (70, 60)):
        ideal = it.sum() / 64 * c_iter + it.size / 64 * c_setup
        s = static_cost(it, c_iter, c_setup)
        print(f'c_iter {c_iter} c_setup {c_setup}: ideal {ideal/len(cars):.0f}  static {s/len(cars):.0f} ({s/ideal:.2f}x ideal)')
        for th in (8, 16, 24, 32, 48):
            r = refill_cost(it, c_iter, c_setup, th)
            print(f'    refill thresh {th}: {r/len(cars):.0f} ({r/ideal:.2f}x ideal, {s/r:.2f}x faster than static)')
