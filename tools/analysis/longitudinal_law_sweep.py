"""sweep the FORM of the longitudinal law with the reference's trained agents (G10's machinery).

The reference's Dreamer checkpoints (ros_agent/checkpoints) carry a reward head trained on the reference simulator's rewards
(100 x progress per agent step at action_repeat 4).  Driving here it predicts 0.25 per agent step where this env pays 0.15
(DESIGN.md 2.2): per agent step and at the same commands the reference car covers ~1.66 x more track.  Candidates, each run
with the austria agent on austria (8 cars from the grid, sampled policy, action_repeat 4, NumPy oracle with the law patched):
  L0  the spec: dv/dt = sign(m) |m| A - D v, A = 4, D = A / MAX_VEL (throttle m settles at m x MAX_VEL)
  L1  force-limited drive TOWARD max_velocity for any m > 0 (SURVEY.md appendix A's recollection of racecar_gym's motor:
      velocity-control joint, target max_velocity, force |m| x max_force): dv/dt = |m| A - d v while v < MAX_VEL (m > 0),
      - |m| A - d v towards 0 (m < 0); small drag d
  L2  L0 with the top speed free (MAX_VEL' in place of 5 m/s, D = A / MAX_VEL')
Reported: wall contacts, mean speed, mean motor command, reward paid / predicted per agent step and their ratio, correlation,
the world model's surprise KL(posterior || prior) in nats, lap + progress of the first cars.
    python tools/analysis/longitudinal_law_sweep.py [steps] > profiles/r04_h_longitudinal_law_sweep.txt
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import make_oracle                                   # noqa: E402
from oracle import racecar_oracle as ro                           # noqa: E402
from oracle.dreamer_policy_port import DreamerPolicy, elu, softplus   # noqa: E402
from racing_dreamer_amd.track_assets import load_track            # noqa: E402

f32 = np.float32
GOLDEN = os.path.join(ROOT, "tests", "golden")


def prior_stats(policy, state_prev, action_prev):
    """img_step of the RSSM (ros_agent/models/dreamer/models.py:78-87): the prior over the next stochastic state."""
    w = policy.w
    x = elu(np.concatenate([state_prev["stoch"], action_prev], 1) @ w["img1_w"] + w["img1_b"])
    deter = policy._gru(x, state_prev["deter"])
    x = elu(deter @ w["img2_w"] + w["img2_b"])
    x = x @ w["img3_w"] + w["img3_b"]
    return x[:, :30], softplus(x[:, 30:]) + f32(0.1)


def posterior_stats(policy, state_prev, action_prev, scan):
    w = policy.w
    x = elu(np.concatenate([state_prev["stoch"], action_prev], 1) @ w["img1_w"] + w["img1_b"])
    deter = policy._gru(x, state_prev["deter"])
    x = elu(np.concatenate([deter, policy.preprocess(scan)], 1) @ w["obs1_w"] + w["obs1_b"])
    x = x @ w["obs2_w"] + w["obs2_b"]
    return x[:, :30], softplus(x[:, 30:]) + f32(0.1)


def kl_normal(mp, sp, mq, sq):
    return (np.log(sq / sp) + (sp ** 2 + (mp - mq) ** 2) / (2 * sq ** 2) - 0.5).sum(1)


def patched_substep(law):
    """OracleRaceEnv._substep with the longitudinal update replaced by `law(m, v) -> (acc, v_new)`."""
    orig_clamp = ro.clamp32

    def _substep(self, envs, motor, steer):
        # run the original sub-step, then redo the speed-dependent part with the candidate law: simplest is to patch the
        # constants the original reads and intercept clamp32 on the speed update - instead the law is applied by rewriting
        # v before the original's integration of the pose: the original computes v itself, so it is disabled (A = D = 0,
        # the clamp kept) and the candidate's v is installed first
        A = self.A
        for a in range(A):
            c = envs * A + a
            m = motor[:, a]
            self.v[c] = law(m, self.v[c])
        saved = ro.ACCEL_MAX, ro.DRAG
        ro.ACCEL_MAX, ro.DRAG = f32(0.0), f32(0.0)
        try:
            return _substep.orig(self, envs, motor, steer)
        finally:
            ro.ACCEL_MAX, ro.DRAG = saved
    _substep.orig = ro.OracleRaceEnv._substep
    return _substep


def law_L0(A, vmax):
    D = A / vmax
    return lambda m, v: np.clip(v + (np.where(m >= 0, np.abs(m) * A, -np.abs(m) * A) - D * v) * 0.01, 0.0, vmax).astype(f32)


def law_L1(A, d, vmax=5.0):
    def f(m, v):
        acc = np.where(m > 0, np.abs(m) * A, -np.abs(m) * A) - d * v
        return np.clip(v + acc * 0.01, 0.0, vmax).astype(f32)
    return f


def run(name, law, steps, n=8, agent="austria", track="austria", repeat=4):
    t0 = time.time()
    orig = ro.OracleRaceEnv._substep
    ro.OracleRaceEnv._substep = patched_substep(law)
    try:
        env = make_oracle(load_track(track), num_envs=n, auto_reset=True, remap_actions=True)
        policy = DreamerPolicy(np.load(os.path.join(GOLDEN, f"dreamer_policy_{agent}.npz")), sample=True, seed=0)
        out = env.reset(mode=ro.RESET_GRID, seed=1)
        state = policy.initial(n)
        crashes, speeds, motors, paid, pred, kls = 0, [], [], [], [], []
        for k in range(steps):
            scan = np.asarray(out["lidar"]).reshape(n, ro.N_BEAMS)
            fresh = np.asarray(out["fresh"]).reshape(n) != 0
            prev = state
            if k and fresh.any():
                keep = (~fresh)[:, None].astype(f32)
                prev = {kk: vv * keep for kk, vv in state.items()}
            mp, sp = posterior_stats(policy, prev, prev["action"], scan)
            mq, sq = prior_stats(policy, prev, prev["action"])
            action, state = policy.act(scan, state, reset=fresh if k else None)
            if k >= 40:
                kls.append(kl_normal(mp, sp, mq, sq))
                pred.append(policy.predicted_reward(state))
                paid.append(np.asarray(out["reward"]).reshape(n).copy())
            out = env.step(action, repeat=repeat)
            crashes += int(np.count_nonzero(np.asarray(out["wall_collision"])))
            speeds.append(float(np.asarray(out["speed"]).mean()))
            motors.append(float(((action[:, 0] + 1) / 2 * (1 - 0.005) + 0.005).mean()))
        paid, pred = np.concatenate(paid), np.concatenate(pred)
        laps = (np.asarray(out["lap"]).reshape(n) + np.asarray(out["progress"]).reshape(n))[:4]
        print(f"{name:44s} contacts {crashes:3d}; speed {np.mean(speeds[50:]):.2f} m/s; motor {np.mean(motors[50:]):.2f}; reward paid {paid.mean():.3f} "
              f"predicted {pred.mean():.3f} ratio {pred.mean() / max(paid.mean(), 1e-9):.2f} corr {np.corrcoef(pred, paid)[0, 1]:.2f}; "
              f"KL {np.mean(np.concatenate(kls)):.1f} nats; lap+progress {np.round(laps, 2)}  ({time.time() - t0:.0f}s)", flush=True)
    finally:
        ro.OracleRaceEnv._substep = orig


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 700
    print(f"# austria agent on austria, 8 cars from the grid, {steps} agent steps, action_repeat 4, sampled policy (tools/analysis/longitudinal_law_sweep.py)")
    run("L0 spec: A 4, settles at m x 5 m/s", law_L0(4.0, 5.0), steps)
    for vmax in (6.5, 8.3):
        run(f"L2 spec form, top speed {vmax}", law_L0(4.0 * vmax / 5.0, vmax), steps)
        run(f"L2 spec form, top speed {vmax}, A 4", law_L0(4.0, vmax), steps)
    for A in (1.0, 2.0, 4.0):
        for d in (0.0, 0.2, 0.5):
            run(f"L1 towards 5 m/s: A {A}, drag {d}", law_L1(A, d), steps)
    run("L1 towards 8.3 m/s: A 2, drag 0.2", law_L1(2.0, 0.2, 8.3), steps)
    print("# treitlstrasse agent on treitlstrasse_v2")
    run("L0 spec", law_L0(4.0, 5.0), steps, agent="treitlstrasse", track="treitlstrasse_v2")
    run("L1 towards 5 m/s: A 2, drag 0.2", law_L1(2.0, 0.2), steps, agent="treitlstrasse", track="treitlstrasse_v2")
