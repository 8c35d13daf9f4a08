#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own wrapper layer.

Runs only in the build container (needs /root/reference).  The reference's simulator core
(racecar_gym / PyBullet) is not installable, but its wrapper layer is plain NumPy/SciPy/PIL and
imports once `gym` and `racecar_gym` are present in sys.modules - exactly what SURVEY.md §8c
probes.  The stubs below provide only the *names* the reference modules import (gym.Wrapper,
gym.spaces.Box/Dict/flatten..., racecar_gym.Task/register_task); every number stored in the
fixtures is computed by the reference's code:

  G1 ReduceActionSpace._normalize .......... dreamer/wrappers.py:119-134
  G2 ActionRepeat.step (Dreamer, baselines)  dreamer/wrappers.py:98-116, baselines/.../single_agent.py:25-40
  G3 TimeLimit.step/reset .................. dreamer/wrappers.py:137-158
  G4 RaceCarWrapper speed / action space ... dreamer/wrappers.py:22-77
  G5 Collect transitions / episode dict .... dreamer/wrappers.py:198-250
  G6 OccupancyMapObs.step patches .......... dreamer/wrappers.py:372-414
  G7 MaximizeSpeed.reward .................. baselines/racing/environment/tasks.py:4-22
  G8 NormalizeObservations + Flatten ....... baselines/racing/environment/single_agent.py:43-99

Fixtures hold inputs and expected outputs only (data, no source).  Library versions used are
recorded inside each file (the reference pins numpy 1.18.5 / scipy 1.5.4 / Pillow 7.2.0,
dreamer/requirements.txt:3,16,17; this container has newer ones).
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


# ----------------------------------------------------------------------------- stubs (names only)
class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            low, high = np.asarray(low), np.asarray(high)
            shape = low.shape
        self.low = np.broadcast_to(np.asarray(low, dtype=np.float64), shape).astype(dtype) if np.isfinite(np.asarray(low, dtype=np.float64)).all() else np.full(shape, low, dtype=dtype)
        self.high = np.broadcast_to(np.asarray(high, dtype=np.float64), shape).astype(dtype) if np.isfinite(np.asarray(high, dtype=np.float64)).all() else np.full(shape, high, dtype=dtype)
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)


class Dict:
    def __init__(self, spaces):
        # gym.spaces.Dict sorts plain-dict keys; (name, space) lists keep their order
        self.spaces = dict(sorted(spaces.items())) if isinstance(spaces, dict) else dict(spaces)

    def __getitem__(self, k):
        return self.spaces[k]

    def __contains__(self, x):          # gym.Space.__contains__ = "x is a valid sample of this space"
        return isinstance(x, dict) and set(x) == set(self.spaces)


def _flatten_space(space):
    if isinstance(space, Box):
        return Box(space.low.ravel(), space.high.ravel(), dtype=space.dtype)
    lows = [_flatten_space(s).low for s in space.spaces.values()]
    highs = [_flatten_space(s).high for s in space.spaces.values()]
    return Box(np.concatenate(lows), np.concatenate(highs), dtype=np.float32)


def _flatten(space, x):
    if isinstance(space, Box):
        return np.asarray(x, dtype=np.float32).ravel()
    return np.concatenate([_flatten(s, x[k]) for k, s in space.spaces.items()])


def _unflatten(space, x):
    if isinstance(space, Box):
        return np.asarray(x, dtype=np.float32).reshape(space.shape)
    out, off = {}, 0
    for k, s in space.spaces.items():
        n = int(np.prod(_flatten_space(s).shape))
        out[k] = _unflatten(s, x[off:off + n])
        off += n
    return out


class Wrapper:
    def __init__(self, env):
        self.env = env
        self.observation_space = getattr(env, "observation_space", None)
        self.action_space = getattr(env, "action_space", None)

    def step(self, action):
        return self.env.step(action)

    def reset(self, **kw):
        return self.env.reset(**kw)

    def __getattr__(self, name):
        return getattr(self.env, name)


class ObservationWrapper(Wrapper):
    def step(self, action):
        o, r, d, i = self.env.step(action)
        return self.observation(o), r, d, i

    def reset(self, **kw):
        return self.observation(self.env.reset(**kw))


def install_stubs():
    gym = types.ModuleType("gym")
    spaces = types.ModuleType("gym.spaces")
    spaces.Box, spaces.Dict = Box, Dict
    spaces.flatten_space, spaces.flatten, spaces.unflatten = _flatten_space, _flatten, _unflatten
    gym.spaces, gym.Wrapper, gym.ObservationWrapper, gym.Env = spaces, Wrapper, ObservationWrapper, object
    wr = types.ModuleType("gym.wrappers")
    gym.wrappers = wr
    rg = types.ModuleType("racecar_gym")
    rg.Task = object
    rg.register_task = lambda name, task: None
    envs = types.ModuleType("racecar_gym.envs")
    mar = types.ModuleType("racecar_gym.envs.multi_agent_race")
    mar.MultiAgentScenario = mar.MultiAgentRaceEnv = object
    rg.envs = envs
    envs.multi_agent_race = mar
    sys.modules.update({"gym": gym, "gym.spaces": spaces, "gym.wrappers": wr, "racecar_gym": rg,
                        "racecar_gym.envs": envs, "racecar_gym.envs.multi_agent_race": mar})


def versions():
    import PIL
    import scipy
    return np.array(f"numpy {np.__version__}; scipy {scipy.__version__}; Pillow {PIL.__version__}")


# ----------------------------------------------------------------------------- scripted fake envs
class ScriptedMultiEnv:
    """Multi-agent env with the racecar_gym return convention, driven by scripted sequences."""

    def __init__(self, agent_ids, rewards, dones, velocities=None):
        self.agent_ids = list(agent_ids)
        self._rewards, self._dones = rewards, dones
        self._vel = velocities
        self.t = 0
        self.calls = []
        lid = Box(0.0, 15.0, (1080,))
        self.observation_space = Dict({a: Dict({"lidar": lid, "pose": Box(-100, 100, (6,)),
                                                "velocity": Box(-10, 10, (6,))}) for a in self.agent_ids})
        self.action_space = Dict({a: Dict({"motor": Box(-1.0, 1.0, (1,)), "steering": Box(-1.0, 1.0, (1,))})
                                  for a in self.agent_ids})

    def _obs(self):
        return {a: {"lidar": np.full(1080, float(self.t), np.float64), "pose": np.arange(6) + 0.5 * self.t,
                    "velocity": np.zeros(6)} for a in self.agent_ids}

    def reset(self, **kw):
        self.reset_kwargs = kw
        return self._obs()

    def step(self, action):
        self.calls.append(action)
        k = self.t
        self.t += 1
        rew = {a: float(self._rewards[k][i]) for i, a in enumerate(self.agent_ids)}
        done = {a: bool(self._dones[k][i]) for i, a in enumerate(self.agent_ids)}
        vel = self._vel[k] if self._vel is not None else np.zeros((len(self.agent_ids), 6))
        info = {a: {"pose": np.arange(6) + 0.5 * self.t, "velocity": np.asarray(vel[i], np.float64),
                    "lap": 1 + (self.t // 5), "progress": 0.1 * (self.t % 5), "time": 0.01 * self.t,
                    "wrong_way": False, "wall_collision": done[a]} for i, a in enumerate(self.agent_ids)}
        return self._obs(), rew, done, info


class ScriptedSingleEnv:
    def __init__(self, rewards, dones):
        self._r, self._d, self.t = rewards, dones, 0
        self.observation_space = Dict({"lidar": Box(0.0, 15.0, (4,))})
        self.action_space = Dict({"motor": Box(-1.0, 1.0, (1,)), "steering": Box(-1.0, 1.0, (1,))})

    def reset(self, **kw):
        return {"lidar": np.zeros(4)}

    def step(self, action):
        self.last_action = action
        k = self.t
        self.t += 1
        return {"lidar": np.full(4, float(self.t))}, float(self._r[k]), bool(self._d[k]), {"t": self.t}


# ----------------------------------------------------------------------------- generators
def gen_wrappers(W, SA, TK, out):
    rng = np.random.default_rng(20211003)
    # G1 -----------------------------------------------------------------------------------------
    a = rng.uniform(-1, 1, (256, 2)).astype(np.float32)
    red = W.ReduceActionSpace(types.SimpleNamespace(agent_ids=["A"]), low=[0.005, -1.0], high=[1.0, 1.0])
    g1 = np.stack([red._normalize(x) for x in a])
    # G2 Dreamer variant --------------------------------------------------------------------------
    cases, g2_rew, g2_done, g2_n, g2_last = [], [], [], [], []
    for done_at in (None, 0, 1, 3, 5):
        for amount in (1, 4, 8):
            rew = rng.uniform(-1, 1, (16, 2))
            don = np.zeros((16, 2), bool)
            if done_at is not None:
                don[done_at, done_at % 2] = True
            env = ScriptedMultiEnv(["A", "B"], rew, don)
            obs, tot, dones, info = W.ActionRepeat(env, amount).step({"A": np.zeros(2), "B": np.zeros(2)})
            cases.append((amount, -1 if done_at is None else done_at))
            g2_rew.append(np.concatenate([rew.ravel(), [tot["A"], tot["B"]]]))
            g2_done.append([dones["A"], dones["B"]])
            g2_n.append(env.t)
            g2_last.append(obs["A"]["lidar"][0])
    # G2 baselines variant ------------------------------------------------------------------------
    b_cases, b_out = [], []
    for done_at in (None, 0, 1, 3):
        for n in (1, 4):
            rew = rng.uniform(-1, 1, 16)
            don = np.zeros(16, bool)
            if done_at is not None:
                don[done_at] = True
            env = ScriptedSingleEnv(rew, don)
            obs, tot, done, info = SA.ActionRepeat(env, n).step(np.zeros(2))
            b_cases.append((n, -1 if done_at is None else done_at))
            b_out.append(np.concatenate([rew, [tot, float(done), env.t, obs["lidar"][0]]]))
    # G3 ------------------------------------------------------------------------------------------
    env = ScriptedMultiEnv(["A"], np.zeros((40, 1)), np.zeros((40, 1), bool))
    tl = W.TimeLimit(env, duration=7)
    try:
        tl.step({"A": np.zeros(2)})
        raised = ""
    except AssertionError as e:
        raised = str(e)
    tl.reset()
    g3 = []
    for k in range(7):
        g3.append(tl.step({"A": np.zeros(2)})[2]["A"])
    try:
        tl.step({"A": np.zeros(2)})
        raised2 = ""
    except AssertionError as e:
        raised2 = str(e)
    # G4 ------------------------------------------------------------------------------------------
    vel = rng.normal(0, 2, (32, 1, 6))
    env = ScriptedMultiEnv(["A"], np.zeros((32, 1)), np.zeros((32, 1), bool), velocities=vel)
    rw = W.RaceCarWrapper(env, agent_id="A")
    speeds = [rw.step({"A": np.array([0.3, -0.2])})[0]["A"]["speed"] for _ in range(32)]
    asp = rw.action_space["A"]
    reset_speed = rw.reset(mode="grid")["A"]["speed"]
    sent = env.calls[0]["A"]
    # G5 ------------------------------------------------------------------------------------------
    T = 7
    rew = rng.uniform(-1, 1, (T, 1))
    don = np.zeros((T, 1), bool)
    don[-1, 0] = True
    env = W.RaceCarWrapper(ScriptedMultiEnv(["A"], rew, don, velocities=rng.normal(0, 1, (T, 1, 6))), "A")
    episodes_out = []
    col = W.Collect(env, callbacks=[lambda eps: episodes_out.append(eps)], precision=32)
    first = col.reset()
    acts = rng.uniform(-1, 1, (T, 2)).astype(np.float32)
    returned_keys = None
    for k in range(T):
        o, r, d, i = col.step({"A": acts[k]})
        returned_keys = sorted(o["A"].keys())
    ep = episodes_out[0][0]
    g5 = {f"g5_ep_{k}": v for k, v in ep.items()}
    g5["g5_actions"] = acts
    g5["g5_rewards"] = rew[:, 0]
    g5["g5_returned_keys"] = np.array(",".join(returned_keys))
    g5["g5_dtypes"] = np.array(";".join(f"{k}:{v.dtype}" for k, v in sorted(ep.items())))
    # G7 ------------------------------------------------------------------------------------------
    task = TK.MaximizeSpeed.__new__(TK.MaximizeSpeed)
    st = rng.uniform(-1, 1, 200)
    v = rng.uniform(0, 5, 200)
    wall = rng.uniform(size=200) < 0.2
    g7 = np.array([task.reward("A", {"A": {"wall_collision": bool(w), "velocity": np.array([vv, 0, 0, 0, 0, 0])}},
                               {"steering": np.array([s])}) for s, vv, w in zip(st, v, wall)])
    # G8 ------------------------------------------------------------------------------------------
    env = ScriptedSingleEnv(np.zeros(8), np.zeros(8, bool))
    env.observation_space = Dict({"lidar": Box(0.25, 15.0, (4,))})
    chain = SA.NormalizeObservations(SA.Flatten(env, flatten_obs=True, flatten_actions=True))
    chain.reset()
    raw_actions = np.array([[0.3, -0.7], [1.5, -2.0], [-1.2, 0.4]], np.float32)
    g8_obs, g8_act = [], []
    for a_ in raw_actions:
        o, r, d, i = chain.step(a_)
        g8_obs.append(np.asarray(o, np.float64))
        g8_act.append([float(env.last_action["motor"][0]), float(env.last_action["steering"][0])])
    np.savez_compressed(
        out, versions=versions(),
        g1_in=a, g1_out=g1,
        g2_cases=np.array(cases), g2_rew=np.array(g2_rew), g2_done=np.array(g2_done), g2_calls=np.array(g2_n),
        g2_last_lidar=np.array(g2_last), g2b_cases=np.array(b_cases), g2b=np.array(b_out),
        g3_dones=np.array(g3), g3_assert_before_reset=np.array(raised), g3_assert_after_limit=np.array(raised2),
        g4_vel=vel[:, 0], g4_speed=np.array(speeds, np.float64), g4_low=asp.low, g4_high=asp.high,
        g4_reset_speed=np.float64(reset_speed), g4_sent_motor=np.float64(sent["motor"]),
        g4_sent_steering=np.float64(sent["steering"]),
        g7_steering=st, g7_velocity=v, g7_wall=wall, g7_reward=g7,
        g8_actions=raw_actions, g8_obs=np.array(g8_obs), g8_sent=np.array(g8_act),
        **g5)


class FullFrameMap:
    """GridMap stand-in over the full source image frame (north-up), as OccupancyMapObs indexes it
    (dreamer/wrappers.py:376,396-399); to_pixel is racecar_gym's convention."""

    def __init__(self, track):
        r0, c0, fh, fw = track.crop
        full = np.zeros((fh, fw), bool)
        full[r0:r0 + track.height, c0:c0 + track.width] = track.drivable[::-1]
        self._map = full
        self.res = track.resolution
        self.ox = track.origin[0] - c0 * self.res
        self.oy = track.origin[1] - (fh - (r0 + track.height)) * self.res
        self.h = fh

    def to_pixel(self, pose):
        x, y = pose[0], pose[1]
        return int(self.h - (y - self.oy) / self.res), int((x - self.ox) / self.res)


def gen_patches(W, out):
    from racing_dreamer_amd.track_assets import load_track
    rng = np.random.default_rng(7)
    data = {"versions": versions()}
    # (columbia_slam = docs/maps/maps/columbia.pgm, "columbia" until round 5; columbia = columbia_small, appended: the random stream
    # of the first three tracks is what it was)
    for name in ("austria", "treitlstrasse_v2", "columbia_slam", "columbia"):
        t = load_track(name)
        gm = FullFrameMap(t)
        # >= 64 poses per track: 64 along the track with lateral / heading jitter, 24 pushed up to and
        # past the track border (the patch then shows mostly wall), 8 fixed yaws incl. both ends of the (-pi, pi] range
        n, n_border = 160, 60                    # candidates; the first 64 + 24 + 8 whose crop window fits are kept
        idx = rng.integers(0, len(t.centerline), n + n_border + 8)
        poses = t.centerline[idx, :3].astype(np.float64)
        poses[:n, 0] += rng.uniform(-0.3, 0.3, n)
        poses[:n, 1] += rng.uniform(-0.3, 0.3, n)
        poses[:n, 2] += rng.uniform(-0.6, 0.6, n)
        side = rng.choice([-1.0, 1.0], n_border) * rng.uniform(0.5, 1.3, n_border)      # metres off the centre line
        th = poses[n:n + n_border, 2]
        poses[n:n + n_border, 0] += -np.sin(th) * side
        poses[n:n + n_border, 1] += np.cos(th) * side
        poses[n:n + n_border, 2] += rng.uniform(-3.14, 3.14, n_border)
        poses[n + n_border:, 2] = [0.0, np.pi / 4, np.pi / 2, np.pi, -np.pi + 1e-6, -np.pi / 2, np.pi - 1e-3, -2.5]
        poses[:, 2] = (poses[:, 2] + np.pi) % (2 * np.pi) - np.pi
        poses[n + n_border + 3, 2] = np.pi
        ok = []
        for p in poses:   # the reference slices without bounds handling: keep poses whose window fits
            pr, pc = gm.to_pixel(p)
            ok.append(110 <= pr < gm._map.shape[0] - 110 and 110 <= pc < gm._map.shape[1] - 110)
        ok = np.array(ok)
        keep = np.concatenate([np.nonzero(ok[:n])[0][:64], n + np.nonzero(ok[n:n + n_border])[0][:24],
                               n + n_border + np.nonzero(ok[n + n_border:])[0]])
        poses = poses[keep]
        patches = []
        for p in poses:
            pose6 = np.array([p[0], p[1], 0, 0, 0, p[2]])

            class Inner:
                agent_ids = ["A"]
                scenario = types.SimpleNamespace(world=types.SimpleNamespace(_maps={"occupancy": gm}))

                def step(self, action):
                    return {"A": {}}, {"A": 0.0}, {"A": False}, {"A": {"pose": pose6}}
            obs = W.OccupancyMapObs(Inner()).step({"A": None})[0]
            patches.append(obs["A"]["lidar_occupancy"])
        data[f"{name}_poses"] = poses
        data[f"{name}_patches"] = np.packbits(np.array(patches, np.uint8) > 0, axis=-2)
        data[f"{name}_raw_max"] = np.array(patches).max()
        print(name, len(poses), "patches; value set", np.unique(np.array(patches)))
    np.savez_compressed(out, **data)


def main():
    install_stubs()
    sys.path.insert(0, os.path.join(REF, "dreamer"))
    import wrappers as W                                             # dreamer/wrappers.py
    sys.path.insert(0, os.path.join(REF, "baselines"))
    import importlib.util

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m
    SA = load("ref_single_agent", os.path.join(REF, "baselines/racing/environment/single_agent.py"))
    TK = load("ref_tasks", os.path.join(REF, "baselines/racing/environment/tasks.py"))
    gen_wrappers(W, SA, TK, os.path.join(HERE, "wrappers_golden.npz"))
    gen_patches(W, os.path.join(HERE, "occupancy_patch_golden.npz"))
    for f in ("wrappers_golden.npz", "occupancy_patch_golden.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
