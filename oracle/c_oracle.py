"""ctypes driver of the plain-C oracle (oracle/racecar_oracle.c) - TEST INFRASTRUCTURE, NOT PRODUCT.

Same interface as racecar_oracle.OracleRaceEnv (reset / step / outputs); env ranges are spread over
a thread pool (ctypes releases the GIL), which is what bench.py's cpu_baseline leg times.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import racecar_oracle as ro

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libracecar_oracle.so")

_fp, _ip, _bp, _up = (C.POINTER(t) for t in (C.c_float, C.c_int32, C.c_uint8, C.c_uint32))


class OcTrack(C.Structure):
    _fields_ = [("occ", _bp), ("ring", _bp), ("drv", _bp), ("progress", _fp), ("centerline", _fp), ("beams", _fp),
                ("foot", _fp), ("h", C.c_int32), ("w", C.c_int32), ("n_centerline", C.c_int32),
                ("org_x", C.c_float), ("org_y", C.c_float), ("res", C.c_float), ("inv_res", C.c_float),
                ("tmax", C.c_float), ("spawn_w", _fp), ("spawn_safe", _ip), ("spawn_rows", _fp)]


class OcFrame(C.Structure):
    _fields_ = [("fh", C.c_int32), ("r_top", C.c_int32), ("c0", C.c_int32), ("ox", C.c_double), ("oy", C.c_double), ("res", C.c_double)]


class OcCfg(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("cars_per_env", C.c_int32), ("first_env", C.c_uint32), ("task", C.c_int32),
                ("laps", C.c_int32), ("terminate_on_collision", C.c_int32), ("remap_actions", C.c_int32),
                ("time_limit_steps", C.c_int32), ("auto_reset", C.c_int32), ("time_limit", C.c_float),
                ("collision_reward", C.c_float), ("act_lo", C.c_float * 2), ("act_hi", C.c_float * 2),
                ("reset_mode", C.c_int32), ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32),
                ("car_task", C.c_int32 * 4), ("n_steps", C.c_int32)]


_STATE_F = ["x", "y", "theta", "ct", "st", "v", "delta", "omega", "accel", "progress"]
_STATE_I = ["lap", "cp"]
_STATE_B = ["wall", "opp", "wrong", "done", "trunc", "fresh"]
_ENV_I = ["steps", "agent_steps"]
_RES_F = ["reward", "discount", "progress_total", "time", "action", "out_progress"]
_RES_I = ["out_lap", "out_cp"]
_RES_B = ["out_done", "out_trunc", "out_wall", "out_opp", "out_wrong"]


class OcState(C.Structure):
    _fields_ = ([(n, _fp) for n in _STATE_F] + [(n, _ip) for n in _STATE_I] + [(n, _bp) for n in _STATE_B]
                + [(n, _ip) for n in _ENV_I] + [("episode", _up)]
                + [(n, _fp) for n in ["reward", "discount", "progress_total", "time", "action"]]
                + [("out_progress", _fp)] + [(n, _ip) for n in _RES_I] + [(n, _bp) for n in _RES_B]
                + [("nstep_hist", _fp)])


def build_library() -> str:
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "racecar_oracle.c")):
        subprocess.run(["make", "-C", HERE], check=True, stdout=subprocess.DEVNULL)
    return LIB


_lib = None


def load():
    global _lib
    if _lib is None:
        lib = C.CDLL(build_library())
        P = C.POINTER
        lib.oc_reset.argtypes = [P(OcTrack), P(OcCfg), P(OcState), C.c_void_p]
        lib.oc_step_range.argtypes = [P(OcTrack), P(OcCfg), P(OcState), C.c_void_p, C.c_int, C.c_int, C.c_int]
        lib.oc_raycast_range.argtypes = [P(OcTrack), P(OcCfg), P(OcState), C.c_void_p, C.c_int, C.c_int]
        lib.oc_patch_range.argtypes = [P(OcTrack), P(OcState), C.c_void_p, C.c_int, C.c_int]
        lib.oc_random_actions.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        lib.oc_spawn_width.argtypes = [P(OcTrack), C.c_void_p]
        lib.oc_spawn_safe.argtypes = [P(OcTrack), C.c_void_p]
        lib.oc_spawn_rows.argtypes = [P(OcTrack), C.c_void_p]
        lib.oc_patch_exact_range.argtypes = [P(OcTrack), P(OcFrame), P(OcState), C.c_void_p, C.c_int, C.c_int]
        lib.oc_patch_exact_range.restype = None
        lib.oc_resize_coefficients.argtypes = [C.c_void_p, C.c_void_p]
        lib.oc_resize_coefficients.restype = None
        for f in (lib.oc_reset, lib.oc_step_range, lib.oc_raycast_range, lib.oc_patch_range, lib.oc_random_actions, lib.oc_spawn_width,
                  lib.oc_spawn_safe, lib.oc_spawn_rows):
            f.restype = None
        lib.oc_set_dynamics.argtypes = [C.c_float] * 5
        lib.oc_set_dynamics.restype = None
        lib.oc_spin.argtypes = [C.c_uint64]
        lib.oc_spin.restype = C.c_uint64
        _lib = lib
    return _lib


def _ptr(a, typ):
    return a.ctypes.data_as(typ)


class COracleEnv:
    def __init__(self, occ, drivable, progress, centerline, origin, resolution, cfg: ro.OracleConfig, threads=1):
        self.lib = load()
        self.cfg = cfg
        self.threads = max(1, int(threads))
        self.pool = ThreadPoolExecutor(self.threads) if self.threads > 1 else None
        occ = np.asarray(occ, bool).copy()
        ring = np.zeros_like(occ)
        ring[0, :] = ring[-1, :] = ring[:, 0] = ring[:, -1] = True
        occ |= ring
        self.H, self.W = occ.shape
        self._keep = dict(
            occ=np.ascontiguousarray(occ, np.uint8), ring=np.ascontiguousarray(ring, np.uint8),
            drv=np.ascontiguousarray(np.asarray(drivable, bool) & ~ring, np.uint8),   # spec: the ring is not drivable
            progress=np.ascontiguousarray(progress, np.float32),
            centerline=np.ascontiguousarray(centerline, np.float32),
            beams=np.ascontiguousarray(np.stack(ro.beam_table(), axis=1), np.float32),
            foot=np.ascontiguousarray(ro.footprint_table(), np.float32))
        k = self._keep
        inv_res = np.float32(1.0 / resolution)
        self.trk = OcTrack(_ptr(k["occ"], _bp), _ptr(k["ring"], _bp), _ptr(k["drv"], _bp), _ptr(k["progress"], _fp),
                           _ptr(k["centerline"], _fp), _ptr(k["beams"], _fp), _ptr(k["foot"], _fp), self.H, self.W,
                           len(k["centerline"]), np.float32(origin[0]), np.float32(origin[1]),
                           np.float32(resolution), inv_res, np.float32(ro.MAX_RANGE * inv_res), None, None, None)
        k["spawn_w"] = np.zeros(len(k["centerline"]), np.float32)      # the C port builds its own table (oc_spawn_width)
        self.lib.oc_spawn_width(C.byref(self.trk), k["spawn_w"].ctypes.data)
        self.trk.spawn_w = _ptr(k["spawn_w"], _fp)
        k["spawn_safe"] = np.zeros(len(k["centerline"]), np.int32)     # ... and the anchors of multi-car starts (oc_spawn_safe)
        self.lib.oc_spawn_safe(C.byref(self.trk), k["spawn_safe"].ctypes.data)
        self.trk.spawn_safe = _ptr(k["spawn_safe"], _ip)
        k["spawn_rows"] = np.zeros((len(k["centerline"]), 5), np.float32)      # ... and the table itself (oc_spawn_rows)
        self.lib.oc_spawn_rows(C.byref(self.trk), k["spawn_rows"].ctypes.data)
        self.trk.spawn_rows = _ptr(k["spawn_rows"], _fp)
        self.B, self.A = cfg.num_envs, cfg.cars_per_env
        n = self.NC = self.B * self.A
        self.ccfg = OcCfg(self.B, self.A, cfg.first_env & 0xFFFFFFFF, cfg.task, cfg.laps,
                          int(cfg.terminate_on_collision), int(cfg.remap_actions), cfg.time_limit_steps,
                          int(cfg.auto_reset), cfg.time_limit, cfg.collision_reward,
                          (C.c_float * 2)(*cfg.action_low), (C.c_float * 2)(*cfg.action_high), 0, 0, 0,
                          (C.c_int32 * 4)(*[(-1 if cfg.car_tasks is None or a >= len(cfg.car_tasks) else int(cfg.car_tasks[a]))
                                            for a in range(4)]), int(cfg.n_steps))
        self.arr = {}
        st = OcState()
        for name in _STATE_F:
            self.arr[name] = np.zeros(n, np.float32)
        for name in _STATE_I:
            self.arr[name] = np.zeros(n, np.int32)
        for name in _STATE_B:
            self.arr[name] = np.zeros(n, np.uint8)
        for name in _ENV_I:
            self.arr[name] = np.zeros(self.B, np.int32)
        self.arr["episode"] = np.zeros(self.B, np.uint32)
        for name in _RES_F:
            self.arr[name] = np.zeros(n * (2 if name == "action" else 1), np.float32)
        for name in _RES_I:
            self.arr[name] = np.zeros(n, np.int32)
        for name in _RES_B:
            self.arr[name] = np.zeros(n, np.uint8)
        self.arr["nstep_hist"] = np.zeros(n * ro.NSTEP_MAX, np.float32)
        for name, typ in OcState._fields_:
            setattr(st, name, _ptr(self.arr[name], typ))
        self.state = st
        self.lidar = np.zeros((n, ro.N_BEAMS), np.float32)
        self.patch = np.zeros((n, ro.PATCH, ro.PATCH), np.uint8)
        self.was_reset = False

    # ------------------------------------------------------------------
    def _split(self, n):
        # several chunks per thread: the pool hands them out as threads become free (rays differ in length); not so many
        # that handing them out costs more than they take
        k = self.threads if self.threads == 1 else max(self.threads, min(n // 8, self.threads * 4))
        k = max(1, min(k, n))
        edges = [n * i // k for i in range(k + 1)]
        return [(a, b) for a, b in zip(edges[:-1], edges[1:]) if b > a]

    def _par(self, fn, n):
        if self.pool is None:
            fn(0, n)
        else:
            list(self.pool.map(lambda ab: fn(*ab), self._split(n)))

    def _observe(self):
        self._par(lambda a, b: self.lib.oc_raycast_range(C.byref(self.trk), C.byref(self.ccfg), C.byref(self.state),
                                                         self.lidar.ctypes.data, a, b), self.NC)
        if self.cfg.render_occupancy == "reference":       # obs_type lidar_occupancy_reference (oracle/patch_reference.py)
            self._par(lambda a, b: self.lib.oc_patch_exact_range(C.byref(self.trk), C.byref(self.frame), C.byref(self.state),
                                                                 self.patch.ctypes.data, a, b), self.NC)
        elif self.cfg.render_occupancy:
            self._par(lambda a, b: self.lib.oc_patch_range(C.byref(self.trk), C.byref(self.state),
                                                           self.patch.ctypes.data, a, b), self.NC)

    def set_frame(self, track):
        """Where the track's grid lies in the source image (for render_occupancy='reference')."""
        from . import patch_reference as px
        fh, ox, oy, r_top, c0 = px.frame_of(track)
        self.frame = OcFrame(fh, r_top, c0, ox, oy, float(track.resolution))

    def reset(self, mask=None, mode=ro.RESET_GRID, seed=0):
        self.ccfg.reset_mode = int(mode)
        self.ccfg.seed_lo, self.ccfg.seed_hi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
        m = None if mask is None else np.ascontiguousarray(np.asarray(mask).astype(np.uint8))
        self.lib.oc_reset(C.byref(self.trk), C.byref(self.ccfg), C.byref(self.state),
                          None if m is None else m.ctypes.data)
        self.was_reset = True
        self._observe()
        return self.outputs()

    def step(self, actions, repeat=1, outputs=True):
        """outputs=False: the results stay in the env's own arrays (self.arr, self.lidar, self.patch) and no per-step
        copies are made - what the timed cpu_baseline leg uses, so that it times the simulator and not NumPy copies."""
        assert self.was_reset, "Must reset environment."
        act = np.ascontiguousarray(np.asarray(actions, np.float32).reshape(self.NC, 2))
        self._par(lambda a, b: self.lib.oc_step_range(C.byref(self.trk), C.byref(self.ccfg), C.byref(self.state),
                                                      act.ctypes.data, int(repeat), a, b), self.B)
        self._observe()
        return self.outputs() if outputs else None

    def random_actions(self, seed, step):
        out = np.zeros((self.NC, 2), np.float32)
        self.lib.oc_random_actions(out.ctypes.data, self.NC, (self.cfg.first_env * self.A) & 0xFFFFFFFF,
                                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, step)
        return out

    def outputs(self):
        a, n = self.arr, self.NC
        pose = np.zeros((n, 6), np.float32)
        vel = np.zeros((n, 6), np.float32)
        pose[:, 0], pose[:, 1], pose[:, 5] = a["x"], a["y"], a["theta"]
        vel[:, 0], vel[:, 5] = a["v"], a["omega"]
        d = dict(action=a["action"].reshape(n, 2).copy(), reward=a["reward"].copy(), discount=a["discount"].copy(),
                 progress_total=a["progress_total"].copy(), time=a["time"].copy(), progress=a["out_progress"].copy(),
                 lap=a["out_lap"].copy(), checkpoint=a["out_cp"].copy(), done=a["out_done"].copy(),
                 truncated=a["out_trunc"].copy(), wall_collision=a["out_wall"].copy(),
                 opponent_collision=a["out_opp"].copy(), wrong_way=a["out_wrong"].copy(),
                 lidar=self.lidar.copy(), pose=pose, velocity=vel, speed=np.abs(a["v"]),
                 acceleration=a["accel"].copy(), steering_angle=a["delta"].copy(), fresh=a["fresh"].copy())
        if self.cfg.render_occupancy:
            d["lidar_occupancy"] = self.patch.copy()
        return d


def set_dynamics(accel_max=None, drag=None, max_vel=None, steer_gain=None, steer_step=None):
    """Calibration sweeps only (tools/analysis/agent_calibration.py): the integrator's free parameters of EVERY C-oracle env of
    this process; None = the spec's value (racecar_oracle.py).  `set_dynamics()` restores the spec."""
    v = lambda x, d: float(d if x is None else x)
    load().oc_set_dynamics(v(accel_max, ro.ACCEL_MAX), v(drag, ro.DRAG), v(max_vel, ro.MAX_VEL), v(steer_gain, ro.STEER_GAIN), v(steer_step, ro.STEER_STEP))
