"""BatchedRaceEnv: the Python host of the MI355X batched racing environment.

Thin layer over the C-ABI (include/racecar_hip.h): torch allocates the output arena and the
stream, the HIP library does all the work, and every output is a zero-copy torch view of the
arena.  The surface mirrors what the reference's callers consume from racecar_gym
(SURVEY.md §8b): ``reset(mode=...)`` / ``step(actions)`` returning ``lidar``, ``pose``,
``velocity``, ``speed``, ``lidar_occupancy``, ``reward``, ``done`` and the info keys
``progress``, ``lap``, ``time``, ``wrong_way``, ``wall_collision``
(dreamer/wrappers.py:62-77,210-226; baselines/.../sb_experiment.py:82-88), batched over
``num_envs x cars_per_env``.

The returned tensors are views of buffers that the next step()/reset() overwrites; clone what
must be kept.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Union

import numpy as np
import torch

from . import _lib as L
from . import spec
from .track_assets import Track, load_track

_FIELD_VIEWS = {
    # name: (field id, torch dtype, trailing shape)
    "lidar": (L.F_LIDAR, torch.float32, (L.RC_N_BEAMS,)),
    "pose": (L.F_POSE, torch.float32, (6,)),
    "velocity": (L.F_VELOCITY, torch.float32, (6,)),
    "speed": (L.F_SPEED, torch.float32, ()),
    "action": (L.F_ACTION, torch.float32, (2,)),
    "reward": (L.F_REWARD, torch.float32, ()),
    "discount": (L.F_DISCOUNT, torch.float32, ()),
    "progress_total": (L.F_PROGRESS_TOTAL, torch.float32, ()),
    "time": (L.F_TIME, torch.float32, ()),
    "lidar_occupancy": (L.F_OCCUPANCY, torch.uint8, (L.RC_PATCH, L.RC_PATCH, 1)),
    "progress": (L.F_PROGRESS, torch.float32, ()),
    "lap": (L.F_LAP, torch.int32, ()),
    "checkpoint": (L.F_CHECKPOINT, torch.int32, ()),
    "done": (L.F_DONE, torch.uint8, ()),
    "truncated": (L.F_TRUNCATED, torch.uint8, ()),
    "wall_collision": (L.F_WALL_COLLISION, torch.uint8, ()),
    "opponent_collision": (L.F_OPPONENT_COLLISION, torch.uint8, ()),
    "wrong_way": (L.F_WRONG_WAY, torch.uint8, ()),
    "fresh": (L.F_FRESH, torch.uint8, ()),
    "acceleration": (L.F_ACCELERATION, torch.float32, ()),
    "steering_angle": (L.F_STEERING_ANGLE, torch.float32, ()),
    "action_in": (L.F_ACTION_IN, torch.float32, (2,)),
}

# lidar_occupancy_reference: the 64 x 64 patch computed EXACTLY as the reference's OccupancyMapObs.step does (dreamer/wrappers.py:
# 396-406; spline rotation + antialiased bicubic resize restated to the binary64 operation) instead of by the one-tap sampler
OBS_TYPES = {"lidar": 0, "lidar_occupancy": 1, "lidar_occupancy_reference": 2}
# scaling fused into the scan's store: metres | dreamer (x/15 - 0.5, tools.py:274) | unit (x/15, single_agent.py:92-99)
LIDAR_TRANSFORMS = {"metres": 0, "dreamer": 1, "unit": 2}
TASKS = {"maximize_progress": spec.TASK_MAX_PROGRESS, "max_progress": spec.TASK_MAX_PROGRESS,
         "max_speed": spec.TASK_MAX_SPEED, "n_step_progress": spec.TASK_N_STEP_PROGRESS}


def _ensure_lab() -> None:
    """Build the lab library (scan variants 0-6, the stamps build) if the one on disk does not belong to the sources - the
    request that `libracecar_lab.so` exists for.  libracecar_hip.so loads it itself, on the first launch of a lab kernel."""
    from . import build
    if build.lab_needs_build():
        build.build_lab(verbose=False)


class BatchedRaceEnv:
    def __init__(self, track: Union[str, Track], num_envs: int, cars_per_env: int = 1, obs_type: str = "lidar",
                 action_repeat: int = 1, seed: int = 0, device: int = 0, first_env: int = 0,
                 task: str = "maximize_progress", laps: int = 10, time_limit: float = 180.0,
                 terminate_on_collision: bool = True, collision_reward: float = -1.0,
                 remap_actions: bool = False, action_low=spec.ACTION_LOW, action_high=spec.ACTION_HIGH,
                 time_limit_steps: int = 0, auto_reset: bool = False, profiling: bool = False,
                 lidar_transform: str = "metres", car_tasks=None, n_steps: int = 10,
                 shared_arena: Optional[torch.Tensor] = None, arena_total_cars: int = 0, arena_first_car: int = 0,
                 stream: Optional[torch.cuda.Stream] = None):
        """car_tasks: optional task name per car slot (agents A, B, ... of a scenario yml; None entries = `task`), e.g.
        ["maximize_progress", "n_step_progress", ...] for baselines/scenarios/max_progress/columbia.yml; n_steps: the
        window of `n_step_progress` in sub-steps.  shared_arena / arena_total_cars / arena_first_car / stream: this env
        fills cars [arena_first_car, ...) of an arena laid out for arena_total_cars cars and runs on the given stream -
        how `MixedTrackEnv` puts one handle per track behind one set of output tensors."""
        if obs_type not in OBS_TYPES:
            raise ValueError(f"obs_type must be one of {sorted(OBS_TYPES)}, got {obs_type!r}")
        if task not in TASKS:
            raise ValueError(f"task must be one of {sorted(TASKS)}, got {task!r}")
        self._lib = L.load_library()                     # raises if the HIP extension is missing
        if not torch.cuda.is_available():
            raise L.RacecarHipError("no HIP device visible to torch; BatchedRaceEnv has no CPU fallback")
        self.track = load_track(track) if isinstance(track, str) else track
        self.num_envs, self.cars_per_env = int(num_envs), int(cars_per_env)
        self.n_cars = self.num_envs * self.cars_per_env
        self.obs_type, self.action_repeat, self.seed = obs_type, int(action_repeat), int(seed)
        if self.track.open and task != "max_speed":
            import warnings
            warnings.warn(f"track {self.track.name!r} is not a loop (tracks/index.json: open): progress never comes round, so no lap is ever "
                          f"completed and `lap > laps` never ends an episode there", stacklevel=2)
        self.device = torch.device("cuda", device)
        self.first_env = int(first_env)

        cfg = L.RcConfig()
        self._lib.rc_default_config(C.byref(cfg))
        cfg.device, cfg.num_envs, cfg.cars_per_env, cfg.first_env = device, self.num_envs, self.cars_per_env, first_env
        cfg.obs_type, cfg.task, cfg.laps, cfg.time_limit = OBS_TYPES[obs_type], TASKS[task], laps, time_limit
        cfg.terminate_on_collision, cfg.collision_reward = int(terminate_on_collision), collision_reward
        cfg.remap_actions = int(remap_actions)
        cfg.action_low[:] = [float(v) for v in action_low]
        cfg.action_high[:] = [float(v) for v in action_high]
        cfg.time_limit_steps, cfg.auto_reset = int(time_limit_steps), int(auto_reset)
        for a, name in enumerate(car_tasks or ()):
            if name is not None:
                if name not in TASKS:
                    raise ValueError(f"car_tasks[{a}] must be one of {sorted(TASKS)}, got {name!r}")
                cfg.car_task[a] = TASKS[name]
        cfg.n_steps = int(n_steps)
        if lidar_transform not in LIDAR_TRANSFORMS:
            raise ValueError(f"lidar_transform must be one of {sorted(LIDAR_TRANSFORMS)}, got {lidar_transform!r}")
        cfg.lidar_transform = LIDAR_TRANSFORMS[lidar_transform]
        cfg.arena_total_cars, cfg.arena_first_car = int(arena_total_cars), int(arena_first_car)
        nbytes = self._lib.rc_arena_bytes(C.byref(cfg))
        if shared_arena is not None:
            if shared_arena.dtype != torch.uint8 or shared_arena.numel() < nbytes or shared_arena.data_ptr() % 64:
                raise ValueError(f"shared_arena must be a 64-byte aligned uint8 tensor of at least {nbytes} bytes")
            self.arena, self._arena_view = shared_arena, shared_arena[:nbytes]
            self.stream = stream if stream is not None else torch.cuda.Stream(device=self.device)
        else:
            with torch.cuda.device(self.device):
                self.arena = torch.zeros(nbytes + 64, dtype=torch.uint8, device=self.device)
                self.stream = stream if stream is not None else torch.cuda.Stream(device=self.device)
            pad = (-self.arena.data_ptr()) % 64
            self._arena_view = self.arena[pad:pad + nbytes]
        cfg.external_arena = self._arena_view.data_ptr()
        cfg.external_arena_bytes = nbytes
        cfg.stream = self.stream.cuda_stream
        self._cfg = cfg
        self._h = C.c_void_p()
        L.check(self._lib.rc_create(C.byref(cfg), C.byref(self._h)))
        self._load_track(self.track)
        self.views: Dict[str, torch.Tensor] = {}
        self._host_layout = {}
        base = self._arena_view.data_ptr()
        for name, (fid, dtype, tail) in _FIELD_VIEWS.items():
            if fid == L.F_OCCUPANCY and obs_type == "lidar":
                continue
            ptr, nb = C.c_void_p(), C.c_size_t()
            L.check(self._lib.rc_get(self._h, fid, C.byref(ptr), C.byref(nb)))
            off = ptr.value - base
            t = self._arena_view[off:off + nb.value].view(dtype)
            self.views[name] = t.view(self.num_envs, self.cars_per_env, *tail)
            self._host_layout[name] = (off, nb.value, str(dtype).replace("torch.", ""), tail)
        self.slab = self.summary_slab = None
        if not (arena_total_cars and arena_total_cars != self.n_cars):     # (a slice of a shared arena has no slab of its own)
            ptr, nb = C.c_void_p(), C.c_size_t()
            L.check(self._lib.rc_trajectory_slab(self._h, C.byref(ptr), C.byref(nb)))
            self.slab = self._arena_view[ptr.value - base:ptr.value - base + nb.value]
            # the record without the bulky observations: pose .. time (76 B per car), contiguous in the arena
            p0, n0, p1, n1 = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_size_t()
            L.check(self._lib.rc_get(self._h, L.F_POSE, C.byref(p0), C.byref(n0)))
            L.check(self._lib.rc_get(self._h, L.F_TIME, C.byref(p1), C.byref(n1)))
            self.summary_slab = self._arena_view[p0.value - base:p1.value - base + n1.value]
        self._own_views = self.views
        if profiling:
            self.set_profiling(True)

    # ------------------------------------------------------------------ plumbing
    def _load_track(self, t: Track) -> None:
        occ = np.ascontiguousarray(t.occ_words, np.uint32)
        drv = np.ascontiguousarray(t.drv_words, np.uint32)
        prog = np.ascontiguousarray(t.progress, np.float32)
        cl = np.ascontiguousarray(t.centerline, np.float32)
        L.check(self._lib.rc_load_track(
            self._h, occ.ctypes.data, drv.ctypes.data, prog.ctypes.data, t.height, t.width, t.pitch,
            np.float32(t.resolution), np.float32(t.origin[0]), np.float32(t.origin[1]), cl.ctypes.data, len(cl)))
        if self.obs_type == "lidar_occupancy_reference":
            # where the grid lies in the source image the reference's GridMap.to_pixel indexes (compat/racecar_gym/core/gridmaps.py)
            r0, c0, fh, _fw = t.crop
            res = float(t.resolution)
            L.check(self._lib.rc_set_source_frame(self._h, int(fh), int(r0 + t.height - 1), int(c0), C.c_double(float(t.origin[0]) - c0 * res),
                                                  C.c_double(float(t.origin[1]) - (fh - (r0 + t.height)) * res), C.c_double(res)))

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.rc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self) -> None:
        L.check(self._lib.rc_sync(self._h))

    def _enter(self) -> None:
        # order the env's stream after whatever produced the actions on torch's current stream; nothing to do
        # when the caller already works on the env's stream (`with torch.cuda.stream(env.stream): ...`), which
        # saves two cross-stream event waits per call (a few microseconds each on the device queue)
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream != self.stream.cuda_stream:
            self.stream.wait_stream(cur)

    def _exit(self) -> None:
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream != self.stream.cuda_stream:
            cur.wait_stream(self.stream)

    # ------------------------------------------------------------------ env API
    def reset(self, mask: Optional[Union[np.ndarray, torch.Tensor]] = None, mode: str = "grid",
              seed: Optional[int] = None) -> Dict[str, torch.Tensor]:
        if mode not in spec.RESET_MODES:
            raise ValueError(f"reset mode must be one of {sorted(spec.RESET_MODES)}, got {mode!r}")
        if seed is not None:
            self.seed = int(seed)
        mptr = None
        if mask is not None:
            m = np.ascontiguousarray(torch.as_tensor(mask).cpu().numpy().astype(np.uint8).reshape(self.num_envs))
            mptr = m.ctypes.data
        self._enter()
        L.check(self._lib.rc_reset(self._h, mptr, spec.RESET_MODES[mode], C.c_uint64(self.seed)))
        self._exit()
        return self.views

    def step(self, actions: Optional[torch.Tensor] = None, repeat: Optional[int] = None) -> Dict[str, torch.Tensor]:
        """One agent step.  ``actions``: float32 [num_envs, cars_per_env, 2] = (motor, steering) on the
        env's device, or None to use the device-side ``action_in`` buffer."""
        repeat = self.action_repeat if repeat is None else int(repeat)
        ptr = None
        if actions is not None:
            if actions.device != self.device:
                actions = actions.to(self.device, non_blocking=True)
            actions = actions.to(torch.float32).contiguous()
            if actions.numel() != self.n_cars * 2:
                raise ValueError(f"actions must hold {self.n_cars} x 2 values, got shape {tuple(actions.shape)}")
            ptr = actions.data_ptr()
        self._enter()
        L.check(self._lib.rc_step(self._h, ptr, repeat))
        self._exit()
        return self.views

    def step_random(self, seed: int, step: int, repeat: Optional[int] = None) -> Dict[str, torch.Tensor]:
        """``fill_random_actions(seed, step)`` + ``step(None, repeat)`` in one pass of the dynamics kernel (identical
        results): the step of a synthetic random-action rollout (the reference's default prefill policy,
        dreamer/dream.py:207-210)."""
        repeat = self.action_repeat if repeat is None else int(repeat)
        self._enter()
        L.check(self._lib.rc_step_random(self._h, C.c_uint64(seed), C.c_uint32(step), repeat))
        self._exit()
        return self.views

    def set_pose(self, xyyaw) -> Dict[str, torch.Tensor]:
        """Teleport every car: float32 [num_envs, cars_per_env, 3] = x, y, yaw; recomputes the observation."""
        a = np.ascontiguousarray(np.asarray(xyyaw, np.float32).reshape(self.n_cars, 3))
        self._enter()
        L.check(self._lib.rc_set_pose(self._h, a.ctypes.data))
        self._exit()
        return self.views

    def set_raycast_variant(self, variant: int) -> None:
        """0 = plain traversal, 1 = free-rectangle skipping, 2 = tuned skipping, 3 = tuned + packed block table
        in LDS, 4 = the same reading the table through L1/L2, 5 = per-cell distance table through L1/L2,
        6 = per-cell, per-quadrant free rectangles through L1/L2,
        7 = the same with one wave per car (default).  All variants return identical results.  Variants 0-6 are not in the
        shipped library: they live in the lab library (csrc/racecar_lab.hip), which is compiled here on first request."""
        if int(variant) != 7:
            _ensure_lab()
        L.check(self._lib.rc_set_raycast_variant(self._h, int(variant)))

    # ------------------------------------------------------------------ half-size record + multi-GPU gather
    def enable_compact(self, buffers: int = 2) -> None:
        """Make the scan also store the LiDAR row as uint16, followed by the 76 B/car summary (`rc_set_compact_slab`):
        the 2 236 B/car record of the `full-u16` gather.  `buffers` slabs take turns (`rotate_compact`), so a gather
        of slab k overlaps the step that fills slab k + 1."""
        nbytes = self._lib.rc_compact_bytes(C.byref(self._cfg))
        slabs = []
        for _ in range(max(1, int(buffers))):
            raw = torch.zeros(nbytes + 64, dtype=torch.uint8, device=self.device)
            pad = (-raw.data_ptr()) % 64
            slabs.append((raw, raw[pad:pad + nbytes]))
        L.check(self._lib.rc_set_compact_slab(self._h, slabs[0][1].data_ptr(), nbytes))
        a, b, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
        L.check(self._lib.rc_compact_layout(self._h, C.byref(a), C.byref(b), C.byref(c)))
        self.compact_layout = (a.value, b.value, c.value)       # uint16 bytes, summary offset, summary bytes
        self._compact, self._compact_k = slabs, 0
        self.compact = slabs[0][1]

    def rotate_compact(self) -> torch.Tensor:
        """Point the next step's compact record at the next slab of the set; returns the slab just completed."""
        done = self.compact
        self._compact_k = (self._compact_k + 1) % len(self._compact)
        self.compact = self._compact[self._compact_k][1]
        L.check(self._lib.rc_set_compact_slab(self._h, self.compact.data_ptr(), self.compact.numel()))
        return done

    def disable_compact(self) -> None:
        L.check(self._lib.rc_set_compact_slab(self._h, None, 0))
        self.compact = None

    def gather_source(self, mode: str) -> torch.Tensor:
        """The bytes one gather mode sends: 'full' (fp32 record), 'full-u16' (compact slab), 'summary' (pose..time)."""
        if mode == "full":
            return self.slab
        if mode == "summary":
            return self.summary_slab
        if mode == "full-u16":
            if getattr(self, "compact", None) is None:
                raise L.RacecarHipError("gather mode 'full-u16' needs enable_compact() first")
            return self.compact
        raise ValueError(f"unknown gather mode {mode!r}")

    def comm_init(self, unique_id: bytes, rank: int, world: int) -> None:
        """RCCL communicator of this handle (`rc_comm_init`); `unique_id` from `comm_unique_id()` on rank 0."""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        L.check(self._lib.rc_comm_init(self._h, buf, 128, int(rank), int(world)))
        self.comm_world = int(world)

    @staticmethod
    def comm_unique_id() -> bytes:
        lib = L.load_library()
        buf = C.create_string_buffer(128)
        L.check(lib.rc_comm_unique_id(buf, 128))
        return buf.raw

    def gather(self, mode: str, dst: torch.Tensor) -> None:
        """`rc_gather_trajectory`: asynchronous RCCL all-gather of the last step's record into `dst` (uint8,
        world x gather_bytes(mode))."""
        L.check(self._lib.rc_gather_trajectory(self._h, L.GATHER_MODES[mode], dst.data_ptr(), dst.numel()))

    def gather_bytes(self, mode: str) -> int:
        return int(self._lib.rc_gather_bytes(self._h, L.GATHER_MODES[mode]))

    def gather_wait(self, host_sync: bool = True) -> None:
        L.check(self._lib.rc_gather_wait(self._h, int(bool(host_sync))))

    # ---- the same gather as direct peer copies (rc_gather_trajectory_p2p: hipIpc handles, one copy stream per peer) ----
    def p2p_setup(self, mode: str, rank: int, world: int) -> bytes:
        """Allocate this rank's destination and flags for peer-copy gathers (sized for the largest payload) and select
        `mode`; returns the 256-byte blob the other ranks need.  Hand every rank's blob (in rank order) to `p2p_connect`
        on every rank.  Called again it only switches the payload: same buffers, same blob."""
        buf = C.create_string_buffer(L.P2P_EXPORT_BYTES)
        L.check(self._lib.rc_p2p_setup(self._h, L.GATHER_MODES[mode], int(rank), int(world), buf, L.P2P_EXPORT_BYTES))
        self._p2p_mode, self._p2p_world = mode, int(world)
        return buf.raw

    def p2p_connect(self, blobs) -> None:
        raw = b"".join(bytes(b) for b in blobs)
        L.check(self._lib.rc_p2p_connect(self._h, C.create_string_buffer(raw, len(raw)), len(raw)))

    def gather_p2p(self, mode: Optional[str] = None) -> None:
        """Send the last step's record into every rank's buffer (asynchronous, behind the work on the env's stream)."""
        if mode is not None and mode != self._p2p_mode:
            raise ValueError(f"the peer-copy gather was set up for {self._p2p_mode!r}, not {mode!r}")
        L.check(self._lib.rc_gather_trajectory_p2p(self._h))

    def gather_p2p_wait(self, host_sync: bool = True):
        """Order the env's stream (and the host) behind the arrival of the last issued gather; returns (device pointer,
        bytes) of the gathered slot: rank r's record at r * bytes / world."""
        ptr, nb = C.c_void_p(), C.c_size_t()
        L.check(self._lib.rc_gather_p2p_wait(self._h, int(bool(host_sync)), C.byref(ptr), C.byref(nb)))
        return ptr.value, nb.value

    def gathered_p2p_host(self, back: int = 0) -> np.ndarray:
        """Host copy of the last gathered slot (back = 1: of the gather before it) as uint8 [world, bytes per rank]
        (synchronising)."""
        ptr, nb = self.gather_p2p_wait(host_sync=True)
        if back:
            p2, n2 = C.c_void_p(), C.c_size_t()
            L.check(self._lib.rc_p2p_slot(self._h, int(back), C.byref(p2), C.byref(n2)))
            ptr, nb = p2.value, n2.value
        out = np.empty(nb, np.uint8)
        L.check(self._lib.rc_copy_from_device(self._h, ptr, out.ctypes.data, nb))
        # the slot's entries are sized for the largest payload: the current one fills the head of each
        return out.reshape(self._p2p_world, -1)[:, :self.gather_bytes(self._p2p_mode)]

    def p2p_disconnect(self) -> None:
        """Wait for this rank's copies and unmap the peers' buffers.  Every rank disconnects, the ranks synchronise (the
        caller's barrier), then they `p2p_teardown()`: exported memory must not be freed while a peer still maps it."""
        L.check(self._lib.rc_p2p_disconnect(self._h))

    def p2p_teardown(self) -> None:
        L.check(self._lib.rc_p2p_teardown(self._h))

    def comm_count(self) -> int:
        n = C.c_int32()
        L.check(self._lib.rc_comm_count(self._h, C.byref(n)))
        return int(n.value)

    def scan_kernel_name(self) -> str:
        buf = C.create_string_buffer(128)
        L.check(self._lib.rc_scan_kernel_name(self._h, buf, 128))
        return buf.value.decode()

    def scan_overruns(self) -> int:
        """Waves of this handle's BOUNDED scans that used up a round's trip budget (0 unless a band was mis-set)."""
        n = C.c_uint64()
        L.check(self._lib.rc_scan_overruns(self._h, C.byref(n)))
        return int(n.value)

    def debug_set(self, knob: str, value: int) -> None:
        """Experiment / validation knobs of the scan (`rc_debug_set`; 0 = production behaviour): ray_threads,
        ray_split, ray_wg_per_cu, band_log2.  Used by tools/knob_sweep.sh and the band-sensitivity check of
        tests/test_gpu_parity.py; the library itself reads nothing from the process environment."""
        L.check(self._lib.rc_debug_set(self._h, L.DEBUG_KNOBS[knob], int(value)))

    def debug_scan_stamps(self, n_waves: int = 0) -> Optional[torch.Tensor]:
        """In-kernel time stamps of the default scan (`rc_debug_scan_stamps`, analysis only; tools/scan_stamps.py): with
        n_waves > 0 the following scans run the instrumented kernel and its first n_waves waves fill the returned
        int64 [n_waves, 32] device tensor; n_waves = 0 switches back to the production kernel."""
        if n_waves <= 0:
            L.check(self._lib.rc_debug_scan_stamps(self._h, None, 0))
            self._stamps = None
            return None
        _ensure_lab()                    # the instrumented build is a lab kernel
        self._stamps = torch.zeros((n_waves, 32), dtype=torch.int64, device=self.device)
        L.check(self._lib.rc_debug_scan_stamps(self._h, self._stamps.data_ptr(), int(n_waves)))
        return self._stamps

    def follow_the_gap(self, motor_straight: float = 0.6, motor_corner: float = 0.3) -> torch.Tensor:
        """Batched follow-the-gap agent (dreamer/dream.py:211-216 prefill): fills and returns `action_in`
        from the current LiDAR scans; pass None to step() to apply it."""
        self._enter()
        L.check(self._lib.rc_follow_the_gap(self._h, motor_straight, motor_corner))
        self._exit()
        return self.views["action_in"]

    def follow_the_gap_reference(self, dt: Optional[float] = None, detail: bool = False):
        """The reference's own follow-the-gap law (ros_agent/agents/follow_the_gap/src/agent.py:128-234) as a device agent:
        fills and returns `action_in`; with detail=True also a float32 [n_cars, 4] tensor of heading, free distance,
        steering angle and speed.  dt defaults to the env's agent step (0.01 s x action_repeat)."""
        dt = 0.01 * self.action_repeat if dt is None else float(dt)
        det = torch.empty((self.n_cars, 4), dtype=torch.float32, device=self.device) if detail else None
        self._enter()
        L.check(self._lib.rc_follow_the_gap_reference(self._h, dt, det.data_ptr() if detail else None))
        self._exit()
        return (self.views["action_in"], det) if detail else self.views["action_in"]

    def fill_random_actions(self, seed: int, step: int) -> None:
        L.check(self._lib.rc_fill_random_actions(self._h, C.c_uint64(seed), C.c_uint32(step)))

    # ------------------------------------------------------------------ profiling
    def set_profiling(self, on, kernels=None) -> None:
        """HIP-event timing of the kernels: on/off, or only the listed kernel ids (L.K_*)."""
        mask = int(bool(on))
        if on and kernels is not None:
            mask = 0
            for k in kernels:
                mask |= 1 << k
            mask |= 1 << 31 if mask == 1 else 0      # keep a single-kernel mask distinct from "1 = all"
        L.check(self._lib.rc_set_profiling(self._h, mask))

    def reset_kernel_times(self) -> None:
        L.check(self._lib.rc_reset_kernel_times(self._h))

    def kernel_times(self) -> Dict[str, Dict[str, float]]:
        out = {}
        for k, name in L.KERNEL_NAMES.items():
            ms, n = C.c_double(), C.c_uint64()
            L.check(self._lib.rc_kernel_time(self._h, k, C.byref(ms), C.byref(n)))
            out[name] = {"total_ms": ms.value, "launches": int(n.value),
                         "avg_ms": ms.value / n.value if n.value else 0.0}
        return out

    @property
    def arena_nbytes(self) -> int:
        return int(self._arena_view.numel())

    def views_of(self, arena: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Typed views [num_envs, cars_per_env, ...] of every output field inside another arena-sized uint8 buffer."""
        out = {}
        for name, (off, nb, dtype, tail) in self._host_layout.items():
            out[name] = arena[off:off + nb].view(getattr(torch, dtype)).view(self.num_envs, self.cars_per_env, *tail)
        return out

    def set_arena(self, arena: Optional[torch.Tensor], views: Optional[Dict[str, torch.Tensor]] = None) -> None:
        """Re-point the outputs of the following reset()/step() calls at `arena` (uint8, >= arena_nbytes, 64-byte
        aligned, on the env's device); None = back to the env's own arena.  `action_in` stays where it is.  This is
        how `replay.TrajectoryRing` records trajectories on the device without copies."""
        if arena is None:
            L.check(self._lib.rc_set_arena(self._h, None, 0))
            self.views = self._own_views
            return
        if arena.dtype != torch.uint8 or arena.device != self.device or not arena.is_contiguous():
            raise ValueError("arena must be a contiguous uint8 tensor on the env's device")
        L.check(self._lib.rc_set_arena(self._h, arena.data_ptr(), arena.numel()))
        new = dict(views) if views is not None else self.views_of(arena)
        new["action_in"] = self._own_views["action_in"]
        self.views = new

    def gather_rows(self, ring: torch.Tensor, slot_bytes: int, slots: torch.Tensor, cars: torch.Tensor, names) -> Dict[str, torch.Tensor]:
        """`rc_gather_rows`: for every row r the record of car `cars[r]` in ring slot `slots[r]` (int32 device tensors),
        the fields `names`, as one launch.  Returns name -> tensor [rows, ...] (views of one fresh buffer)."""
        unknown = [n for n in names if not (n in _FIELD_VIEWS and n != "action_in" and n in self._host_layout)]
        if unknown:
            raise ValueError(f"fields {unknown} are not recorded by this env")
        order = sorted(names, key=lambda n: _FIELD_VIEWS[n][0])
        mask = 0
        for n in order:
            mask |= 1 << _FIELD_VIEWS[n][0]
        rows = int(slots.numel())
        nbytes = int(self._lib.rc_gather_rows_bytes(self._h, mask, rows))
        out = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        slots = slots.to(torch.int32).contiguous()
        cars = cars.to(torch.int32).contiguous()
        self._enter()
        L.check(self._lib.rc_gather_rows(self._h, ring.data_ptr(), int(slot_bytes), slots.data_ptr(), cars.data_ptr(), rows, mask,
                                         out.data_ptr(), nbytes))
        self._exit()
        res, off = {}, 0
        for n in order:
            fid, dtype, tail = _FIELD_VIEWS[n]
            per = self._host_layout[n][1] // self.n_cars
            res[n] = out[off:off + per * rows].view(dtype).view(rows, *tail)
            off = (off + per * rows + 63) // 64 * 64
        return res

    def sample_windows(self, ring: torch.Tensor, slot_bytes: int, capacity: int, oldest: int, count: int, length: int,
                       n_windows: int, seed: int, draw: int, max_tries: int = 16) -> Dict[str, torch.Tensor]:
        """`rc_sample_windows`: window starts of a replay sampler drawn on the device - no host round trip.  Returns int32
        device tensors `slots`, `slots_obs`, `cars` [n_windows * length] (rows for `gather_rows`), `meta` [n_windows, 4] =
        (t0, car, terminal, first) and `failed` [1] (windows that found no episode-internal start in `max_tries` draws)."""
        i32 = dict(dtype=torch.int32, device=self.device)
        out = dict(slots=torch.empty(n_windows * length, **i32), slots_obs=torch.empty(n_windows * length, **i32),
                   cars=torch.empty(n_windows * length, **i32), meta=torch.empty((n_windows, 4), **i32),
                   failed=torch.zeros(1, dtype=torch.int32, device=self.device))
        self._enter()
        L.check(self._lib.rc_sample_windows(self._h, ring.data_ptr(), int(slot_bytes), int(capacity), int(oldest), int(count),
                                            int(length), int(n_windows), C.c_uint64(int(seed) & (2 ** 64 - 1)), C.c_uint32(int(draw) & 0xffffffff),
                                            int(max_tries), out["slots"].data_ptr(), out["slots_obs"].data_ptr(),
                                            out["cars"].data_ptr(), out["meta"].data_ptr(), out["failed"].data_ptr()))
        self._exit()
        return out

    def sample_batch_layout(self, names, n_windows: int, length: int) -> Dict[str, object]:
        """Layout of the ONE buffer `sample_batch` fills (include/racecar_hip.h, rc_sample_batch): field sections in field
        order (64-byte aligned), then meta int32 [n_windows, 4], then the 64-byte failure block - `payload` bytes in all, what
        a sharded store exchanges - then the sampler's row indices; `total` bytes to allocate."""
        unknown = [n for n in names if not (n in _FIELD_VIEWS and n != "action_in" and n in self._host_layout)]
        if unknown:
            raise ValueError(f"fields {unknown} are not recorded by this env")
        order = sorted(names, key=lambda n: _FIELD_VIEWS[n][0])
        mask = 0
        for n in order:
            mask |= 1 << _FIELD_VIEWS[n][0]
        payload, meta_off = C.c_size_t(), C.c_size_t()
        total = int(self._lib.rc_sample_batch_bytes(self._h, mask, int(n_windows), int(length), C.byref(payload), C.byref(meta_off)))
        if total == 0:
            raise ValueError(f"no recorded field among {list(names)}")
        rows, off, fields = int(n_windows) * int(length), 0, {}
        for n in order:
            fid, dtype, tail = _FIELD_VIEWS[n]
            per = self._host_layout[n][1] // self.n_cars
            fields[n] = (off, per * rows, dtype, tail)
            off = (off + per * rows + 63) // 64 * 64
        assert off == meta_off.value, (off, meta_off.value)
        return {"mask": mask, "total": total, "payload": int(payload.value), "meta": int(meta_off.value), "failed": int(meta_off.value) +
                (16 * int(n_windows) + 63) // 64 * 64, "fields": fields, "n_windows": int(n_windows), "length": int(length)}

    def sample_batch(self, ring: torch.Tensor, slot_bytes: int, capacity: int, oldest: int, count: int, layout: Dict[str, object],
                     seed: int, draw: int, reset_rows: bool = True, max_tries: int = 16, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`rc_sample_batch`: one training batch - windows drawn, rows gathered, reset rows written - as one call into one
        packed uint8 buffer (`sample_batch_layout`); returns the buffer (`out`, or a fresh one)."""
        if out is None:
            out = torch.empty(layout["total"] + 64, dtype=torch.uint8, device=self.device)
            out = out[(-out.data_ptr()) % 64:][:layout["total"]]
        self._enter()
        L.check(self._lib.rc_sample_batch(self._h, ring.data_ptr(), int(slot_bytes), int(capacity), int(oldest), int(count), layout["length"],
                                          layout["n_windows"], C.c_uint64(int(seed) & (2 ** 64 - 1)), C.c_uint32(int(draw) & 0xffffffff),
                                          int(max_tries), layout["mask"], int(bool(reset_rows)), out.data_ptr(), int(out.numel())))
        self._exit()
        return out

    def host_snapshot(self) -> Dict[str, np.ndarray]:
        """Every output field on the host from ONE device-to-host copy of the arena (for small batches, e.g. the
        single-env shim): dict of NumPy views [num_envs, cars_per_env, ...] into one host buffer."""
        self.sync()
        buf = self._arena_view.cpu().numpy()
        out = {}
        for name, (off, nb, dtype, tail) in self._host_layout.items():
            out[name] = buf[off:off + nb].view(dtype).reshape(self.num_envs, self.cars_per_env, *tail)
        return out

    def host(self, name: str) -> np.ndarray:
        """Synchronised host copy of one output field."""
        self.sync()
        return self.views[name].cpu().numpy()


class MixedTrackEnv:
    """A batch that mixes tracks by blocks of envs (SURVEY.md 8e "per-env track"; BASELINE configs[4]'s track mix on ONE GPU):
    one `BatchedRaceEnv` handle per track, all of them filling their slice of ONE output arena, so the caller
    sees a single set of tensors `[total_envs, cars_per_env, ...]` and `track_id[total_envs]` (each handle on a stream of its own,
    forked from / joined to the caller's current stream around every call).  Env e of the batch is env e
    of the job: its reset stream is keyed by `first_env + e` whatever block it lies in, so a block of track T behaves
    exactly like the same envs in a single-track batch.  A step is one (dynamics, scan[, render]) launch set per track.

        env = MixedTrackEnv(["columbia", "austria", "barcelona"], [21846, 21845, 21845], auto_reset=True)
        out = env.reset(mode="random", seed=0); out = env.step(actions)       # actions float32 [65536, 1, 2] on the device
    """

    def __init__(self, tracks, envs_per_track, cars_per_env: int = 1, obs_type: str = "lidar", device: int = 0,
                 first_env: int = 0, **kw):
        if len(tracks) != len(envs_per_track) or not tracks:
            raise ValueError("one env count per track")
        self._lib = L.load_library()
        if not torch.cuda.is_available():
            raise L.RacecarHipError("no HIP device visible to torch; MixedTrackEnv has no CPU fallback")
        self.device = torch.device("cuda", device)
        self.num_envs, self.cars_per_env = int(sum(envs_per_track)), int(cars_per_env)
        self.n_cars = self.num_envs * self.cars_per_env
        cfg = L.RcConfig()
        self._lib.rc_default_config(C.byref(cfg))
        cfg.num_envs, cfg.cars_per_env, cfg.obs_type = self.num_envs, self.cars_per_env, OBS_TYPES[obs_type]
        nbytes = self._lib.rc_arena_bytes(C.byref(cfg))
        with torch.cuda.device(self.device):
            raw = torch.zeros(nbytes + 64, dtype=torch.uint8, device=self.device)
            self.stream = torch.cuda.Stream(device=self.device)      # a stream for callers that want one (bench.py works on it)
        pad = (-raw.data_ptr()) % 64
        self._raw, self.arena = raw, raw[pad:pad + nbytes]
        self.parts, self.blocks = [], []
        e0 = 0
        for track, n in zip(tracks, envs_per_track):
            self.parts.append(BatchedRaceEnv(track, int(n), cars_per_env, obs_type=obs_type, device=device, first_env=first_env + e0,
                                             shared_arena=self.arena, arena_total_cars=self.n_cars,
                                             arena_first_car=e0 * self.cars_per_env, stream=self.stream, **kw))
            self.blocks.append((e0, e0 + int(n)))
            e0 += int(n)
        self.track_id = torch.cat([torch.full((b - a,), i, dtype=torch.int32) for i, (a, b) in enumerate(self.blocks)]).to(self.device)
        self.views: Dict[str, torch.Tensor] = {}
        off, per = C.c_size_t(), C.c_size_t()
        for name, (fid, dtype, tail) in _FIELD_VIEWS.items():
            L.check(self._lib.rc_field_layout(C.byref(cfg), fid, C.byref(off), C.byref(per)))
            if per.value == 0:
                continue
            t = self.arena[off.value:off.value + per.value * self.n_cars].view(dtype)
            self.views[name] = t.view(self.num_envs, self.cars_per_env, *tail)

    @classmethod
    def from_track_ids(cls, tracks, track_id, **kw):
        """An ARBITRARY per-env track assignment (`track_id[e]` indexes `tracks`): the envs are laid out in the arena sorted by
        track (stable), because a workgroup of the render shares one bitmap in LDS and a wave's table lines should be its
        neighbours'; `row_of_env[e]` is the arena row of the caller's env e, `env_of_row` its inverse, and `to_rows` /
        `to_envs` reorder a leading-axis tensor between the two orders (one indexed copy).  Row r draws from reset stream
        `first_env + r`."""
        ids = torch.as_tensor(track_id, dtype=torch.int64).cpu().reshape(-1)
        if ids.numel() == 0 or int(ids.min()) < 0 or int(ids.max()) >= len(tracks):
            raise ValueError("track_id must index tracks")
        used = [i for i in range(len(tracks)) if bool((ids == i).any())]
        env = cls([tracks[i] for i in used], [int((ids == i).sum()) for i in used], **kw)
        env.env_of_row = torch.argsort(ids, stable=True).to(env.device)
        env.row_of_env = torch.empty_like(env.env_of_row)
        env.row_of_env[env.env_of_row] = torch.arange(ids.numel(), device=env.device)
        env.track_id = torch.as_tensor(used, dtype=torch.int32, device=env.device)[env.track_id.long()]   # ids as the caller numbered them
        return env

    def to_rows(self, x: torch.Tensor) -> torch.Tensor:
        """Caller order [num_envs, ...] -> arena order (e.g. actions before `step`)."""
        return x.to(self.device)[self.env_of_row]

    def to_envs(self, x: torch.Tensor) -> torch.Tensor:
        """Arena order -> caller order (e.g. `to_envs(out["lidar"])`)."""
        return x[self.row_of_env]

    # All blocks work on ONE stream (`self.stream`), ordered behind and before the caller's current stream around each call.  A
    # step is ONE dynamics and ONE scan launch over all blocks (`rc_step_group`: every wave works from the parameters of the
    # block it lies in): with a launch pair per block on streams of their own, each block's small dynamics kernel waited behind
    # the previous block's scan, the three scans shared the chip and 33 us passed between the join of one step and the first
    # kernel of the next (0.258 -> 0.208 ms per step of three blocks of 21 845 envs; the three tracks alone average 0.20;
    # EXPERIMENTS I.10).
    def _ordered(self, call):
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream != self.stream.cuda_stream:
            self.stream.wait_stream(cur)
        L.check(call())
        if cur.cuda_stream != self.stream.cuda_stream:
            cur.wait_stream(self.stream)
        return self.views

    def _fork_join(self, call):
        def every_block():
            for p, blk in zip(self.parts, self.blocks):
                rc = call(p, blk)
                if rc:
                    return rc
            return 0
        return self._ordered(every_block)

    def _handles(self):
        if getattr(self, "_handle_array", None) is None:
            self._handle_array = (C.c_void_p * len(self.parts))(*[p._h for p in self.parts])
        return self._handle_array

    def reset(self, mode: str = "grid", seed: Optional[int] = None):
        if mode not in spec.RESET_MODES:
            raise ValueError(f"reset mode must be one of {sorted(spec.RESET_MODES)}, got {mode!r}")
        for p in self.parts:
            if seed is not None:
                p.seed = int(seed)
        return self._fork_join(lambda p, blk: p._lib.rc_reset(p._h, None, spec.RESET_MODES[mode], C.c_uint64(p.seed)))

    def step(self, actions: Optional[torch.Tensor] = None, repeat: Optional[int] = None):
        """actions: float32 [total_envs, cars_per_env, 2] on the device (None: each block's `action_in`)."""
        if actions is not None:
            actions = actions.to(self.device, torch.float32).reshape(self.num_envs, self.cars_per_env, 2).contiguous()
        rep = self.parts[0].action_repeat if repeat is None else int(repeat)
        if len(self.parts) > 8:             # (more blocks than a group launch carries: one launch pair per block)
            return self._fork_join(lambda p, blk: p._lib.rc_step(p._h, None if actions is None else actions[blk[0]:blk[1]].data_ptr(), rep))
        return self._ordered(lambda: self._lib.rc_step_group(self._handles(), len(self.parts),
                                                             None if actions is None else actions.data_ptr(), rep))

    def step_random(self, seed: int, step: int, repeat: Optional[int] = None):
        rep = self.parts[0].action_repeat if repeat is None else int(repeat)
        if len(self.parts) > 8:
            return self._fork_join(lambda p, blk: p._lib.rc_step_random(p._h, C.c_uint64(seed), C.c_uint32(step), rep))
        return self._ordered(lambda: self._lib.rc_step_random_group(self._handles(), len(self.parts), C.c_uint64(seed), C.c_uint32(step), rep))

    def follow_the_gap_reference(self, dt: Optional[float] = None):
        self._fork_join(lambda p, blk: p._lib.rc_follow_the_gap_reference(p._h, 0.01 * p.action_repeat if dt is None else float(dt), None))
        return self.views["action_in"]

    def sync(self):
        for p in self.parts:
            p.sync()

    def close(self):
        for p in self.parts:
            p.close()
