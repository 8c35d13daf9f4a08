"""ctypes binding of libracecar_hip.so (include/racecar_hip.h).

There is no CPU fallback: if the library has not been built, importing the binding raises with
the build command.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``python -m racing_dreamer_amd.build``.
"""
from __future__ import annotations

import ctypes as C
import os

LIB_DIR = os.path.join(os.path.dirname(__file__), "lib")
LIB_PATH = os.path.join(LIB_DIR, "libracecar_hip.so")

RC_ABI_VERSION = 3          # include/racecar_hip.h
RC_N_BEAMS = 1080
RC_PATCH = 64
RC_MAX_CARS = 4

# rc_field
(F_LIDAR, F_POSE, F_VELOCITY, F_SPEED, F_ACTION, F_REWARD, F_DISCOUNT, F_PROGRESS_TOTAL, F_TIME, F_OCCUPANCY,
 F_PROGRESS, F_LAP, F_CHECKPOINT, F_DONE, F_TRUNCATED, F_WALL_COLLISION, F_OPPONENT_COLLISION, F_WRONG_WAY,
 F_FRESH, F_ACCELERATION, F_STEERING_ANGLE, F_ACTION_IN, F_COUNT) = range(23)

GATHER_FULL, GATHER_FULL_U16, GATHER_SUMMARY = range(3)
GATHER_MODES = {"full": GATHER_FULL, "full-u16": GATHER_FULL_U16, "summary": GATHER_SUMMARY}

# rc_debug_set knobs (experiments / validation only; all 0 in production)
DBG_RAY_THREADS, DBG_RAY_SPLIT, DBG_RAY_WG_PER_CU, DBG_BAND_LOG2, DBG_PATCH_VARIANT, DBG_SCAN_BOUNDED, DBG_SCAN_ORDER, DBG_EXACT_CHUNK = range(8)
DEBUG_KNOBS = {"ray_threads": DBG_RAY_THREADS, "ray_split": DBG_RAY_SPLIT, "ray_wg_per_cu": DBG_RAY_WG_PER_CU,
               "band_log2": DBG_BAND_LOG2, "patch_variant": DBG_PATCH_VARIANT, "scan_bounded": DBG_SCAN_BOUNDED,
               "scan_order": DBG_SCAN_ORDER, "exact_chunk": DBG_EXACT_CHUNK}
P2P_EXPORT_BYTES = 256

K_DYNAMICS, K_RAYCAST, K_PATCH, K_RESET, K_ACTIONS, K_FTG, K_COUNT = range(7)
KERNEL_NAMES = {K_DYNAMICS: "rc_dynamics_kernel", K_RAYCAST: "rc_raycast_kernel", K_PATCH: "rc_patch_kernel",
                K_RESET: "rc_reset_kernel", K_ACTIONS: "rc_random_actions_kernel", K_FTG: "rc_ftg_kernel"}


class RcConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("device", C.c_int32), ("num_envs", C.c_int32), ("cars_per_env", C.c_int32),
        ("first_env", C.c_int64), ("obs_type", C.c_int32), ("task", C.c_int32), ("laps", C.c_int32),
        ("time_limit", C.c_float), ("terminate_on_collision", C.c_int32), ("collision_reward", C.c_float),
        ("remap_actions", C.c_int32), ("action_low", C.c_float * 2), ("action_high", C.c_float * 2),
        ("time_limit_steps", C.c_int32), ("auto_reset", C.c_int32), ("lidar_transform", C.c_int32),
        ("external_arena", C.c_void_p),
        ("external_arena_bytes", C.c_size_t), ("stream", C.c_void_p),
        ("car_task", C.c_int32 * 4), ("n_steps", C.c_int32),
        ("arena_total_cars", C.c_int32), ("arena_first_car", C.c_int32),
    ]


# every symbol include/racecar_hip.h declares: name -> (restype, argtypes)
_P = C.POINTER
SYMBOLS = {
    "rc_default_config": (None, [_P(RcConfig)]),
    "rc_arena_bytes": (C.c_size_t, [_P(RcConfig)]),
    "rc_field_layout": (C.c_int, [_P(RcConfig), C.c_int32, _P(C.c_size_t), _P(C.c_size_t)]),
    "rc_create": (C.c_int, [_P(RcConfig), _P(C.c_void_p)]),
    "rc_destroy": (None, [C.c_void_p]),
    "rc_load_track": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int32]),
    "rc_set_source_frame": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double]),
    "rc_reset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint64]),
    "rc_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "rc_step_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "rc_set_pose": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rc_follow_the_gap": (C.c_int, [C.c_void_p, C.c_float, C.c_float]),
    "rc_follow_the_gap_reference": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p]),
    "rc_fill_random_actions": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32]),
    "rc_step_random": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32]),
    "rc_step_group": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]),
    "rc_step_random_group": (C.c_int, [C.c_void_p, C.c_int32, C.c_uint64, C.c_uint32, C.c_int32]),
    "rc_get": (C.c_int, [C.c_void_p, C.c_int32, _P(C.c_void_p), _P(C.c_size_t)]),
    "rc_copy_out": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t]),
    "rc_trajectory_slab": (C.c_int, [C.c_void_p, _P(C.c_void_p), _P(C.c_size_t)]),
    "rc_gather_rows_bytes": (C.c_size_t, [C.c_void_p, C.c_uint32, C.c_int32]),
    "rc_gather_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p, C.c_size_t]),
    "rc_sample_windows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64,
                                    C.c_uint32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rc_sample_batch_bytes": (C.c_size_t, [C.c_void_p, C.c_uint32, C.c_int32, C.c_int32, _P(C.c_size_t), _P(C.c_size_t)]),
    "rc_sample_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64,
                                  C.c_uint32, C.c_int32, C.c_uint32, C.c_int32, C.c_void_p, C.c_size_t]),
    "rc_sync": (C.c_int, [C.c_void_p]),
    "rc_stream": (C.c_void_p, [C.c_void_p]),
    "rc_set_profiling": (C.c_int, [C.c_void_p, C.c_int32]),
    "rc_kernel_time": (C.c_int, [C.c_void_p, C.c_int32, _P(C.c_double), _P(C.c_uint64)]),
    "rc_reset_kernel_times": (C.c_int, [C.c_void_p]),
    "rc_set_raycast_variant": (C.c_int, [C.c_void_p, C.c_int32]),
    "rc_debug_set": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "rc_debug_scan_stamps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "rc_scan_overruns": (C.c_int, [C.c_void_p, _P(C.c_uint64)]),
    "rc_scan_kernel_name": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "rc_comm_count": (C.c_int, [C.c_void_p, _P(C.c_int32)]),
    "rc_p2p_setup": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t]),
    "rc_p2p_connect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "rc_gather_trajectory_p2p": (C.c_int, [C.c_void_p]),
    "rc_gather_p2p_wait": (C.c_int, [C.c_void_p, C.c_int32, _P(C.c_void_p), _P(C.c_size_t)]),
    "rc_p2p_slot": (C.c_int, [C.c_void_p, C.c_int32, _P(C.c_void_p), _P(C.c_size_t)]),
    "rc_p2p_disconnect": (C.c_int, [C.c_void_p]),
    "rc_p2p_teardown": (C.c_int, [C.c_void_p]),
    "rc_device_alloc": (C.c_int, [C.c_void_p, C.c_size_t, _P(C.c_void_p)]),
    "rc_device_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rc_copy_from_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rc_compact_bytes": (C.c_size_t, [_P(RcConfig)]),
    "rc_set_compact_slab": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "rc_compact_layout": (C.c_int, [C.c_void_p, _P(C.c_size_t), _P(C.c_size_t), _P(C.c_size_t)]),
    "rc_comm_library": (C.c_int, [C.c_char_p]),
    "rc_comm_unique_id": (C.c_int, [C.c_void_p, C.c_size_t]),
    "rc_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_int32]),
    "rc_gather_bytes": (C.c_size_t, [C.c_void_p, C.c_int32]),
    "rc_gather_trajectory": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t]),
    "rc_gather_wait": (C.c_int, [C.c_void_p, C.c_int32]),
    "rc_spec_tables": (None, [C.c_void_p, C.c_void_p]),
    "rc_set_arena": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "rc_selftest_reciprocal": (C.c_int, [C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rc_selftest_sqrt": (C.c_int, [C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rc_selftest_div6": (C.c_int, [C.c_int32, _P(C.c_uint64), _P(C.c_uint64)]),
    "rc_selftest_exact_estimate": (C.c_int, [C.c_void_p, _P(C.c_uint64)]),
    "rc_last_error": (C.c_char_p, []),
    "rc_abi_version": (C.c_int, []),
    "rc_build_id": (C.c_char_p, []),
}


class RacecarHipError(RuntimeError):
    pass


_lib = None


def load_library(path: str = LIB_PATH) -> C.CDLL:
    """dlopen the HIP library and bind every declared symbol.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise RacecarHipError(
            f"{path} not found: the HIP extension is not built. Run `python -m racing_dreamer_amd.build` "
            "(needs hipcc; cross-compiles for gfx950 without a GPU). There is no CPU fallback.")
    if path == LIB_PATH and not os.environ.get("RC_ALLOW_STALE_LIBRARY"):
        # A library that does not belong to the sources beside it would let every test pass on yesterday's kernels: refuse it.
        # (The identity is a hash of flags and file CONTENTS compiled into the library, racing_dreamer_amd/build.py; the A/B
        # scripts under tools/, which put variant builds in the library's place, set RC_ALLOW_STALE_LIBRARY=1.)
        from . import build as _build
        if os.path.isdir(_build.CSRC) and _build.needs_build(path):
            raise RacecarHipError(
                f"{path} was built from other sources than the ones in {_build.CSRC} (its build id {_build.library_build_id(path)}, "
                f"theirs {_build.source_hash()}): run `python -m racing_dreamer_amd.build` (or __graft_entry__.build()) first.")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)     # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.rc_abi_version() != RC_ABI_VERSION:
        raise RacecarHipError(f"ABI version mismatch: {path} reports {lib.rc_abi_version()}, this binding expects "
                              f"{RC_ABI_VERSION} - rebuild with `python -m racing_dreamer_amd.build --force`")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load_library().rc_last_error()
        raise RacecarHipError(f"libracecar_hip error {rc}: {msg.decode() if msg else '?'}")
