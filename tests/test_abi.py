"""The C-ABI library loads on a CPU-only box and exports every symbol include/racecar_hip.h declares.
No compute call is made here (there is no GPU); error paths that do not need a device are exercised."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "racecar_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rc_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(hip_lib):
    from racing_dreamer_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(hip_lib, name), f"{name} declared in racecar_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == declared, "ctypes binding and header disagree"
    assert hip_lib.rc_abi_version() == _lib.RC_ABI_VERSION == 3


def test_config_struct_matches_header(hip_lib):
    from racing_dreamer_amd import _lib
    cfg = _lib.RcConfig()
    hip_lib.rc_default_config(C.byref(cfg))
    assert cfg.struct_size == C.sizeof(_lib.RcConfig)
    # defaults = dreamer/scenarios/max_progress/columbia.yml:10 and dream.py:138
    assert (cfg.laps, cfg.time_limit, cfg.terminate_on_collision, cfg.collision_reward) == (10, 180.0, 1, -1.0)
    assert np.allclose(list(cfg.action_low), [0.005, -1.0]) and list(cfg.action_high) == [1.0, 1.0]
    assert cfg.num_envs == 1 and cfg.cars_per_env == 1


def test_arena_layout_is_the_trajectory_record(hip_lib):
    from racing_dreamer_amd import _lib, spec
    cfg = _lib.RcConfig()
    hip_lib.rc_default_config(C.byref(cfg))
    cfg.num_envs = 65536
    per_car = hip_lib.rc_arena_bytes(C.byref(cfg)) / 65536
    assert 4 * spec.RECORD_FLOATS == 4396                       # dreamer/wrappers.py:213-219 record
    assert 4396 <= per_car <= 4396 + 40                         # record + flags/info, no padding waste
    cfg.obs_type = 1
    assert hip_lib.rc_arena_bytes(C.byref(cfg)) / 65536 - per_car == 4096


def test_spec_tables_match_python_spec(hip_lib):
    from racing_dreamer_amd import spec
    beams = np.zeros((1080, 2), np.float32)
    foot = np.zeros((34, 2), np.float32)
    hip_lib.rc_spec_tables(beams.ctypes.data, foot.ctypes.data)
    assert np.array_equal(beams, spec.beam_table())
    assert np.array_equal(foot, spec.footprint_table())
    # beam 0 at +135 deg, clockwise sweep to -135 deg (dreamer/tools.py:84-86)
    ang = np.degrees(np.arctan2(beams[:, 1], beams[:, 0]))
    assert abs(ang[0] - 135) < 1e-4 and abs(ang[-1] + 135) < 1e-4 and np.all(np.diff(ang) < 0)


def test_errors_are_codes_with_messages_not_exceptions(hip_lib):
    from racing_dreamer_amd import _lib
    cfg = _lib.RcConfig()
    hip_lib.rc_default_config(C.byref(cfg))
    h = C.c_void_p()
    cfg.cars_per_env = 9
    assert hip_lib.rc_create(C.byref(cfg), C.byref(h)) == -1 and b"cars_per_env" in hip_lib.rc_last_error()
    cfg.cars_per_env = 1
    cfg.struct_size = 8
    assert hip_lib.rc_create(C.byref(cfg), C.byref(h)) == -1 and b"ABI mismatch" in hip_lib.rc_last_error()
    assert hip_lib.rc_step(None, None, 1) == -1
    assert hip_lib.rc_sync(None) == -1
    assert h.value is None


def test_product_never_imports_the_oracle():
    """The product path must not route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "racing_dreamer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "libracecar_oracle" not in src and "c_oracle" not in src and "import_module" not in src, f


def test_missing_library_fails_loudly(tmp_path):
    from racing_dreamer_amd import _lib
    saved = _lib._lib
    _lib._lib = None
    try:
        try:
            _lib.load_library(str(tmp_path / "nope.so"))
            raised = False
        except _lib.RacecarHipError as e:
            raised = "no CPU fallback" in str(e)
        assert raised
    finally:
        _lib._lib = saved


def test_a_library_that_does_not_belong_to_the_sources_is_refused(monkeypatch):
    """Round 6: a stale libracecar_hip.so (built, then a header edited) would let the GPU suite pass on yesterday's kernels.  The
    loader compares the build id compiled into the library with the hash of the sources beside it and refuses a mismatch -
    unless the A/B scripts' RC_ALLOW_STALE_LIBRARY says a variant build stands in the library's place."""
    import pytest
    from racing_dreamer_amd import _lib, build
    saved = _lib._lib
    _lib._lib = None
    try:
        monkeypatch.delenv("RC_ALLOW_STALE_LIBRARY", raising=False)
        monkeypatch.setattr(build, "needs_build", lambda *a, **k: True)
        with pytest.raises(_lib.RacecarHipError, match="built from other sources"):
            _lib.load_library()
        monkeypatch.setenv("RC_ALLOW_STALE_LIBRARY", "1")
        assert _lib.load_library() is not None
    finally:
        _lib._lib = saved


def test_build_refuses_a_scan_kernel_that_spills():
    """The scan hands a register to an asynchronous load in inline assembly and waits for it in a later statement: a spill
    of that register would store it before the load has landed.  The build parses the compiler's resource
    remarks and refuses such a library."""
    import pytest
    from racing_dreamer_amd import build
    head = "k.hip:1:1: remark: Function Name: _ZN12_GLOBAL__N_121rc_raycast_car_kernelILi1ELb0ELb0EEEv8RcParamsi [-Rpass-analysis=kernel-resource-usage]\n"
    patch = ("k.hip:1:1: remark: Function Name: _ZN12_GLOBAL__N_119rc_patch_car_kernelILb1EEEv8RcParams [-Rpass-analysis=kernel-resource-usage]\n"
             "k.hip:1:1: remark:     VGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]\n"
             "k.hip:1:1: remark:     Occupancy [waves/SIMD]: 8 [-Rpass-analysis=kernel-resource-usage]\n")
    ok = head + ("k.hip:1:1: remark:     ScratchSize [bytes/lane]: 0 [-Rpass-analysis=kernel-resource-usage]\n"
                 "k.hip:1:1: remark:     Occupancy [waves/SIMD]: 8 [-Rpass-analysis=kernel-resource-usage]\n"
                 "k.hip:1:1: remark:     VGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]\n") + patch
    build.check_resource_usage(ok)
    with pytest.raises(RuntimeError, match="VGPRs Spill = 3"):
        build.check_resource_usage(ok.replace("VGPRs Spill: 0", "VGPRs Spill: 3", 1))
    with pytest.raises(RuntimeError, match="ScratchSize"):
        build.check_resource_usage(ok.replace("ScratchSize [bytes/lane]: 0", "ScratchSize [bytes/lane]: 16"))
    with pytest.raises(RuntimeError, match="7 waves/SIMD < 8"):
        build.check_resource_usage(ok.replace("Occupancy [waves/SIMD]: 8", "Occupancy [waves/SIMD]: 7", 1))
    with pytest.raises(RuntimeError, match="not found"):
        build.check_resource_usage(patch)
    other = ok + ("k.hip:9:1: remark: Function Name: _ZN12_GLOBAL__N_118rc_dynamics_kernelILi4EEEv8RcParamsPfijjj [-Rpass-analysis=kernel-resource-usage]\n"
                  "k.hip:9:1: remark:     VGPRs Spill: 12 [-Rpass-analysis=kernel-resource-usage]\n")
    build.check_resource_usage(other)           # kernels without such loads may spill
    # the exact render's prefilter keeps a line per lane in registers: one wave per SIMD is its design, a spill is refused
    exact = ok + ("k.h:9:1: remark: Function Name: _Z31rc_patch_exact_prefilter_kernel13RcExactParams [-Rpass-analysis=kernel-resource-usage]\n"
                  "k.h:9:1: remark:     Occupancy [waves/SIMD]: 1 [-Rpass-analysis=kernel-resource-usage]\n"
                  "k.h:9:1: remark:     VGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]\n")
    build.check_resource_usage(exact, required=("rc_raycast_car_kernel", "rc_patch_car_kernel", "rc_patch_exact_prefilter_kernel"))
    with pytest.raises(RuntimeError, match="rc_patch_exact_prefilter_kernel.*VGPRs Spill = 15"):
        build.check_resource_usage(exact.replace("k.h:9:1: remark:     VGPRs Spill: 0", "k.h:9:1: remark:     VGPRs Spill: 15"))


def test_build_refuses_a_copy_of_a_register_with_a_load_in_flight():
    """The other half of the same hazard: no spill, but an instruction that reads (or overwrites) the destination of the scan's
    inline-assembly table request before the `s_waitcnt vmcnt(0)` that belongs to it.  The build walks the kernels' assembly."""
    import pytest
    from racing_dreamer_amd import build
    ok = """
_ZN12_GLOBAL__N_121rc_raycast_car_kernelILi1ELb0ELb0EEEv8RcParamsi:
	v_mad_u32_u24 v49, v48, s36, v49
	;;#ASMSTART
	global_load_ushort v49, v49, s[16:17]
	;;#ASMEND
	v_cndmask_b32_e64 v55, v50, v54, s[2:3]
	v_add_f32_e32 v55, 0.5, v55
	s_and_saveexec_b64 s[28:29], vcc
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	v_mov_b32_e32 v51, v49
	s_endpgm
"""
    assert build.check_async_load_registers(ok) == 1
    with pytest.raises(RuntimeError, match="touches v49"):
        build.check_async_load_registers(ok.replace("v_add_f32_e32 v55, 0.5, v55", "v_mov_b32_e32 v51, v49"))
    with pytest.raises(RuntimeError, match="touches v49"):
        build.check_async_load_registers(ok.replace("v_cndmask_b32_e64 v55, v50, v54, s[2:3]", "v_fma_f32 v[48:49], v1, v2, v3"))
    with pytest.raises(RuntimeError, match="does not see"):
        build.check_async_load_registers(ok.replace("global_load_ushort", "global_load_dword"))
    # the same instructions in another kernel are nobody's business
    other = ok.replace("rc_raycast_car_kernel", "rc_dynamics_kernel").replace("v_add_f32_e32 v55, 0.5, v55", "v_mov_b32_e32 v51, v49")
    assert build.check_async_load_registers(ok + other) == 1


def test_a_changed_source_is_rebuilt_whatever_the_file_times_say(hip_lib, tmp_path):
    """build.py decides by CONTENT: the library carries the hash of the flags and of every source and header it was made
    from (`rc_build_id()`).  A copy of the tree whose library is NEWER than its sources is still rebuilt once a comment in a
    .hip file changes (file times said "up to date" - the round-3 rule), and is reused when nothing changed, however old."""
    import shutil
    import time
    from racing_dreamer_amd import build
    assert hip_lib.rc_build_id().decode() == build.source_hash() == build.library_build_id()
    assert not build.needs_build()
    pkg = tmp_path / "racing_dreamer_amd"
    shutil.copytree(os.path.join(ROOT, "racing_dreamer_amd", "csrc"), pkg / "csrc")
    shutil.copytree(os.path.join(ROOT, "include"), tmp_path / "include")
    lib = pkg / "lib" / "libracecar_hip.so"
    os.makedirs(lib.parent)
    shutil.copy(build.LIB_PATH, lib)
    csrc = str(pkg / "csrc")
    assert build.source_hash(csrc) == build.source_hash() and not build.needs_build(str(lib), csrc)
    src = pkg / "csrc" / "racecar_abi.hip"
    src.write_text(src.read_text() + "\n// a comment: nothing the compiler sees, but not the source this library was built from\n")
    old = time.time() - 86400
    os.utime(src, (old, old))                                   # the source now looks a day OLDER than the library
    assert os.path.getmtime(src) < os.path.getmtime(lib)
    assert build.needs_build(str(lib), csrc)
    build.build(verbose=False, csrc=csrc, lib_path=str(lib))
    assert build.LAST_BUILD["action"] == "compiled"
    assert build.library_build_id(str(lib)) == build.source_hash(csrc) != build.source_hash()
    build.build(verbose=False, csrc=csrc, lib_path=str(lib))
    assert build.LAST_BUILD["action"] == "reused"
    # a library without an id (or a truncated file) is never taken for current
    (pkg / "lib" / "junk.so").write_bytes(b"\x7fELF" + b"\0" * 64)
    assert build.library_build_id(str(pkg / "lib" / "junk.so")) is None and build.needs_build(str(pkg / "lib" / "junk.so"), csrc)


def test_the_shipped_library_carries_no_lab_kernels(hip_lib):
    """the superseded scan variants (rc_raycast_kernel<A, 0..6>), the instrumented "stamps" build and the dropped
    direction-table experiment are not compiled into libracecar_hip.so; they live in libracecar_lab.so (csrc/racecar_lab.hip),
    built on request with the same flags and the same refusals, found by the shipped library next to itself on first use."""
    from racing_dreamer_amd import build
    with open(build.LIB_PATH, "rb") as f:
        shipped = f.read()
    for name in (b"rc_raycast_kernelILi", b"28rc_raycast_car_stamps_kernelE", b"rc_build_dir_table_kernel", b"g_dir_table"):     # (mangled: kernel symbols)
        assert name not in shipped, name
    assert b"rc_raycast_car_kernelILi1ELb0ELb0E" in shipped                # the scan that is shipped
    with open(os.path.join(build.CSRC, "racecar_kernels.hip")) as f:
        text = f.read()
    assert "RC_EXP_DIR_TABLE" not in text and "cast_ray_dda" not in text
    lab = build.build_lab(verbose=False)
    assert not build.lab_needs_build() and build.library_build_id(lab) == build.source_hash(sources=build.LAB_SOURCES)
    import ctypes
    h = ctypes.CDLL(lab)
    for sym in ("rclab_launch_raycast", "rclab_set_lds_limits", "rclab_build_id"):
        assert hasattr(h, sym), sym
    with open(lab, "rb") as f:
        blob = f.read()
    assert b"rc_raycast_kernelILi1ELi0E" in blob and b"28rc_raycast_car_stamps_kernelE" in blob
    # the two libraries are identified separately: a change of the lab's source does not make the shipped library stale
    assert build.source_hash() != build.source_hash(sources=build.LAB_SOURCES)
