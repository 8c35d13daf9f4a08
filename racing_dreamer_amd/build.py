"""Build libracecar_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libracecar_hip.so")
SOURCES = ["racecar_kernels.hip", "racecar_abi.hip"]
HEADERS = ["racecar_device.h", "racecar_internal.h", "racecar_spec.h", "racecar_scan.h", "racecar_patch_exact.h", os.path.join("..", "..", "include", "racecar_hip.h")]
# The lab library: scan variants 0-6 and the instrumented build of the scan (racecar_lab.hip).  NOT part of the shipped
# library; built by build_lab() - which tools/ and the variant tests call - and loaded by libracecar_hip.so on first use.
LAB_PATH = os.path.join(LIB_DIR, "libracecar_lab.so")
LAB_SOURCES = ["racecar_lab.hip"]
# -ffp-contract=off: the env spec is "one IEEE fp32 operation per written operator" (DESIGN.md §2);
# a fused multiply-add would break bit-exact parity with the CPU oracle.
# -fno-slp-vectorize: packed fp32 (v_pk_*_f32) issues at half rate on gfx950, i.e. buys nothing over two scalar
# operations, and costs register-pair copies, explicit |x| / -x operands and wait states (tools/ubench/valu_issue4.hip);
# the places where a packed form does help are written as explicit vector types.
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared", "-std=c++17", "-Wall",
         "-Wno-unused-function"]


def find_hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked on PATH and in /opt/rocm/bin)")


BUILD_ID_MARK = b"RC_BUILD_ID="


def source_hash(csrc: str = CSRC, flags=None, sources=None) -> str:
    """sha256 over the flags and the CONTENT of every source and header the library is made of (in a fixed order, each
    preceded by its name): the identity of a build.  It is compiled into the library (`-DRC_BUILD_ID`, `rc_build_id()`),
    so whether a library on disk belongs to the sources beside it is a question about contents, not about file times - a
    tree pushed with a stale `.so` that happens to be newer than its sources is rebuilt."""
    import hashlib
    h = hashlib.sha256()
    h.update("\0".join(FLAGS if flags is None else flags).encode())
    for name in (SOURCES if sources is None else sources) + HEADERS:
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(b"\0" + os.path.basename(name).encode() + b"\0" + f.read())
    return h.hexdigest()[:32]


def headers_hash(csrc: str = CSRC) -> str:
    """sha256 over the CONTENT of the headers alone: what the shipped library and the lab library must have in common (RcParams
    and RcLaunchInfo cross between them by pointer).  Compiled into both as -DRC_HEADERS_ID; the loader compares."""
    import hashlib
    h = hashlib.sha256()
    for name in HEADERS:
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(b"\0" + os.path.basename(name).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def _tmp_beside(path: str) -> str:
    """A temporary name next to `path` that no other process uses: N ranks may build the same library at once (the lab on first
    use of a variant), and a shared `.new` let one replace or remove the file another was still writing."""
    import tempfile
    fd, tmp = tempfile.mkstemp(prefix=os.path.basename(path) + ".", suffix=".new", dir=os.path.dirname(path))
    os.close(fd)
    return tmp


def library_build_id(path: str = LIB_PATH):
    """The build id a library file carries (the string behind `rc_build_id()`), read from the file's bytes - no dlopen, so
    asking does not load a stale library into the process.  None if the file is missing or carries none."""
    try:
        with open(path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    at = blob.find(BUILD_ID_MARK)
    if at < 0:
        return None
    tail = blob[at + len(BUILD_ID_MARK):at + len(BUILD_ID_MARK) + 32]
    return tail.decode("ascii", "replace") if len(tail) == 32 and all(c in b"0123456789abcdef" for c in tail) else None


def needs_build(lib_path: str = LIB_PATH, csrc: str = CSRC) -> bool:
    return library_build_id(lib_path) != source_hash(csrc)


# Kernels that hand a register to an asynchronous load through inline assembly and wait for it in a LATER assembly
# statement (the scan's table request, racecar_kernels.hip: trip_head / trip_tail): between the two the compiler believes
# the value is there.  That is safe as long as it keeps the value where the load will write it - it does not copy a live
# register without need - but a spill of that register to scratch would store it BEFORE the load has landed.  So the build
# refuses a library in which one of these kernels spills or uses scratch at all, and one that has fallen below the
# occupancy the launch geometry assumes.
NO_SPILL_KERNELS = ("rc_raycast_car_kernel", "rc_raycast_car_stamps_kernel", "rc_raycast_kernel", "rc_patch_car_kernel",
                    "rc_patch_exact_prefilter_kernel", "rc_patch_exact_sample_kernel")
# (the exact prefilter holds a line per lane in 161 + registers: one wave per SIMD by design - what it must not do is spill)
MIN_WAVES_PER_SIMD = {"rc_raycast_car_kernel": 8, "rc_patch_car_kernel": 8, "rc_patch_exact_sample_kernel": 4}


def check_resource_usage(remarks: str, required=("rc_raycast_car_kernel", "rc_patch_car_kernel"), min_waves=None) -> None:
    """Parse `-Rpass-analysis=kernel-resource-usage` remarks; raise if a kernel of NO_SPILL_KERNELS spills."""
    min_waves = MIN_WAVES_PER_SIMD if min_waves is None else min_waves
    import re
    name, problems, seen = None, [], set()
    for line in remarks.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            continue
        m = re.search(r"remark:\s+(ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|Occupancy \[waves/SIMD\]): (\d+)", line)
        if not m or name is None:
            continue
        kernel = next((k for k in NO_SPILL_KERNELS if k + "I" in name or name.endswith(k) or k + "E" in name or re.search(k + r"\d", name)), None)
        if kernel is None:
            continue
        seen.add(kernel)
        key, val = m.group(1), int(m.group(2))
        if key.startswith("Occupancy"):
            if val < min_waves.get(kernel, 1):
                problems.append(f"{name}: {val} waves/SIMD < {min_waves[kernel]}")
        elif key != "SGPRs Spill" and val != 0:
            problems.append(f"{name}: {key} = {val}")
    missing = [k for k in required if k not in seen]
    if missing:
        raise RuntimeError(f"resource-usage remarks not found for {missing}: cannot verify that the scan does not spill")
    if problems:
        raise RuntimeError("build refused (see racing_dreamer_amd/build.py, NO_SPILL_KERNELS):\n  " + "\n  ".join(problems))


def check_async_load_registers(asm_text: str, kernels=("rc_raycast_car_kernel", "rc_raycast_car_stamps_kernel", "rc_raycast_kernel")) -> int:
    """The scan requests a table entry with an inline-assembly `global_load_ushort` and waits for it in a LATER statement
    (`s_waitcnt vmcnt(0)`): until then the destination register is not the compiler's to read, copy or overwrite, and
    nothing tells it so.  This walks the generated assembly of the scan kernels and raises if any instruction between such
    a load and the next wait for it (in layout order) touches the destination register.  Returns the number of loads
    checked."""
    import re
    checked, problems = 0, []
    kernel, pending, in_asm = None, None, False
    for raw in asm_text.splitlines():
        line = raw.strip()
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel = m.group(1) if any(k in m.group(1) for k in kernels) else None
            pending = None
            continue
        if kernel is None or not line or line.startswith((";", ".", "//")) and "#ASM" not in line:
            if ";;#ASMSTART" in line:
                in_asm = True
            elif ";;#ASMEND" in line:
                in_asm = False
            continue
        if ";;#ASMSTART" in line:
            in_asm = True
            continue
        if ";;#ASMEND" in line:
            in_asm = False
            continue
        if "s_endpgm" in line:
            kernel, pending = None, None
            continue
        code = line.split(";")[0].strip()
        if not code or code.endswith(":"):
            continue
        if pending is not None:
            if code.startswith("s_waitcnt") and "vmcnt(0)" in code:
                pending = None
            else:
                regs = set(int(r) for r in re.findall(r"\bv(\d+)\b", code))
                for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", code):
                    regs.update(range(int(a), int(b) + 1))
                if pending[0] in regs:
                    problems.append(f"{kernel}: `{code}` touches v{pending[0]} while `{pending[1]}` is in flight")
                    pending = None
        m = re.match(r"^global_load_ushort v(\d+),", code)
        if m and in_asm:
            pending = (int(m.group(1)), code)
            checked += 1
    if problems:
        raise RuntimeError("build refused: a register with an asynchronous load in flight is used before its wait\n  " + "\n  ".join(problems[:8]))
    if checked == 0:
        raise RuntimeError("no inline-assembly table request found in the scan kernels: the check does not see what it should")
    return checked


def verify_scan_assembly(verbose: bool = True, csrc: str = CSRC, source: str = None) -> int:
    """Compile the kernels to assembly once more (device only) and run check_async_load_registers on it."""
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "kernels.s")
        flags = [f for f in FLAGS if f not in ("-fPIC", "-shared")]
        cmd = [find_hipcc(), *flags, "-S", "--cuda-device-only", os.path.join(csrc, source or SOURCES[0]), "-o", out]
        r = subprocess.run(cmd, cwd=csrc, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-4000:])
            raise subprocess.CalledProcessError(r.returncode, cmd)
        with open(out) as f:
            n = check_async_load_registers(f.read())
    if verbose:
        print(f"[racing_dreamer_amd.build] scan assembly: {n} in-flight table requests, none touched before its wait", flush=True)
    return n


LAST_BUILD = {"action": None, "build_id": None}      # what the last build() call did: "compiled" | "reused"


def build(force: bool = False, verbose: bool = True, csrc: str = CSRC, lib_path: str = LIB_PATH) -> str:
    """Compile `csrc` into `lib_path` unless the library there already carries the hash of those sources and the flags
    (`LAST_BUILD["action"]` says which happened).  The defaults are the package's own tree; the tests build copies."""
    want = source_hash(csrc)
    LAST_BUILD.update(action="reused", build_id=want)
    if not force and library_build_id(lib_path) == want:
        if verbose:
            print(f"[racing_dreamer_amd.build] reused {lib_path}: its build id {want} is the hash of the sources and flags", flush=True)
        return lib_path
    os.makedirs(os.path.dirname(lib_path), exist_ok=True)
    tmp = _tmp_beside(lib_path)
    cmd = [find_hipcc(), *FLAGS, f'-DRC_BUILD_ID="{want}"', f'-DRC_HEADERS_ID="{headers_hash(csrc)}"', "-Rpass-analysis=kernel-resource-usage",
           *[os.path.join(csrc, s) for s in SOURCES], "-o", tmp]
    if verbose:
        print("[racing_dreamer_amd.build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, cwd=csrc, capture_output=True, text=True)
    other = [l for l in r.stderr.splitlines() if "kernel-resource-usage" not in l and not l.startswith(("      |", " ")) and "hip-link" not in l]
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-8000:])
        if os.path.exists(tmp):
            os.remove(tmp)
        raise subprocess.CalledProcessError(r.returncode, cmd)
    if other and verbose:
        print("\n".join(other), file=sys.stderr)
    try:
        check_resource_usage(r.stderr, required=("rc_raycast_car_kernel", "rc_patch_car_kernel", "rc_patch_exact_prefilter_kernel",
                                                 "rc_patch_exact_sample_kernel"))
        verify_scan_assembly(verbose, csrc)
    except RuntimeError:
        os.remove(tmp)
        raise
    os.chmod(tmp, 0o755)
    os.replace(tmp, lib_path)
    LAST_BUILD.update(action="compiled")
    if library_build_id(lib_path) != want:
        raise RuntimeError(f"{lib_path} does not carry the build id it was compiled with ({library_build_id(lib_path)} != {want})")
    if verbose:
        print(f"[racing_dreamer_amd.build] compiled {lib_path} (build id {want})", flush=True)
    return lib_path


def build_lab(force: bool = False, verbose: bool = True, csrc: str = CSRC, lab_path: str = LAB_PATH) -> str:
    """Compile the lab library (scan variants 0-6, the stamps build) unless the one at `lab_path` already carries the hash of
    its sources: same flags, same refusals (no spills in a scan kernel, no register touched under an asynchronous load)."""
    import time
    want = source_hash(csrc, sources=LAB_SOURCES)
    if not force and library_build_id(lab_path) == want:
        if verbose:
            print(f"[racing_dreamer_amd.build] reused {lab_path}: its build id {want} is the hash of the lab's sources and flags", flush=True)
        return lab_path
    os.makedirs(os.path.dirname(lab_path), exist_ok=True)
    tmp = _tmp_beside(lab_path)
    cmd = [find_hipcc(), *FLAGS, f'-DRC_BUILD_ID="{want}"', f'-DRC_HEADERS_ID="{headers_hash(csrc)}"', "-Rpass-analysis=kernel-resource-usage",
           *[os.path.join(csrc, s) for s in LAB_SOURCES], "-o", tmp]
    if verbose:
        print("[racing_dreamer_amd.build]", " ".join(cmd), flush=True)
    t0 = time.time()
    r = subprocess.run(cmd, cwd=csrc, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-8000:])
        if os.path.exists(tmp):
            os.remove(tmp)
        raise subprocess.CalledProcessError(r.returncode, cmd)
    try:
        check_resource_usage(r.stderr, required=("rc_raycast_car_stamps_kernel", "rc_raycast_kernel"), min_waves={})
        verify_scan_assembly(verbose, csrc, source=LAB_SOURCES[0])
    except RuntimeError:
        os.remove(tmp)
        raise
    os.chmod(tmp, 0o755)
    os.replace(tmp, lab_path)
    if verbose:
        print(f"[racing_dreamer_amd.build] compiled {lab_path} in {time.time() - t0:.1f} s (build id {want})", flush=True)
    return lab_path


def lab_needs_build(lab_path: str = LAB_PATH, csrc: str = CSRC) -> bool:
    return library_build_id(lab_path) != source_hash(csrc, sources=LAB_SOURCES)


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
    if "--lab" in sys.argv:
        print(build_lab(force="--force" in sys.argv))
