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
HEADERS = ["racecar_device.h", "racecar_internal.h", "racecar_spec.h", os.path.join("..", "..", "include", "racecar_hip.h")]
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


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [find_hipcc(), *FLAGS, *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB_PATH]
    if verbose:
        print("[racing_dreamer_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
