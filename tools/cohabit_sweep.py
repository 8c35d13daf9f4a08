#!/usr/bin/env python3
"""What a co-tenant costs the scan (run on the GPU box): python tools/cohabit_sweep.py [--envs 65536] [--track austria]

The N > 1 run keeps a collective's workgroups on the chip beside the scan.  With ONE rank the "collective" is a 5 MB copy that sits
there for the scan's whole duration and the scan is 12 % slower for it (profiles/r04_d_sharded_timeline.txt); how much of that is
the copy's own doing, and what a collective of RCCL's shape - a few dozen workgroups that stay for the length of the transfer -
would cost, nobody had measured.  This tool launches a kernel of chosen shape from the lab library (`rclab_launch_cohabit`:
W workgroups x T threads, asleep or copying, resident for a set time) on a second stream, runs the production step beside it on the
env's stream and reads the scan's duration from the launch-attached events.  Analysis only."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from racing_dreamer_amd import build  # noqa: E402
from racing_dreamer_amd.batched_env import BatchedRaceEnv  # noqa: E402
from racing_dreamer_amd.track_assets import load_track  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--track", default="austria")
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--shapes", default="1x64,8x256,32x256,64x256,32x512,64x512,128x512,256x256,256x1024,1024x256")
    ap.add_argument("--mb", type=int, default=64, help="size of the buffer the copying co-tenant works on")
    a = ap.parse_args()
    if build.lab_needs_build():
        build.build_lab(verbose=False)
    lab = C.CDLL(build.LAB_PATH)
    lab.rclab_launch_cohabit.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    lab.rclab_launch_cohabit.restype = C.c_int
    env = BatchedRaceEnv(load_track(a.track), a.envs, 1, auto_reset=True)
    torch.cuda.set_stream(env.stream)
    env.reset(mode="random", seed=0)
    for k in range(150):
        env.step_random(0, k)
    side = torch.cuda.Stream(device=env.device)
    buf = torch.zeros(a.mb << 20, dtype=torch.uint8, device=env.device)
    copied = torch.zeros(1, dtype=torch.int64, device=env.device)
    torch.cuda.synchronize()

    def timed(shape, mode):
        """Mean scan duration over `steps` steps with the co-tenant resident throughout; the co-tenant's own traffic [GB/s]."""
        env.reset_kernel_times()
        env.set_profiling(True)
        us = min(100000, int(a.steps * 300))      # one launch that outlasts the window (0.3 ms per step is generous), at most 100 ms
        if shape is not None:
            w, t = shape
            copied.zero_()
            torch.cuda.synchronize()
            rc = lab.rclab_launch_cohabit(C.c_void_p(side.cuda_stream), w, t, us, C.c_void_p(buf.data_ptr()), buf.numel(), mode,
                                          C.c_void_p(copied.data_ptr()))
            if rc != 0:
                raise RuntimeError(f"rclab_launch_cohabit: {rc}")
        for k in range(a.steps):
            env.step_random(0, 1000 + k)
        env.stream.synchronize()
        still = shape is not None and not side.query()          # the co-tenant outlived the window: it was there for all of it
        torch.cuda.synchronize()
        t = env.kernel_times()
        env.set_profiling(False)
        gbs = float(copied.item()) * 32.0 / (us * 1e-6) / 1e9 if shape is not None else 0.0       # 16 B read + 16 B written per vector
        return t["rc_raycast_kernel"]["avg_ms"] * 1e3, gbs, still

    shapes = [tuple(int(v) for v in s.split("x")) for s in a.shapes.split(",")]
    print(f"{a.envs} envs on {a.track}, {a.steps} steps per point; scan / dynamics on the launch-attached events, step = wall time of the window / steps [us]")
    modes = ((0, "asleep"), (1, "copying"), (2, "copying, non-temporal"))
    print("scan duration [us]; every cell: beside the co-tenant (per cent against the mean of the scan alone measured right before and right "
          "after; the chip's clocks wander by +- 5 % with what ran last), the co-tenant's own traffic while the scan runs")
    print(f"{'co-tenant':>14s} " + " ".join(f"{n:>34s}" for _, n in modes) + "   waves (share of the wave slots)")
    before = timed(None, 0)[0]
    for shape in shapes:
        cells = []
        for mode, _ in modes:
            s_co, gbs, still = timed(shape, mode)
            after = timed(None, 0)[0]
            ref = 0.5 * (before + after)
            cells.append(f"{s_co:7.2f} ({(s_co / ref - 1) * 100:+6.1f} %) {gbs:7.0f} GB/s{'' if still else ' !'}")
            before = after
        waves = shape[0] * shape[1] // 64
        print(f"{shape[0]:7d} x {shape[1]:4d} " + " ".join(f"{c:>34s}" for c in cells) + f"   {waves} ({waves / 8192 * 100:.1f} %)", flush=True)
    print(f"(! = the co-tenant had left before the window ended; the scan alone at the end: {before:.2f} us)")

if __name__ == "__main__":
    main()
