"""Throughput when the caller hands over HOST buffers (actions in, observations out over PCIe every step):
rc_step_host + one device-to-host copy of the arena.  The headline number in bench.py keeps everything in HBM."""
import sys, time
sys.path.insert(0, '.')
import ctypes as C
import numpy as np
import torch
from racing_dreamer_amd import _lib as L
from racing_dreamer_amd.batched_env import BatchedRaceEnv

for n in (4096, 65536):
    env = BatchedRaceEnv("austria", n, 1, auto_reset=True)
    env.reset(mode="random", seed=0)
    act = np.random.default_rng(0).uniform(-1, 1, (n, 1, 2)).astype(np.float32)
    host = torch.empty(env.arena_nbytes, dtype=torch.uint8).pin_memory()
    lidar = torch.empty(n * 1080, dtype=torch.float32).pin_memory()
    def full():
        L.check(env._lib.rc_step_host(env._h, act.ctypes.data, 1))
        host.copy_(env._arena_view, non_blocking=True); torch.cuda.synchronize()
    def lidar_only():
        L.check(env._lib.rc_step_host(env._h, act.ctypes.data, 1))
        lidar.copy_(env.views["lidar"].reshape(-1), non_blocking=True); torch.cuda.synchronize()
    for name, fn in (("whole record", full), ("lidar only", lidar_only)):
        for _ in range(5): fn()
        t0 = time.perf_counter()
        for _ in range(30): fn()
        dt = (time.perf_counter() - t0) / 30
        nbytes = env.arena_nbytes if name == "whole record" else n * 4320
        print(f"{n} envs, host actions in + {name} out ({nbytes/1e6:.1f} MB/step, pinned): {dt*1e3:.2f} ms/step = "
              f"{n/dt/1e6:.2f} M env-steps/s ({nbytes/dt/1e9:.1f} GB/s D2H)")
    env.close()
