"""Step time of the default bench workload with and without the per-kernel HIP-event timers (run on the GPU box)."""
import sys, time
sys.path.insert(0, '.')
import torch
from racing_dreamer_amd.batched_env import BatchedRaceEnv
from racing_dreamer_amd import _lib as L

env = BatchedRaceEnv("austria", 65536, 1, auto_reset=True, profiling=False)
env.reset(mode="random", seed=0)
torch.cuda.set_stream(env.stream)
for mode in ("off", "raycast", "raycast+dynamics", "off"):
    if mode == "off":
        env.set_profiling(False)
    else:
        env.set_profiling(True, kernels=[L.K_RAYCAST] + ([L.K_DYNAMICS] if "dyn" in mode else []))
    for k in range(20):
        env.step_random(1, k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(300):
        env.step_random(1, 100 + k)
    env.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"profiling {mode:18s}: {dt / 300 * 1e3:.4f} ms/step", flush=True)
    env.set_profiling(False)
env.close()
