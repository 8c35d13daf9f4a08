"""MixedTrackEnv under `rocprofv3 --kernel-trace`: 30 steps of 65 536 envs in three track blocks (python3 tools/mixed_trace.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from racing_dreamer_amd.batched_env import MixedTrackEnv
env = MixedTrackEnv(["columbia", "austria", "barcelona"], [21846, 21845, 21845], auto_reset=True)
env.reset(mode="random", seed=0)
torch.cuda.set_stream(env.stream)
for k in range(150):
    env.step_random(seed=2, step=k)
env.sync(); torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for k in range(30):
    env.step_random(seed=1, step=k)
env.sync(); torch.cuda.synchronize()
print("ms per step", (time.perf_counter() - t0) / 30 * 1e3)
env.close()
