"""Does the scan run slower for a while after the GPU has idled (or has only moved memory)?  python3 tools/idle_ramp_trace.py under
`rocprofv3 --kernel-trace`: 150 settle steps, 0.5 s of host sleep, 150 steps, an 18 GB memset, 150 steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from racing_dreamer_amd.batched_env import BatchedRaceEnv
env = BatchedRaceEnv("austria", 65536, 1, auto_reset=True)
torch.cuda.set_stream(env.stream)
env.reset(mode="random", seed=0)
for k in range(300):
    env.step_random(seed=2, step=k)
env.sync()
time.sleep(0.5)
for k in range(150):
    env.step_random(seed=1, step=k)
env.sync()
big = torch.zeros(18 * 1024 ** 3, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for k in range(150):
    env.step_random(seed=1, step=200 + k)
env.sync()
