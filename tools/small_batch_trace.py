"""configs[1] under `rocprofv3 --kernel-trace` (tools/small_batch.sh): 4 096 envs on columbia, 150 settle steps, 200 steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from racing_dreamer_amd.batched_env import BatchedRaceEnv
env = BatchedRaceEnv("columbia", 4096, 1, auto_reset=True)
torch.cuda.set_stream(env.stream)
env.reset(mode="random", seed=0)
for k in range(350):
    env.step_random(seed=2, step=k)
env.sync()
