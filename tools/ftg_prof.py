"""The reference follow-the-gap agent's kernel under the profiler (GPU box): cars settled on the racing line by the law itself,
then N agent steps.  `python tools/ftg_prof.py [n_steps] [settle]` prints the kernel's mean time from launch-attached events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from racing_dreamer_amd import _lib as L
from racing_dreamer_amd.batched_env import BatchedRaceEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
settle = int(sys.argv[2]) if len(sys.argv) > 2 else 150
env = BatchedRaceEnv("austria", 65536, 1, auto_reset=True)
env.reset(mode="random", seed=0)
torch.cuda.set_stream(env.stream)
for k in range(settle):
    env.follow_the_gap_reference()
    env.step(None)
env.sync()
env.reset_kernel_times()
env.set_profiling(True, kernels=[L.K_RAYCAST, L.K_FTG])
for k in range(n):
    env.follow_the_gap_reference()
    env.step(None)
env.sync()
env.set_profiling(False)
print({k: round(v["avg_ms"], 4) for k, v in env.kernel_times().items() if v["launches"]})
env.close()
