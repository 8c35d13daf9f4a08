"""Rollout throughput with the device-resident TrajectoryRing attached, and window-sampling throughput."""
import sys, time
sys.path.insert(0, '.')
import torch
from racing_dreamer_amd.batched_env import BatchedRaceEnv
from racing_dreamer_amd.replay import TrajectoryRing

n, cap, steps = 65536, 128, 300
env = BatchedRaceEnv("austria", n, 1, auto_reset=True)
torch.cuda.set_stream(env.stream)
env.reset(mode="random", seed=0)
for k in range(20):
    env.fill_random_actions(seed=1, step=k); env.step(None)
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(steps):
    env.fill_random_actions(seed=1, step=k); env.step(None)
torch.cuda.synchronize(); plain = n * steps / (time.perf_counter() - t0)
ring = TrajectoryRing(env, cap)
ring.reset(mode="random", seed=0)
for k in range(cap):
    env.fill_random_actions(seed=1, step=k); ring.step(None)
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(steps):
    env.fill_random_actions(seed=1, step=k); ring.step(None)
torch.cuda.synchronize(); ringed = n * steps / (time.perf_counter() - t0)
g = torch.Generator(device="cuda").manual_seed(0)
ring.sample(50, 50, generator=g); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50):
    b = ring.sample(50, 50, generator=g)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print(f"plain {plain/1e6:.1f} M env-steps/s, with ring ({cap} slots, {ring.buffer.numel()/2**30:.1f} GiB) {ringed/1e6:.1f} M env-steps/s; "
      f"sample(50 x 50 windows) {dt*1e3:.2f} ms = {50*50/dt/1e6:.2f} M transitions/s")
