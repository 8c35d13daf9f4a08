"""What runs in the first 20 steps after a reset?  python3 tools/fresh_window_trace.py [timers] under `rocprofv3 --kernel-trace`
(tools/fresh_window.sh): 150 settle steps, reset, synchronise, 20 steps between two stream events - with `timers`, the scan's
launch-attached timer on, as bench.py's fresh_reset leg has it; then 40 steady steps for comparison."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from racing_dreamer_amd import _lib as L
from racing_dreamer_amd.batched_env import BatchedRaceEnv
timers = len(sys.argv) > 1 and sys.argv[1] == "timers"
env = BatchedRaceEnv("austria", 65536, 1, auto_reset=True)
torch.cuda.set_stream(env.stream)
env.reset(mode="random", seed=0)
for k in range(150):
    env.step_random(seed=3, step=1000 + k)
env.reset(mode="random", seed=0)
if timers:
    env.reset_kernel_times()
    env.set_profiling(True, kernels=[L.K_RAYCAST])
env.sync(); torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
ev0.record(env.stream)
for k in range(20):
    env.step_random(seed=3, step=k)
t1 = time.perf_counter()
ev1.record(env.stream)
env.sync(); torch.cuda.synchronize()
t2 = time.perf_counter()
if timers:
    env.set_profiling(False)
print(f"fresh window ({'scan timer on' if timers else 'no timers'}): host enqueue {(t1 - t0) * 1e3:.3f} ms, host total {(t2 - t0) * 1e3:.3f} ms, "
      f"between the stream events {ev0.elapsed_time(ev1):.3f} ms = {ev0.elapsed_time(ev1) / 20:.4f} per step", flush=True)
for k in range(40):
    env.step_random(seed=3, step=20 + k)
env.sync()
