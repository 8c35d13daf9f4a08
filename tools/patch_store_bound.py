"""How fast can rc_patch_car_kernel's stores alone go?  After reset() every car's observation is the first of an episode: the
render writes all-zero patches (the same 4 KB per car, the same store instructions, no taps).  python tools/patch_store_bound.py"""
import sys
sys.path.insert(0, ".")
import torch
from racing_dreamer_amd.batched_env import BatchedRaceEnv
track = sys.argv[1] if len(sys.argv) > 1 else "austria"
env = BatchedRaceEnv(track, 65536, 1, obs_type="lidar_occupancy", auto_reset=True)
print(track)
torch.cuda.set_stream(env.stream)
env.reset(mode="random", seed=0)
env.sync()
env.reset_kernel_times(); env.set_profiling(True)
for k in range(20):
    env.reset(mode="random", seed=k)
env.sync(); env.set_profiling(False)
print("all-zero patches (stores only):", {k: round(v["avg_ms"], 4) for k, v in env.kernel_times().items() if v["launches"]})
for k in range(150):
    env.step_random(seed=2, step=k)
env.sync(); env.reset_kernel_times(); env.set_profiling(True)
for k in range(50):
    env.step_random(seed=1, step=k)
env.sync(); env.set_profiling(False)
print("rendered patches:", {k: round(v["avg_ms"], 4) for k, v in env.kernel_times().items() if v["launches"]})
for variant in (4, 0):
    env.debug_set("patch_variant", variant)
    for k in range(20):
        env.step_random(seed=1, step=100 + k)
    env.sync(); env.reset_kernel_times(); env.set_profiling(True)
    for k in range(50):
        env.step_random(seed=1, step=200 + k)
    env.sync(); env.set_profiling(False)
    print("patch_variant", variant, "(4 = the unpadded bitmap):", {k: round(v["avg_ms"], 4) for k, v in env.kernel_times().items() if v["launches"]})
