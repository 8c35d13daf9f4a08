"""obs_type lidar_occupancy_reference: ms per step and the exact render's own time at some batch sizes (GPU box):
    python tools/time_exact_render.py [n_envs ...]"""
import os, sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from racing_dreamer_amd.batched_env import BatchedRaceEnv
from racing_dreamer_amd import _lib as L
sizes = [int(a) for a in sys.argv[1:]] or [2048, 16384, 65536]
for n in sizes:
    env = BatchedRaceEnv("austria", n, 1, obs_type="lidar_occupancy_reference", auto_reset=True)
    if os.environ.get("RC_EXACT_CHUNK"):
        env.debug_set("exact_chunk", int(os.environ["RC_EXACT_CHUNK"]))
    env.reset(mode="random", seed=0)
    torch.cuda.set_stream(env.stream)
    for k in range(5): env.step_random(seed=1, step=k)
    env.sync(); env.reset_kernel_times(); env.set_profiling(True, kernels=[L.K_PATCH])
    t0 = time.perf_counter()
    m = 10
    for k in range(m): env.step_random(seed=1, step=5 + k)
    env.sync(); dt = time.perf_counter() - t0
    env.set_profiling(False)
    print(n, "envs: step", dt / m * 1e3, "ms;", n * m / dt / 1e6, "M env-steps/s; exact render", env.kernel_times()["rc_patch_kernel"], flush=True)
    env.close()
