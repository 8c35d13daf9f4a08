"""TrajectoryRing on the GPU box: step time and per-kernel times when every step writes its record into the next ring slot
(rc_set_arena) against stepping in place, and what a sample(50, 50) costs.   python tools/ring_sample_bench.py [envs] [slots]"""
import sys, time
sys.path.insert(0, ".")
import torch
from racing_dreamer_amd.batched_env import BatchedRaceEnv
from racing_dreamer_amd.replay import TrajectoryRing

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 64
env = BatchedRaceEnv("austria", n, 1, auto_reset=True)
torch.cuda.set_stream(env.stream)
env.reset(mode="random", seed=0)


def timed(name, fn, reps=100):
    for k in range(10):
        fn(k)
    env.sync(); env.reset_kernel_times(); env.set_profiling(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(reps):
        fn(10 + k)
    env.sync(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps * 1e3
    env.set_profiling(False)
    kt = {k: round(v["avg_ms"], 4) for k, v in env.kernel_times().items() if v["launches"]}
    print(f"{name}: {dt:.4f} ms per call  {kt}")


timed("step in place          ", lambda k: env.step_random(seed=1, step=k))
ring = TrajectoryRing(env, slots)
ring.reset(mode="random", seed=0)
for k in range(slots):                      # every slot once: their views are built on first use
    ring.step_random(seed=1, step=k)
timed(f"step into a {slots}-slot ring", lambda k: ring.step_random(seed=1, step=k))
gen = torch.Generator(device=env.device); gen.manual_seed(1)
fields = ("lidar", "action", "reward", "discount")
if slots >= 52:
    timed("sample(50, 50)         ", lambda k: ring.sample(50, 50, fields=fields, generator=gen), reps=20)
if slots >= 52:
    lay = env.sample_batch_layout(fields, 50, 50)
    out = torch.empty(lay["total"] + 64, dtype=torch.uint8, device=env.device)
    out = out[(-out.data_ptr()) % 64:][:lay["total"]]
    timed("sample_packed(50, 50)  ", lambda k: ring.sample_packed(50, 50, fields=fields, generator=gen, out=out, layout=lay), reps=50)
    import time as _t
    torch.cuda.synchronize(); t0 = _t.perf_counter()
    for k in range(200):
        ring.sample_packed(50, 50, fields=fields, generator=gen, out=out, layout=lay)
    host = (_t.perf_counter() - t0) / 200 * 1e3
    torch.cuda.synchronize()
    print(f"sample_packed: {host:.4f} ms of host time per call (enqueue only)")
    t0 = _t.perf_counter()
    for k in range(50):
        ring.sample(50, 50, fields=fields, generator=gen, check=False)
    host = (_t.perf_counter() - t0) / 50 * 1e3
    torch.cuda.synchronize()
    print(f"sample(check=False): {host:.4f} ms of host time per call (enqueue only)")
