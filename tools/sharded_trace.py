"""bench.ShardedCollector under `rocprofv3 --kernel-trace` with ONE rank over real RCCL (python3 tools/sharded_trace.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29544")):
    os.environ.setdefault(k, v)
import torch
import torch.distributed as dist
import bench
from racing_dreamer_amd.batched_env import BatchedRaceEnv
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
prio = os.environ.get("ENV_STREAM_PRIORITY")
env = BatchedRaceEnv("austria", 65536, 1, auto_reset=True, stream=None if prio is None else torch.cuda.Stream(priority=int(prio)))
print("stream priority range", torch.cuda.Stream.priority_range(), "env stream priority", env.stream.priority, flush=True)
torch.cuda.set_stream(env.stream)
env.reset(mode="random", seed=0)
for k in range(150):
    env.step_random(seed=2, step=k)
col = bench.ShardedCollector(env, dist, 0, summary=os.environ.get("NO_SUMMARY") is None)
k0 = col.prefill(0)
for k in range(20):
    col.step(k0 + k)
col.wait(); env.sync(); torch.cuda.synchronize()
n = int(os.environ.get("TRACE_STEPS", "40"))
marker = torch.zeros(1, device=env.device)
marker.fill_(1.0)                     # (a fill kernel: marks the start of the timed window in the trace)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(n):
    col.step(k0 + 20 + k)
t1 = time.perf_counter()
col.wait(); env.sync(); torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"ms per step {(t2 - t0) / n * 1e3:.4f}  host enqueue {(t1 - t0) / n * 1e3:.4f}  drain {(t2 - t1) * 1e3:.3f} ms", flush=True)
col.close()
dist.barrier(); dist.destroy_process_group()
