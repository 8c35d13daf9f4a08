"""What the pieces of bench.py's N > 1 headline (`--gather sharded`) cost on ONE rank over real RCCL (run on the GPU box under
`python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/sharded_loop_bench.py`):
the step in place, into the ring, + the per-step summary all-gather, + a ShardedReplay batch every 10th step, both."""
import os, sys, time
sys.path.insert(0, ".")
import torch
import torch.distributed as dist
from racing_dreamer_amd.batched_env import BatchedRaceEnv
from racing_dreamer_amd.distributed import TrajectoryGather
from racing_dreamer_amd.replay import ShardedReplay, TrajectoryRing

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
env = BatchedRaceEnv("austria", n, 1, auto_reset=True)
torch.cuda.set_stream(env.stream)
env.reset(mode="random", seed=0)
for k in range(150):
    env.step_random(seed=2, step=k)
ring = TrajectoryRing(env, 64)
rep = ShardedReplay(ring)
gen = torch.Generator(device=env.device); gen.manual_seed(1)
fields = ("lidar", "action", "reward", "discount")
off = env.summary_slab.data_ptr() - env._arena_view.data_ptr()
nb = env.summary_slab.numel()
tg = TrajectoryGather(env.summary_slab, stage=False, depth=4)
for k in range(70):
    ring.step_random(seed=1, step=k)
rep.sample(50, 50, fields=fields, generator=gen)
ring.detach()
state = {"n": 0}


def loop(name, summary, batch, in_ring=True, local=False):
    def one(k):
        if in_ring:
            ring.step_random(seed=1, step=k)
        else:
            env.step_random(seed=1, step=k)
        if summary:
            tg.launch(ring.slot(ring.head)[off:off + nb])
        state["n"] += 1
        if batch and state["n"] % 10 == 0:
            if local:
                ring.sample(50, 50, fields=fields, generator=gen, check=False)
            else:
                rep.sample(50, 50, fields=fields, generator=gen, check=False)
    for k in range(20):
        one(k)
    tg.wait(); env.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        one(20 + k)
    t_host = time.perf_counter() - t0
    tg.wait(); env.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if in_ring:
        ring.detach()
    print(f"{name:46s} {dt / steps * 1e3:.4f} ms per step   (host enqueue {t_host / steps * 1e3:.4f})", flush=True)


loop("step in place", False, False, in_ring=False)
loop("step into the ring", False, False)
loop("ring + summary all-gather per step", True, False)
loop("ring + local sample(50, 50) every 10th", False, True, local=True)
loop("ring + ShardedReplay batch every 10th", False, True)
loop("ring + summary + batch (the headline)", True, True)
side = torch.cuda.Stream(device=env.device)
evb = torch.cuda.Event()


def batch_variant(name, exchange, on_side, record=True):
    state["n"] = 0

    def one(k):
        ring.step_random(seed=1, step=k)
        state["n"] += 1
        if state["n"] % 10 == 0:
            local = rep.draw(50, 50, fields=fields, generator=gen, check=False)
            if not exchange:
                return
            if on_side:
                if record:
                    for t in local.values():
                        t.record_stream(side)
                evb.record(env.stream)
                with torch.cuda.stream(side):
                    side.wait_event(evb)
                    state["b"] = rep.exchange(local)
            else:
                state["b"] = rep.exchange(local)
    for k in range(20):
        one(k)
    env.stream.wait_stream(side); env.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        one(20 + k)
    t_host = time.perf_counter() - t0
    env.stream.wait_stream(side); env.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ring.detach()
    print(f"{name:46s} {dt / steps * 1e3:.4f} ms per step   (host enqueue {t_host / steps * 1e3:.4f})", flush=True)


batch_variant("draw only", False, False)
batch_variant("draw + packed exchange on the env stream", True, False)
batch_variant("draw + packed exchange on a side stream", True, True)
batch_variant("the same without record_stream", True, True, record=False)
import bench       # the collector bench.py runs: side stream, staged summaries, one packed collective per batch
for summary in (True, False):
    col = bench.ShardedCollector(env, dist, 0, summary=summary)
    k0 = 1000 + col.prefill(1000)
    for k in range(20):
        col.step(k0 + k)
    col.wait(); env.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        col.step(k0 + 20 + k)
    t_host = time.perf_counter() - t0
    col.wait(); env.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{'bench.ShardedCollector' + ('' if summary else ' without the summary'):46s} {dt / steps * 1e3:.4f} ms per step   (host enqueue {t_host / steps * 1e3:.4f})", flush=True)
    col.close()
dist.barrier()
dist.destroy_process_group()
