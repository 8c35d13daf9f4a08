#!/usr/bin/env python3
"""Benchmark of the hot path: env-steps/s of the batched racing env on N MI355X (one process per GPU).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 200 --warmup 20

A "step" is one pass of the hot path over one batch: random actions (Philox, on device) ->
integrator + collision + progress/reward/done + auto-reset -> 1080-beam LiDAR scan, for 65 536 envs
per GPU (weak scaling), action_repeat 1, so one step = one simulator sub-step (dt = 0.01 s) of every
env.  For N > 1 every step's record also goes into the overlapped RCCL all-gather of the trajectory slab.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
RAYCAST_BYTES_PER_CAR = 4 * 1080 + 16       # lidar row written + (x, y, cos, sin) read, DESIGN.md §5
STEP_BYTES_PER_CAR = 4 * 1080 + 159         # whole env-step, obs_type=lidar
PATCH_BYTES_PER_CAR = 4096


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--cars", type=int, default=1)
    ap.add_argument("--track", default="austria")
    ap.add_argument("--mixed-tracks", action="store_true",
                    help="BASELINE.json configs[4]: rank r runs track [columbia, austria, barcelona][r mod 3]")
    ap.add_argument("--obs-type", default="lidar", choices=["lidar", "lidar_occupancy"])
    ap.add_argument("--repeat", type=int, default=1, help="action repeat (sub-steps per step)")
    ap.add_argument("--gather", default="summary", choices=["summary", "full", "none"],
                    help="N>1: what the per-step RCCL all-gather carries. summary = the transition record without "
                         "the LiDAR row (pose, velocity, speed, action, reward, discount, progress, time: 76 B/car; "
                         "the scans stay sharded in each rank's HBM); full = the whole 4 396 B/car record (xGMI-bound: "
                         "DESIGN.md §6); none = no collective")
    ap.add_argument("--gather-every", type=int, default=4,
                    help="N>1: steps per all-gather (each collective carries that many per-step records; same bytes, fewer launches)")
    ap.add_argument("--no-gather", action="store_true", help="same as --gather none")
    ap.add_argument("--force-gather", action="store_true",
                    help="run the all-gather path even with one rank (needs a torch.distributed.run launch)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl = RCCL; gloo only for functional tests on one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ftg", dest="no_cpu_baseline_ftg", action="store_true", help="skip the follow-the-gap secondary figure")
    ap.add_argument("--cpu-envs", type=int, default=0, help="envs in the CPU baseline sample (0 = auto)")
    ap.add_argument("--raycast-variant", type=int, default=None)
    ap.add_argument("--debug-knob", action="append", default=[], metavar="NAME=VALUE",
                    help="experiment knob passed to rc_debug_set (ray_threads, ray_split, ray_wg_per_cu, band_log2)")
    return ap.parse_args()


def cpu_baseline(track, cars, obs_type, repeat, n_envs):
    """The CPU oracle timed on this host's cores on a bounded sample of the same workload."""
    from oracle import cpu_baseline as cb
    return cb.run(track, cars=cars, occupancy=(obs_type == "lidar_occupancy"), repeat=repeat, n_envs=n_envs)


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
            sys.exit(2)
    distributed = world > 1 or (args.force_gather and "RANK" in os.environ)
    dev = local_rank % max(torch.cuda.device_count(), 1)      # ranks > GPUs only in --backend gloo functional tests
    torch.cuda.set_device(dev)
    if distributed:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo")

    from racing_dreamer_amd.batched_env import BatchedRaceEnv
    from racing_dreamer_amd.distributed import TrajectoryGather, shard_envs
    from racing_dreamer_amd.track_assets import load_track

    track_name = ["columbia", "austria", "barcelona"][rank % 3] if args.mixed_tracks else args.track
    track = load_track(track_name)
    shard = shard_envs(args.envs * world, rank, world)
    env = BatchedRaceEnv(track, shard.num_envs, args.cars, obs_type=args.obs_type, action_repeat=args.repeat,
                         device=dev, first_env=shard.first_env, auto_reset=True, profiling=False)
    if args.raycast_variant is not None:
        from racing_dreamer_amd import _lib as L
        L.check(env._lib.rc_set_raycast_variant(env._h, args.raycast_variant))
    for kv in args.debug_knob:
        name, _, val = kv.partition("=")
        env.debug_set(name, int(val))
    env.reset(mode="random", seed=0)
    gather_mode = "none" if (args.no_gather or not distributed) else args.gather
    gather_src = {"none": None, "full": env.slab, "summary": env.summary_slab}[gather_mode]
    gather = TrajectoryGather(gather_src, every=max(1, args.gather_every)) if gather_src is not None else None

    # the rollout loop works on the env's own stream (no cross-stream event waits between the step's kernels and
    # the staging copy of the gather); `barrier()` synchronises the whole device
    torch.cuda.set_stream(env.stream)

    def one_step(k, repeat=None):
        env.step_random(seed=1, step=k, repeat=repeat)        # actions drawn inside the dynamics kernel (Philox, on device)
        if gather is not None:
            gather.launch(gather_src)

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        one_step(k)
    if gather is not None:
        gather.wait()
    env.sync()
    env.reset_kernel_times()
    from racing_dreamer_amd import _lib as L
    # start / stop timestamps attached to every launch of the DOMINANT kernel (the scan) on the stream it runs on;
    # timing the two small kernels as well would cost the timed region 6 us per step, so they get a pass of their
    # own after it
    env.set_profiling(True, kernels=[L.K_RAYCAST])
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step(args.warmup + k)
    if gather is not None:
        gather.wait()
    env.sync()
    barrier()
    dt = time.perf_counter() - t0
    env.set_profiling(False)
    ktimes = env.kernel_times()
    # the other kernels of the step: a short untimed pass with all timers on
    env.reset_kernel_times()
    env.set_profiling(True, kernels=[L.K_PATCH, L.K_DYNAMICS])
    for k in range(min(args.steps, 50)):
        one_step(args.warmup + args.steps + k)
    if gather is not None:
        gather.wait()
    env.sync()
    env.set_profiling(False)
    for name, v in env.kernel_times().items():
        if name != "rc_raycast_kernel":
            ktimes[name] = v

    # secondary figure: the reference's own setting, action_repeat 4 with the scan once per agent step
    # (dreamer/dream.py:55) - a quarter of the steps, same barriers
    r4_steps = max(args.steps // 4, 5)
    barrier()
    t1 = time.perf_counter()
    for k in range(r4_steps):
        one_step(args.warmup + 2 * args.steps + k, repeat=4)
    if gather is not None:
        gather.wait()
    env.sync()
    barrier()
    dt4 = time.perf_counter() - t1

    # secondary figure: the long-ray case of SURVEY.md 8d - cars driven along the track by the follow-the-gap agent
    # (the reference's other prefill policy, dreamer/dream.py:211-216) instead of crashing into walls with random
    # actions: rank 0 only, single-GPU runs only (it is a property of the scan, not of the scaling)
    ftg = None
    if world == 1 and not args.no_cpu_baseline_ftg:
        mean_range_random = float(env.views["lidar"].float().mean().item())
        env.reset(mode="random", seed=0)
        for k in range(150):                      # let the cars settle on the racing line
            env.follow_the_gap()
            env.step(None)
        env.sync()
        env.reset_kernel_times()
        env.set_profiling(True, kernels=[L.K_RAYCAST])
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        n_ftg = max(args.steps // 4, 5)
        for k in range(n_ftg):
            env.follow_the_gap()
            env.step(None)
        env.sync()
        torch.cuda.synchronize()
        dtf = time.perf_counter() - t2
        env.set_profiling(False)
        kt = env.kernel_times()
        ftg = {"env_steps_per_s": args.envs * n_ftg * args.repeat / dtf, "steps": n_ftg,
               "raycast_ms": round(kt["rc_raycast_kernel"]["avg_ms"], 4),
               "mean_range_m": float(env.views["lidar"].float().mean().item()),
               "mean_range_m_random_actions": mean_range_random,
               "note": "same envs driven by the device follow-the-gap agent after 150 settling steps (cars on the racing "
                       "line, long rays) instead of random actions; includes the agent's kernel"}

    if distributed:
        tmax = torch.tensor([dt, dt4], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt, dt4 = float(tmax[0].item()), float(tmax[1].item())

    total_envs = args.envs * world
    env_steps = total_envs * args.steps * args.repeat
    value = env_steps / dt
    if rank == 0:
        n_cars = shard.num_envs * args.cars
        ray = ktimes["rc_raycast_kernel"]
        # the symbol rocprofv3 lists: the default scan (variant 7) is rc_raycast_car_kernel<A>, variants 0-6 rc_raycast_kernel<A, V>
        variant = 7 if args.raycast_variant is None else args.raycast_variant
        ray_symbol = f"rc_raycast_car_kernel<{args.cars}>" if variant == 7 else f"rc_raycast_kernel<{args.cars}, {variant}>"
        ray_s = ray["avg_ms"] * 1e-3
        achieved = RAYCAST_BYTES_PER_CAR * n_cars / ray_s / 1e9 if ray_s > 0 else 0.0
        traffic = valu = None
        tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tp):
            with open(tp) as f:
                prof = json.load(f)
            traffic = prof.get(f"{track_name}:{shard.num_envs}x{args.cars}:{args.obs_type}")
            valu = prof.get("_valu_wave_insts", {}).get(f"{track_name}:{shard.num_envs}x{args.cars}:{args.obs_type}")
        out = {
            "metric": "env-steps/sec at 65 536 parallel envs, 1080-beam LiDAR, 1/2/4/8 MI355X",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{args.envs} envs/GPU x {args.cars} car, track "
                            f"{'mixed columbia/austria/barcelona by rank' if args.mixed_tracks else args.track}, obs_type={args.obs_type}, "
                            f"1080-beam lidar every sub-step, random-action rollouts (Philox on device), "
                            f"auto-reset, action_repeat {args.repeat}",
                "envs_per_gpu": args.envs, "total_envs": total_envs, "cars_per_env": args.cars,
                "track": "mixed: [columbia, austria, barcelona][rank mod 3]" if args.mixed_tracks else args.track,
                "obs_type": args.obs_type, "action_repeat": args.repeat,
                "parallelism": f"env-sharded x{world}" + ("" if gather is None else
                                f" + overlapped RCCL all-gather of the {gather_mode} trajectory record of every step, "
                                f"one collective per {max(1, args.gather_every)} steps ({gather_src.numel()} B per GPU per step)"),
                "gather": gather_mode,
            },
            "roofline": {
                "bound": "hbm", "kernel": ray_symbol, "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": RAYCAST_BYTES_PER_CAR * n_cars,
                "avg_launch_ms": ray["avg_ms"], "launches": ray["launches"],
                "rays_per_s": n_cars * 1080 / ray_s if ray_s > 0 else 0.0,
                # what actually bounds the scan: wave-level VALU instructions per launch from the PMC
                # profile, and the rate they retire at per SIMD (1 024 SIMDs) at the duration measured here
                "valu_wave_insts_per_launch": valu,
                "valu_insts_per_simd_per_us": (valu / 1024 / (ray_s * 1e6)) if (valu and ray_s > 0) else None,
                "note": "compulsory HBM traffic is ~4.3 KB per car-scan, so the scan is bound by VALU work "
                        "of the grid traversal, not by HBM (SURVEY.md §8d); see DESIGN.md §5",
            },
            "kernels_ms": {k: round(v["avg_ms"], 4) for k, v in ktimes.items() if v["launches"]},
            "action_repeat_4": {"env_steps_per_s": total_envs * r4_steps * 4 / dt4,
                                "agent_steps_per_s": total_envs * r4_steps / dt4, "steps": r4_steps,
                                "note": "same workload with action_repeat 4, LiDAR once per agent step (dreamer/dream.py:55)"},
        }
        if ftg is not None:
            out["follow_the_gap"] = ftg
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(track, args.cars, args.obs_type, args.repeat, args.cpu_envs)
        print(json.dumps(out), flush=True)
    env.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
