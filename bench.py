#!/usr/bin/env python3
"""Benchmark of the hot path: env-steps/s of the batched racing env on N MI355X (one process per GPU).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python bench.py --gpus 8 --steps 200 --warmup 20          (starts its own 8 ranks, see self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 200 --warmup 20

A "step" is one pass of the hot path over one batch: random actions (Philox, on device) ->
integrator + collision + progress/reward/done + auto-reset -> 1080-beam LiDAR scan, for 65 536 envs
per GPU (weak scaling), action_repeat 1, so one step = one simulator sub-step (dt = 0.01 s) of every
env.  For N > 1 every step's trajectory record also goes into the overlapped RCCL all-gather; the
payload is named in config.workload (`full-u16` by default: the whole record with the LiDAR row as
uint16 written by the scan - never the LiDAR-less `summary` unless asked for).
Prints ONE JSON line on rank 0; at N = 1 the line also carries the other single-GPU configurations
of BASELINE.json (`configs`) and the CPU baseline.
"""
from __future__ import annotations

import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
RAYCAST_BYTES_PER_CAR = 4 * 1080 + 16       # lidar row written + (x, y, cos, sin) read, DESIGN.md §5
STEP_BYTES_PER_CAR = 4 * 1080 + 159         # whole env-step, obs_type=lidar
PATCH_BYTES_PER_CAR = 4096
GATHER_MODES = ("full-u16", "full", "summary", "none")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle", type=int, default=150,
                    help="untimed steps between reset and the warm-up in which the random-action rollout spreads from the "
                         "spawn poses (part of preparing the synthetic data; reported as config.settle_steps)")
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--cars", type=int, default=1)
    ap.add_argument("--track", default="austria")
    ap.add_argument("--mixed-tracks", action="store_true",
                    help="BASELINE.json configs[4]: rank r runs track [columbia, austria, barcelona][r mod 3]")
    ap.add_argument("--obs-type", default="lidar", choices=["lidar", "lidar_occupancy"])
    ap.add_argument("--repeat", type=int, default=1, help="action repeat (sub-steps per step)")
    ap.add_argument("--gather", default="full-u16", choices=GATHER_MODES,
                    help="N>1: what the per-step RCCL all-gather carries. full-u16 (default) = the whole transition record "
                         "with the LiDAR row as uint16 written by the scan's store path (2 236 B/car); full = the fp32 "
                         "record (4 396 B/car); summary = the record without the LiDAR row (76 B/car, the scans stay "
                         "sharded in each rank's HBM); none = no collective.  All but `none` are xGMI-bound at this "
                         "simulation rate: DESIGN.md §6 has the table")
    ap.add_argument("--gather-every", type=int, default=1,
                    help="N>1: steps per all-gather (each collective carries that many per-step records: same bytes, "
                         "fewer launches; needs staging copies, so 1 - no copy, the collective reads the record in "
                         "place - is the default)")
    ap.add_argument("--gather-via", default="torch", choices=["torch", "abi", "p2p"],
                    help="transport of the gather: torch.distributed (RCCL inside PyTorch), the C-ABI's own "
                         "rc_gather_trajectory (RCCL bound by libracecar_hip.so; unique id passed through torch.distributed) "
                         "or rc_gather_trajectory_p2p (direct peer copies over hipIpc handles, one copy stream per peer)")
    ap.add_argument("--no-gather", action="store_true", help="same as --gather none")
    ap.add_argument("--no-gather-modes", action="store_true", help="N>1: skip the short legs that time the other gather modes")
    ap.add_argument("--no-gather-check", action="store_true",
                    help="N>1: skip the self-check of every gathered payload (each rank's shard against the sender's checksum)")
    ap.add_argument("--force-gather", action="store_true",
                    help="run the all-gather path even with one rank (needs a torch.distributed.run launch)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl = RCCL; gloo only for functional tests on one GPU)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="N>1 started without a launcher: seconds after which the ranks this process started are killed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ftg", dest="no_cpu_baseline_ftg", action="store_true", help="skip the follow-the-gap secondary figure")
    ap.add_argument("--no-configs", action="store_true", help="N=1: skip the other single-GPU configurations of BASELINE.json")
    ap.add_argument("--cpu-envs", type=int, default=0, help="envs in the CPU baseline sample (0 = auto)")
    ap.add_argument("--numpy-envs", type=int, default=4096, help="batch of the vectorised-NumPy CPU leg (SURVEY.md 8d)")
    ap.add_argument("--no-numpy-baseline", action="store_true", help="skip the vectorised-NumPy CPU leg (one step takes seconds)")
    ap.add_argument("--raycast-variant", type=int, default=None)
    ap.add_argument("--debug-knob", action="append", default=[], metavar="NAME=VALUE",
                    help="experiment knob passed to rc_debug_set (ray_threads, ray_split, ray_wg_per_cu, band_log2)")
    return ap.parse_args()


def self_launch(n_ranks, timeout_s):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves - one child running
    `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` - BEFORE this process has touched the
    GPU (nothing above imports torch), pass the ranks' output through (rank 0 prints the JSON line), and exit with the
    child's code: non-zero if any rank failed, 124 if the ranks did not finish within `timeout_s` (the whole process
    group this function started is then killed - by its id, nothing else).  Never re-executes a process that has
    initialised the GPU."""
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL and hipIpc* need on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    child = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        rc = child.wait(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        print(f"bench.py: the {n_ranks} ranks did not finish within {timeout_s:.0f} s - killing them", file=sys.stderr)
        try:
            os.killpg(child.pid, signal.SIGTERM)
            child.wait(timeout=10)
        except (ProcessLookupError, subprocess.TimeoutExpired):
            try:
                os.killpg(child.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
        rc = 124
    except KeyboardInterrupt:
        os.killpg(child.pid, signal.SIGTERM)
        rc = 130
    sys.exit(rc)


def cpu_baseline(track, cars, obs_type, repeat, n_envs):
    """The CPU oracle timed on this host's cores on a bounded sample of the same workload."""
    from oracle import cpu_baseline as cb
    return cb.run(track, cars=cars, occupancy=(obs_type == "lidar_occupancy"), repeat=repeat, n_envs=n_envs)


class Gatherer:
    """One gather mode of the N > 1 run: where the record lies, how it is sent, what it costs on the links.

    With one collective per step (`every == 1`) EVERY payload is read in place from a double-buffered source - the compact
    slab pair for `full-u16` (`rotate_compact`), a pair of output arenas for `full` and `summary` (`rc_set_arena`) - so the
    step that follows a collective writes the OTHER buffer and the collective never reads a record that is being
    overwritten (the header of rc_gather_trajectory demands exactly that of its caller).  `every > 1` goes through
    TrajectoryGather's staging copies."""

    def __init__(self, env, mode, every, via, dist_mod):
        import torch
        from racing_dreamer_amd.distributed import TrajectoryGather, gather_link_model
        self.env, self.mode, self.via = env, mode, via
        self.world = dist_mod.get_world_size()
        if mode == "full-u16":
            if getattr(env, "compact", None) is None:
                env.enable_compact(buffers=2)
        elif getattr(env, "compact", None) is not None:
            env.disable_compact()               # this leg's scan does not write the uint16 rows, its step does not copy the summary
        self.in_place = every == 1 or via != "torch"
        self.arenas, self._k, self._sent, self._last_dst = None, 0, 0, None
        if self.in_place and mode in ("full", "summary"):
            second = torch.zeros(env.arena_nbytes + 64, dtype=torch.uint8, device=env.device)
            pad = (-second.data_ptr()) % 64
            self.arenas = [None, second[pad:pad + env.arena_nbytes]]       # None = the env's own arena
            self._keep = second
        src = env.gather_source(mode)
        self.bytes = int(src.numel())
        self.model = gather_link_model(self.bytes, self.world)
        self.tg = None
        if via == "torch":
            self.tg = TrajectoryGather(src, every=every, stage=not self.in_place)
        elif via == "abi":
            self.dst = [torch.empty(self.world * self.bytes, dtype=torch.uint8, device=env.device) for _ in range(2)]
        self.includes = {"full-u16": "scan writes fp32 + uint16 rows; 76 B/car summary copied behind it; collective of 2 236 B/car",
                         "full": "collective of the fp32 record (4 396 B/car) read in place from alternating arenas",
                         "summary": "collective of pose..time (76 B/car) read in place from alternating arenas"}[mode]

    def _source(self):
        env = self.env
        if self.mode == "full-u16":
            return env.compact
        view = env._arena_view if self.arenas[self._k] is None else self.arenas[self._k]
        if self.mode == "full":
            return view[:env.slab.numel()]
        off = env.summary_slab.data_ptr() - env._arena_view.data_ptr()
        return view[off:off + env.summary_slab.numel()]

    def after_step(self):
        env = self.env
        if self.via == "abi":
            # the collective before last wrote dst[k]: order this one behind it, then send the record just produced
            env.gather_wait(host_sync=False)
            self._last_dst = self.dst[self._sent & 1]
            env.gather(self.mode, self._last_dst)
            self._sent += 1
        elif self.via == "p2p":
            env.gather_p2p(self.mode)
        else:
            self.tg.launch(self._source())
        if self.mode == "full-u16":
            env.rotate_compact()
        elif self.arenas is not None:
            self._k ^= 1
            env.set_arena(self.arenas[self._k])

    def wait(self):
        if self.via == "abi":
            self.env.gather_wait(host_sync=True)
        elif self.via == "p2p":
            self.env.gather_p2p_wait(host_sync=True)
        else:
            self.tg.wait()

    @staticmethod
    def _checksum(t):
        """Two sums over the record's 32-bit words (plain, and weighted by position), in wrapping int64."""
        import torch
        w = t.contiguous().view(torch.int32).to(torch.int64)
        pos = torch.arange(w.numel(), device=w.device, dtype=torch.int64) % 65521 + 1
        return [int(w.sum().item()), int((w * pos).sum().item())]

    def check(self, step, k0, dist_mod):
        """Self-check of the collective on whatever hardware this run is on: three steps in the timed loop's own rhythm (each
        followed by its collective, the next step launched behind it into the other buffer of the pair), one more step to
        overwrite what a late read would see, then every rank compares EVERY rank's shard of the last gathered record with the
        checksum that rank took of the record before it was sent.  Returns {"ok", "ranks", ...}; the same on all ranks."""
        import torch
        if not self.in_place:
            return {"ok": None, "skipped": "staged batches (--gather-every > 1)"}
        mine = None
        for j in range(3):
            step(k0 + j)
            mine = self._checksum(self._source())
            if j == 2 and os.environ.get("RC_BENCH_CORRUPT_GATHER") == str(dist_mod.get_rank()):
                self._source()[5] ^= 1           # tests: this check must see a single flipped bit in one rank's record
            self.after_step()
        step(k0 + 3)
        self.wait()
        self.env.sync()
        if self.via == "abi":
            got = self._last_dst.view(self.world, -1)
        elif self.via == "p2p":
            got = torch.from_numpy(self.env.gathered_p2p_host())
        else:
            got = self.tg.wait()
        sums = [None] * self.world
        dist_mod.all_gather_object(sums, mine)
        bad = [r for r in range(self.world) if self._checksum(got[r]) != sums[r]]
        oks = [None] * self.world
        dist_mod.all_gather_object(oks, not bad)
        out = {"ok": all(oks), "ranks": self.world, "records_in_flight": 3, "bytes_per_rank": self.bytes, "via": self.via}
        if bad:
            out["bad_shards_seen_by_this_rank"] = bad
        return out

    def close(self):
        """Leave the env as it was found: outputs in its own arena."""
        self.wait()
        if self.arenas is not None:
            self.env.set_arena(None)
            self._k = 0


def time_config(name, track_name, envs, cars, obs_type, steps, warmup, mode="random", settle=150):
    """One of BASELINE.json's other single-GPU configurations on the current device: ms per step, per-kernel times
    from launch-attached HIP events, whole-step and per-kernel HBM-roofline fractions."""
    import torch
    from racing_dreamer_amd.batched_env import BatchedRaceEnv
    env = BatchedRaceEnv(track_name, envs, cars, obs_type=obs_type, auto_reset=True)
    env.reset(mode=mode, seed=0)
    torch.cuda.set_stream(env.stream)
    for k in range(settle):
        env.step_random(seed=2, step=k)
    for k in range(warmup):
        env.step_random(seed=1, step=k)
    env.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        env.step_random(seed=1, step=warmup + k)
    env.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    env.reset_kernel_times()
    env.set_profiling(True)
    for k in range(min(steps, 40)):
        env.step_random(seed=1, step=warmup + steps + k)
    env.sync()
    env.set_profiling(False)
    kt = {k: round(v["avg_ms"], 4) for k, v in env.kernel_times().items() if v["launches"]}
    env.close()
    n_cars = envs * cars
    step_bytes = (STEP_BYTES_PER_CAR + (PATCH_BYTES_PER_CAR if obs_type == "lidar_occupancy" else 0)) * n_cars
    ms = dt / steps * 1e3
    out = {"workload": name, "envs": envs, "cars_per_env": cars, "track": track_name, "obs_type": obs_type,
           "steps": steps, "ms_per_step": ms, "env_steps_per_s": envs * steps / dt, "kernels_ms": kt,
           "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                        "step_bytes": step_bytes, "step_achieved": step_bytes / (ms * 1e-3) / 1e9,
                        "step_frac": step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}
    if kt.get("rc_raycast_kernel"):
        a = RAYCAST_BYTES_PER_CAR * n_cars / (kt["rc_raycast_kernel"] * 1e-3) / 1e9
        out["roofline"]["raycast_achieved"], out["roofline"]["raycast_frac"] = a, a / HBM_PEAK_GBS
    if kt.get("rc_patch_kernel"):
        a = PATCH_BYTES_PER_CAR * n_cars / (kt["rc_patch_kernel"] * 1e-3) / 1e9
        out["roofline"]["patch_achieved"], out["roofline"]["patch_frac"] = a, a / HBM_PEAK_GBS
    return out


def time_mixed_tracks(names, envs, steps, warmup, settle=150):
    """BASELINE.json configs[4]'s track mix inside ONE batch on the current device (not the 8-GPU run): `envs` envs in equal
    blocks on `names`, one handle per track filling one arena (MixedTrackEnv), random-action rollouts."""
    import torch
    from racing_dreamer_amd.batched_env import MixedTrackEnv
    per = [envs // len(names) + (1 if i < envs % len(names) else 0) for i in range(len(names))]
    env = MixedTrackEnv(list(names), per, auto_reset=True)
    env.reset(mode="random", seed=0)
    torch.cuda.set_stream(env.stream)
    for k in range(settle):
        env.step_random(seed=2, step=k)
    for k in range(warmup):
        env.step_random(seed=1, step=k)
    env.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        env.step_random(seed=1, step=warmup + k)
    env.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lead = env.parts[0]                     # (a step is one launch per kernel over all blocks, timed on the first handle)
    lead.reset_kernel_times()
    lead.set_profiling(True)
    for k in range(min(steps, 40)):
        env.step_random(seed=1, step=warmup + steps + k)
    env.sync()
    lead.set_profiling(False)
    kt = {k: round(v["avg_ms"], 4) for k, v in lead.kernel_times().items() if v["launches"]}
    env.close()
    ms = dt / steps * 1e3
    step_bytes = STEP_BYTES_PER_CAR * envs
    return {"workload": f"configs[4]'s track mix on ONE GPU: {envs} envs in {len(names)} blocks ({', '.join(names)}), one handle per "
                        f"track filling one arena, one launch per kernel over all blocks (MixedTrackEnv, rc_step_group); the 8-GPU "
                        f"run itself is `--gpus 8 --mixed-tracks`",
            "envs": envs, "cars_per_env": 1, "track": "mixed: " + " / ".join(names), "obs_type": "lidar", "steps": steps,
            "ms_per_step": ms, "env_steps_per_s": envs * steps / dt, "kernels_ms": kt,
            "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "step_bytes": step_bytes,
                         "step_achieved": step_bytes / (ms * 1e-3) / 1e9, "step_frac": step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus, args.launch_timeout)          # does not return
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)", file=sys.stderr)
        sys.exit(2)
    distributed = world > 1 or (args.force_gather and "RANK" in os.environ)
    dev = local_rank % max(torch.cuda.device_count(), 1)      # ranks > GPUs only in --backend gloo functional tests
    torch.cuda.set_device(dev)
    comm_ranks = None
    if distributed:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo")
        # what the communicator itself says about its size: one contribution per rank, summed by the collective
        one = torch.ones(1, dtype=torch.int32, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(one)
        comm_ranks = int(one.item())

    from racing_dreamer_amd import _lib as L
    from racing_dreamer_amd.batched_env import BatchedRaceEnv
    from racing_dreamer_amd.distributed import gather_link_model, shard_envs
    from racing_dreamer_amd.track_assets import load_track

    track_name = ["columbia", "austria", "barcelona"][rank % 3] if args.mixed_tracks else args.track
    track = load_track(track_name)
    shard = shard_envs(args.envs * world, rank, world)
    env = BatchedRaceEnv(track, shard.num_envs, args.cars, obs_type=args.obs_type, action_repeat=args.repeat,
                         device=dev, first_env=shard.first_env, auto_reset=True, profiling=False)
    if args.raycast_variant is not None:
        L.check(env._lib.rc_set_raycast_variant(env._h, args.raycast_variant))
    for kv in args.debug_knob:
        name, _, val = kv.partition("=")
        env.debug_set(name, int(val))
    env.reset(mode="random", seed=0)
    # the synthetic data is a random-action rollout that HAS SETTLED: right after a reset every car stands on the centre line
    # looking along the track (longer rays, a scan 5 % slower than in the spread of poses a long run is made of)
    for k in range(args.settle):
        env.step_random(seed=2, step=k)
    env.sync()
    gather_mode = "none" if (args.no_gather or not distributed) else args.gather
    via = args.gather_via
    abi_ranks = None
    if distributed and via == "abi":
        if args.backend != "nccl":
            via = "torch"                       # RCCL wants one GPU per rank; the gloo functional tests share one
        else:
            ids = [BatchedRaceEnv.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            env.comm_init(ids[0], rank, world)
            abi_ranks = env.comm_count()
    every = max(1, args.gather_every)
    if via == "p2p" and every != 1:
        raise SystemExit("--gather-via p2p sends one record per gather (--gather-every 1)")

    def p2p_close():
        # exported memory must not be freed while a peer still maps it: everybody unmaps, THEN everybody frees
        env.p2p_disconnect()
        dist.barrier()
        env.p2p_teardown()
        env._p2p_mode = None

    def make_gatherer(mode):
        if mode == "none":
            return None
        if via == "p2p":                        # buffers and mappings are made once; later legs only switch the payload
            first = getattr(env, "_p2p_mode", None) is None
            blob = env.p2p_setup(mode, rank, world)
            if first:
                blobs = [None] * world
                dist.all_gather_object(blobs, blob)
                env.p2p_connect(blobs)
        return Gatherer(env, mode, every, via, dist)

    gather = make_gatherer(gather_mode)

    # the rollout loop works on the env's own stream (no cross-stream event waits between the step's kernels and
    # the collective's dependency on them); `barrier()` synchronises the whole device
    torch.cuda.set_stream(env.stream)

    def one_step(k, repeat=None, g=None):
        env.step_random(seed=1, step=k, repeat=repeat)        # actions drawn inside the dynamics kernel (Philox, on device)
        if g is not None:
            g.after_step()

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def finish(g):
        if g is not None:
            g.wait()
        env.sync()

    for k in range(args.warmup):
        one_step(k, g=gather)
    finish(gather)
    env.reset_kernel_times()
    # start / stop timestamps attached to every launch of the DOMINANT kernel (the scan) on the stream it runs on;
    # timing the two small kernels as well would cost the timed region 6 us per step, so they get a pass of their
    # own after it
    env.set_profiling(True, kernels=[L.K_RAYCAST])
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step(args.warmup + k, g=gather)
    finish(gather)
    barrier()
    dt = time.perf_counter() - t0
    env.set_profiling(False)
    ktimes = env.kernel_times()
    scan_symbol = env.scan_kernel_name()
    # the other kernels of the step: a short untimed pass with all timers on
    env.reset_kernel_times()
    env.set_profiling(True, kernels=[L.K_PATCH, L.K_DYNAMICS])
    for k in range(min(args.steps, 50)):
        one_step(args.warmup + args.steps + k, g=gather)
    finish(gather)
    env.set_profiling(False)
    for name, v in env.kernel_times().items():
        if name != "rc_raycast_kernel":
            ktimes[name] = v
    step_no = args.warmup + 2 * args.steps

    # secondary figure: the reference's own setting, action_repeat 4 with the scan once per agent step
    # (dreamer/dream.py:55) - a quarter of the steps, same barriers, same payload as the headline
    r4_steps = max(args.steps // 4, 5)
    barrier()
    t1 = time.perf_counter()
    for k in range(r4_steps):
        one_step(step_no + k, repeat=4, g=gather)
    finish(gather)
    barrier()
    dt4 = time.perf_counter() - t1
    step_no += r4_steps
    headline_bytes = gather.bytes if gather is not None else 0
    headline_in_place = gather.in_place if gather is not None else True
    gather_checks = {}
    if gather is not None:
        if not args.no_gather_check:
            gather_checks[gather_mode] = gather.check(lambda k: env.step_random(seed=1, step=k), step_no, dist)
            step_no += 4
        gather.close()

    # N > 1: the same loop with each of the other payloads, short legs with the same barriers (every rank runs the same
    # sequence): what the headline's choice of payload costs, measured rather than argued.  Each leg sets the env up for
    # its own payload only (`includes` says what its step carried); `none` is the pure simulation rate.
    mode_legs, leg_includes = {}, {}
    if distributed and not args.no_gather_modes:
        n_leg = max(args.steps // 4, 5)
        for m in GATHER_MODES:
            if m == gather_mode:
                continue
            if m == "none" and getattr(env, "compact", None) is not None:
                env.disable_compact()
            g = make_gatherer(m)
            leg_includes[m] = g.includes if g is not None else "no collective, no uint16 rows, no summary copy: the simulation alone"
            for k in range(3):
                one_step(step_no + k, g=g)
            finish(g)
            barrier()
            t1 = time.perf_counter()
            for k in range(n_leg):
                one_step(step_no + 3 + k, g=g)
            finish(g)
            barrier()
            mode_legs[m] = [time.perf_counter() - t1, n_leg]
            step_no += 3 + n_leg
            if g is not None:
                if not args.no_gather_check:
                    gather_checks[m] = g.check(lambda k: env.step_random(seed=1, step=k), step_no, dist)
                    step_no += 4
                g.close()
    if getattr(env, "_p2p_mode", None) is not None:
        p2p_close()

    # N > 1: the payload DESIGN.md 6 recommends instead of per-step records - every rank keeps its records in a
    # device-resident ring and what crosses the links is the TRAINING BATCH: ShardedReplay.sample(50 windows x 50 steps)
    # once per step (far more often than a learner asks for one)
    batch_leg = None
    if distributed and not args.no_gather_modes:
        from racing_dreamer_amd.replay import ShardedReplay, TrajectoryRing
        if getattr(env, "compact", None) is not None:
            env.disable_compact()
        length, cap = 50, 64
        batch = -(-50 // world) * world
        ring = TrajectoryRing(env, cap)
        for k in range(length + 4):                   # fill: a window needs `length` records
            ring.step_random(seed=1, step=step_no + k)
        step_no += length + 4
        rep = ShardedReplay(ring)
        gen = torch.Generator(device=env.device)
        gen.manual_seed(1234 + rank)
        fields = ("lidar", "action", "reward", "discount")
        for k in range(3):                            # warm-up: the sampler's kernels, the collectives' first use per dtype and size
            ring.step_random(seed=1, step=step_no + k)
            out = rep.sample(batch, length, fields=fields, generator=gen)
        step_no += 3
        batch_bytes = sum(int(out[f].numel() * out[f].element_size()) for f in fields) // world
        n_leg = max(args.steps // 4, 5)
        env.sync()
        barrier()
        t1 = time.perf_counter()
        for k in range(n_leg):
            ring.step_random(seed=1, step=step_no + k)
            rep.sample(batch, length, fields=fields, generator=gen)
        env.sync()
        barrier()
        batch_leg = [time.perf_counter() - t1, n_leg, batch_bytes, batch, length]
        step_no += n_leg
        ring.detach()
        del rep, ring

    # secondary figure: cars driven along the track at speed by the reference's follow-the-gap law (its other prefill
    # policy, dreamer/dream.py:211-216) instead of crawling under random actions: rank 0 only, single-GPU runs only
    ftg = None
    if world == 1 and not args.no_cpu_baseline_ftg:
        mean_range_random = float(env.views["lidar"].float().mean().item())
        env.reset(mode="random", seed=0)
        for k in range(150):                      # let the cars settle on the racing line
            env.follow_the_gap_reference()
            env.step(None)
        env.sync()
        env.reset_kernel_times()
        env.set_profiling(True, kernels=[L.K_RAYCAST, L.K_FTG])
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        n_ftg = max(args.steps // 4, 5)
        for k in range(n_ftg):
            env.follow_the_gap_reference()
            env.step(None)
        env.sync()
        torch.cuda.synchronize()
        dtf = time.perf_counter() - t2
        env.set_profiling(False)
        kt = env.kernel_times()
        ftg = {"env_steps_per_s": args.envs * n_ftg * args.repeat / dtf, "steps": n_ftg,
               "raycast_ms": round(kt["rc_raycast_kernel"]["avg_ms"], 4),
               "agent_kernel_ms": round(kt["rc_ftg_kernel"]["avg_ms"], 4),
               "mean_range_m": float(env.views["lidar"].float().mean().item()),
               "mean_speed_m_s": float(env.views["speed"].float().mean().item()),
               "mean_range_m_random_actions": mean_range_random,
               "note": "same envs driven by rc_follow_the_gap_reference - the law of the reference's own follow-the-gap node "
                       "(ros_agent/agents/follow_the_gap/src/agent.py:128-234) as a device agent - after 150 settling steps "
                       "(cars at 4 m/s instead of crawling under random actions); includes the agent's kernel"}

    if distributed:
        times = [dt, dt4] + [v[0] for v in mode_legs.values()] + ([batch_leg[0]] if batch_leg else [])
        tmax = torch.tensor(times, dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        times = [float(v) for v in tmax.tolist()]
        dt, dt4 = times[0], times[1]
        for (m, v), t in zip(mode_legs.items(), times[2:]):
            v[0] = t
        if batch_leg:
            batch_leg[0] = times[-1]

    total_envs = args.envs * world
    env_steps = total_envs * args.steps * args.repeat
    value = env_steps / dt
    n_cars = shard.num_envs * args.cars
    out = None
    if rank == 0:
        ray = ktimes["rc_raycast_kernel"]
        ray_s = ray["avg_ms"] * 1e-3
        achieved = RAYCAST_BYTES_PER_CAR * n_cars / ray_s / 1e9 if ray_s > 0 else 0.0
        # HBM traffic and instruction counts of the scan are PMC figures: they cannot be collected inside this run (the
        # counters need rocprofv3 passes of their own), so they are quoted from the committed profile of the SAME
        # workload (track, batch, obs_type, default scan) and are null for any other - `traffic_source` says which file
        traffic = valu = traffic_source = None
        tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        key = f"{track_name}:{shard.num_envs}x{args.cars}:{args.obs_type}"
        if os.path.exists(tp) and args.raycast_variant is None and not args.debug_knob and scan_symbol.endswith("false, false>"):
            with open(tp) as f:
                prof = json.load(f)
            traffic = prof.get(key)
            valu = prof.get("_valu_wave_insts", {}).get(key)
            if traffic is not None:
                traffic_source = prof.get("_source", "profiles/hbm_traffic.json") + " (builder-run rocprofv3 --pmc passes of this workload; not measured in this run)"
        if not distributed:
            gather_txt = "no collective (one rank)"
        elif gather_mode == "none":
            gather_txt = "NO trajectory gather (--gather none)"
        else:
            how = {"torch": "RCCL all-gather through torch.distributed", "abi": "RCCL all-gather through rc_gather_trajectory",
                   "p2p": "direct peer copies through rc_gather_trajectory_p2p (hipIpc, one copy stream per peer)"}[via]
            gather_txt = (f"every step's trajectory record gathered on every rank as `{gather_mode}` ({headline_bytes} B per GPU "
                          f"per step, {'read in place from a double-buffered source' if headline_in_place else 'from staging copies'}, "
                          f"one gather per {every} step{'s' if every > 1 else ''}, {how}, overlapped with the "
                          f"following step; the gathered buffer is overwritten two gathers later - no consumer in this benchmark)")
        out = {
            "metric": "env-steps/sec at 65 536 parallel envs, 1080-beam LiDAR, 1/2/4/8 MI355X",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32",
            "data": f"synthetic (random-action rollout, {args.settle} untimed settling steps after reset, then the warm-up)",
            "config": {
                "workload": f"{args.envs} envs/GPU x {args.cars} car, track "
                            f"{'mixed columbia/austria/barcelona by rank' if args.mixed_tracks else args.track}, obs_type={args.obs_type}, "
                            f"1080-beam lidar every sub-step, random-action rollouts (Philox on device), "
                            f"auto-reset, action_repeat {args.repeat}; {gather_txt}",
                "envs_per_gpu": args.envs, "total_envs": total_envs, "cars_per_env": args.cars,
                "track": "mixed: [columbia, austria, barcelona][rank mod 3]" if args.mixed_tracks else args.track,
                "obs_type": args.obs_type, "action_repeat": args.repeat, "settle_steps": args.settle,
                "parallelism": f"env-sharded x{world}", "gather": gather_mode, "gather_via": via if distributed else None,
                # the size of the job as the communicator reports it (sum over ranks of 1 through the backend's own
                # all-reduce; ncclCommCount of the C-ABI's communicator when that transport is used)
                "rccl_ranks": comm_ranks if (distributed and args.backend == "nccl") else None,
                "comm_backend": args.backend if distributed else None, "comm_ranks": comm_ranks, "abi_comm_ranks": abi_ranks,
            },
            "roofline": {
                "bound": "hbm", "kernel": scan_symbol, "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": RAYCAST_BYTES_PER_CAR * n_cars,
                "avg_launch_ms": ray["avg_ms"], "launches": ray["launches"],
                "rays_per_s": n_cars * 1080 / ray_s if ray_s > 0 else 0.0,
                # what actually bounds the scan: wave-level VALU instructions per launch from the PMC
                # profile, and the rate they retire at per SIMD (1 024 SIMDs) at the duration measured here
                "valu_wave_insts_per_launch": valu,
                "valu_insts_per_simd_per_us": (valu / 1024 / (ray_s * 1e6)) if (valu and ray_s > 0) else None,
                "note": "compulsory HBM traffic is ~4.3 KB per car-scan, so the scan is bound by the instruction stream "
                        "of the grid traversal, not by HBM (SURVEY.md §8d); the >= 40 % HBM target is NOT met; DESIGN.md §4.2",
            },
            "kernels_ms": {k: round(v["avg_ms"], 4) for k, v in ktimes.items() if v["launches"]},
            "action_repeat_4": {"env_steps_per_s": total_envs * r4_steps * 4 / dt4,
                                "agent_steps_per_s": total_envs * r4_steps / dt4, "steps": r4_steps,
                                "note": "same workload with action_repeat 4, LiDAR once per agent step (dreamer/dream.py:55)"},
        }
        if distributed:
            # every payload next to its link bound (xGMI full mesh, 7 links x 76.8 GB/s inbound per GPU)
            sizes = {"full": int(env.slab.numel()), "summary": int(env.summary_slab.numel()),
                     "full-u16": int(env._lib.rc_compact_bytes(env._cfg)), "none": 0}
            table = {}
            for m in GATHER_MODES:
                e = gather_link_model(sizes[m], world)
                if m == gather_mode:
                    e.update(ms_per_step=dt / args.steps * 1e3, env_steps_per_s=value, steps=args.steps, headline=True)
                elif m in mode_legs:
                    t, n = mode_legs[m]
                    e.update(ms_per_step=t / n * 1e3, env_steps_per_s=total_envs * n * args.repeat / t, steps=n,
                             includes=leg_includes.get(m))
                table[m] = e
            if batch_leg:
                t, n, nbytes, batch, length = batch_leg
                e = gather_link_model(nbytes, world)
                e.update(ms_per_step=t / n * 1e3, env_steps_per_s=total_envs * n * args.repeat / t, steps=n,
                         includes=f"records kept in a {64}-slot device ring per rank (rc_set_arena, no copy); every step "
                                  f"ShardedReplay.sample({batch} windows x {length} steps: lidar, action, reward, discount) "
                                  f"all-gathered over torch.distributed - {nbytes} B per rank per sample; windows drawn "
                                  f"and gathered on the device (rc_sample_windows, rc_gather_rows), one host read of the failure count per sample")
                table["batch"] = e
            out["gather_modes"] = table
        if gather_checks:
            # every payload's collective checked on THIS hardware: each rank's shard of the last of three in-flight records
            # equals the checksum its sender took before sending (Gatherer.check)
            out["gather_check"] = dict(ok=all(c["ok"] is not False for c in gather_checks.values()), payloads=gather_checks)
        if ftg is not None:
            out["follow_the_gap"] = ftg
    env.close()
    if rank == 0:
        if world == 1 and not args.no_configs:
            # BASELINE.json configs[1..3], each a few ms of GPU time (configs[0] is the CPU plumbing case, configs[4]
            # the 8-GPU run: `--gpus 8 --mixed-tracks`)
            cfgs = [("configs[1]: 4 096 envs, columbia, 1080-beam lidar", "columbia", 4096, 1, "lidar", 400, 40, "random"),
                    ("configs[2]: 65 536 envs, austria, obs_type=lidar_occupancy (64x64 render)", "austria", 65536, 1,
                     "lidar_occupancy", 100, 10, "random"),
                    ("configs[3]: 32 768 envs x 2 cars, treitlstrasse_v2, inter-car raycast + collision",
                     "treitlstrasse_v2", 32768, 2, "lidar", 100, 10, "random_ball")]
            out["configs"] = [time_config(*c, settle=args.settle) for c in cfgs]
            out["configs"].append(time_mixed_tracks(("columbia", "austria", "barcelona"), 65536, 100, 10, settle=args.settle))
        if not args.no_cpu_baseline and world == 1:
            from oracle import cpu_baseline as cb
            out["cpu_baseline"] = cpu_baseline(track, args.cars, args.obs_type, args.repeat, args.cpu_envs)
            out["cpu_baseline"]["single_env"] = cb.run_single_env()      # BASELINE.json configs[0]: the B = 1 CPU step()
            if not args.no_numpy_baseline:
                out["cpu_baseline"]["numpy_batch"] = cb.run_numpy_batch(track, n_envs=args.numpy_envs)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if any(c["ok"] is False for c in gather_checks.values()):
        print(f"bench.py: rank {rank}: a gathered record differs from what its sender sent: {gather_checks}", file=sys.stderr)
        sys.exit(4)


if __name__ == "__main__":
    main()
