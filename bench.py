#!/usr/bin/env python3
"""Benchmark of the hot path: env-steps/s of the batched racing env on N MI355X (one process per GPU).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python bench.py --gpus 8 --steps 200 --warmup 20          (starts its own 8 ranks, see self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 200 --warmup 20

A "step" is one pass of the hot path over one batch: random actions (Philox, on device) ->
integrator + collision + progress/reward/done + auto-reset -> 1080-beam LiDAR scan, for 65 536 envs
per GPU (weak scaling), action_repeat 1, so one step = one simulator sub-step (dt = 0.01 s) of every
env.  For N > 1 the headline (`--gather sharded`) is the trajectory store at the granularity
its consumer reads it: every rank's records go straight into its own device-resident TrajectoryRing,
every step's 76 B/car summary (pose .. time: the record without the LiDAR row) is all-gathered over
RCCL, and every 10th step one training batch of 50 windows x 50 steps (dreamer/dream.py:80-83: 100
batches per 1 000 env steps) is drawn from the rings and all-gathered.  The per-step gathers of whole
records (`full-u16`, `full`) are timed beside it as secondary legs with their link bound.
Prints ONE JSON line on rank 0 - whatever happens after the headline leg: every secondary leg runs
under a deadline and an error guard (LineGuard), and the line is printed with what has been measured
when one of them fails.  At N = 1 the line also carries the other single-GPU configurations of
BASELINE.json (`configs`), the scan on the other tracks (`tracks`), the first steps after a reset
(`fresh_reset`) and the CPU baseline.
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
# dmabuf IPC (what RCCL and hipIpc* need on this driver): also for ranks an outside launcher started, and before anything
# initialises the HIP runtime (torch is imported inside main)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

T_START = time.time()            # when THIS process started; the outermost one hands its own down to its ranks (RC_BENCH_T0)
from racing_dreamer_amd.bench_guard import (EXIT_CHECK_MISMATCH, EXIT_LEG_LOST, T0_ENV, LineGuard,   # noqa: E402
                                            die_with_parent, self_launch)

HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
RAYCAST_BYTES_PER_CAR = 4 * 1080 + 16       # lidar row written + (x, y, cos, sin) read, DESIGN.md §5
STEP_BYTES_PER_CAR = 4 * 1080 + 159         # whole env-step, obs_type=lidar
PATCH_BYTES_PER_CAR = 4096
GATHER_MODES = ("sharded", "full-u16", "full", "summary", "none")
SIMDS = 1024                     # 256 CUs x 4
CLOCK_GHZ = 2.4
VALU_CYCLES_FULL_RATE = 2.35     # issue cost of a full-rate vector instruction with 8 waves per SIMD (tools/ubench/valu_issue4.hip,
                                 # profiles/r02_c_valu_issue4_ubench.txt); half-rate forms cost 4.3
BATCH_EVERY, BATCH_WINDOWS, BATCH_LENGTH = 10, 50, 50      # dreamer/dream.py:80-83: batch 50 x 50, 100 train steps per 1 000 env steps


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
    ap.add_argument("--gather", default="sharded", choices=GATHER_MODES,
                    help="N>1: what crosses the links. sharded (default) = records stay in each rank's device-resident "
                         "TrajectoryRing; per step the 76 B/car summary is all-gathered, every 10th step one training batch "
                         "of 50 x 50 windows (ShardedReplay; dreamer/dream.py:80-83).  full-u16 = every step's whole "
                         "transition record with the LiDAR row as uint16 (2 236 B/car); full = the fp32 record "
                         "(4 396 B/car); summary = the 76 B/car alone; none = no collective.  The per-step gathers of "
                         "whole records are xGMI-bound by an order of magnitude at this simulation rate: DESIGN.md 6")
    ap.add_argument("--leg-timeout", type=float, default=150.0,
                    help="the most seconds ONE leg may take (capped by what is left of --time-budget) before the line is printed "
                         "with what has been measured and the run ends (exit 5; exit 3 if that leg was the rendezvous or the "
                         "headline itself): a collective that has never run across devices may hang, and must not cost the line")
    ap.add_argument("--time-budget", type=float, default=520.0,
                    help="seconds the WHOLE run may take, counted from the start of the outermost process (the driver ends a run "
                         "after 600 s): legs are given deadlines from what is left, and legs for which too little is left are "
                         "skipped and listed in `legs_skipped`")
    ap.add_argument("--observable-seconds", type=float, default=12.0,
                    help="N = 1: wall time of the `steady` window (the headline's loop continued), sized from the measured rate so "
                         "that an outside sampler polling every 5 s sees the GPU busy at least twice; 0 = the 2 000-step window only")
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
    ap.add_argument("--launch-timeout", type=float, default=540.0,
                    help="N>1 started without a launcher: seconds after the start at which the ranks this process started are "
                         "ended (SIGTERM to their process group - rank 0 prints the line it has - then SIGKILL); inside the "
                         "driver's 600 s, behind --time-budget")
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


def cpu_baseline(track, cars, obs_type, repeat, n_envs):
    """The CPU oracle timed on this host's cores on a bounded sample of the same workload."""
    from oracle import cpu_baseline as cb
    return cb.run(track, cars=cars, occupancy=(obs_type == "lidar_occupancy"), repeat=repeat, n_envs=n_envs)


def _checksum(t):
    """Two sums over a record's 32-bit words (plain, and weighted by position), in wrapping int64."""
    import torch
    w = t.contiguous().view(-1).view(torch.uint8)
    w = w[:w.numel() // 4 * 4].view(torch.int32).to(torch.int64)
    pos = torch.arange(w.numel(), device=w.device, dtype=torch.int64) % 65521 + 1
    return [int(w.sum().item()), int((w * pos).sum().item())]


class Gatherer:
    """One per-step gather mode of the N > 1 run: where the record lies, how it is sent, what it costs on the links.

    With one collective per step (`every == 1`) EVERY payload is read in place from a double-buffered source - the compact
    slab pair for `full-u16` (`rotate_compact`), a pair of output arenas for `full` and `summary` (`rc_set_arena`) - so the
    step that follows a collective writes the OTHER buffer, and the step after that - which rewrites the source - is queued
    behind the collective's completion (every transport: torch's work.wait(), rc_gather_wait, rc_gather_p2p_wait put the
    env's stream behind collective k before collective k + 1 is issued).  `every > 1` goes through TrajectoryGather's
    staging copies."""

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
        self.arenas, self._k, self._sent, self._dsts = None, 0, 0, []
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
        view = env._arena_view if (self.arenas is None or self.arenas[self._k] is None) else self.arenas[self._k]
        if self.mode == "full":
            return view[:env.slab.numel()]
        off = env.summary_slab.data_ptr() - env._arena_view.data_ptr()
        return view[off:off + env.summary_slab.numel()]

    def after_step(self):
        env = self.env
        if self.via == "abi":
            # the collective before last wrote dst[k]: order this one behind it, then send the record just produced
            env.gather_wait(host_sync=False)
            self._dsts = (self._dsts + [self.dst[self._sent & 1]])[-2:]
            env.gather(self.mode, self._dsts[-1])
            self._sent += 1
        elif self.via == "p2p":
            if self._sent:
                env.gather_p2p_wait(host_sync=False)     # THE RULE (racecar_hip.h): the env's stream behind gather k - 1, inbound and outbound
            env.gather_p2p(self.mode)
            self._sent += 1
        else:
            self.tg.launch(self._source())
        if self.mode == "full-u16":
            env.rotate_compact()
        elif self.arenas is not None:
            self._k ^= 1
            env.set_arena(self.arenas[self._k])

    def step(self, k, repeat=None):
        self.env.step_random(seed=1, step=k, repeat=repeat)       # actions drawn inside the dynamics kernel (Philox, on device)
        self.after_step()

    def wait(self):
        if self.via == "abi":
            self.env.gather_wait(host_sync=True)
        elif self.via == "p2p":
            if self._sent:
                self.env.gather_p2p_wait(host_sync=True)
        else:
            self.tg.wait()

    def _gathered(self, back):
        import torch
        if self.via == "abi":
            return self._dsts[-1 - back].view(self.world, -1)
        if self.via == "p2p":
            return torch.from_numpy(self.env.gathered_p2p_host(back=back))
        return self.tg.recent(back)

    def check(self, k0, dist_mod):
        """Self-check of the collective on whatever hardware this run is on: three steps in the timed loop's own rhythm (each
        followed by its collective, the next step launched behind it into the other buffer of the pair) and one more step,
        which REWRITES the source of the second record.  Then every rank compares every rank's shard of the last TWO gathered
        records with the checksums their senders took before sending: the last one, whose source is still untouched, and
        the one before it, whose source has been overwritten since - a collective that read its source late shows there.
        Returns {"ok", "ranks", ...}; the same on all ranks."""
        if not self.in_place:
            return {"ok": None, "skipped": "staged batches (--gather-every > 1)"}
        sums = []
        for j in range(3):
            self.env.step_random(seed=1, step=k0 + j)
            sums.append(_checksum(self._source()))
            if j == 2 and os.environ.get("RC_BENCH_CORRUPT_GATHER") == str(dist_mod.get_rank()):
                self._source()[5] ^= 1           # tests: this check must see a single flipped bit in one rank's record
            self.after_step()
        self.env.step_random(seed=1, step=k0 + 3)
        self.wait()
        self.env.sync()
        all_sums = [None] * self.world
        dist_mod.all_gather_object(all_sums, sums)
        bad = []
        for back, j in ((0, 2), (1, 1)):
            got = self._gathered(back)
            bad += [[r, j] for r in range(self.world) if _checksum(got[r][:self.bytes]) != all_sums[r][j]]
        oks = [None] * self.world
        dist_mod.all_gather_object(oks, not bad)
        out = {"ok": all(oks), "ranks": self.world, "records_in_flight": 3, "records_verified": 2,
               "bytes_per_rank": self.bytes, "via": self.via}
        if bad:
            out["bad_rank_record_pairs_seen_by_this_rank"] = bad
        return out

    def close(self):
        """Leave the env as it was found: outputs in its own arena."""
        self.wait()
        if self.arenas is not None:
            self.env.set_arena(None)
            self._k = 0


class ShardedCollector:
    """The N > 1 headline: the concat of `Collect` -> `save_episodes` -> `load_episodes`
    (dreamer/wrappers.py:213-219, dreamer/tools.py:235-264) at the granularity its consumer reads it.  Every rank's records
    go straight into its own `TrajectoryRing` (`rc_set_arena` before each step: no copy); what crosses the links is
      * every step's 76 B/car summary (pose .. time), `summary_every` steps per collective: each step's summary is copied out
        of its ring slot into a staging buffer (5 MB) and every `summary_every`-th step ONE all-gather over torch.distributed
        (nccl = RCCL) carries them - the same bytes per step on the links, a tenth of the launches;
      * every `batch_every`-th step one training batch, `windows` windows x `length` steps of lidar, action, reward,
        discount drawn on the device from the ranks' rings (`ShardedReplay.draw`) and all-gathered as ONE packed buffer
        (`exchange`): the reference's cadence of 100 batches of 50 x 50 per 1 000 env steps (dreamer/dream.py:80-83).
    Copies and collectives run on a SIDE stream behind one event per step: the env's stream records events and never waits
    for the links - except before it rewrites a ring slot: then it waits for the event behind the RETIREMENT of the collective
    that read that slot (see `step`)."""

    FIELDS = ("lidar", "action", "reward", "discount")

    def __init__(self, env, dist_mod, rank, batch_every=BATCH_EVERY, windows=BATCH_WINDOWS, length=BATCH_LENGTH,
                 capacity=64, summary_every=1, summary=True):
        import torch
        from racing_dreamer_amd.distributed import TrajectoryGather, gather_link_model
        from racing_dreamer_amd.replay import ShardedReplay, TrajectoryRing
        self.torch = torch
        self.env, self.dist, self.mode, self.via = env, dist_mod, "sharded", "torch"
        self.world = dist_mod.get_world_size()
        if getattr(env, "compact", None) is not None:
            env.disable_compact()
        self.batch_every, self.length, self.capacity = int(batch_every), int(length), int(capacity)
        self.windows = -(-int(windows) // self.world) * self.world
        # one ring per env, reused by every leg that collects on it: a 64-slot ring of 65 536 records is 18.4 GB, and a fresh one
        # per leg left the freed ones cached in torch's allocator (the 6-rank rehearsal on one GPU ran out of memory in its
        # third leg: profiles/r06_c_*)
        self.ring = getattr(env, "_bench_ring", None)
        if self.ring is None or self.ring.capacity != capacity:
            self.ring = env._bench_ring = TrajectoryRing(env, capacity)
        else:
            self.ring.clear()
        self.rep = ShardedReplay(self.ring)
        self.gen = torch.Generator(device=env.device)
        self.gen.manual_seed(1234 + rank)
        self.side = torch.cuda.Stream(device=env.device)
        self.ev_step = [torch.cuda.Event() for _ in range(capacity)]
        self.ev_done = [None] * capacity
        self.ev_batch = torch.cuda.Event()
        # a training batch is ONE native call into ONE packed buffer (rc_sample_batch) and ONE collective (exchange_packed):
        # two preallocated buffer pairs take turns, the env's stream waits for the collective of two batches ago before it
        # draws into that pair again
        self.layout = env.sample_batch_layout(self.FIELDS, self.windows // self.world, self.length)
        def aligned(nbytes):
            raw = torch.empty(nbytes + 64, dtype=torch.uint8, device=env.device)
            return raw[(-raw.data_ptr()) % 64:][:nbytes]
        self.batch_src = [aligned(self.layout["total"]) for _ in range(2)]
        self.batch_dst = [aligned(self.world * self.layout["payload"]) for _ in range(2)]
        self.batch_done = [None, None]
        self.off = env.summary_slab.data_ptr() - env._arena_view.data_ptr()
        self.bytes = int(env.summary_slab.numel()) if summary else 0
        self.summary_every = int(summary_every)
        # one collective per step reads the summary IN PLACE from its ring slot (a slot is not rewritten for `capacity` steps,
        # so several may be in flight); more steps per collective go through staging copies.  A short timed window ends
        # with the last collective's whole latency in it: the fewer steps it carries, the shorter that tail
        self.tg = None
        if summary:
            self.tg = (TrajectoryGather(env.summary_slab, stage=False, depth=min(8, capacity - 4)) if self.summary_every == 1 else
                       TrajectoryGather(env.summary_slab, every=self.summary_every, stage=True, depth=2))
        # hand-overs between the one that issues a slot's in-place collective and the first whose event is behind it (see step)
        self.lag = self.tg.depth if (self.tg is not None and not self.tg.stage) else 0
        if self.lag + 1 >= capacity:
            raise ValueError(f"a ring of {capacity} slots cannot keep {self.lag} in-place collectives in flight")
        self.n, self.batches, self.last_batch = 0, 0, None
        per_car = {"lidar": 4320, "action": 8, "reward": 4, "discount": 4}
        self.batch_bytes = self.windows // self.world * self.length * sum(per_car[f] for f in self.FIELDS)
        self.model = gather_link_model(self.bytes + self.batch_bytes // self.batch_every, self.world)
        self.in_place = True
        self.includes = (f"records kept in a {capacity}-slot device ring per rank (rc_set_arena, no copy)"
                         + (f"; every step's 76 B/car summary ({self.bytes} B per rank per step) all-gathered "
                            + ("in place from its ring slot, one collective per step, several in flight" if self.summary_every == 1 else
                               f"through staging copies, {self.summary_every} steps per collective") if summary else "; no per-step exchange")
                         + f"; every {self.batch_every}th step ShardedReplay batch of {self.windows} windows x {self.length} steps "
                           f"({', '.join(self.FIELDS)}) drawn on the device (rc_sample_windows, rc_gather_rows) and all-gathered as one "
                           f"packed buffer - {self.batch_bytes} B per rank per batch; copies and collectives on a side stream behind "
                           f"one event per step, no host read in the loop")

    def _summary(self):
        return self.ring.slot(self.ring.head)[self.off:self.off + self.bytes]

    def prefill(self, k0):
        """`length` + 4 untimed records so that windows exist, and one batch to set up its collective."""
        n = self.length + 4
        for k in range(n):
            self.ring.step_random(seed=1, step=k0 + k)
        # one batch through the very path the loop takes (first use of a collective of this size, of the side stream, of the
        # sampler's kernels: none of that belongs into a timed window)
        buf, _ = self.rep.draw_packed(self.windows, self.length, fields=self.FIELDS, generator=self.gen, out=self.batch_src[1], layout=self.layout)
        self.ev_batch.record(self.env.stream)
        with self.torch.cuda.stream(self.side):
            self.side.wait_event(self.ev_batch)
            self.last_batch = self.rep.exchange_packed(buf, self.layout, out=self.batch_dst[1])
        self.env.stream.wait_stream(self.side)
        return n

    def _send_summary(self):
        torch, slot = self.torch, self.ring.head
        ev = self.ev_step[slot]
        ev.record(self.env.stream)
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            self.tg.launch(self._summary())
            done = self.ev_done[slot] or torch.cuda.Event()
            done.record(self.side)
            self.ev_done[slot] = done

    def step(self, k, repeat=None):
        torch = self.torch
        nxt = (self.ring.head + 1) % self.capacity
        # Before slot `nxt` is rewritten, whatever read it `capacity` steps ago must be over.  ev_done[s] is recorded on the side
        # stream right after the hand-over of slot s, and by then the side stream is only behind the collectives
        # `TrajectoryGather._retire` waited for: those up to `depth` BEFORE the one just issued (an async collective runs on the
        # process group's own stream; issuing it does not put the side stream behind it).  An in-place collective reads the ring
        # slot itself, so the event that covers slot nxt's collective is the one recorded `depth` hand-overs after it (ADVICE r4:
        # waiting on ev_done[nxt] alone let a rank that runs > capacity - depth steps ahead of a peer overwrite the source of a
        # pending all-gather).  A staged hand-over copies the slot on the side stream at once: there ev_done[nxt] is the copy.
        covered = (nxt + self.lag) % self.capacity
        if self.ev_done[covered] is not None and self.ev_done[nxt] is not None:
            self.env.stream.wait_event(self.ev_done[covered])
        self.ring.step_random(seed=1, step=k, repeat=repeat)
        if self.tg is not None:
            self._send_summary()
        self.n += 1
        if self.n % self.batch_every == 0:
            i = self.batches & 1
            if self.batch_done[i] is not None:
                self.env.stream.wait_event(self.batch_done[i])
            buf, _ = self.rep.draw_packed(self.windows, self.length, fields=self.FIELDS, generator=self.gen, out=self.batch_src[i],
                                          layout=self.layout)                      # env's stream: a memset and two launches
            self.ev_batch.record(self.env.stream)
            with torch.cuda.stream(self.side):
                self.side.wait_event(self.ev_batch)
                self.last_batch = self.rep.exchange_packed(buf, self.layout, out=self.batch_dst[i])     # views [world, windows / world, ...]
                done = self.batch_done[i] or torch.cuda.Event()
                done.record(self.side)
                self.batch_done[i] = done
            self.batches += 1

    def wait(self):
        with self.torch.cuda.stream(self.side):
            if self.tg is not None:
                self.tg.wait()
        self.env.stream.wait_stream(self.side)

    def check(self, k0, dist_mod):
        """Three steps in the loop's own rhythm, each followed by its summary hand-over, and one more; the collective is then
        flushed and every rank compares every rank's shard of ALL THREE gathered records with the checksums their senders
        took before sending - and every rank's rows of one gathered training batch with the checksum their owner holds of
        the same rows."""
        out = {"ok": True, "ranks": self.world, "via": "torch"}
        bad = []
        self.wait()
        if self.tg is not None:
            sums = []
            for j in range(3):
                self.ring.step_random(seed=1, step=k0 + j)
                sums.append(_checksum(self._summary()))
                if j == 2 and os.environ.get("RC_BENCH_CORRUPT_GATHER") == str(dist_mod.get_rank()):
                    self._summary()[5] ^= 1          # tests: this check must see a single flipped bit in one rank's record
                self._send_summary()
            self.ring.step_random(seed=1, step=k0 + 3)
            self.wait()
            self.env.sync()
            all_sums = [None] * self.world
            dist_mod.all_gather_object(all_sums, sums)
            if self.summary_every == 1:
                bad += [[r, 2 - back] for back in range(3) for r in range(self.world)
                        if _checksum(self.tg.recent(back)[r]) != all_sums[r][2 - back]]
            else:
                got = self.tg.recent(0)                      # [world, 3 snapshots, bytes]
                bad += [[r, j] for r in range(self.world) for j in range(3) if _checksum(got[r, j]) != all_sums[r][j]]
            out.update(records_in_flight=3, records_verified=3, bytes_per_rank=self.bytes)
        buf, _ = self.rep.draw_packed(self.windows, self.length, fields=self.FIELDS, generator=self.gen, layout=self.layout)
        batch = self.rep.exchange_packed(buf, self.layout)
        self.env.sync()
        self.torch.cuda.synchronize()
        me = dist_mod.get_rank()
        local = self.ring.unpack(buf, self.layout)           # what this rank drew, before it went through the collective
        mine = {f: _checksum(local[f]) for f in self.FIELDS + ("meta",)}
        owners = [None] * self.world
        dist_mod.all_gather_object(owners, mine)
        bad_rows = [[r, f] for r in range(self.world) for f in self.FIELDS + ("meta",) if _checksum(batch[f][r]) != owners[r][f]]
        out["batch_windows_without_a_start"] = int(batch["failed"].sum().item())
        oks = [None] * self.world
        dist_mod.all_gather_object(oks, not bad and not bad_rows)
        out.update(ok=all(oks), batch_rows_verified=self.windows, batch_bytes_per_rank=self.batch_bytes)
        if bad:
            out["bad_rank_record_pairs_seen_by_this_rank"] = bad
        if bad_rows:
            out["bad_batch_rows_seen_by_this_rank"] = bad_rows
        return out

    def close(self):
        self.wait()
        self.env.sync()
        self.ring.detach()


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
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(env.stream)
    for k in range(steps):
        env.step_random(seed=1, step=warmup + k)
    ev1.record(env.stream)
    env.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gpu_ms = ev0.elapsed_time(ev1) / steps
    # Per-kernel times: ONE kernel per pass, as the headline does for its scan.  With the launch-attached timers on every kernel
    # of the step at once, each kernel's start stamp is taken when its packet is picked up - while the kernel before it still
    # drains - so at small batches the per-kernel figures summed to MORE than the step (BENCH_r04: configs[1] 0.0093 + 0.0255 =
    # 0.0348 ms against a 0.0311 ms step).  One timer per pass leaves the other kernels untimed and the
    # step's rhythm as in the timed window.
    from racing_dreamer_amd import _lib as L
    # The launch-attached events read 1.2-1.6 us MORE per launch than rocprofv3's dispatch stamps of the same kernels
    # (profiles/r05_h_small_batch_stamps.log: 23.36 + 7.54 = 30.90 us of a 31.33 us step at 4 096 envs, where the events read
    # 24.6 + 9.1): 8 % of the step there, under 1 % at 65 536 envs.  The line carries the step between two stream events
    # (gpu_ms_per_step) beside the sum, and the excess per launch as stamp_overhead_ms_per_launch.
    kt, k0 = {}, warmup + steps
    n_pass = min(steps, 100)
    for kid in (L.K_RAYCAST, L.K_DYNAMICS) + ((L.K_PATCH,) if obs_type == "lidar_occupancy" else ()):
        env.reset_kernel_times()
        env.set_profiling(True, kernels=[kid])
        env.sync()
        for k in range(n_pass):
            env.step_random(seed=1, step=k0 + k)
        k0 += n_pass
        env.sync()
        env.set_profiling(False)
        v = env.kernel_times()[L.KERNEL_NAMES[kid]]
        if v["launches"]:
            kt[L.KERNEL_NAMES[kid]] = round(v["avg_ms"], 4)
    env.close()
    n_cars = envs * cars
    step_bytes = (STEP_BYTES_PER_CAR + (PATCH_BYTES_PER_CAR if obs_type == "lidar_occupancy" else 0)) * n_cars
    ms = dt / steps * 1e3
    out = {"workload": name, "envs": envs, "cars_per_env": cars, "track": track_name, "obs_type": obs_type,
           "steps": steps, "ms_per_step": ms, "env_steps_per_s": envs * steps / dt, "kernels_ms": kt,
           "gpu_ms_per_step": round(gpu_ms, 4),
           "kernels_sum_ms": round(sum(kt.values()), 4), "kernels_over_step": round(sum(kt.values()) / gpu_ms, 3),
           "stamp_overhead_ms_per_launch": round(max(sum(kt.values()) - gpu_ms, 0.0) / max(len(kt), 1), 4),
           "kernels_note": "each kernel timed in a pass of its own (launch-attached events on that kernel only); the events read "
                           "1.2-1.6 us more per launch than rocprofv3's stamps of the same dispatches (profiles/r05_h_small_batch_stamps.log), "
                           "so at small batches the sum exceeds gpu_ms_per_step (the step between two stream events) by that much per kernel",
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
    from racing_dreamer_amd import _lib as L
    kt, k0 = {}, warmup + steps
    for kid in (L.K_RAYCAST, L.K_DYNAMICS):               # one kernel per pass: see time_config
        lead.reset_kernel_times()
        lead.set_profiling(True, kernels=[kid])
        for k in range(min(steps, 40)):
            env.step_random(seed=1, step=k0 + k)
        k0 += min(steps, 40)
        env.sync()
        lead.set_profiling(False)
        v = lead.kernel_times()[L.KERNEL_NAMES[kid]]
        if v["launches"]:
            kt[L.KERNEL_NAMES[kid]] = round(v["avg_ms"], 4)
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


def time_track(track_name, envs, steps, warmup, settle):
    """The scan on another track at the headline's batch size (SURVEY.md 8d asked for the worst case, VERDICT r3 for the
    range across tracks): ms per step, the scan's launch-attached time and its fraction of the HBM roofline."""
    c = time_config(f"{envs} envs, {track_name}, 1080-beam lidar", track_name, envs, 1, "lidar", steps, warmup, settle=settle)
    return {"track": track_name, "envs": envs, "ms_per_step": c["ms_per_step"], "env_steps_per_s": c["env_steps_per_s"],
            "raycast_ms": c["kernels_ms"].get("rc_raycast_kernel"), "raycast_frac": c["roofline"].get("raycast_frac"),
            "dynamics_ms": c["kernels_ms"].get("rc_dynamics_kernel")}


def main():
    args = parse_args()
    t0_run = float(os.environ.get(T0_ENV, T_START))           # the outermost process's start: the time budget counts from there
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus, args.launch_timeout, __file__, sys.argv[1:], t0=T_START)          # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)", file=sys.stderr)
        sys.exit(2)
    distributed = world > 1 or (args.force_gather and "RANK" in os.environ)
    # from here on a signal, a deadline or a failed leg leaves through the guard: rank 0 prints the line it has (none yet)
    guard = LineGuard(rank, world if distributed else 1, args.leg_timeout, deadline=t0_run + args.time_budget)
    guard.install()
    if "RANK" in os.environ:
        die_with_parent()                                      # a launcher that is killed outright takes this rank with it
    try:
        run(args, guard, rank, local_rank, world, distributed, t0_run)
    except Exception as exc:                                   # noqa: BLE001
        # an exception OUTSIDE a guarded leg (the code between two legs): once there is a line, it is printed - with what failed -
        # and the exit code is the guard's (3 before the headline, 5 after it); before that an exception is an exception
        if guard.armed:
            import traceback
            traceback.print_exc()
            guard.finalise(guard.leg_name or "between legs", f"rank {rank}: {type(exc).__name__}: {exc}")
        raise


def run(args, guard, rank, local_rank, world, distributed, t0_run):
    # test hook (tests/test_distributed.py, no GPU there): the local stage is a stand-in, the rendezvous is real, and the
    # headline leg has no env to run on - what is exercised is the launcher, the guard and the exit codes
    fake_local = os.environ.get("RC_BENCH_FAKE_LOCAL") == "1" and distributed
    import torch
    import torch.distributed as dist

    dev = local_rank % max(torch.cuda.device_count(), 1)      # ranks > GPUs only in --backend gloo functional tests
    if not fake_local:
        torch.cuda.set_device(dev)
    comm_ranks = None
    store = None

    from racing_dreamer_amd import _lib as L
    from racing_dreamer_amd.batched_env import BatchedRaceEnv
    from racing_dreamer_amd.distributed import gather_link_model, shard_envs
    from racing_dreamer_amd.track_assets import load_track

    track_name = ["columbia", "austria", "barcelona"][rank % 3] if args.mixed_tracks else args.track
    track = load_track(track_name)
    shard = shard_envs(args.envs * world, rank, world)
    def make_env():
        e = BatchedRaceEnv(track, shard.num_envs, args.cars, obs_type=args.obs_type, action_repeat=args.repeat,
                           device=dev, first_env=shard.first_env, auto_reset=True, profiling=False)
        if args.raycast_variant is not None:
            e.set_raycast_variant(args.raycast_variant)      # (variants 0-6: lab kernels, built on this request)
        for kv in args.debug_knob:
            name, _, val = kv.partition("=")
            e.debug_set(name, int(val))
        e.reset(mode="random", seed=0)
        # every loop below works on the env's own stream: with torch's current stream another one, each step pays two cross-stream
        # event waits (BatchedRaceEnv._enter / _exit) - what the 25 us per step of "non-scan time" in round 4's fresh_reset leg were
        # (that leg ran before this call; `profiles/r05_e_fresh_window*.log` has the kernel timeline of the window)
        torch.cuda.set_stream(e.stream)
        return e

    env = None
    if not fake_local:
        env = make_env()
    # the first steps after a reset, timed on their own (N = 1): what `--settle 0` would put into the timed window
    fresh = None
    if not distributed and not args.no_configs:
        n_fresh = 20
        for k in range(150):                 # the chip at its working clocks first: what is measured is the poses, not the ramp
            env.step_random(seed=3, step=1000 + k)
        env.reset(mode="random", seed=0)
        env.reset_kernel_times()
        env.set_profiling(True, kernels=[L.K_RAYCAST])
        env.sync()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(env.stream)
        for k in range(n_fresh):
            env.step_random(seed=3, step=k)
        ev1.record(env.stream)
        env.sync()
        torch.cuda.synchronize()
        dtf = time.perf_counter() - t0
        env.set_profiling(False)
        fresh_scan = env.kernel_times()["rc_raycast_kernel"]["avg_ms"]
        # the same 20 steps once more for the dynamics kernel alone (one timer per pass, see time_config)
        env.reset(mode="random", seed=0)
        env.reset_kernel_times()
        env.set_profiling(True, kernels=[L.K_DYNAMICS])
        for k in range(n_fresh):
            env.step_random(seed=3, step=k)
        env.sync()
        env.set_profiling(False)
        fresh_dyn = env.kernel_times()["rc_dynamics_kernel"]["avg_ms"]
        gpu_ms = ev0.elapsed_time(ev1) / n_fresh
        fresh = {"steps": n_fresh, "ms_per_step": dtf / n_fresh * 1e3, "env_steps_per_s": shard.num_envs * n_fresh / dtf,
                 "raycast_ms": round(fresh_scan, 4), "dynamics_ms": round(fresh_dyn, 4),
                 "gpu_ms_per_step": round(gpu_ms, 4),
                 "host_overhead_ms_per_step": round(dtf / n_fresh * 1e3 - gpu_ms, 4),
                 "other_gpu_ms_per_step": round(gpu_ms - fresh_scan - fresh_dyn, 4),
                 "note": "the first 20 steps after reset(mode='random') - no settling of the poses; the GPU itself kept busy by 150 steps "
                         "before that reset - host-timed with the scan's launch timers on.  gpu_ms_per_step = the same window between two "
                         "events on the env's stream: what is left of ms_per_step beyond it is the host's share of a 4 ms window (first "
                         "launch, the closing synchronisation); other_gpu_ms_per_step = the window's GPU time beyond scan + dynamics: the "
                         "start of the first kernel on an idle GPU and the 5 us gap either side of every TIMED launch (the scan's timer "
                         "is on in this window; profiles/r05_e_fresh_window_timers.log)"}
        env.reset(mode="random", seed=0)
    gather_mode = "none" if (args.no_gather or not distributed) else args.gather
    via = args.gather_via
    abi_ranks = None
    total_envs = args.envs * world

    def contract_line(value, seconds_per_step, gather_txt):
        """The keys the driver's contract names, for the provisional line and for the measured one."""
        return {
            "metric": "env-steps/sec at 65 536 parallel envs, 1080-beam LiDAR, 1/2/4/8 MI355X",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": seconds_per_step * 1e3, "higher_is_better": True, "scaling": "weak",
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
                "parallelism": f"env-sharded x{world}", "gather": gather_mode,
            },
        }

    # ------------------------------------------------------------------ N > 1, stage 1: what this rank can do ALONE
    # A simulation-only window on this rank's own GPU - no collective, nothing a peer can hold up - gives rank 0 something to
    # print BEFORE the first collective this code has ever issued across devices: the guard is armed with a PROVISIONAL line
    # (`headline_pending`), and the rendezvous, the ring prefill and the timed headline window then run under deadlines of their
    # own.  If one of them hangs or raises, that line is printed with `aborted` and every rank exits 3 (VERDICT r5 #1b).
    step_no = 0
    local = None
    if distributed:
        if fake_local:
            dt_local = 1e-3 * args.steps
        else:
            for k in range(args.settle):
                env.step_random(seed=2, step=k)
            for k in range(args.warmup):
                env.step_random(seed=1, step=args.settle + k)
            env.sync()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(args.steps):
                env.step_random(seed=1, step=args.settle + args.warmup + k)
            env.sync()
            torch.cuda.synchronize()
            dt_local = time.perf_counter() - t0
            step_no = args.settle + args.warmup + args.steps
        local = {"steps": args.steps, "ms_per_step": dt_local / args.steps * 1e3,
                 "env_steps_per_s_this_rank": shard.num_envs * args.steps * args.repeat / dt_local,
                 "note": "rank 0's own simulation-only window (settle, warm-up, then the timed steps; no collective), taken before the "
                         "rendezvous"}
        prov = contract_line(total_envs * args.steps * args.repeat / dt_local, dt_local / args.steps,
                             "PROVISIONAL: rank 0's own simulation-only window x the number of ranks - printed only if the run "
                             "ends before the headline has been measured")
        prov["provisional"] = local
        guard.arm(prov if rank == 0 else None, pending=True)
        # ---------------------------------------------------------------- stage 2a: the rendezvous, under its own deadline
        with guard.leg("rendezvous", budget_s=min(args.leg_timeout, 120.0)):
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
            else:
                dist.init_process_group("gloo")
            # what the communicator itself says about its size: one contribution per rank, summed by the collective
            one = torch.ones(1, dtype=torch.int32, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(one)
            comm_ranks = int(one.item())
            try:
                store = dist.distributed_c10d._get_default_store()
            except Exception:                                      # noqa: BLE001 - without it a failed leg is still caught by the deadline
                store = None
            guard.set_store(store)
        # The env of stage 1 has done its work.  The headline runs on an env - and so on a stream - created AFTER the communicator:
        # with the env's stream older than the communicator's, the sharded loop ran 0.237 ms per step where it takes 0.207 the other
        # way round (one rank over RCCL, alternating on one box: profiles/r06_i_ab_stream_order_one_rank.txt; the scan itself is
        # the same - it is the exchange beside it that lands differently on the hardware queues).  The track's tables are shared
        # between the two handles (the new one is created before the old one goes), so this costs an arena and a reset.
        if not fake_local and not os.environ.get("RC_EXP_KEEP_ENV"):
            with guard.leg("second_env", budget_s=min(args.leg_timeout, 60.0)):
                old_env, env = env, make_env()
                old_env.close()
                step_no = 0
        if via == "abi":                                # the C-ABI's own communicator belongs to the env the headline runs on
            with guard.leg("abi_communicator", budget_s=min(args.leg_timeout, 60.0)):
                if args.backend != "nccl":
                    via = "torch"                       # RCCL wants one GPU per rank; the gloo functional tests share one
                else:
                    ids = [BatchedRaceEnv.comm_unique_id() if rank == 0 else None]
                    dist.broadcast_object_list(ids, src=0)
                    env.comm_init(ids[0], rank, world)
                    abi_ranks = env.comm_count()
        if rank == 0:
            print(f"bench.py: {world} ranks joined ({args.backend}), {guard.time_left():.0f} s of the time budget left; rank 0 alone: "
                  f"{local['env_steps_per_s_this_rank'] / 1e6:.1f} M env-steps/s", file=sys.stderr, flush=True)
    every = max(1, args.gather_every)
    if via == "p2p" and every != 1:
        raise SystemExit("--gather-via p2p sends one record per gather (--gather-every 1)")

    def p2p_close():
        # exported memory must not be freed while a peer still maps it: everybody unmaps, THEN everybody frees
        env.p2p_disconnect()
        dist.barrier()
        env.p2p_teardown()
        env._p2p_mode = None

    class Plain:
        """No collective: the simulation alone."""
        mode, via, bytes, in_place = "none", None, 0, True
        includes = "no collective, no uint16 rows, no summary copy: the simulation alone"

        def step(self, k, repeat=None):
            env.step_random(seed=1, step=k, repeat=repeat)

        def wait(self):
            pass

        def close(self):
            pass

    def make_collector(mode):
        if mode == "none":
            if getattr(env, "compact", None) is not None:
                env.disable_compact()
            return Plain()
        if mode == "sharded":
            return ShardedCollector(env, dist, rank)
        if via == "p2p":                        # buffers and mappings are made once; later legs only switch the payload
            first = getattr(env, "_p2p_mode", None) is None
            blob = env.p2p_setup(mode, rank, world)
            if first:
                blobs = [None] * world
                dist.all_gather_object(blobs, blob)
                env.p2p_connect(blobs)
        return Gatherer(env, mode, every, via, dist)

    # the rollout loop works on the env's own stream (no cross-stream event waits between the step's kernels and
    # the collective's dependency on them); `barrier()` synchronises the whole device
    if env is not None:
        torch.cuda.set_stream(env.stream)

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def finish(g):
        g.wait()
        env.sync()

    def max_over_ranks(seconds):
        if not distributed:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(g, k0, n, repeat=None):
        """n steps of collector g between barriers; seconds = MAX over ranks."""
        barrier()
        t0 = time.perf_counter()
        for k in range(n):
            g.step(k0 + k, repeat=repeat)
        t1 = time.perf_counter()
        finish(g)
        t2 = time.perf_counter()
        barrier()
        t3 = time.perf_counter()
        if os.environ.get("RC_BENCH_TRACE_TIMED"):           # where a timed region's wall time goes (analysis; tools/fixed_cost.sh)
            print(f"bench.py: rank {rank}: timed[{getattr(g, 'mode', '?')}, {n} steps] enqueue {(t1 - t0) * 1e3:.3f} ms, "
                  f"drain {(t2 - t1) * 1e3:.3f} ms, closing barrier {(t3 - t2) * 1e3:.3f} ms", file=sys.stderr, flush=True)
        return max_over_ranks(t3 - t0)

    # ------------------------------------------------------------------ the headline leg (a failure here fails the run)
    # The synthetic data is a random-action rollout that HAS SETTLED, and the chip is at its working clocks: `--settle` untimed
    # steps of the very loop that is timed, right in front of the warm-up.  Two things ride on them: the cars
    # leave the post-reset poses (long rays), and the GPU leaves the power state an idle spell puts it into - after 0.5 s
    # without work (a ring allocation, a collective's set-up) the scan runs up to 30 % slower for about 25 ms
    # (profiles/r04_l_idle_ramp.txt).  Every secondary leg does the same (`preheat`).
    def preheat(g, k0, n=None):
        n = args.settle if n is None else n
        for k in range(n):
            g.step(k0 + k)
        return n

    # N > 1, stage 2b: ring prefill, settling, warm-up and the timed window - the first collectives on the data path - under a
    # deadline; the guard still holds the provisional line.  N = 1: the guard is not armed, an exception here is an exception.
    with guard.leg("headline"):
        if fake_local:
            raise RuntimeError("RC_BENCH_FAKE_LOCAL: there is no env to measure a headline on")
        gather = make_collector(gather_mode)
        if gather_mode == "sharded":
            step_no += gather.prefill(step_no)
        step_no += preheat(gather, step_no)
        for k in range(args.warmup):
            gather.step(step_no + k)
        step_no += args.warmup
        finish(gather)
        env.reset_kernel_times()
        # start / stop timestamps attached to every launch of the DOMINANT kernel (the scan) on the stream it runs on;
        # timing the two small kernels as well would cost the timed region 6 us per step, so they get a pass of their
        # own after it
        env.set_profiling(True, kernels=[L.K_RAYCAST])
        dt = timed(gather, step_no, args.steps)
        step_no += args.steps
        env.set_profiling(False)
        ktimes = env.kernel_times()
        scan_symbol = env.scan_kernel_name()
    # a long steady-state figure of the SAME env and loop right behind the driver's window (a 20-step window is
    # 3.7 ms, too short for an outside observer; this one is >= 2 000 steps, ~0.4 s, and contains its share of order re-sorts)
    steady = steady_long = None
    if not distributed and not args.no_configs:
        def window(n):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ev0.record(env.stream)
            for k in range(n):
                gather.step(step_no + k)
            ev1.record(env.stream)
            finish(gather)
            torch.cuda.synchronize()
            dts = time.perf_counter() - t0
            return {"steps": n, "seconds": dts, "ms_per_step": dts / n * 1e3, "env_steps_per_s": shard.num_envs * n * args.repeat / dts,
                    "gpu_ms_per_step": ev0.elapsed_time(ev1) / n}
        n_steady = max(2000, 10 * args.steps)
        steady = window(n_steady)
        step_no += n_steady
        steady["note"] = ("the headline's env and loop, continued for a window an observer can see (no timers on); host-timed like the "
                          "headline, gpu_ms_per_step between two events on the env's stream")
        # ... and one an OUTSIDE observer can see: the driver's sampler polls the GPU's busy figure every ~5 s and
        # has never caught this benchmark at work (the 2 000-step window is 0.36 s).  The same loop for --observable-seconds of
        # wall time, sized from the rate just measured: two or more polls fall inside it.
        if args.observable_seconds > 0 and guard.time_left() > 4 * args.observable_seconds + 120:
            n_long = int(args.observable_seconds / (steady["ms_per_step"] * 1e-3)) + 1
            steady_long = window(n_long)
            step_no += n_long
            steady_long["note"] = (f"the same loop for about {args.observable_seconds:.0f} s of wall time without a pause (no timers, no host "
                                   f"synchronisation inside): long enough for a sampler that polls the GPU every 5 s to see it busy twice")
    # the other kernels of the step: a short untimed pass with all timers on
    env.reset_kernel_times()
    env.set_profiling(True, kernels=[L.K_PATCH, L.K_DYNAMICS])
    n_pass = min(args.steps, 50)
    for k in range(n_pass):
        gather.step(step_no + k)
    step_no += n_pass
    finish(gather)
    env.set_profiling(False)
    for name, v in env.kernel_times().items():
        if name != "rc_raycast_kernel":
            ktimes[name] = v

    value = total_envs * args.steps * args.repeat / dt
    n_cars = shard.num_envs * args.cars
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
        gather_txt = "NO trajectory exchange (--gather none)"
    elif gather_mode == "sharded":
        gather_txt = ("`value` is timed with THIS payload on the links - trajectory store sharded: " + gather.includes
                      + "; the per-step gathers of whole records are the `full-u16` / `full` legs of gather_modes (a record per "
                        "sub-step) and of gather_modes_repeat_4 (a record per agent step of 4 sub-steps: the reference's cadence); "
                        "BASELINE configs[4]'s track mix is the leg configs4_track_mix")
    else:
        how = {"torch": "RCCL all-gather through torch.distributed", "abi": "RCCL all-gather through rc_gather_trajectory",
               "p2p": "direct peer copies through rc_gather_trajectory_p2p (hipIpc, one copy stream per peer)"}[via]
        gather_txt = (f"every step's trajectory record gathered on every rank as `{gather_mode}` ({gather.bytes} B per GPU "
                      f"per step, {'read in place from a double-buffered source' if gather.in_place else 'from staging copies'}, "
                      f"one gather per {every} step{'s' if every > 1 else ''}, {how}, overlapped with the "
                      f"following step; the gathered buffer is overwritten two gathers later - no consumer in this benchmark)")
    out = contract_line(value, dt / args.steps, gather_txt)
    out["config"].update({
        "gather_via": (gather.via if distributed else None), "gather_detail": (gather.includes if distributed else None),
        # the size of the job as the communicator reports it (sum over ranks of 1 through the backend's own
        # all-reduce; ncclCommCount of the C-ABI's communicator when that transport is used)
        "rccl_ranks": comm_ranks if (distributed and args.backend == "nccl") else None,
        "comm_backend": args.backend if distributed else None, "comm_ranks": comm_ranks, "abi_comm_ranks": abi_ranks,
        "time_budget_s": args.time_budget,
    })
    out.update({
        "roofline": {
            "bound": "hbm", "kernel": scan_symbol, "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
            "algorithmic_bytes_per_launch": RAYCAST_BYTES_PER_CAR * n_cars,
            "avg_launch_ms": ray["avg_ms"], "launches": ray["launches"],
            "rays_per_s": n_cars * 1080 / ray_s if ray_s > 0 else 0.0,
            # what actually bounds the scan: wave-level VALU instructions per launch from the PMC profile
            # against the issue slots of 1 024 SIMDs over the duration measured here
            "valu_wave_insts_per_launch": valu,
            "valu_insts_per_simd_per_us": (valu / SIMDS / (ray_s * 1e6)) if (valu and ray_s > 0) else None,
            "issue_frac": (valu * VALU_CYCLES_FULL_RATE / (SIMDS * ray_s * CLOCK_GHZ * 1e9)) if (valu and ray_s > 0) else None,
            "issue_frac_model": f"VALU wave-instructions x {VALU_CYCLES_FULL_RATE} cycles (issue cost of a full-rate instruction, "
                                f"tools/ubench/valu_issue4.hip; half-rate forms cost 4.3, so this is a lower bound) / ({SIMDS} SIMDs x "
                                f"kernel cycles at {CLOCK_GHZ} GHz)",
            "note": "compulsory HBM traffic is ~4.3 KB per car-scan, so the scan is bound by the instruction stream "
                    "of the grid traversal, not by HBM (SURVEY.md 8d); the >= 40 % HBM target is NOT met; DESIGN.md 4.2",
        },
        "kernels_ms": {k: round(v["avg_ms"], 4) for k, v in ktimes.items() if v["launches"]},
        "kernels_sum_ms": round(sum(v["avg_ms"] for v in ktimes.values() if v["launches"]), 4),
    })
    if steady is not None:
        out["steady"] = steady
    if steady_long is not None:
        out["steady_long"] = steady_long
    if local is not None:
        out["rank0_alone_before_the_rendezvous"] = local
        # which number answers the north star: the north star names "an RCCL all-gather over xGMI only for
        # the trajectory-buffer concat"; `value` is timed with the concat at the granularity its consumer reads it
        out["north_star_answer"] = (
            "`value` = all ranks' env-steps/s with the trajectory store SHARDED (every step's 76 B/car summary all-gathered, one 50 x 50 "
            "training batch all-gathered every 10th step; whole records stay in the rank that produced them) - the scaling figure to "
            "compare with N = 1.  The literal concat of WHOLE per-step records on every rank is `gather_modes.full` / `full-u16` (a "
            "record per sub-step) and `gather_modes_repeat_4` (a record per agent step of 4 sub-steps, the reference's cadence, "
            "dreamer/wrappers.py:107-116,213-219): measured in this run beside their xGMI link bounds, link-bound by an order of "
            "magnitude at this simulation rate (DESIGN.md 6).  BASELINE configs[4] (rank r on [columbia, austria, barcelona][r mod 3]) "
            "is `configs4_track_mix`.")
    if fresh is not None:
        out["fresh_reset"] = fresh
    if distributed:
        # cpu_baseline is timed on rank 0 of the N = 1 run only (the driver's contract): the N > 1 line points at it
        cb = None
        cbp = os.path.join(ROOT, "profiles", "cpu_baseline_n1.json")
        if os.path.exists(cbp):
            with open(cbp) as f:
                cb = json.load(f)
        out["cpu_baseline"] = {"measured_in_this_run": False,
                               "see": "the N = 1 line of this bench (`python bench.py`): the CPU oracle is timed there, on rank 0, "
                                      "in the same run as the GPU figure; quoted below from the committed N = 1 line",
                               "quoted": cb}
    sizes = None
    if distributed:
        sizes = {"full": int(env.slab.numel()), "summary": int(env.summary_slab.numel()),
                 "full-u16": int(env._lib.rc_compact_bytes(env._cfg)), "none": 0}
        out["gather_modes"] = {gather_mode: dict(getattr(gather, "model", None) or gather_link_model(0, world),
                                                 ms_per_step=dt / args.steps * 1e3, env_steps_per_s=value, steps=args.steps,
                                                 headline=True, includes=gather.includes)}
        if gather_mode == "sharded":
            out["gather_modes"]["sharded"].update(batches_in_timed_window=gather.batches, batch_every=gather.batch_every,
                                                  batch_windows=gather.windows, batch_length=gather.length)
    # N > 1, stage 3: the measured headline replaces the provisional line
    if distributed:
        guard.promote(out if rank == 0 else None)
    else:
        guard.arm(out if rank == 0 else None, store)
    if rank == 0 and distributed:
        print("bench.py headline (the one JSON line on stdout follows at the end of the run): "
              + json.dumps({k: out[k] for k in ("value", "unit", "n_gpus", "ms_per_step")} | {"gather": gather_mode}),
              file=sys.stderr, flush=True)

    # the headline payload's self-check: the only thing after the timed region that can fail the run
    checks = {}
    if distributed and gather_mode != "none" and not args.no_gather_check:
        with guard.leg("headline_check"):        # (an exception or a hang in the check itself ends the run with the line; a MISMATCH fails it)
            checks[gather_mode] = gather.check(step_no, dist)
            step_no += 8
            if rank == 0:
                out["gather_check"] = dict(ok=checks[gather_mode]["ok"] is not False, payloads=checks)
    with guard.leg("close_headline"):
        gather.close()
        gather = None
    if checks and checks[gather_mode]["ok"] is False:
        guard.emit()
        print(f"bench.py: rank {rank}: a gathered record differs from what its sender sent: {checks}", file=sys.stderr, flush=True)
        guard.done()                   # (a failed self-check of the headline payload: exit 4)
        if distributed:
            dist.barrier()
        sys.exit(EXIT_CHECK_MISMATCH)

    # ------------------------------------------------------------------ secondary legs: each under the guard, each started only
    # if the time budget has room for it (`guard.go`: rank 0 decides, every rank takes the same branch), in the order of what the
    # line can least do without: N = 1 - the CPU baseline (the contract names it), then the other configurations; N > 1 - BASELINE
    # configs[4] itself, the steady state of the headline payload, the whole-record concat at the reference's cadence, the
    # per-sub-step payloads, and the simulation alone last.
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        with guard.leg("cpu_baseline"):
            from oracle import cpu_baseline as cb
            out["cpu_baseline"] = cpu_baseline(track, args.cars, args.obs_type, args.repeat, args.cpu_envs)
            out["cpu_baseline"]["single_env"] = cb.run_single_env()      # BASELINE.json configs[0]: the B = 1 CPU step()
            if not args.no_numpy_baseline:
                out["cpu_baseline"]["numpy_batch"] = cb.run_numpy_batch(track, n_envs=args.numpy_envs)

    # BASELINE.json configs[4] inside the plain `--gpus N` run: rank r on [columbia, austria, barcelona][r mod 3],
    # the headline's payload (the sharded store) over the mix.  Each rank builds a SECOND env on its track of the mix (the
    # headline's env stays as it is); ranks on different tracks run steps of different length, so the closing barrier waits for
    # the slowest track - which is the point of the configuration.
    if distributed and not args.mixed_tracks and not args.no_gather_modes and guard.go("configs4_track_mix", 40):
        with guard.leg("configs4_track_mix"):
            mix = ["columbia", "austria", "barcelona"]
            mine = mix[rank % 3]
            env_mix = BatchedRaceEnv(load_track(mine), shard.num_envs, args.cars, obs_type=args.obs_type, action_repeat=args.repeat,
                                     device=dev, first_env=shard.first_env, auto_reset=True, profiling=False)
            env_mix.reset(mode="random", seed=0)
            head_env, env = env, env_mix              # (the collectors close over `env`)
            try:
                torch.cuda.set_stream(env.stream)
                g = make_collector(gather_mode)
                k = 0
                if gather_mode == "sharded":
                    k += g.prefill(k)
                k += preheat(g, k)
                finish(g)
                n_leg = max(args.steps, 2 * BATCH_EVERY)
                t = timed(g, k, n_leg)
                g.close()
            finally:
                env = head_env
                torch.cuda.set_stream(env.stream)
                env_mix.close()
                env_mix._bench_ring = None          # (its ring goes back to the allocator)
            tracks_by_rank = [None] * world
            dist.all_gather_object(tracks_by_rank, mine)
            if rank == 0:
                out["configs4_track_mix"] = {
                    "workload": f"BASELINE.json configs[4]: {total_envs} envs over {world} ranks, rank r on [columbia, austria, barcelona][r mod 3], "
                                f"payload = the headline's ({gather_mode})",
                    "tracks_by_rank": tracks_by_rank, "steps": n_leg, "ms_per_step": t / n_leg * 1e3,
                    "env_steps_per_s": total_envs * n_leg * args.repeat / t, "gather": gather_mode}

    # the headline payload over a window ten times as long: a 20-step window (3 - 5 ms) carries the pipeline's start and
    # drain and the closing barrier at full weight; this is the steady state next to it
    def leg_weight(mode, n_steps):
        """bytes a payload leg moves per rank: what its duration scales with when the links (or a host-staged test backend) bound it"""
        per_step = {"sharded": sizes["summary"], "batch": 0, "none": 0}.get(mode, sizes.get(mode, 0)) if sizes else 0
        return float(max(per_step, 1 << 20)) * n_steps

    n_long = max(10 * args.steps, 200)
    if distributed and gather_mode != "none" and guard.go("steady_state", 15, kind="gather", weight=leg_weight(gather_mode, n_long + args.settle)):
        with guard.leg("steady_state", kind="gather", weight=leg_weight(gather_mode, n_long + args.settle)):
            g = make_collector(gather_mode)
            if gather_mode == "sharded":
                step_no += g.prefill(step_no)
            step_no += preheat(g, step_no)
            finish(g)
            t = timed(g, step_no, n_long)
            step_no += n_long
            g.close()
            if rank == 0:
                out["gather_modes"][gather_mode]["steady_state"] = {"steps": n_long, "ms_per_step": t / n_long * 1e3,
                                                                    "env_steps_per_s": total_envs * n_long * args.repeat / t}

    # The whole-record gathers at the REFERENCE's cadence: Collect records one transition per AGENT step
    # (dreamer/wrappers.py:107-116 ActionRepeat inside, :213-219 Collect outside), the reference runs action_repeat 4
    # (dreamer/dream.py:55) - so one record crosses the links per 4 sub-steps, the scan runs once per record, and the link
    # bound is set against a step of four dynamics kernels + one scan.  (The legs below gather a record per sub-step.)
    if distributed and not args.no_gather_modes:
        out["gather_modes_repeat_4"] = {} if rank == 0 else None
        for m in ("full-u16", "full"):
            n_leg = max(args.steps // 4, 5)
            wgt = leg_weight(m, n_leg + 40 + 4)
            if not guard.go(m + "@repeat4", 12, kind="gather", weight=wgt):
                continue
            with guard.leg(m + "@repeat4", kind="gather", weight=wgt):
                g = make_collector(m)
                step_no += preheat(g, step_no, 40)
                finish(g)
                t = timed(g, step_no, n_leg, repeat=4)
                step_no += n_leg
                e = dict(getattr(g, "model", None) or gather_link_model(sizes.get(m, 0), world))
                e["link_bound_ms_per_agent_step"] = e.pop("link_bound_ms_per_step")
                e["bytes_per_gpu_per_agent_step"] = e.pop("bytes_per_gpu_per_step")
                e["inbound_bytes_per_gpu_per_agent_step"] = e.pop("inbound_bytes_per_gpu_per_step")
                e.update(action_repeat=4, ms_per_agent_step=t / n_leg * 1e3, agent_steps_per_s=total_envs * n_leg / t,
                         env_steps_per_s=total_envs * n_leg * 4 / t, agent_steps=n_leg, includes=g.includes,
                         cadence="one record per agent step of 4 sub-steps, the scan once per record (dreamer/wrappers.py:107-116,213-219; dream.py:55)")
                if not args.no_gather_check:
                    c = g.check(step_no, dist)
                    step_no += 8
                    e["check"] = c
                    checks[m + "@repeat4"] = c
                g.close()
                if rank == 0:
                    out["gather_modes_repeat_4"][m] = e
                    out["gather_check"] = dict(ok=all(c["ok"] is not False for c in checks.values()), payloads=checks)

    # N > 1: the same loop with each of the other payloads, short legs with the same barriers (every rank runs the same
    # sequence): what the headline's choice costs or saves, measured rather than argued, each next to its link bound
    # (xGMI full mesh, 7 links x 76.8 GB/s inbound per GPU).  Each leg sets the env up for its own payload only
    # (`includes` says what its step carried); `none` is the pure simulation rate; `batch` = sharded without the per-step
    # summary.  Cheapest and most informative first: a leg that fails ends the run with the legs before it in the line.
    if distributed and not args.no_gather_modes:
        order = [m for m in ("none", "batch", "sharded", "summary", "full-u16", "full") if m != gather_mode]
        for m in order:
            wgt = leg_weight(m, max(args.steps // 4, 5 if m not in ("batch", "sharded") else 2 * BATCH_EVERY) + args.settle + 4)
            if not guard.go(m, 12, kind="gather", weight=wgt):
                continue
            with guard.leg(m, kind="gather", weight=wgt):
                if m == "batch":
                    g = ShardedCollector(env, dist, rank, summary=False)
                else:
                    g = make_collector(m)
                n_leg = max(args.steps // 4, 5)
                if m in ("batch", "sharded"):
                    step_no += g.prefill(step_no)
                    n_leg = max(n_leg, 2 * g.batch_every)        # (a leg without a batch in it would not be this payload)
                step_no += preheat(g, step_no)
                finish(g)
                t = timed(g, step_no, n_leg)
                step_no += n_leg
                e = dict(getattr(g, "model", None) or gather_link_model(sizes.get(m, 0), world))
                e.update(ms_per_step=t / n_leg * 1e3, env_steps_per_s=total_envs * n_leg * args.repeat / t, steps=n_leg,
                         includes=g.includes)
                if m not in ("none",) and not args.no_gather_check:
                    c = g.check(step_no, dist)
                    step_no += 8
                    e["check"] = c
                    checks[m] = c
                g.close()
                if rank == 0:
                    out["gather_modes"][m] = e
                    out["gather_check"] = dict(ok=all(c["ok"] is not False for c in checks.values()), payloads=checks)
    if getattr(env, "_p2p_mode", None) is not None:
        with guard.leg("p2p_close"):
            p2p_close()

    # the reference's own setting, action_repeat 4 with the scan once per agent step (dreamer/dream.py:55)
    if guard.go("action_repeat_4", 8):
        with guard.leg("action_repeat_4"):
            r4_steps = max(args.steps // 4, 5)
            g = make_collector("none")
            step_no += preheat(g, step_no, 40)
            dt4 = timed(g, step_no, r4_steps, repeat=4)
            step_no += r4_steps
            if rank == 0:
                out["action_repeat_4"] = {"env_steps_per_s": total_envs * r4_steps * 4 / dt4,
                                          "agent_steps_per_s": total_envs * r4_steps / dt4, "steps": r4_steps,
                                          "note": "same envs with action_repeat 4, LiDAR once per agent step (dreamer/dream.py:55), no exchange"}

    # secondary figure: cars driven along the track at speed by the reference's follow-the-gap law (its other prefill
    # policy, dreamer/dream.py:211-216) instead of crawling under random actions: single-GPU runs only
    if world == 1 and not args.no_cpu_baseline_ftg and guard.go("follow_the_gap", 10):
        with guard.leg("follow_the_gap"):
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
            out["follow_the_gap"] = {
                "env_steps_per_s": args.envs * n_ftg * args.repeat / dtf, "steps": n_ftg,
                "raycast_ms": round(kt["rc_raycast_kernel"]["avg_ms"], 4),
                "agent_kernel_ms": round(kt["rc_ftg_kernel"]["avg_ms"], 4),
                "mean_range_m": float(env.views["lidar"].float().mean().item()),
                "mean_speed_m_s": float(env.views["speed"].float().mean().item()),
                "mean_range_m_random_actions": mean_range_random,
                "note": "same envs driven by rc_follow_the_gap_reference - the law of the reference's own follow-the-gap node "
                        "(ros_agent/agents/follow_the_gap/src/agent.py:128-234) as a device agent - after 150 settling steps "
                        "(cars at 4 m/s instead of crawling under random actions); includes the agent's kernel"}
    env.close()
    if rank == 0 and world == 1 and not args.no_configs:
        # BASELINE.json configs[1..3], each a few ms of GPU time (configs[0] is the CPU plumbing case, configs[4]
        # the 8-GPU run: `--gpus 8 --mixed-tracks`)
        cfgs = [("configs[1]: 4 096 envs, columbia, 1080-beam lidar", "columbia", 4096, 1, "lidar", 400, 40, "random"),
                ("configs[2]: 65 536 envs, austria, obs_type=lidar_occupancy (64x64 render)", "austria", 65536, 1,
                 "lidar_occupancy", 100, 10, "random"),
                ("configs[3]: 32 768 envs x 2 cars, treitlstrasse_v2, inter-car raycast + collision",
                 "treitlstrasse_v2", 32768, 2, "lidar", 100, 10, "random_ball")]
        if guard.go("configs", 40):
            with guard.leg("configs"):
                out["configs"] = [time_config(*c, settle=args.settle) for c in cfgs]
                out["configs"].append(time_mixed_tracks(("columbia", "austria", "barcelona"), 65536, 100, 10, settle=args.settle))
        # obs_type lidar_occupancy_reference: the patch computed exactly as the reference's OccupancyMapObs.step does (binary64 spline
        # rotation + Pillow's integer resize, racecar_patch_exact.h) - opt-in, ~600 x the fast sampler's cost: a few steps suffice
        if guard.go("exact_render", 15):
            with guard.leg("exact_render"):
                n_x = min(args.envs, 16384)
                ex = BatchedRaceEnv(track, n_x, 1, obs_type="lidar_occupancy_reference", device=dev, auto_reset=True)
                ex.reset(mode="random", seed=0)
                torch.cuda.set_stream(ex.stream)
                for k in range(3):
                    ex.step_random(seed=2, step=k)
                ex.sync()
                ex.reset_kernel_times()
                ex.set_profiling(True, kernels=[L.K_PATCH])
                t0 = time.perf_counter()
                for k in range(4):
                    ex.step_random(seed=1, step=3 + k)
                ex.sync()
                dtx = time.perf_counter() - t0
                ex.set_profiling(False)
                kx = ex.kernel_times()["rc_patch_kernel"]["avg_ms"]
                ex.close()
                out["exact_render"] = {
                    "workload": f"{n_x} envs, {track_name}, obs_type=lidar_occupancy_reference (dreamer/wrappers.py:396-406 restated to the binary64 operation)",
                    "steps": 4, "ms_per_step": dtx / 4 * 1e3, "env_steps_per_s": n_x * 4 / dtx, "render_ms": round(kx, 3),
                    "render_us_per_car": round(kx * 1e3 / n_x, 3),
                    "note": "identical to the reference's own patches (G6); the reference's wrapper takes 5.6 ms per call on one core (SURVEY.md 6)"}
        # the scan across tracks at the headline's batch size: small tables (columbia) to the largest (gbr: 506 MB of
        # first-trip table) - the range the headline's one track sits in
        if guard.go("tracks", 40):
            with guard.leg("tracks"):
                out["tracks"] = [time_track(t, args.envs, 60, 10, args.settle) for t in ("columbia", "barcelona", "gbr") if t != track_name]
                out["tracks"].insert(0, {"track": track_name, "envs": args.envs, "ms_per_step": out["ms_per_step"],
                                         "env_steps_per_s": value, "raycast_ms": round(ray["avg_ms"], 4),
                                         "raycast_frac": achieved / HBM_PEAK_GBS, "headline": True})
                rm = [t["raycast_ms"] for t in out["tracks"] if t.get("raycast_ms")]
                out["roofline"]["raycast_ms_range_over_tracks"] = [min(rm), max(rm)]
                out["roofline"]["frac_range_over_tracks"] = [RAYCAST_BYTES_PER_CAR * args.envs / (max(rm) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                             RAYCAST_BYTES_PER_CAR * args.envs / (min(rm) * 1e-3) / 1e9 / HBM_PEAK_GBS]
    if rank == 0:
        out["seconds_since_start"] = round(time.time() - t0_run, 1)
    guard.emit()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    guard.done()
    if any(c["ok"] is False for c in checks.values()):
        print(f"bench.py: rank {rank}: a gathered record of a secondary payload differs from what its sender sent: {checks}", file=sys.stderr)
        sys.exit(EXIT_LEG_LOST)
    sys.exit(guard.exit_code())           # 0, or 5 when a leg after the headline failed (N = 1: recorded in `leg_errors`, the run went on)


if __name__ == "__main__":
    main()
