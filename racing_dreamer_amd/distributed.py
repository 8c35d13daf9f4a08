"""Multi-GPU: env-index sharding and the trajectory all-gather.

Envs are independent, so the job is sharded by contiguous env-index blocks, one process per GPU,
with no data-path collective: rank r owns envs [r * per_rank, (r + 1) * per_rank) and passes
``first_env`` to the HIP library so the per-env Philox streams are keyed by the *global* env id
(results do not depend on the number of ranks).  The only exchange is the all-gather that
concatenates every rank's trajectory slab (the per-step record `Collect.step` builds in the
reference, dreamer/wrappers.py:213-219) - issued through ``torch.distributed`` (backend "nccl" is
RCCL over xGMI on ROCm; "gloo" on CPU for the tests) on staging copies, several steps per collective, so that it
overlaps with the following steps' kernels, which are VALU-bound and leave HBM and the links idle.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class Shard:
    rank: int
    world_size: int
    first_env: int
    num_envs: int
    total_envs: int


def shard_envs(total_envs: int, rank: int, world_size: int) -> Shard:
    """Contiguous block partition of `total_envs`; the first `total_envs % world_size` ranks get one more."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    if total_envs < world_size:
        raise ValueError(f"cannot shard {total_envs} envs over {world_size} ranks")
    base, rem = divmod(total_envs, world_size)
    n = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return Shard(rank, world_size, first, n, total_envs)


class TrajectoryGather:
    """All-gather of equally sized per-rank slabs, overlapped with the steps that follow.

    ``launch(slab)`` hands over one record.  With ``stage=True`` (default) it is first snapshot into a staging buffer,
    so the producer may overwrite it at once, and every ``every``-th call starts ONE asynchronous collective over the
    ``every`` snapshots taken since the last one (same bytes on the links, 1 / every of the launches); ``depth + 1``
    staging buffers take turns: the buffer being filled for collective k + 1 was the source of collective k - depth, which
    `_issue` retired before it issued collective k (two buffers were enough for depth 1 only: with depth 2 the copies for
    collective k + 1 went into the source collective k - 1 could still be reading - ADVICE r4).  With
    ``stage=False`` (``every`` must be 1) the collective reads ``slab`` IN PLACE - no copy - and the caller alternates
    between source buffers (`BatchedRaceEnv.rotate_compact`, a pair of arenas, `TrajectoryRing`).
    ``depth`` = collectives allowed in flight: before collective k is issued, the caller's stream is put behind
    collective k - depth.  With the default 1 collective k - 1 has finished before the step that follows launch(k) starts
    - what a PAIR of source buffers needs; a source that is not rewritten for ``depth + 1`` steps (a ring slot) may use
    more, and then a collective that runs late does not stall the steps behind it.  ``depth + 1`` gathered buffers take
    turns.  ``consumer(view)``, if given, receives every completed batch (``[world, slab_bytes]`` for ``every == 1``, else
    ``[world, n_snapshots, slab_bytes]``) before its buffer is reused.
    ``wait()`` flushes a partial batch, waits for everything, and returns the gathered buffer of the last collective;
    ``recent(back)`` the view of the collective `back` issues ago (0 = last; valid for back <= depth).
    """

    def __init__(self, slab_like: torch.Tensor, group: Optional[dist.ProcessGroup] = None, every: int = 1,
                 stage: bool = True, consumer=None, depth: int = 1):
        if every < 1:
            raise ValueError("every must be >= 1")
        if depth < 1:
            raise ValueError("depth must be >= 1")
        if not stage and every != 1:
            raise ValueError("in-place gathers (stage=False) carry one record per collective")
        self.group = group
        self.every = int(every)
        self.stage = bool(stage)
        self.depth = int(depth)
        self.consumer = consumer
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        flat = slab_like.reshape(-1)
        self.slab_numel = flat.numel()
        self.staging = [torch.empty(self.every * flat.numel(), dtype=flat.dtype, device=flat.device)
                        for _ in range(self.depth + 1)] if stage else None
        self.gathered_bufs = [torch.empty(self.world * self.every * flat.numel(), dtype=flat.dtype, device=flat.device)
                              for _ in range(self.depth + 1)]
        self._pending = []       # [(work or None, view, source)] oldest first: issued, not yet waited for
        self._views = []         # views of the last depth + 1 collectives, newest last
        self._cur = 0            # staging buffer being filled
        self._k = 0              # snapshots in it
        self._g = 0              # gathered buffer the NEXT collective writes

    @property
    def gathered(self) -> torch.Tensor:
        return self.gathered_bufs[(self._g - 1) % len(self.gathered_bufs)]      # the buffer of the most recently issued collective

    def launch(self, slab: torch.Tensor) -> None:
        n = self.slab_numel
        if not self.stage:
            self._issue(slab.reshape(-1), 1)
            return
        self.staging[self._cur][self._k * n:(self._k + 1) * n].copy_(slab.reshape(-1), non_blocking=True)
        self._k += 1
        if self._k == self.every:
            k = self._k
            src = self.staging[self._cur][:k * n]
            self._cur = (self._cur + 1) % len(self.staging)
            self._k = 0
            self._issue(src, k)

    def _retire(self, keep: int) -> None:
        """Put the caller's stream behind every pending collective but the newest `keep` (work.wait() is a stream wait for
        the nccl backend, a host wait for gloo)."""
        while len(self._pending) > keep:
            work, view, _src = self._pending.pop(0)
            if work is not None:
                work.wait()
            if self.consumer is not None:
                self.consumer(view)

    def _issue(self, src: torch.Tensor, k: int) -> None:
        self._retire(self.depth - 1)         # collective k - depth: its source and its gathered buffer are free again
        n = self.slab_numel
        dst = self.gathered_bufs[self._g][:self.world * k * n]
        self._g = (self._g + 1) % len(self.gathered_bufs)
        view = dst.view(self.world, -1) if self.every == 1 else dst.view(self.world, k, -1)
        self._views = (self._views + [view])[-(self.depth + 1):]
        if dist.get_backend(self.group) == "gloo" and src.is_cuda:
            # gloo has no device all-gather: stage through the host (functional tests only)
            host = src.cpu()
            out = torch.empty(self.world * host.numel(), dtype=host.dtype)
            dist.all_gather_into_tensor(out, host, group=self.group)
            dst.copy_(out)
            self._pending.append((None, view, src))
            return
        self._pending.append((dist.all_gather_into_tensor(dst, src, group=self.group, async_op=True), view, src))

    def pending_sources(self) -> List[torch.Tensor]:
        """The source buffers of the collectives issued and not yet retired (what nobody may write to now)."""
        return [src for _w, _v, src in self._pending]

    def recent(self, back: int = 0) -> torch.Tensor:
        if not 0 <= back < len(self._views):
            raise IndexError(f"only the last {len(self._views)} collectives are kept")
        return self._views[-1 - back]

    def wait(self) -> torch.Tensor:
        if self.stage and self._k:           # a partial batch (every rank holds the same number of snapshots)
            k, n = self._k, self.slab_numel
            src = self.staging[self._cur][:k * n]
            self._cur = (self._cur + 1) % len(self.staging)
            self._k = 0
            self._issue(src, k)
        self._retire(0)
        if not self._views:
            g = self.gathered[:self.world * self.every * self.slab_numel]
            return g.view(self.world, -1) if self.every == 1 else g.view(self.world, self.every, -1)
        return self._views[-1]


SUMMARY_FIELDS = [("pose", 24, torch.float32, (6,)), ("velocity", 24, torch.float32, (6,)),
                  ("speed", 4, torch.float32, ()), ("action", 8, torch.float32, (2,)), ("reward", 4, torch.float32, ()),
                  ("discount", 4, torch.float32, ()), ("progress_total", 4, torch.float32, ()),
                  ("time", 4, torch.float32, ())]

# uint16 LiDAR code -> value in the row's own units: q / scale - off (include/racecar_hip.h, rc_set_compact_slab)
LIDAR_U16_CODE = {"metres": (0.0, 65535.0 / 15.0), "dreamer": (0.5, 65535.0), "unit": (0.0, 65535.0)}


def dequantise_lidar(q: torch.Tensor, transform: str = "metres") -> torch.Tensor:
    off, scale = LIDAR_U16_CODE[transform]
    wide = q.view(torch.int16).to(torch.int32) & 0xFFFF        # (uint16 arithmetic is not implemented on every device)
    return wide.to(torch.float32) / scale - off


def _sections(buf: torch.Tensor, n_cars: int, fields, off: int = 0) -> dict:
    out = {}
    for name, per_car, dtype, tail in fields:
        nb = per_car * n_cars
        out[name] = buf[off:off + nb].view(dtype).view(n_cars, *tail)
        off = (off + nb + 63) // 64 * 64
    return out


def summary_field_views(summary_rank: torch.Tensor, n_cars: int) -> dict:
    """Typed views of one rank's POSE..TIME bytes (gather mode 'summary')."""
    return _sections(summary_rank, n_cars, SUMMARY_FIELDS)


def compact_field_views(compact_rank: torch.Tensor, n_cars: int) -> dict:
    """Typed views of one rank's compact slab (gather mode 'full-u16'): `lidar_u16` uint16 [n, 1080] + the summary."""
    from . import _lib as L
    nb = n_cars * L.RC_N_BEAMS * 2
    out = {"lidar_u16": compact_rank[:nb].view(torch.uint16).view(n_cars, L.RC_N_BEAMS)}
    out.update(_sections(compact_rank, n_cars, SUMMARY_FIELDS, (nb + 63) // 64 * 64))
    return out


def gather_link_model(bytes_per_gpu_per_step: int, world: int, link_gbs: float = 76.8, links: int = 7) -> dict:
    """Lower bound of an all-gather on the xGMI full mesh: every GPU receives (world - 1) shards, at best one per link
    in parallel (MI355X: 7 links x 153.6 GB/s bidirectional = 76.8 GB/s inbound each), so the step cannot be shorter
    than shard bytes / link rate however the collective is scheduled."""
    peers = max(world - 1, 0)
    inbound = bytes_per_gpu_per_step * peers
    t = bytes_per_gpu_per_step / (link_gbs * 1e9) * max(1.0, peers / links) if peers else 0.0
    return {"bytes_per_gpu_per_step": int(bytes_per_gpu_per_step), "inbound_bytes_per_gpu_per_step": int(inbound),
            "link_bound_ms_per_step": t * 1e3, "assumed_link_GBps_inbound": link_gbs, "links": links}


def slab_field_views(slab_rank: torch.Tensor, n_cars: int, occupancy: bool) -> dict:
    """Typed views of one rank's slab bytes (layout of include/racecar_hip.h: LIDAR..TIME[, OCCUPANCY],
    every section 64-byte aligned)."""
    from . import _lib as L
    sizes = [("lidar", L.RC_N_BEAMS * 4, torch.float32, (L.RC_N_BEAMS,)), ("pose", 24, torch.float32, (6,)),
             ("velocity", 24, torch.float32, (6,)), ("speed", 4, torch.float32, ()),
             ("action", 8, torch.float32, (2,)), ("reward", 4, torch.float32, ()),
             ("discount", 4, torch.float32, ()), ("progress_total", 4, torch.float32, ()),
             ("time", 4, torch.float32, ())]
    if occupancy:
        sizes.append(("lidar_occupancy", L.RC_PATCH * L.RC_PATCH, torch.uint8, (L.RC_PATCH, L.RC_PATCH, 1)))
    out, off = {}, 0
    for name, per_car, dtype, tail in sizes:
        nb = per_car * n_cars
        out[name] = slab_rank[off:off + nb].view(dtype).view(n_cars, *tail)
        off = (off + nb + 63) // 64 * 64
    return out
