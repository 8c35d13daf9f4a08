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

    ``launch(slab)`` snapshots the slab into a staging buffer (so the producer may overwrite it); every ``every``-th
    call starts ONE asynchronous collective over the ``every`` snapshots taken since the last one (same bytes on the
    links, 1 / every of the launches: at 0.2 ms per step a collective per step is mostly fixed cost).  Two staging
    buffers take turns, so a collective has ``every`` steps to finish before its buffer is written again.
    ``wait()`` flushes a partial batch, waits, and returns the gathered buffer of the last collective:
    ``[world_size, slab_bytes]`` for ``every == 1``, else ``[world_size, n_snapshots, slab_bytes]``.
    """

    def __init__(self, slab_like: torch.Tensor, group: Optional[dist.ProcessGroup] = None, every: int = 1):
        if every < 1:
            raise ValueError("every must be >= 1")
        self.group = group
        self.every = int(every)
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        flat = slab_like.reshape(-1)
        self.slab_numel = flat.numel()
        self.staging = [torch.empty(self.every * flat.numel(), dtype=flat.dtype, device=flat.device) for _ in range(2)]
        self.gathered = torch.empty(self.world * self.every * flat.numel(), dtype=flat.dtype, device=flat.device)
        self._work = None
        self._cur = 0            # staging buffer being filled
        self._k = 0              # snapshots in it
        self._last_n = self.every

    def launch(self, slab: torch.Tensor) -> None:
        n = self.slab_numel
        self.staging[self._cur][self._k * n:(self._k + 1) * n].copy_(slab.reshape(-1), non_blocking=True)
        self._k += 1
        if self._k == self.every:
            self._issue()

    def _issue(self) -> None:
        if self._work is not None:       # the previous collective (it read the OTHER staging buffer, wrote `gathered`)
            self._work.wait()
            self._work = None
        n, k = self.slab_numel, self._k
        src = self.staging[self._cur][:k * n]
        dst = self.gathered[:self.world * k * n]
        self._last_n = k
        self._cur ^= 1
        self._k = 0
        if dist.get_backend(self.group) == "gloo" and src.is_cuda:
            # gloo has no device all-gather: stage through the host (functional tests only)
            host = src.cpu()
            out = torch.empty(self.world * host.numel(), dtype=host.dtype)
            dist.all_gather_into_tensor(out, host, group=self.group)
            dst.copy_(out)
            return
        self._work = dist.all_gather_into_tensor(dst, src, group=self.group, async_op=True)

    def wait(self) -> torch.Tensor:
        if self._k:
            self._issue()                # a partial batch (every rank holds the same number of snapshots)
        if self._work is not None:
            self._work.wait()
            self._work = None
        g = self.gathered[:self.world * self._last_n * self.slab_numel]
        return g.view(self.world, -1) if self.every == 1 else g.view(self.world, self._last_n, -1)


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
