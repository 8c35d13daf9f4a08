"""Device-resident trajectory ring: the replay store of the batched env.

The reference keeps experience as episode files: `Collect` appends one transition per agent step
(dreamer/wrappers.py:213-219), `save_episodes` writes `.npz` files (dreamer/callbacks.py:41-53) and
`load_episodes` samples fixed-length windows out of them for the learner (dreamer/tools.py:235-264).  With tens of
thousands of envs on one MI355X the same role is played by a ring of the last `capacity` step records of EVERY
car, resident in HBM (288 GB hold ~900 steps of 65 536 cars with full 1080-beam scans): before each step the env's
output arena is re-pointed at the next slot (`rc_set_arena`), so the kernels write the record straight into the
ring - no copy, no host - and `sample()` gathers `[batch, length, ...]` windows with one indexed read per field.

Window semantics follow the reference's dataset (episode files sliced by `load_episodes`): a window never crosses
an episode boundary.  With `auto_reset` the record of an episode's LAST step carries the terminal reward / discount 0
/ done together with the first observation of the next episode (`fresh` = 1): such a record may END a window - it
is the terminal transition the learner's discount head trains on (dreamer/models.py:103) - and then keeps its reward,
discount, progress and time while its observation fields are replaced by the previous row's (the terminal
observation is gone, as in vectorised gym envs; `EpisodeRecorder` does the same).  A window that STARTS on a fresh
record gets the reference's reset row there (action 0, reward 0, discount 1, progress -1, time 0 -
wrappers.py:232-236).  A fresh record anywhere else would splice two episodes, so such windows are rejected.
`EpisodeRecorder` (trajectory.py) remains the way to write reference-format episode files for a subset of cars.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch

DEFAULT_SAMPLE_FIELDS = ("lidar", "lidar_occupancy", "action", "reward", "discount", "progress_total", "time",
                         "speed", "done", "fresh")
# the observation part of a record (what a terminal row borrows from the row before it)
OBSERVATION_FIELDS = ("lidar", "lidar_occupancy", "pose", "velocity", "speed", "acceleration", "steering_angle")


class TrajectoryRing:
    def __init__(self, env, capacity: int):
        """`env`: a BatchedRaceEnv (anything with `_host_layout`, `arena_nbytes`, `device`, `num_envs`,
        `cars_per_env`, `set_arena()`, `reset()`, `step()`).  Allocates capacity x arena bytes on the env's device."""
        if capacity < 2:
            raise ValueError("capacity must be >= 2")
        self.env = env
        self.capacity = int(capacity)
        self.slot_bytes = (env.arena_nbytes + 63) // 64 * 64
        raw = torch.zeros(self.capacity * self.slot_bytes + 64, dtype=torch.uint8, device=env.device)
        pad = (-raw.data_ptr()) % 64
        self._raw = raw
        self.buffer = raw[pad:pad + self.capacity * self.slot_bytes]
        self.head = -1            # slot of the latest record
        self.count = 0            # records held (<= capacity)
        self.steps_written = 0
        self._slot_views: Dict[int, Dict[str, torch.Tensor]] = {}
        # one strided tensor per field over all slots: [capacity, num_envs, cars_per_env, ...]
        self.fields: Dict[str, torch.Tensor] = {}
        for name, (off, nb, dtype, tail) in env._host_layout.items():
            if name == "action_in":
                continue
            flat = torch.as_strided(self.buffer, (self.capacity, nb), (self.slot_bytes, 1), off)
            self.fields[name] = flat.view(getattr(torch, dtype)).view(self.capacity, env.num_envs, env.cars_per_env, *tail)

    # ------------------------------------------------------------------ writing
    def slot(self, k: int) -> torch.Tensor:
        return self.buffer[k * self.slot_bytes:(k + 1) * self.slot_bytes]

    def _advance(self) -> None:
        self.head = (self.head + 1) % self.capacity
        views = self._slot_views.get(self.head)
        if views is None:
            views = self.env.views_of(self.slot(self.head))
            self._slot_views[self.head] = views
        self.env.set_arena(self.slot(self.head), views)
        self.count = min(self.count + 1, self.capacity)
        self.steps_written += 1

    def reset(self, *args, **kwargs) -> Dict[str, torch.Tensor]:
        """env.reset() of EVERY env recorded as the next ring record.  Masked resets are refused: the reset kernel
        writes nothing for the envs that keep running, so their part of the new slot would hold whatever was there
        `capacity` steps ago - record finished envs with `auto_reset=True` (the terminal transition and the new
        episode's first observation then share one record, see `sample`)."""
        mask = kwargs.get("mask", args[0] if args else None)
        if mask is not None:
            raise ValueError("TrajectoryRing.reset cannot record a masked reset; build the env with auto_reset=True")
        self._advance()
        return self.env.reset(*args, **kwargs)

    def step(self, actions=None, repeat=None) -> Dict[str, torch.Tensor]:
        """env.step() recorded as the next ring record.  Device-side agents that read the current observation
        (`env.follow_the_gap()`) must run BEFORE this call: it re-points the outputs first."""
        self._advance()
        return self.env.step(actions, repeat=repeat)

    def step_random(self, seed: int, step: int, repeat=None) -> Dict[str, torch.Tensor]:
        """env.step_random() (actions drawn on the device) recorded as the next ring record."""
        self._advance()
        return self.env.step_random(seed, step, repeat=repeat)

    def latest(self) -> Dict[str, torch.Tensor]:
        return self.env.views

    def detach(self) -> None:
        """Give the env its own arena back (the ring keeps its contents)."""
        self.env.set_arena(None)

    def clear(self) -> None:
        """Forget the records held (the memory stays): the next record goes to slot 0, windows are drawn from what is written
        from now on.  For a caller that runs several collections over one env and wants ONE ring's worth of memory."""
        self.head, self.count, self.steps_written = -1, 0, 0

    # ------------------------------------------------------------------ reading
    def _field_names(self, fields: Optional[Sequence[str]]) -> list:
        """The fields of a draw: the default set narrowed to what this ring records, or exactly what the caller named - a
        name the ring does not record is an error, not a silently shorter batch."""
        if fields is None:
            return [f for f in DEFAULT_SAMPLE_FIELDS if f in self.fields]
        missing = [f for f in fields if f not in self.fields]
        if missing:
            raise ValueError(f"fields {missing} are not recorded by this ring (recorded: {sorted(self.fields)})")
        return list(fields)

    def window_starts(self, length: int) -> int:
        """Number of distinct start times of a `length`-step window inside the filled part of the ring."""
        return max(self.count - length + 1, 0)

    def _sample_native(self, batch, length, names, seed, reset_rows, max_tries, check, defer=False):
        """The whole draw on the device: `rc_sample_windows` (a wave per window: draw, test the episode boundary, emit the
        rows) and two `rc_gather_rows` launches; no host round trip unless `check`.  With `defer` only those three launches
        are queued (on the env's stream) and the small fix-ups - the reference's reset row, the per-window integers - are
        left to `finish_sample`, which a caller may run on another stream."""
        env = self.env
        oldest = (self.head + 1) % self.capacity if self.count == self.capacity else 0
        self._draws = getattr(self, "_draws", 0) + 1
        w = env.sample_windows(self.buffer, self.slot_bytes, self.capacity, oldest, self.count, length, batch, seed, self._draws, max_tries)
        if check and int(w["failed"].item()):
            raise RuntimeError(f"could not find {batch} windows of {length} records without an episode boundary")
        obs_names = [n for n in names if n in OBSERVATION_FIELDS]
        rest = [n for n in names if n not in OBSERVATION_FIELDS]
        rows = {}
        if obs_names:
            rows.update(env.gather_rows(self.buffer, self.slot_bytes, w["slots_obs"], w["cars"], obs_names))
        if rest:
            rows.update(env.gather_rows(self.buffer, self.slot_bytes, w["slots"], w["cars"], rest))
        out = {n: v.view(batch, length, *v.shape[1:]) for n, v in rows.items()}
        out["_meta"], out["_reset_rows"] = w["meta"], reset_rows
        return out if defer else self.finish_sample(out)

    def finish_sample(self, out: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """The fix-ups of a deferred native draw, queued on torch's current stream (the caller orders it behind the draw)."""
        out = dict(out)
        meta, reset_rows = out.pop("_meta"), out.pop("_reset_rows")
        if reset_rows:                      # the reference's reset row (wrappers.py:221-226): one in-place masked fill per field
            first = meta[:, 3] != 0
            for name, value in (("action", 0.0), ("reward", 0.0), ("discount", 1.0), ("time", 0.0), ("progress_total", -1.0)):
                if name in out:
                    head = out[name][:, 0]
                    head.masked_fill_(first.view(-1, *([1] * (head.dim() - 1))), value)
        car = meta[:, 1].long()
        out["env"], out["car"] = car // self.env.cars_per_env, car % self.env.cars_per_env
        out["t0"], out["terminal"], out["first"] = meta[:, 0].long(), meta[:, 2] != 0, meta[:, 3] != 0
        return out

    def sample_packed(self, batch: int, length: int, fields: Optional[Sequence[str]] = None, generator: Optional[torch.Generator] = None,
                      reset_rows: bool = True, max_tries: int = 16, out: Optional[torch.Tensor] = None, layout=None):
        """The same draw as `sample(native=True)` as ONE native call into ONE packed buffer (`rc_sample_batch`: a memset and two
        launches on the env's stream, no torch kernels, no host read): returns (buffer, layout) - `unpack(buffer, layout)`
        gives the field views.  The form a sharded store sends: `ShardedReplay.exchange_packed`."""
        if self.window_starts(length) <= 0:
            raise ValueError(f"ring holds {self.count} records, a window needs {length}")
        names = self._field_names(fields)
        if layout is None:
            layout = self.env.sample_batch_layout(names, batch, length)
        oldest = (self.head + 1) % self.capacity if self.count == self.capacity else 0
        self._draws = getattr(self, "_draws", 0) + 1
        seed = generator.initial_seed() if generator is not None else 0x5eed
        buf = self.env.sample_batch(self.buffer, self.slot_bytes, self.capacity, oldest, self.count, layout, seed, self._draws,
                                    reset_rows=reset_rows, max_tries=max_tries, out=out)
        return buf, layout

    @staticmethod
    def unpack(buf: torch.Tensor, layout, lead=()) -> Dict[str, torch.Tensor]:
        """Views of a packed batch (`buf` [..., >= payload] uint8, leading axes `lead`): field -> [*lead, windows, length, ...],
        `meta` int32 [*lead, windows, 4] = (t0, car, terminal, starts an episode), `failed` int32 [*lead] (windows that found
        no episode-internal start)."""
        nw, ln = layout["n_windows"], layout["length"]
        out = {}
        for name, (off, nb, dtype, tail) in layout["fields"].items():
            out[name] = buf[..., off:off + nb].view(dtype).view(*lead, nw, ln, *tail)
        out["meta"] = buf[..., layout["meta"]:layout["meta"] + 16 * nw].view(torch.int32).view(*lead, nw, 4)
        out["failed"] = buf[..., layout["failed"]:layout["failed"] + 4].view(torch.int32).reshape(tuple(lead))
        return out

    def sample(self, batch: int, length: int, fields: Optional[Sequence[str]] = None,
               generator: Optional[torch.Generator] = None, reset_rows: bool = True,
               max_tries: int = 16, native: Optional[bool] = None, check: bool = True, defer: bool = False) -> Dict[str, torch.Tensor]:
        """`batch` windows of `length` consecutive records of one car each, uniformly over (time, env, car) among
        the windows that stay inside one episode: no fresh record strictly inside, and a fresh LAST record only if it
        is the terminal transition of the window's episode (done = 1, written by auto-reset).  Returns field ->
        [batch, length, ...] (copies, on the ring's device) plus `env`, `car`, `t0` (ring age of the first record,
        0 = oldest), `terminal` (bool [batch]: the last row is such a terminal transition) and `first` (bool [batch]: the
        window's first record starts an episode - the row `reset_rows` rewrites).  On a device ring the draw
        itself runs on the device (`native`, default there; `check=False` also drops the one host read of the failure count);
        `native=False` is the tensor-indexing form with torch's generator."""
        nstart = self.window_starts(length)
        if nstart <= 0:
            raise ValueError(f"ring holds {self.count} records, a window needs {length}")
        names = self._field_names(fields)
        if native is None:
            native = hasattr(self.env, "sample_windows") and self.buffer.is_cuda
        if native:                          # (the draw comes from Philox on the device, keyed by the generator's seed and a counter)
            seed = generator.initial_seed() if generator is not None else 0x5eed
            if defer or not hasattr(self.env, "sample_batch"):
                return self._sample_native(batch, length, names, seed, reset_rows, max_tries, check, defer)
            # one native call into one packed buffer (rc_sample_batch), then views: 0.017 ms for 50 x 50 windows where the
            # three launches + torch fix-ups of `_sample_native` take 0.14
            buf, layout = self.sample_packed(batch, length, fields=names, generator=generator, reset_rows=reset_rows, max_tries=max_tries)
            out = self.unpack(buf, layout)
            meta = out.pop("meta")
            failed = out.pop("failed")
            if check and int(failed.item()):
                raise RuntimeError(f"could not find {batch} windows of {length} records without an episode boundary")
            car = meta[:, 1].long()
            out["env"], out["car"] = car // self.env.cars_per_env, car % self.env.cars_per_env
            out["t0"], out["terminal"], out["first"] = meta[:, 0].long(), meta[:, 2] != 0, meta[:, 3] != 0
            return out
        dev = self.buffer.device
        oldest = (self.head + 1) % self.capacity if self.count == self.capacity else 0
        ar = torch.arange(length, device=dev)
        fresh, done = self.fields["fresh"], self.fields["done"]
        keep_t, keep_e, keep_c, have = [], [], [], 0
        for _ in range(max_tries):
            m = max(2 * (batch - have), 16)
            t0 = torch.randint(0, nstart, (m,), device=dev, generator=generator)
            e = torch.randint(0, self.env.num_envs, (m,), device=dev, generator=generator)
            c = torch.randint(0, self.env.cars_per_env, (m,), device=dev, generator=generator)
            slots = (oldest + t0[:, None] + ar[None, :]) % self.capacity                    # [m, length]
            ok = torch.ones(m, dtype=torch.bool, device=dev)
            if length > 2:
                ok &= ~(fresh[slots[:, 1:-1], e[:, None], c[:, None]] != 0).any(1)
            if length > 1:                                    # a fresh last record must be the episode's terminal one
                ok &= (fresh[slots[:, -1], e, c] == 0) | (done[slots[:, -1], e, c] != 0)
            keep_t.append(t0[ok]); keep_e.append(e[ok]); keep_c.append(c[ok])
            have += int(ok.sum())
            if have >= batch:
                break
        else:
            raise RuntimeError(f"could not find {batch} windows of {length} records without an episode boundary")
        t0 = torch.cat(keep_t)[:batch]; e = torch.cat(keep_e)[:batch]; c = torch.cat(keep_c)[:batch]
        slots = (oldest + t0[:, None] + ar[None, :]) % self.capacity
        terminal = torch.zeros(batch, dtype=torch.bool, device=dev)
        if length > 1:
            terminal = (fresh[slots[:, -1], e, c] != 0) & (done[slots[:, -1], e, c] != 0)
        if hasattr(self.env, "gather_rows") and self.buffer.is_cuda:
            # one launch per group of fields (rc_gather_rows: a wave per row) instead of one indexing kernel chain per field.
            # A terminal row takes its observation from the record before it (the new episode's observation is not this
            # episode's): for the observation fields that row simply reads the previous slot.
            cars = (e * self.env.cars_per_env + c)[:, None].expand(batch, length).reshape(-1)
            slots_obs = slots.clone()
            if length > 1:
                slots_obs[:, -1] = torch.where(terminal, slots[:, -2], slots[:, -1])
            obs_names = [n for n in names if n in OBSERVATION_FIELDS]
            rest = [n for n in names if n not in OBSERVATION_FIELDS]
            rows = {}
            if obs_names:
                rows.update(self.env.gather_rows(self.buffer, self.slot_bytes, slots_obs.reshape(-1), cars, obs_names))
            if rest:
                rows.update(self.env.gather_rows(self.buffer, self.slot_bytes, slots.reshape(-1), cars, rest))
            out = {n: v.view(batch, length, *v.shape[1:]) for n, v in rows.items()}
        else:
            out = {name: self.fields[name][slots, e[:, None], c[:, None]] for name in names}
            if length > 1:
                for name in OBSERVATION_FIELDS:               # the new episode's observation is not this episode's
                    if name in out:
                        out[name][terminal, -1] = out[name][terminal, -2]
        first = self.fields["fresh"][slots[:, 0], e, c] != 0                   # window starts an episode
        if reset_rows:
            if "action" in out:
                out["action"][first, 0] = 0.0
            if "reward" in out:
                out["reward"][first, 0] = 0.0
            if "discount" in out:
                out["discount"][first, 0] = 1.0
            if "time" in out:
                out["time"][first, 0] = 0.0
            if "progress_total" in out:
                out["progress_total"][first, 0] = -1.0
        out["env"], out["car"], out["t0"], out["terminal"], out["first"] = e, c, t0, terminal, first
        return out


class ShardedReplay:
    """The multi-GPU form of the replay store that needs no per-step collective: every rank keeps the
    records of ITS envs in its own `TrajectoryRing`, and what crosses the links is the training batch - each rank draws
    `batch / world` windows from its shard and an all-gather over `torch.distributed` (backend "nccl" = RCCL; "gloo" in
    the CPU tests) hands every rank the same global batch: 50 x 50 windows of the 2 236-byte record are 5.6 MB per
    train step, where gathering every step's records of 65 536 envs is 147 MB per GPU per 0.23 ms.  This is the
    concat of `Collect` -> `save_episodes` -> `load_episodes` (dreamer/wrappers.py:213-219, dreamer/tools.py:235-264)
    at the granularity its consumer reads it."""

    def __init__(self, ring: TrajectoryRing, group=None):
        import torch.distributed as dist
        self.ring, self.group, self._dist = ring, group, dist
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)

    META = ("env", "car", "t0", "terminal", "first")

    def draw(self, batch: int, length: int, fields: Optional[Sequence[str]] = None,
             generator: Optional[torch.Generator] = None, check: bool = True, defer: bool = False) -> Dict[str, torch.Tensor]:
        """This rank's `batch / world` windows (the local half of `sample`): queued on the env's stream.  `defer` (device
        rings): only the sampler's three launches are queued there; `exchange` does the small fix-ups on ITS stream."""
        if batch % self.world:
            raise ValueError(f"batch {batch} is not a multiple of the world size {self.world}")
        return self.ring.sample(batch // self.world, length, fields=fields, generator=generator, check=check, defer=defer)

    def exchange(self, local: Dict[str, torch.Tensor], flat: bool = True) -> Dict[str, torch.Tensor]:
        """All-gather what `draw` returned: ONE collective over one packed byte buffer (every field and the four per-window
        integers, each section 16-byte aligned), queued on torch's CURRENT stream - a caller may run it on a stream of its
        own behind an event, so that the env's stream never waits for the links (bench.py's sharded headline does).
        flat=True: field -> [batch, ...], rank r's windows at rows [r * batch / world, (r + 1) * batch / world) (one copy
        per field: the collective's output is rank-major).  flat=False: no copy - field -> a VIEW [world, batch / world, ...]
        of the gathered buffer, and `meta` int32 [world, batch / world, 4] = (t0, car, terminal, first record of an episode)
        in place of env / car / t0 / terminal / rank."""
        if "_meta" in local:                                 # a deferred draw: its fix-ups run here, on the current stream
            reset_rows, meta = local["_reset_rows"], local["_meta"]
            local = {k: v for k, v in local.items() if not k.startswith("_")}
            if reset_rows:
                first = meta[:, 3] != 0
                for name, value in (("action", 0.0), ("reward", 0.0), ("discount", 1.0), ("time", 0.0), ("progress_total", -1.0)):
                    if name in local:
                        head = local[name][:, 0]
                        head.masked_fill_(first.view(-1, *([1] * (head.dim() - 1))), value)
        else:
            local = dict(local)
            car = local.pop("car").to(torch.int32) + local.pop("env").to(torch.int32) * self.ring.env.cars_per_env
            term = local.pop("terminal").to(torch.int32)
            first = local.pop("first").to(torch.int32)
            meta = torch.stack([local.pop("t0").to(torch.int32), car, term, first], 1)
        per = int(meta.shape[0])
        batch = per * self.world
        local["meta"] = meta
        names, parts, spans, off = [], [], [], 0
        for name, t in local.items():
            raw = t.contiguous().view(-1).view(torch.uint8)
            pad = (-raw.numel()) % 16
            names.append((name, t.dtype, tuple(t.shape[1:])))
            spans.append((off, raw.numel()))
            parts.append(raw)
            if pad:
                parts.append(raw.new_zeros(pad))
            off += raw.numel() + pad
        flat_src = torch.cat(parts)
        host_bounce = self._dist.get_backend(self.group) == "gloo" and flat_src.is_cuda
        src = flat_src.cpu() if host_bounce else flat_src           # gloo has no device collectives (functional tests only)
        gathered = torch.empty(self.world * off, dtype=torch.uint8, device=src.device)
        self._dist.all_gather_into_tensor(gathered, src, group=self.group)
        gathered = gathered.to(flat_src.device).view(self.world, off)
        out = {}
        for (name, dtype, tail), (o, n) in zip(names, spans):
            v = gathered[:, o:o + n].view(dtype).view(self.world, per, *tail)
            out[name] = v.reshape(batch, *tail) if flat else v
        if not flat:
            return out
        meta = out.pop("meta")
        car = meta[:, 1].long()
        out["env"], out["car"] = car // self.ring.env.cars_per_env, car % self.ring.env.cars_per_env
        out["t0"], out["terminal"], out["first"] = meta[:, 0].long(), meta[:, 2] != 0, meta[:, 3] != 0
        if getattr(self, "_rank_rows", None) is None or self._rank_rows.numel() != batch or self._rank_rows.device != flat_src.device:
            self._rank_rows = torch.arange(self.world, device=flat_src.device).repeat_interleave(per)
        out["rank"] = self._rank_rows
        return out

    def draw_packed(self, batch: int, length: int, fields: Optional[Sequence[str]] = None, generator: Optional[torch.Generator] = None,
                    out: Optional[torch.Tensor] = None, layout=None):
        """This rank's `batch / world` windows as one packed buffer (`TrajectoryRing.sample_packed`): one native call on the
        env's stream."""
        if batch % self.world:
            raise ValueError(f"batch {batch} is not a multiple of the world size {self.world}")
        return self.ring.sample_packed(batch // self.world, length, fields=fields, generator=generator, out=out, layout=layout)

    def exchange_packed(self, buf: torch.Tensor, layout, out: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """ONE all-gather of the packed payload on torch's current stream; returns views of the gathered buffer
        (`TrajectoryRing.unpack` with a leading rank axis): field -> [world, batch / world, length, ...]."""
        n = layout["payload"]
        src = buf[:n]
        host_bounce = self._dist.get_backend(self.group) == "gloo" and src.is_cuda
        if host_bounce:                                       # gloo has no device collectives (functional tests only)
            got = torch.empty(self.world * n, dtype=torch.uint8)
            self._dist.all_gather_into_tensor(got, src.cpu(), group=self.group)
            if out is None:
                out = got.to(buf.device)
            else:
                out[:self.world * n].copy_(got)
        else:
            if out is None:
                out = torch.empty(self.world * n, dtype=torch.uint8, device=buf.device)
            self._dist.all_gather_into_tensor(out[:self.world * n], src, group=self.group)
        return TrajectoryRing.unpack(out[:self.world * n].view(self.world, n), layout, lead=(self.world,))

    def sample(self, batch: int, length: int, fields: Optional[Sequence[str]] = None,
               generator: Optional[torch.Generator] = None, check: bool = True) -> Dict[str, torch.Tensor]:
        """`batch` windows in total (a multiple of the world size), rank r's `batch / world` at rows
        [r * batch / world, (r + 1) * batch / world); `env` holds this rank's LOCAL env indices, `rank` the owner.
        check=False drops the one host read per draw (the count of windows that found no episode-internal start): the
        whole sample is then queued without the host waiting for the device.  = `exchange(draw(...))`."""
        return self.exchange(self.draw(batch, length, fields=fields, generator=generator, check=check))
