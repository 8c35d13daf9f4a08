"""TrajectoryRing (device-resident replay) on a CPU stand-in for the env: slot rotation,
strided field views, window sampling without episode boundaries, the reference's reset row."""
import numpy as np
import pytest
import torch

from racing_dreamer_amd.replay import TrajectoryRing


class FakeEnv:
    """Writes recognisable values into whatever arena it is pointed at: lidar = step + env / 1000, reward = step,
    fresh = 1 on the steps listed per env."""

    def __init__(self, num_envs=6, cars=2, fresh_at=(), done_at=()):
        self.num_envs, self.cars_per_env, self.device = num_envs, cars, torch.device("cpu")
        n = num_envs * cars
        self._host_layout, off = {}, 0
        for name, dtype, tail in (("lidar", "float32", (8,)), ("action", "float32", (2,)), ("reward", "float32", ()),
                                  ("discount", "float32", ()), ("fresh", "uint8", ()), ("done", "uint8", ())):
            nb = n * int(np.prod(tail, dtype=int)) * (4 if dtype == "float32" else 1)
            self._host_layout[name] = (off, nb, dtype, tail)
            off = (off + nb + 63) // 64 * 64
        self.arena_nbytes = off
        self.own = torch.zeros(off, dtype=torch.uint8)
        self.views = self.views_of(self.own)
        self.t = -1
        self.fresh_at = set(fresh_at)
        self.done_at = set(done_at)          # (t, env): auto-reset record - terminal scalars + the next episode's first obs
        self.bound = []

    def views_of(self, arena):
        return {k: arena[o:o + nb].view(getattr(torch, dt)).view(self.num_envs, self.cars_per_env, *tail)
                for k, (o, nb, dt, tail) in self._host_layout.items()}

    def set_arena(self, arena, views=None):
        self.views = self.views_of(self.own) if arena is None else (views or self.views_of(arena))
        self.bound.append(None if arena is None else arena.data_ptr())

    def _write(self):
        self.t += 1
        v = self.views
        env_id = torch.arange(self.num_envs, dtype=torch.float32)[:, None, None]
        v["lidar"][:] = self.t + env_id / 1000.0
        v["action"][:] = 0.5
        v["reward"][:] = float(self.t)
        v["discount"][:] = 0.9
        v["done"][:] = 0
        v["fresh"][:] = 0
        for (t, e) in self.fresh_at:
            if t == self.t:
                v["fresh"][e] = 1
        for (t, e) in self.done_at:
            if t == self.t:
                v["fresh"][e] = 1
                v["done"][e] = 1
                v["discount"][e] = 0.0
                v["reward"][e] = -1.0
                v["lidar"][e] = -7.0          # the NEXT episode's first observation
        return v

    def reset(self, **kw):
        return self._write()

    def step(self, actions=None, repeat=None):
        return self._write()


def test_slots_rotate_and_fields_are_strided_views():
    env = FakeEnv()
    ring = TrajectoryRing(env, capacity=4)
    ring.reset()
    for _ in range(5):
        ring.step()
    # 6 records written into 4 slots: slots hold steps 4, 5, 2, 3
    assert ring.count == 4 and ring.head == 1 and ring.steps_written == 6
    assert [float(ring.fields["reward"][k, 0, 0]) for k in range(4)] == [4.0, 5.0, 2.0, 3.0]
    assert ring.fields["lidar"].shape == (4, 6, 2, 8)
    assert torch.equal(ring.fields["lidar"][1, 3], torch.full((2, 8), 5.0 + 0.003))
    assert len(set(env.bound)) == 4                      # exactly `capacity` distinct arenas were bound
    assert ring.latest()["reward"][0, 0] == 5.0
    ring.detach()
    assert env.bound[-1] is None


def test_windows_are_consecutive_in_time_and_stay_on_one_car():
    env = FakeEnv()
    ring = TrajectoryRing(env, capacity=8)
    ring.reset()
    for _ in range(10):
        ring.step()
    g = torch.Generator().manual_seed(0)
    out = ring.sample(batch=64, length=5, generator=g)
    r = out["reward"]
    assert r.shape == (64, 5) and out["lidar"].shape == (64, 5, 8)
    assert torch.all(r[:, 1:] - r[:, :-1] == 1.0)                       # consecutive steps, also across the wrap
    assert r.min() >= 3.0 and r.max() <= 10.0                            # only the 8 newest records (steps 3..10)
    env_of_row = ((out["lidar"][:, 0, 0] - r[:, 0]) * 1000).round().long()
    assert torch.equal(env_of_row, out["env"])
    assert set(out["t0"].tolist()) <= set(range(ring.window_starts(5)))


def test_windows_never_contain_an_episode_start_after_their_first_record():
    fresh_at = {(4, 1), (7, 2), (2, 0)}
    env = FakeEnv(fresh_at=fresh_at)
    ring = TrajectoryRing(env, capacity=16)
    ring.reset()
    for _ in range(11):
        ring.step()
    g = torch.Generator().manual_seed(1)
    out = ring.sample(batch=256, length=4, generator=g)
    assert not (out["fresh"][:, 1:] != 0).any()
    starts = out["fresh"][:, 0] != 0
    assert starts.any() and (~starts).any()
    # the reference's reset row at an episode start (wrappers.py:232-236); other rows untouched
    assert torch.all(out["action"][starts, 0] == 0.0) and torch.all(out["reward"][starts, 0] == 0.0)
    assert torch.all(out["discount"][starts, 0] == 1.0)
    assert torch.all(out["action"][~starts] == 0.5) and torch.all(out["discount"][:, 1:] == 0.9)
    # the flag itself travels with the batch (consumers of ShardedReplay.exchange(flat=False) read meta[..., 3])
    assert torch.equal(out["first"], starts)
    raw = ring.sample(batch=64, length=4, generator=g, reset_rows=False)
    assert torch.all(raw["action"] == 0.5) and torch.equal(raw["first"], raw["fresh"][:, 0] != 0)
    with pytest.raises(ValueError, match="not recorded"):
        ring.sample(batch=4, length=4, fields=("lidar", "no_such_field"))


def test_terminal_transitions_are_sampled_as_the_last_row_of_a_window():
    """with auto-reset the terminal reward / discount 0 sit in a record that is also `fresh`; the
    learner must see them (pcont trains on `discount`, dreamer/models.py:103).  Such a record may end a window, with
    the previous row's observation; it may start one as a reset row; it may not sit inside one."""
    done_at = {(5, 1), (8, 3), (6, 0)}
    env = FakeEnv(done_at=done_at)
    ring = TrajectoryRing(env, capacity=16)
    ring.reset()
    for _ in range(12):
        ring.step()
    g = torch.Generator().manual_seed(2)
    out = ring.sample(batch=2048, length=4, generator=g)
    term = out["terminal"]
    assert term.any() and (~term).any()
    assert not (out["fresh"][:, 1:-1] != 0).any()
    assert torch.equal(out["fresh"][:, -1] != 0, term) and torch.all(out["done"][term, -1] == 1)
    # terminal rows keep the step's scalars and borrow the observation of the row before
    assert torch.all(out["discount"][term, -1] == 0.0) and torch.all(out["reward"][term, -1] == -1.0)
    assert torch.equal(out["lidar"][term, -1], out["lidar"][term, -2]) and not (out["lidar"][:, 1:] == -7.0).any()
    assert torch.all(out["discount"][~term, -1] == 0.9)
    # expected rate: windows of 4 over 13 records have 10 start times per env; each terminal record ends exactly one
    # window and rules out the two windows that hold it at row 1 or 2
    t_last = out["t0"] + 3
    hits = {(int(t), int(e)) for t, e, m in zip(t_last, out["env"], term) if m}
    assert hits == done_at
    valid = 10 * 6 - 2 * len(done_at)
    assert abs(float(term.float().mean()) - len(done_at) / valid) < 0.03
    # a window starting on the terminal record is the next episode's reset row
    starts = out["fresh"][:, 0] != 0
    assert starts.any()
    assert torch.all(out["discount"][starts, 0] == 1.0) and torch.all(out["reward"][starts, 0] == 0.0)
    assert torch.all(out["lidar"][starts, 0] == -7.0)


def test_masked_reset_is_refused():
    env = FakeEnv()
    ring = TrajectoryRing(env, capacity=4)
    ring.reset()
    with pytest.raises(ValueError, match="masked reset"):
        ring.reset(mask=np.ones(6, np.uint8))
    assert ring.steps_written == 1


def test_errors():
    env = FakeEnv()
    with pytest.raises(ValueError):
        TrajectoryRing(env, capacity=1)
    ring = TrajectoryRing(env, capacity=4)
    ring.reset()
    with pytest.raises(ValueError, match="a window needs"):
        ring.sample(4, 3)
