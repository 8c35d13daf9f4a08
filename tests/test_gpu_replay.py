"""TrajectoryRing on the device: the kernels write step records straight into ring slots (rc_set_arena)."""
import numpy as np
import pytest

from helpers import EXACT_FLOAT, EXACT_INT

pytestmark = pytest.mark.gpu


def test_ring_records_equal_the_oracle_rollout_and_terminal_transitions_are_sampled():
    """Every slot of the ring against the CPU oracle's rollout of the same seeds (not against a second HIP env), then
    the learner-side property ADVICE r1 asked for: sampled batches contain the episodes' terminal transitions
    (discount 0, collision reward) at the rate the ring holds them."""
    import torch
    from oracle import c_oracle, racecar_oracle as ro
    from racing_dreamer_amd import spec
    from racing_dreamer_amd.batched_env import BatchedRaceEnv
    from racing_dreamer_amd.replay import TrajectoryRing
    from racing_dreamer_amd.track_assets import load_track
    track = load_track("columbia")
    n, cap, steps = 192, 16, 48
    env = BatchedRaceEnv(track, n, 1, obs_type="lidar_occupancy", auto_reset=True, action_repeat=4)
    ora = c_oracle.COracleEnv(track.occ, track.drivable, track.progress, track.centerline, track.origin, track.resolution,
                              ro.OracleConfig(num_envs=n, auto_reset=True, render_occupancy=True), threads=8)
    ring = TrajectoryRing(env, capacity=cap)
    ring.reset(mode="random", seed=4)
    want = [ora.reset(mode=spec.RESET_RANDOM, seed=4)]
    for k in range(steps):
        act = ro.random_actions(21, k, n)
        act[:, 0] = 1.0                                     # full throttle, random steering: crashes within a second or two
        ring.step(torch.from_numpy(act).cuda().view(n, 1, 2), repeat=4)
        want.append(ora.step(act, repeat=4))
    torch.cuda.synchronize()
    for age in range(cap):                                  # the newest `cap` records, oldest first
        slot = (ring.head + 1 + age) % cap
        rec = want[steps + 1 - cap + age]
        for name in EXACT_FLOAT + EXACT_INT + ["lidar_occupancy"]:
            got = ring.fields[name][slot].cpu().numpy().reshape(-1)
            exp = np.asarray(rec[name]).reshape(-1)
            assert np.array_equal(got, exp.astype(got.dtype)), (age, name)
    # terminal transitions: windows of 4 ending on a done record
    done = torch.stack([torch.from_numpy(w["done"].astype(np.uint8)) for w in want[-cap:]])      # [cap, n]
    assert int(done.sum()) >= 10
    g = torch.Generator(device="cuda").manual_seed(0)
    batch = ring.sample(batch=4096, length=4, generator=g)
    term = batch["terminal"]
    assert int(term.sum()) > 0
    assert torch.all(batch["discount"][term, -1] == 0.0) and torch.all(batch["done"][term, -1] == 1)
    assert torch.all(batch["reward"][term, -1] < 0.0)               # progress gain of a crash step minus 1 (collision_reward)
    assert torch.equal(batch["lidar"][term, -1], batch["lidar"][term, -2])
    assert torch.all(batch["discount"][:, :-1][batch["fresh"][:, :-1] == 0] == 1.0)
    assert not (batch["fresh"][:, 1:-1] != 0).any()
    # rate: each terminal record at ring age >= 3 ends one window; valid windows = all (start, env) minus rejected ones
    fresh = torch.stack([torch.from_numpy(w["fresh"].astype(np.uint8)) for w in want[-cap:]])
    ok = torch.ones(cap - 3, n, dtype=torch.bool)
    for t0 in range(cap - 3):
        ok[t0] = ~(fresh[t0 + 1:t0 + 3] != 0).any(0) & ((fresh[t0 + 3] == 0) | (done[t0 + 3] != 0))
    expected = float(((done[3:] != 0) & (fresh[3:] != 0) & ok).sum()) / float(ok.sum())
    assert abs(float(term.float().mean()) - expected) < 0.25 * expected + 0.01
    env.close()


def test_ring_records_equal_a_plain_rollout_and_sampling_runs_on_device():
    import torch
    from racing_dreamer_amd.batched_env import BatchedRaceEnv
    from racing_dreamer_amd.replay import TrajectoryRing
    kw = dict(num_envs=256, cars_per_env=2, obs_type="lidar_occupancy", auto_reset=True, action_repeat=4, time_limit_steps=7)
    plain = BatchedRaceEnv("treitlstrasse_v2", **kw)
    env = BatchedRaceEnv("treitlstrasse_v2", **kw)
    ring = TrajectoryRing(env, capacity=10)
    ref = [{k: v.clone() for k, v in plain.reset(mode="random", seed=3).items()}]
    ring.reset(mode="random", seed=3)
    steps = 24
    for k in range(steps):
        plain.fill_random_actions(seed=5, step=k)
        ref.append({n: v.clone() for n, v in plain.step(None).items()})
        env.fill_random_actions(seed=5, step=k)          # action_in does not move with the outputs
        out = ring.step(None)
        assert out["lidar"].data_ptr() == ring.fields["lidar"][ring.head].data_ptr()      # zero-copy: written in place
    torch.cuda.synchronize()
    assert ring.count == 10 and ring.steps_written == steps + 1
    for age in range(10):                                  # the 10 newest records, oldest first
        slot = (ring.head + 1 + age) % 10
        want = ref[steps + 1 - 10 + age]
        for name, f in ring.fields.items():
            assert torch.equal(f[slot], want[name]), (age, name)
    assert sum(int(r["fresh"].sum()) for r in ref[-9:]) > 0          # the window filter below has something to filter
    g = torch.Generator(device="cuda").manual_seed(0)
    batch = ring.sample(batch=128, length=4, generator=g)
    assert batch["lidar"].shape == (128, 4, 1080) and batch["lidar"].is_cuda
    assert batch["lidar_occupancy"].shape == (128, 4, 64, 64, 1) and batch["action"].shape == (128, 4, 2)
    assert not (batch["fresh"][:, 1:-1] != 0).any()
    assert torch.equal(batch["fresh"][:, -1] != 0, batch["terminal"])
    t = batch["time"]
    inner = t[:, 1:-1] - t[:, :-2]
    assert torch.allclose(inner, torch.full_like(inner, 0.04), atol=1e-5)                     # consecutive agent steps
    # detach: the env writes into its own arena again and the ring keeps its contents
    newest = ring.fields["reward"][ring.head].clone()
    ring.detach()
    env.fill_random_actions(seed=5, step=steps)
    env.step(None)
    torch.cuda.synchronize()
    assert torch.equal(ring.fields["reward"][ring.head], newest)
    plain.close(); env.close()


def test_follow_the_gap_reads_the_current_slot_before_the_ring_advances():
    import torch
    from racing_dreamer_amd.batched_env import BatchedRaceEnv
    from racing_dreamer_amd.replay import TrajectoryRing
    plain = BatchedRaceEnv("columbia", 64, 1, auto_reset=True)
    env = BatchedRaceEnv("columbia", 64, 1, auto_reset=True)
    ring = TrajectoryRing(env, capacity=3)
    plain.reset(mode="grid"); ring.reset(mode="grid")
    for _ in range(7):
        plain.follow_the_gap(); a = plain.step(None)
        env.follow_the_gap(); b = ring.step(None)
    torch.cuda.synchronize()
    for name in ("lidar", "pose", "reward", "action", "done"):
        assert torch.equal(a[name], b[name]), name
    plain.close(); env.close()


def test_set_arena_rejects_bad_buffers():
    import torch
    from racing_dreamer_amd import _lib as L
    from racing_dreamer_amd.batched_env import BatchedRaceEnv
    env = BatchedRaceEnv("columbia", 8, 1)
    with pytest.raises(L.RacecarHipError, match="too small"):
        env.set_arena(torch.zeros(64, dtype=torch.uint8, device="cuda"))
    big = torch.zeros(env.arena_nbytes + 128, dtype=torch.uint8, device="cuda")
    off = (-big.data_ptr()) % 64 + 4
    with pytest.raises(L.RacecarHipError, match="aligned"):
        env.set_arena(big[off:off + env.arena_nbytes])
    with pytest.raises(ValueError):
        env.set_arena(torch.zeros(env.arena_nbytes, dtype=torch.uint8))
    env.set_arena(None)
    env.reset()
    env.close()


def test_native_window_gather_equals_tensor_indexing():
    """rc_gather_rows (one wave per row, every field of the mask in one launch) against plain advanced indexing on the ring's
    strided views - the LiDAR rows as 16-byte vectors, 24-byte poses, single-byte flags, the 4 KB patches - and through
    TrajectoryRing.sample: the terminal rows' observations come from the record before them."""
    import torch
    from racing_dreamer_amd.batched_env import BatchedRaceEnv
    from racing_dreamer_amd.replay import OBSERVATION_FIELDS, TrajectoryRing
    from oracle import racecar_oracle as ro
    env = BatchedRaceEnv("columbia", 96, 2, obs_type="lidar_occupancy", auto_reset=True, action_repeat=4)
    ring = TrajectoryRing(env, capacity=12)
    ring.reset(mode="random_ball", seed=1)
    for k in range(60):
        act = ro.random_actions(2, k, 192)
        act[:, 0] = 1.0                                     # full throttle, random steering: episodes end within a second or two
        ring.step(torch.from_numpy(act).cuda().view(96, 2, 2))
    g = torch.Generator(device="cuda").manual_seed(3)
    rows = 500
    slots = torch.randint(0, 12, (rows,), device="cuda", generator=g)
    e = torch.randint(0, 96, (rows,), device="cuda", generator=g)
    c = torch.randint(0, 2, (rows,), device="cuda", generator=g)
    names = ["lidar", "pose", "velocity", "speed", "action", "reward", "discount", "progress_total", "time", "lidar_occupancy",
             "progress", "lap", "done", "fresh", "wall_collision", "steering_angle"]
    got = env.gather_rows(ring.buffer, ring.slot_bytes, slots, e * 2 + c, names)
    torch.cuda.synchronize()
    for n in names:
        assert torch.equal(got[n], ring.fields[n][slots, e, c]), n
    with pytest.raises(Exception, match="NULL argument|empty field mask"):
        env.gather_rows(ring.buffer, ring.slot_bytes, slots, e * 2 + c, [])
    # through the sampler: identical to the indexing form of the same draw
    g1, g2 = torch.Generator(device="cuda").manual_seed(9), torch.Generator(device="cuda").manual_seed(9)
    fields = ("lidar", "lidar_occupancy", "action", "reward", "discount", "time", "speed", "done", "fresh")
    a = ring.sample(768, 3, fields=fields, generator=g1, native=False)
    native = env.gather_rows
    try:
        del BatchedRaceEnv.gather_rows                  # the indexing path of the same sampler
        b = ring.sample(768, 3, fields=fields, generator=g2, native=False)
    finally:
        BatchedRaceEnv.gather_rows = native.__func__
    assert set(a) == set(b) and bool(a["terminal"].any())
    for n in a:
        assert torch.equal(a[n], b[n]), n
    env.close()


def test_windows_drawn_on_the_device_are_the_rows_they_say_and_uniform():
    """rc_sample_windows (the draw, the episode-boundary test and the row list in one launch) through TrajectoryRing.sample:
    every window is what plain indexing of the ring gives at its (t0, env, car) - terminal rows with the previous record's
    observation, reset rows at episode starts -, no window crosses an episode boundary, both draws of one seed differ, and the
    starts are spread evenly over ring age and cars."""
    import torch
    from racing_dreamer_amd.batched_env import BatchedRaceEnv
    from racing_dreamer_amd.replay import OBSERVATION_FIELDS, TrajectoryRing
    from oracle import racecar_oracle as ro
    env = BatchedRaceEnv("columbia", 128, 2, obs_type="lidar_occupancy", auto_reset=True, action_repeat=4)
    ring = TrajectoryRing(env, capacity=16)
    ring.reset(mode="random_ball", seed=1)
    for k in range(40):                                     # more records than slots: the ring has wrapped
        act = ro.random_actions(7, k, 256)
        act[:, 0] = 1.0
        ring.step(torch.from_numpy(act).cuda().view(128, 2, 2))
    g = torch.Generator(device="cuda").manual_seed(11)
    fields = ("lidar", "lidar_occupancy", "action", "reward", "discount", "time", "done", "fresh", "progress_total")
    length, batch = 5, 4096
    a = ring.sample(batch, length, fields=fields, generator=g)
    b = ring.sample(batch, length, fields=fields, generator=g)
    torch.cuda.synchronize()
    assert not torch.equal(a["t0"], b["t0"]) and a["lidar"].shape == (batch, length, 1080)
    oldest = (ring.head + 1) % 16
    slots = (oldest + a["t0"][:, None] + torch.arange(length, device="cuda")[None, :]) % 16
    e, c = a["env"][:, None], a["car"][:, None]
    fresh, done = ring.fields["fresh"][slots, e, c], ring.fields["done"][slots, e, c]
    assert not (fresh[:, 1:-1] != 0).any() and bool((((fresh[:, -1] == 0) | (done[:, -1] != 0))).all())
    term = (fresh[:, -1] != 0) & (done[:, -1] != 0)
    assert torch.equal(a["terminal"], term) and int(term.sum()) > 20
    first = fresh[:, 0] != 0
    assert int(first.sum()) > 20
    for name in fields:
        want = ring.fields[name][slots, e, c].clone()
        if name in OBSERVATION_FIELDS:
            want[term, -1] = want[term, -2]
        if name in ("action", "reward", "time"):
            want[first, 0] = 0.0
        if name == "discount":
            want[first, 0] = 1.0
        if name == "progress_total":
            want[first, 0] = -1.0
        assert torch.equal(a[name], want), name
    # uniform over (start, car) among the admissible windows: compare the start histogram with the admissible counts
    n_start = 16 - length + 1
    idx = (oldest + torch.arange(n_start, device="cuda")[:, None] + torch.arange(length, device="cuda")[None, :]) % 16   # [start, j]
    fr, dn = ring.fields["fresh"][idx], ring.fields["done"][idx]                  # [start, j, env, car]
    ok = ~(fr[:, 1:-1] != 0).any(1) & ((fr[:, -1] == 0) | (dn[:, -1] != 0))      # [start, env, car]
    share = ok.flatten(1).sum(1).float() / ok.sum()
    got = torch.bincount(a["t0"], minlength=n_start).float() / batch
    assert float((got - share).abs().max()) < 0.03, (got, share)
    assert bool(ok[a["t0"], a["env"], a["car"]].all())
    with pytest.raises(Exception, match="does not fit"):
        env.sample_windows(ring.buffer, ring.slot_bytes, 16, oldest, 16, 17, 4, 1, 1)
    # the same draw as ONE native call into ONE packed buffer (rc_sample_batch: the sampler, both kinds of row gather and
    # the reset rows in a memset and two launches): identical fields, identical meta
    ring._draws = 100
    ref = ring.finish_sample(ring.sample(batch, length, fields=fields, generator=g, defer=True))   # the three-launch path + torch fix-ups
    ring._draws = 100
    buf, lay = ring.sample_packed(batch, length, fields=fields, generator=g)
    pk = TrajectoryRing.unpack(buf, lay)
    torch.cuda.synchronize()
    assert lay["payload"] < lay["total"] == buf.numel() and int(pk["failed"]) == 0
    for name in fields:
        assert torch.equal(pk[name], ref[name]), name
    assert torch.equal(pk["meta"][:, 0].long(), ref["t0"]) and torch.equal(pk["meta"][:, 1].long(), ref["env"] * 2 + ref["car"])
    assert torch.equal(pk["meta"][:, 2] != 0, ref["terminal"]) and int((pk["meta"][:, 3] != 0).sum()) > 20
    raw, _ = ring.sample_packed(64, length, fields=("reward", "discount"), generator=g, reset_rows=False)
    rawv = TrajectoryRing.unpack(raw, env.sample_batch_layout(("reward", "discount"), 64, length))
    torch.cuda.synchronize()
    starts = rawv["meta"][:, 3] != 0
    assert bool(starts.any()) and bool((rawv["discount"][starts, 0] == 0.0).any())        # records as stored: the previous episode's terminal row
    with pytest.raises(Exception, match="output too small"):
        env.sample_batch(ring.buffer, ring.slot_bytes, 16, oldest, 16, lay, 1, 1, out=torch.empty(lay["total"] - 64, dtype=torch.uint8, device="cuda"))
    env.close()
