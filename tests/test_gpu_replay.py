"""TrajectoryRing on the device: the kernels write step records straight into ring slots (rc_set_arena)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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
    assert not (batch["fresh"][:, 1:] != 0).any()
    t = batch["time"]
    assert torch.allclose(t[:, 1:] - t[:, :-1], torch.full_like(t[:, 1:], 0.04), atol=1e-5)   # consecutive agent steps
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
