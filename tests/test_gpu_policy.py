"""G10 on the device: the reference's trained austria agent (tests/test_golden_policy.py) drives the HIP env."""
import numpy as np
import pytest

from oracle import racecar_oracle as ro
from oracle.dreamer_policy_port import DreamerPolicy
from test_golden_policy import c_env, drive, weights

pytestmark = pytest.mark.gpu


class DeviceEnv:
    """The oracle's reset / step interface over BatchedRaceEnv (NumPy in, NumPy out)."""
    MODES = {ro.RESET_GRID: "grid", ro.RESET_RANDOM: "random"}

    def __init__(self, track, n):
        from racing_dreamer_amd.batched_env import BatchedRaceEnv
        self.env = BatchedRaceEnv(track, n, 1, auto_reset=True, remap_actions=True)

    def _out(self, views):
        import torch
        torch.cuda.synchronize()
        return {k: views[k].cpu().numpy() for k in ("lidar", "fresh", "wall_collision", "speed", "lap", "progress", "pose")}

    def reset(self, mode, seed):
        return self._out(self.env.reset(mode=self.MODES[mode], seed=seed))

    def step(self, actions, repeat):
        import torch
        return self._out(self.env.step(torch.from_numpy(np.ascontiguousarray(actions, np.float32)).cuda(), repeat=repeat))


def test_reference_agent_on_the_device_is_the_oracle_run_step_for_step():
    """The deterministic form of the agent closes the loop over the HIP env and over the C oracle: bit-identical scans give
    identical commands give identical poses, 300 agent steps long - and no car touches a wall."""
    n = 16
    dev, ora = DeviceEnv("austria", n), c_env("austria", n)
    pd, po = (DreamerPolicy(weights("austria"), sample=False) for _ in range(2))
    od, oo = dev.reset(ro.RESET_GRID, 1), ora.reset(mode=ro.RESET_GRID, seed=1)
    sd, so = pd.initial(n), po.initial(n)
    crashes = 0
    for k in range(300):
        assert np.array_equal(od["lidar"].reshape(n, -1), np.asarray(oo["lidar"]).reshape(n, -1)), f"scan differs at agent step {k}"
        ad, sd = pd.act(od["lidar"].reshape(n, -1), sd)
        ao, so = po.act(np.asarray(oo["lidar"]).reshape(n, -1), so)
        assert np.array_equal(ad, ao)
        od, oo = dev.step(ad, 4), ora.step(ao, repeat=4)
        assert np.array_equal(od["pose"].reshape(n, 6), np.asarray(oo["pose"]).reshape(n, 6))
        crashes += int(np.count_nonzero(od["wall_collision"]))
    assert crashes == 0 and float(od["speed"].mean()) > 3.0
    dev.env.close()


@pytest.mark.gpu_slow          # (22 s, most of it the NumPy policy: `pytest -m gpu_slow`; the closed loop above stays in `-m gpu`)
def test_reference_agent_drives_a_thousand_cars_on_the_device():
    n = 1024
    dev = DeviceEnv("austria", n)
    crashes, speed, laps = drive(dev, DreamerPolicy(weights("austria"), sample=False), n, 300, mode=ro.RESET_RANDOM)
    # 307 200 agent steps (3.4 hours of driving) from random poses on the centre line: fewer than one wall contact per 5 000
    # agent steps (26 when measured: one per 8 minutes); mirrored, half the cars hit a wall within 2.4 s
    assert crashes * 5000 <= n * 300 and speed > 3.0, (crashes, speed)
    crashes_m, _, _ = drive(dev, DreamerPolicy(weights("austria"), sample=False), n, 60, mode=ro.RESET_RANDOM, mirror=True)
    assert crashes_m >= n // 2, crashes_m
    dev.env.close()


def test_g12_the_references_test_protocol_on_the_device():
    """G12 on the HIP env (tests/test_golden_policy.py has the C-oracle run and the published numbers): the austria agent under
    dreamer/dream.py's test protocol - grid start, action_repeat 4, 1000 agent steps = 40 s, max_progress scenario - covers
    1.31 x [0.85, 1.35] laps without touching a wall; the deterministic agent's episode is the C oracle's, bit for bit."""
    from test_golden_policy import G12_BAND, PUBLISHED_DREAMER, _protocol
    ep = _protocol()
    n = 8
    a = ep.run_episodes("austria", "austria", n, repeat=4, max_agent_steps=1000, laps=10, backend="hip")
    assert (a["ended"] == "limit").all(), a["ended"]
    ratio = a["progress"].mean() / PUBLISHED_DREAMER["austria"]
    assert G12_BAND[0] <= ratio <= G12_BAND[1], (a["progress"], ratio)
    d = ep.run_episodes("austria", "austria", 4, repeat=4, max_agent_steps=1000, laps=10, backend="hip", sample=False)
    c = ep.run_episodes("austria", "austria", 4, repeat=4, max_agent_steps=1000, laps=10, backend="c", sample=False)
    assert np.array_equal(d["progress"], c["progress"]) and np.array_equal(d["time"], c["time"])
