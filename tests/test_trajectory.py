"""EpisodeRecorder writes the same episode records as the reference's Collect + save_episodes
(dreamer/wrappers.py:210-250, dreamer/callbacks.py:41-53): keys, dtypes, reset row, discount, file name."""
import os

import numpy as np
import torch

from helpers import make_oracle
from oracle import racecar_oracle as ro
from racing_dreamer_amd.track_assets import synthetic_track
from racing_dreamer_amd.trajectory import EpisodeRecorder, count_steps

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "wrappers_golden.npz"))


def _views(out, B, A):
    v = {}
    for k, a in out.items():
        a = np.asarray(a)
        t = torch.from_numpy(a.reshape(B, A, *a.shape[1:]).copy())
        v[k] = t.unsqueeze(-1) if k == "lidar_occupancy" else t
    return v


def test_episode_files_match_the_reference_record(tmp_path):
    B = 6
    env = make_oracle(synthetic_track(), num_envs=B, auto_reset=True, render_occupancy=True, time_limit_steps=9)
    rec = EpisodeRecorder(B, 1, env_indices=[0, 3, 5], directory=str(tmp_path))
    rec.on_reset(_views(env.reset(mode=ro.RESET_RANDOM, seed=2), B, 1))
    episodes, n_steps = [], 0
    for k in range(30):
        act = ro.random_actions(7, k, B)
        act[:, 0] = 1.0
        episodes += rec.on_step(_views(env.step(act, repeat=4), B, 1))
        n_steps += 1
    assert len(episodes) >= 6                       # time limit 9 -> every recorded env finishes 3 episodes
    golden_keys = sorted(k[6:] for k in G.files if k.startswith("g5_ep_")) + ["lidar_occupancy"]
    for ep in episodes:
        assert sorted(ep) == sorted(golden_keys)
        T = len(ep["reward"]) - 1
        assert 1 <= T <= 9
        for k in ep:
            want = np.uint8 if k == "lidar_occupancy" else np.float32     # Collect._convert, precision 32
            assert ep[k].dtype == want and len(ep[k]) == T + 1, k
        assert ep["lidar"].shape == (T + 1, 1080) and ep["lidar_occupancy"].shape == (T + 1, 64, 64, 1)
        assert ep["pose"].shape == (T + 1, 6) and ep["action"].shape == (T + 1, 2)
        # reset row (wrappers.py:232-236, 74, 413)
        assert ep["progress"][0] == -1.0 and ep["discount"][0] == 1.0 and ep["reward"][0] == 0.0
        assert ep["time"][0] == 0.0 and ep["speed"][0] == 0.0 and not ep["action"][0].any()
        assert not ep["lidar_occupancy"][0].any()
        assert ep["discount"][-1] == 0.0 and np.all(ep["discount"][:-1] == 1.0)
        assert np.allclose(np.diff(ep["time"][1:]), 0.04, atol=1e-6)      # 4 sub-steps of dt per transition
    files = sorted(os.listdir(tmp_path))
    assert len(files) == len(episodes)
    for f in files:                                  # {timestamp}-{uuid}-{length}.npz (callbacks.py:49)
        ts, uid, length = f[:-4].split("-")
        assert len(ts) == 15 and len(uid) == 32
        with np.load(tmp_path / f) as d:
            assert int(length) == len(d["reward"]) and sorted(d.files) == sorted(golden_keys)
    assert count_steps(tmp_path) == sum(len(e["reward"]) - 1 for e in episodes)


def test_recorder_without_auto_reset_keeps_the_terminal_observation():
    env = make_oracle(synthetic_track(), num_envs=2, time_limit_steps=4)
    rec = EpisodeRecorder(2, 1, env_indices=[1])
    rec.on_reset(_views(env.reset(), 2, 1))
    eps = []
    for k in range(4):
        out = env.step(np.array([[0.5, 0.0], [0.5, 0.1]], np.float32))
        eps += rec.on_step(_views(out, 2, 1))
    assert len(eps) == 1 and len(eps[0]["reward"]) == 5
    assert np.array_equal(eps[0]["lidar"][-1], out["lidar"][1])           # terminal observation kept
