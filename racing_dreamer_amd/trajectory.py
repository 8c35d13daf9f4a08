"""Episode store compatible with the reference's `Collect` wrapper + `callbacks.save_episodes`
(SURVEY.md §8f N1).

The reference records, per agent, one transition per agent step - the observation dict plus
``action, reward, discount = 1 - done, progress = lap + progress - 1, time`` - starting with a reset row
(action 0, reward 0, discount 1, progress -1, time 0), casts everything to float32 / int32 / uint8 at episode
end (dreamer/wrappers.py:210-250) and writes ``{timestamp}-{uuid}-{length}.npz`` with those keys
(dreamer/callbacks.py:41-53), which `tools.load_episodes` then samples (dreamer/tools.py:235-264).

`EpisodeRecorder` does the same for a chosen subset of the batched env's cars: after every
`BatchedRaceEnv.reset()/step()` it pulls those cars' record fields out of the output arena (one small
device gather + one host copy), appends them, and on `done` emits the episode dict / file.  It reads the
views only, so it works on any mapping of field name -> tensor [num_envs, cars_per_env, ...].
"""
from __future__ import annotations

import datetime
import pathlib
import uuid
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

OBS_KEYS = ("lidar", "pose", "velocity", "speed", "lidar_occupancy")


class EpisodeRecorder:
    def __init__(self, num_envs: int, cars_per_env: int, env_indices: Sequence[int], car: int = 0,
                 directory: Optional[str] = None, callbacks: Sequence[Callable[[List[Dict[str, np.ndarray]]], None]] = ()):
        self.env_indices = torch.as_tensor(list(env_indices), dtype=torch.long)
        if self.env_indices.numel() and int(self.env_indices.max()) >= num_envs:
            raise ValueError("env index out of range")
        self.car = int(car)
        if not 0 <= self.car < cars_per_env:
            raise ValueError("car index out of range")
        self.directory = pathlib.Path(directory).expanduser() if directory else None
        self.callbacks = list(callbacks)
        self._rows: List[List[Dict[str, np.ndarray]]] = [[] for _ in range(len(self.env_indices))]
        self.episodes_written = 0

    def _pull(self, views: Dict[str, torch.Tensor], keys) -> Dict[str, np.ndarray]:
        out = {}
        for k in keys:
            if k in views:
                idx = self.env_indices.to(views[k].device)
                out[k] = views[k].index_select(0, idx)[:, self.car].cpu().numpy()
        return out

    def on_reset(self, views: Dict[str, torch.Tensor], mask=None) -> None:
        """Call after env.reset(): starts a new episode for the recorded envs that were reset."""
        obs = self._pull(views, OBS_KEYS)
        m = np.ones(len(self._rows), bool) if mask is None else np.asarray(mask, bool)[self.env_indices.numpy()]
        for i in np.nonzero(m)[0]:
            row = {k: v[i] for k, v in obs.items()}
            row["speed"] = np.float32(0.0)                                   # wrappers.py:74
            if "lidar_occupancy" in row:
                row["lidar_occupancy"] = np.zeros_like(row["lidar_occupancy"])   # wrappers.py:413
            row.update(action=np.zeros(2, np.float32), reward=np.float32(0.0), discount=np.float32(1.0),
                       progress=np.float32(-1.0), time=np.float32(0.0))      # wrappers.py:232-236
            self._rows[i] = [row]

    def on_step(self, views: Dict[str, torch.Tensor]) -> List[Dict[str, np.ndarray]]:
        """Call after env.step(): appends one transition per recorded env; returns the episodes that ended.

        With `auto_reset` the observation of a finished env already belongs to the next episode; it becomes
        that episode's reset row (the terminal observation is replaced, as in vectorised gym envs)."""
        rec = self._pull(views, OBS_KEYS + ("action", "reward", "discount", "progress_total", "time", "done", "fresh"))
        finished = []
        for i in range(len(self._rows)):
            if not self._rows[i]:
                continue                                                      # not started (reset not seen)
            row = {k: rec[k][i] for k in OBS_KEYS if k in rec}
            row.update(action=rec["action"][i], reward=rec["reward"][i], discount=rec["discount"][i],
                       progress=rec["progress_total"][i], time=rec["time"][i])
            done = bool(rec["done"][i])
            fresh = bool(rec["fresh"][i]) if "fresh" in rec else False
            if done and fresh:            # auto-reset: keep the terminal scalars, drop the new episode's observation
                prev = self._rows[i][-1]
                for k in OBS_KEYS:
                    if k in row:
                        row[k] = prev[k]
            self._rows[i].append(row)
            if done:
                ep = {k: np.stack([r[k] for r in self._rows[i]]).astype(
                    np.uint8 if k == "lidar_occupancy" else np.float32) for k in self._rows[i][0]}
                finished.append(ep)
                self._rows[i] = []
                if fresh:                 # the env was auto-reset: its new first observation starts the next episode
                    first = {k: rec[k][i] for k in OBS_KEYS if k in rec}
                    first["speed"] = np.float32(0.0)
                    first.update(action=np.zeros(2, np.float32), reward=np.float32(0.0), discount=np.float32(1.0),
                                 progress=np.float32(-1.0), time=np.float32(0.0))
                    self._rows[i] = [first]
        if finished:
            for cb in self.callbacks:
                cb(finished)
            if self.directory is not None:
                save_episodes(self.directory, finished)
            self.episodes_written += len(finished)
        return finished


def save_episodes(directory, episodes) -> List[pathlib.Path]:
    """Write each episode dict as one compressed `.npz` the reference's dataset loader picks up.

    What `dreamer/tools.py` relies on is only the file format: `load_episodes` globs `*.npz` and reads every key
    (tools.py:240-246), `count_episodes` takes the number of steps from the file name's last `-`-separated field
    (tools.py:224-228: `int(stem.rsplit('-', 1)[-1]) - 1`).  Names are `{timestamp}-{unique id}-{rows}.npz` like the
    reference's writer (dreamer/callbacks.py:41-53); the file is written under a temporary name and renamed, so a
    loader that rescans the directory while a rollout is running never sees a partial archive."""
    out_dir = pathlib.Path(directory).expanduser()
    out_dir.mkdir(parents=True, exist_ok=True)
    stamp = datetime.datetime.now().strftime("%Y%m%dT%H%M%S")
    written = []
    for ep in episodes:
        rows = int(np.shape(ep["reward"])[0])
        final = out_dir / f"{stamp}-{uuid.uuid4().hex}-{rows}.npz"
        partial = final.with_suffix(".npz.part")
        with open(partial, "wb") as fh:
            np.savez_compressed(fh, **ep)
        partial.replace(final)
        written.append(final)
    return written


def count_steps(directory) -> int:
    """dreamer/tools.py:231-232: total steps on disk from the `-{length}.npz` suffixes (reset rows excluded)."""
    return sum(int(str(n).split("-")[-1][:-4]) - 1 for n in pathlib.Path(directory).glob("*.npz"))
