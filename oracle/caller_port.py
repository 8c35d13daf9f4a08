"""Restatement of the two CALLERS either side of the env in the reference's Dreamer pipeline - TEST INFRASTRUCTURE,
NOT PRODUCT.  `dreamer/tools.py` cannot be imported here (module-level TensorFlow imports, tools.py:10-15), so the
rollout driver and the dataset reader it defines are restated in plain Python for the tests that exercise the drop-in
env from the caller's side:

  rollout ............... tools.simulate          dreamer/tools.py:154-206  (TensorBoard summaries dropped, :202-204)
  count_episode_files ... tools.count_episodes    dreamer/tools.py:224-228
  episode_windows ....... tools.load_episodes     dreamer/tools.py:235-264

Semantics kept from the reference, each one asserted by tests/test_caller_loop.py:
  * the env is reset whenever ANY agent reported done on the previous step (tools.py:177-178), and on entry when no
    state is passed in (all `dones` start True, tools.py:168);
  * every observation value reaches the policy with a leading batch dimension of 1 (`np.stack([v])`, tools.py:184)
    and the policy's first output row is the action (`actions[id][0]`, tools.py:189);
  * `length` counts the agent steps of the running episode, is added to `step` when the episode ends and cleared
    (tools.py:198-200); the loop runs until `step >= steps` or `episode >= episodes` (tools.py:175);
  * the first agent's statistic per step is `lap + progress - 1` (tools.py:195), its maximum per episode and the
    episode return are collected when the NEXT reset happens (tools.py:179-182);
  * the returned state is `(step - steps, episode - episodes, dones, length, obs, agent_states)` (tools.py:206);
  * the dataset reader draws `rescan` file indices, then a window start in `[0, total - length]` inclusive
    (`randint(0, available + 1)`), or `min(randint(0, total), available)` with `balance`, skips episodes with
    `total - length < 1`, and slices EVERY key by `[index:index + length]` (tools.py:248-263).
"""
from __future__ import annotations

import pathlib

import numpy as np


def rollout(policies, env, agent_ids=("A",), steps=0, episodes=0, state=None):
    """Drive `env` (dict-keyed multi-agent API) with one policy per agent until `steps` agent steps of finished
    episodes or `episodes` episodes have been collected.  policy(obs_batch, done_batch, policy_state) ->
    (action_batch, policy_state).  Returns (resume_state, stats)."""
    ids = list(agent_ids)
    lead = ids[0]
    finished_progress, finished_returns = [], []
    ep_progress, ep_return = [], 0.0
    if state is None:
        step = episode = 0
        dones = {a: True for a in ids}
        length = np.zeros(len(ids), np.int32)
        obs = {a: None for a in ids}
        pol_state = {a: None for a in ids}
    else:
        step, episode, dones, length, obs, pol_state = state
    n_resets = n_env_steps = 0
    while (steps and step < steps) or (episodes and episode < episodes):
        if any(dones.values()):
            obs = env.reset()
            n_resets += 1
            if ep_progress:
                finished_progress.append(max(ep_progress))
                finished_returns.append(ep_return)
            ep_return = 0.0
        batched = {a: {k: np.stack([v]) for k, v in o.items()} for a, o in obs.items()}
        actions = {}
        for i, a in enumerate(ids):
            out, pol_state[a] = policies[i](batched[a], np.stack([dones[a]]), pol_state[a])
            actions[a] = np.array(out[0])
        obs, rewards, dones, infos = env.step(actions)
        n_env_steps += 1
        ep_return = ep_return + rewards[lead]
        ep_progress.append(infos[lead]["lap"] + infos[lead]["progress"] - 1)
        over = any(dones.values())
        episode += int(over)
        length += 1
        step += (int(over) * length).sum()
        length *= (1 - over)
    stats = {"progress": finished_progress, "return": finished_returns, "resets": n_resets, "env_steps": n_env_steps}
    return (step - steps, episode - episodes, dones, length, obs, pol_state), stats


def count_episode_files(directory):
    """(episodes, steps) from the `...-{rows}.npz` names: rows - 1 transitions per file."""
    rows = [int(p.stem.rsplit("-", 1)[-1]) - 1 for p in pathlib.Path(directory).glob("*.npz")]
    return len(rows), sum(rows)


def episode_windows(directory, rescan, length=None, balance=False, seed=0, rounds=1):
    """`rounds` passes of the reference's (endless) generator: yields dicts key -> array[length, ...]."""
    directory = pathlib.Path(directory).expanduser()
    rng = np.random.RandomState(seed)
    cache = {}
    for _ in range(rounds):
        for path in directory.glob("*.npz"):
            if path not in cache:
                with path.open("rb") as f:
                    data = np.load(f)
                    cache[path] = {k: data[k] for k in data.keys()}
        names = list(cache.keys())
        for pick in rng.choice(len(names), rescan):
            episode = cache[names[pick]]
            if length:
                total = len(next(iter(episode.values())))
                available = total - length
                if available < 1:
                    continue
                start = min(rng.randint(0, total), available) if balance else int(rng.randint(0, available + 1))
                episode = {k: v[start:start + length] for k, v in episode.items()}
            yield episode
