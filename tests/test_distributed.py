"""Multi-GPU host logic on CPU: env sharding and the trajectory all-gather over gloo, world_size 2."""
import json
import os
import subprocess
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_envs_partition():
    from racing_dreamer_amd.distributed import shard_envs
    for total, world in ((524288, 8), (65536, 1), (10, 3), (7, 7)):
        shards = [shard_envs(total, r, world) for r in range(world)]
        assert shards[0].first_env == 0 and sum(s.num_envs for s in shards) == total
        for a, b in zip(shards[:-1], shards[1:]):
            assert a.first_env + a.num_envs == b.first_env
        assert max(s.num_envs for s in shards) - min(s.num_envs for s in shards) <= 1
    with pytest.raises(ValueError):
        shard_envs(3, 0, 4)
    with pytest.raises(ValueError):
        shard_envs(8, 8, 8)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total_envs, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import make_oracle
    from oracle import racecar_oracle as ro
    from racing_dreamer_amd.distributed import TrajectoryGather, shard_envs, slab_field_views
    from racing_dreamer_amd.track_assets import synthetic_track
    track = synthetic_track()
    sh = shard_envs(total_envs, rank, world)
    env = make_oracle(track, num_envs=sh.num_envs, auto_reset=True, first_env=sh.first_env)
    env.reset(mode=ro.RESET_RANDOM, seed=4)

    def slab_of(out, n):       # same byte layout as rc_trajectory_slab (sections 64-byte aligned)
        parts = []
        for key in ("lidar", "pose", "velocity", "speed", "action", "reward", "discount", "progress_total", "time"):
            b = np.ascontiguousarray(out[key], np.float32).tobytes()
            parts.append(b + b"\0" * ((-len(b)) % 64))
        raw = b"".join(parts)
        return torch.frombuffer(bytearray(raw[:len(raw) - ((-len(b)) % 64)]), dtype=torch.uint8)

    gather = None
    last = None
    for k in range(3):
        act = ro.random_actions(1, k, sh.num_envs, first_car=sh.first_env)
        out = env.step(act)
        slab = slab_of(out, sh.num_envs)
        if gather is None:
            gather = TrajectoryGather(slab)
        gather.launch(slab)            # overlaps with the next step
        last = out
    g = gather.wait()
    # batched form: one collective per 2 snapshots; 3 launches = a full batch and a flushed partial one
    batched = TrajectoryGather(slab, every=2)
    marks = []
    for k in range(3):
        snap = slab.clone()
        snap[:8] = torch.arange(8, dtype=torch.uint8) + 16 * k + rank
        marks.append(snap)
        batched.launch(snap)
        if k == 1:
            gb = batched.wait()
            assert gb.shape[:2] == (world, 2)
            for r in range(world):
                for j in range(2):
                    assert int(gb[r, j, 0]) == 16 * j + r and (r != rank or torch.equal(gb[r, j, 8:], slab[8:]))
    gb = batched.wait()
    assert gb.shape[:2] == (world, 1) and all(int(gb[r, 0, 0]) == 32 + r for r in range(world))
    # in-place form (no staging copy) with a consumer: every batch is handed over exactly once, in order, before its
    # gathered buffer is reused (49 of 50 batches used to be unobservable)
    seen = []
    inplace = TrajectoryGather(slab, stage=False, consumer=lambda v: seen.append(v[:, 0].clone()))
    bufs = [slab.clone(), slab.clone()]
    for k in range(5):
        b = bufs[k % 2]
        b[0] = 40 + k + rank
        inplace.launch(b)
    final = inplace.wait()
    assert [t.tolist() for t in seen] == [[40 + k + r for r in range(world)] for k in range(5)]
    assert final[:, 0].tolist() == [44 + r for r in range(world)]
    with pytest.raises(ValueError):
        TrajectoryGather(slab, every=2, stage=False)
    # a deeper pipeline over a ring of sources (the N > 1 headline of bench.py: records stay in the rank's ring, `depth`
    # collectives may be in flight, the results of the last depth + 1 stay readable)
    deep = TrajectoryGather(slab, stage=False, depth=3)
    ring = [slab.clone() for _ in range(5)]
    for k in range(7):
        b = ring[k % 5]
        b[0] = 60 + k + rank
        deep.launch(b)
    deep.wait()
    assert [deep.recent(j)[:, 0].tolist() for j in range(4)] == [[66 - j + r for r in range(world)] for j in range(4)]
    with pytest.raises(IndexError):
        deep.recent(4)
    # staged AND deep (bench.py's ShardedCollector with summary_every > 1 builds exactly this): the staging
    # buffer a launch copies into is never the source of a collective still in flight, and every batch arrives whole
    got = []
    sd = TrajectoryGather(slab, every=2, stage=True, depth=2, consumer=lambda v: got.append(v[:, :, 0].clone()))
    assert len(sd.staging) == 3
    for k in range(11):
        snap = slab.clone()
        snap[0] = 100 + k + rank
        busy = {t.untyped_storage().data_ptr() for t in sd.pending_sources()}
        assert sd.staging[sd._cur].untyped_storage().data_ptr() not in busy
        sd.launch(snap)
        assert len(sd.pending_sources()) <= 2
    sd.wait()
    flat = [row for v in got for row in v.transpose(0, 1).tolist()]
    assert flat == [[100 + k + r for r in range(world)] for k in range(11)]
    views = [slab_field_views(g[r], sh.num_envs, False) for r in range(world)]
    lidar = torch.cat([v["lidar"] for v in views]).numpy()
    reward = torch.cat([v["reward"] for v in views]).numpy()
    mine = slice(sh.first_env, sh.first_env + sh.num_envs)
    ok = np.array_equal(lidar[mine], last["lidar"]) and np.array_equal(reward[mine], last["reward"])
    q.put((rank, ok, lidar, reward))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_equals_single_rank_run():
    """world_size 2 over gloo: the gathered buffer equals the unsharded job (global-id RNG streams)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import make_oracle
    from oracle import racecar_oracle as ro
    from racing_dreamer_amd.track_assets import synthetic_track
    total = 12
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = make_oracle(synthetic_track(), num_envs=total, auto_reset=True)
    ref.reset(mode=ro.RESET_RANDOM, seed=4)
    for k in range(3):
        out = ref.step(ro.random_actions(1, k, total))
    for rank, ok, lidar, reward in results:
        assert ok
        assert np.array_equal(lidar, out["lidar"]) and np.array_equal(reward, out["reward"])


def test_compact_and_summary_views_parse_the_documented_layout():
    """include/racecar_hip.h, rc_set_compact_slab: uint16 [n][1080], padding to 64 B, then POSE..TIME as in the arena."""
    from oracle import racecar_oracle as ro
    from racing_dreamer_amd.distributed import (compact_field_views, dequantise_lidar, gather_link_model,
                                                summary_field_views, SUMMARY_FIELDS)
    n = 37                                               # 37 * 2160 is not a multiple of 64: the padding matters
    rng = np.random.default_rng(0)
    lidar = rng.uniform(0, 15, (n, 1080)).astype(np.float32)
    lidar[0, :3] = [0.0, 15.0, 7.5]
    q = ro.quantise_lidar_u16(lidar)
    assert q.dtype == np.uint16 and q[0, 0] == 0 and q[0, 1] == 65535 and q[0, 2] == 32768   # 32767.5 -> even
    fields = {name: rng.standard_normal((n,) + tail).astype(np.float32) for name, _, _, tail in SUMMARY_FIELDS}
    parts = [q.tobytes()]
    parts[0] += b"\0" * ((-len(parts[0])) % 64)
    summary = b""
    for name, per_car, _, _ in SUMMARY_FIELDS:
        b = fields[name].tobytes()
        assert len(b) == per_car * n
        summary += b + b"\0" * ((-len(b)) % 64)
    buf = torch.frombuffer(bytearray(parts[0] + summary), dtype=torch.uint8)
    v = compact_field_views(buf, n)
    assert np.array_equal(v["lidar_u16"].numpy(), q)
    s = summary_field_views(torch.frombuffer(bytearray(summary), dtype=torch.uint8), n)
    for name in fields:
        assert np.array_equal(v[name].numpy(), fields[name]) and np.array_equal(s[name].numpy(), fields[name])
    back = dequantise_lidar(v["lidar_u16"]).numpy()
    assert np.abs(back - lidar).max() <= 15.0 / 65535 / 2 * 1.001
    assert sum(p for _, p, _, _ in SUMMARY_FIELDS) == 76
    # link model: the full fp32 record of 65 536 cars over 7 links
    m = gather_link_model(65536 * 4396, 8)
    assert m["inbound_bytes_per_gpu_per_step"] == 7 * 65536 * 4396 and 3.5 < m["link_bound_ms_per_step"] < 4.0
    assert gather_link_model(65536 * 4396, 1)["link_bound_ms_per_step"] == 0.0
    assert gather_link_model(65536 * 76, 2)["link_bound_ms_per_step"] == pytest.approx(65536 * 76 / 76.8e9 * 1e3)


def _replay_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from test_replay import FakeEnv
    from racing_dreamer_amd.replay import ShardedReplay, TrajectoryRing
    env = FakeEnv(num_envs=4, cars=1, fresh_at={(3, 1), (5, 2), (4, 3)})
    ring = TrajectoryRing(env, capacity=8)
    ring.reset()
    for _ in range(9):
        ring.step()
        ring.fields["lidar"][ring.head] += 100.0 * rank          # mark this rank's records
    g = torch.Generator().manual_seed(10 + rank)
    rep = ShardedReplay(ring)
    batch = rep.sample(batch=12, length=3, generator=g)
    # the view form carries the same windows' per-window integers, "starts an episode" included (it was all zeros)
    local = rep.draw(12, 3, generator=torch.Generator().manual_seed(10 + rank))
    views = rep.exchange(local, flat=False)
    assert views["meta"].shape == (world, 6, 4)
    assert torch.equal(views["meta"][rank, :, 3] != 0, local["first"]) and torch.equal(views["meta"][rank, :, 2] != 0, local["terminal"])
    assert torch.equal(views["meta"][..., 3] != 0, views["fresh"][:, :, 0] != 0)
    q.put((rank, {k: v.numpy() for k, v in batch.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_replay_gathers_the_training_batch_not_the_records():
    """each rank samples batch / world windows from its own ring; every rank ends up with the same global
    batch, rank r's windows in rows [r * batch / world, ...)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_replay_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = results[0], results[1]
    assert sorted(a) == sorted(b)
    for k in a:
        assert np.array_equal(a[k], b[k]), k                      # the same global batch on both ranks
    assert a["lidar"].shape == (12, 3, 8) and a["rank"].tolist() == [0] * 6 + [1] * 6
    assert np.array_equal(a["first"], a["fresh"][:, 0] != 0) and a["first"].any() and not a["first"].all()
    owner = (a["lidar"][:, 0, 0] >= 100.0).astype(int)              # rank 1 marked its records with + 100
    assert owner.tolist() == a["rank"].tolist()
    r = a["reward"]
    assert np.all(r[:, 2] - r[:, 1] == 1.0)                         # windows are consecutive steps ...
    assert np.all((r[:, 1] - r[:, 0] == 1.0) | a["first"]) and np.all(r[a["first"], 0] == 0.0)    # ... behind a reset row or not


def test_bench_starts_its_own_ranks_and_reports_their_failure():
    """`python bench.py --gpus 2` without a launcher: the process starts two ranks itself (before touching any GPU) and
    exits with their code.  On this GPU-less box both ranks fail at the first HIP call - which is the point: the failure
    of a rank reaches the caller as a non-zero exit, not as a hang or a silent success."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: tests/test_gpu_bench.py covers the successful run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
                        "--envs", "64", "--no-cpu-baseline"], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode not in (0, 2), r.stderr[-1500:]          # (2 was round 2's "needs torch.distributed.run")
    # (the launcher ends the second rank as soon as the first has failed: its own message may or may not have been printed)
    assert r.stderr.count("No HIP GPUs are available") >= 1 and "{" not in r.stdout


def _run_guard_snippet(body, timeout=60):
    code = ("import sys, time, json; sys.path.insert(0, %r); import bench\n" % ROOT) + body
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout)


def test_line_guard_prints_the_one_line_whatever_a_later_leg_does():
    """bench.py's LineGuard without a GPU: once armed with the headline, (a) a leg that raises at N = 1 is recorded and the run
    goes on, the final line carries `leg_errors` and the exit code says a leg was lost (5); (b) a leg that overruns its deadline
    ends the process with the line + `aborted`, exit 5; (c) before it is armed - the N = 1 headline leg itself - an exception is
    an exception; (d) a leg for which the time budget has no room is skipped and listed, which is not an error."""
    r = _run_guard_snippet('''
g = bench.LineGuard(0, 1, 30.0)
line = {"metric": "m", "value": 1.0}
g.arm(line)
with g.leg("configs"):
    raise RuntimeError("boom")
with g.leg("tracks"):
    line["tracks"] = [1, 2]
g.emit(shutdown_s=1.0)
time.sleep(5)            # a shutdown that hangs (a rank that never reaches the closing barrier): the process ends, one line
print("not reached")
''')
    assert r.returncode == 5, r.stderr
    out = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(out) == 1 and out[0]["value"] == 1.0 and out[0]["tracks"] == [1, 2]
    assert "boom" in out[0]["leg_errors"]["configs"] and "aborted" not in out[0] and "not reached" not in r.stdout
    r = _run_guard_snippet('''
g = bench.LineGuard(0, 1, 30.0)
g.arm({"metric": "m", "value": 1.5})
g.emit(shutdown_s=1.0)
time.sleep(5)            # the same hung shutdown behind a COMPLETE line: exit 0
''')
    assert r.returncode == 0 and json.loads(r.stdout)["value"] == 1.5, (r.stdout, r.stderr)
    r = _run_guard_snippet('''
g = bench.LineGuard(0, 1, 1.0)
g.arm({"metric": "m", "value": 2.0})
with g.leg("cpu_baseline"):
    time.sleep(30)
print("not reached")
''')
    assert r.returncode == 5 and "not reached" not in r.stdout, (r.stdout, r.stderr)
    out = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(out) == 1 and out[0]["value"] == 2.0 and out[0]["aborted"]["leg"] == "cpu_baseline" and "deadline" in out[0]["aborted"]["reason"]
    assert out[0]["aborted"]["exit_code"] == 5
    r = _run_guard_snippet('''
g = bench.LineGuard(0, 1, 30.0)
with g.leg("headline"):
    raise RuntimeError("the headline itself")
''')
    assert r.returncode != 0 and "the headline itself" in r.stderr and not r.stdout.strip()
    r = _run_guard_snippet('''
g = bench.LineGuard(0, 1, 30.0, deadline=time.time() + 40.0)
line = {"metric": "m", "value": 3.0}
g.arm(line)
assert g.go("quick", 5.0)
with g.leg("quick", kind="gather"):
    time.sleep(0.2)
assert not g.go("long", 60.0)              # 40 s left, 12 s of them reserved for leaving
assert 0.9 < g.budget() <= 30.0 and g.budget(500.0) <= 40.0 - g.RESERVE_S
g.emit()
g.done()
sys.exit(g.exit_code())
''')
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    assert "long" in out["legs_skipped"] and "leg_errors" not in out and "aborted" not in out


def test_a_provisional_line_is_printed_when_the_headline_never_comes():
    """VERDICT r5 #1b: armed with a PROVISIONAL line before the first collective, a rendezvous or headline leg that hangs or
    raises prints that line - `headline_pending`, `aborted` - and exits 3, a code of its own; SIGTERM prints it too."""
    r = _run_guard_snippet('''
g = bench.LineGuard(0, 1, 1.0)
g.install()
g.arm({"metric": "m", "value": 8.0}, pending=True)
with g.leg("rendezvous"):
    time.sleep(30)
''')
    assert r.returncode == 3, (r.stdout, r.stderr)
    out = json.loads(r.stdout)
    assert out["headline_pending"] is True and out["aborted"]["leg"] == "rendezvous" and out["aborted"]["exit_code"] == 3
    assert "BEFORE the headline" in out["aborted"]["note"]
    r = _run_guard_snippet('''
g = bench.LineGuard(0, 1, 30.0)
g.install()
g.arm({"metric": "m", "value": 8.0}, pending=True)
with g.leg("headline"):
    raise RuntimeError("first contact")
''')
    assert r.returncode == 3 and "first contact" in json.loads(r.stdout)["aborted"]["reason"], (r.stdout, r.stderr)
    r = _run_guard_snippet('''
import os, signal
g = bench.LineGuard(0, 1, 30.0)
g.install()
g.arm({"metric": "m", "value": 8.0}, pending=True)
g.promote({"metric": "m", "value": 9.0})
with g.leg("full"):
    os.kill(os.getpid(), signal.SIGTERM)
    time.sleep(30)
''')
    assert r.returncode == 128 + 15, (r.stdout, r.stderr)
    out = json.loads(r.stdout)
    assert out["value"] == 9.0 and "headline_pending" not in out and out["aborted"]["leg"] == "signal"


def _pids_alive(pids):
    alive = []
    for p in pids:
        try:
            with open(f"/proc/{p}/stat") as f:
                if f.read().rsplit(")", 1)[1].split()[0] != "Z":
                    alive.append(p)
        except OSError:
            pass
    return alive


@pytest.mark.parametrize("how", ["SIGTERM", "SIGKILL"])
def test_no_rank_survives_its_launcher_and_the_line_is_printed(how, tmp_path):
    """VERDICT r5 #1a, #1d: `python bench.py --gpus 2` (the self-launcher, as the driver starts it) with both ranks hung in the
    headline leg - the first collective on the data path.  SIGTERM to the LAUNCHER alone (what a driver's time-out sends):
    it forwards the signal to the ranks' session, rank 0 prints the ONE provisional line, every process of the tree is gone
    when the launcher returns, and the exit code is 128 + 15.  SIGKILL to the launcher (no handler can run): the kernel's
    parent-death signal ends the launcher child and the ranks all the same.  No GPU: the local stage is a stand-in
    (RC_BENCH_FAKE_LOCAL), the rendezvous over gloo is real."""
    import signal
    import subprocess
    import sys
    import time
    import psutil
    out, err = tmp_path / "out", tmp_path / "err"
    env = dict(os.environ, RC_BENCH_FAKE_LOCAL="1", RC_BENCH_HANG_LEG="headline")
    with open(out, "w") as fo, open(err, "w") as fe:
        p = subprocess.Popen([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--steps", "4", "--warmup", "1", "--envs", "64"],
                             cwd=ROOT, env=env, stdout=fo, stderr=fe)
        try:
            t_end = time.time() + 240
            while time.time() < t_end and "ranks joined" not in err.read_text():      # both ranks are through the rendezvous
                assert p.poll() is None, err.read_text()[-2000:]
                time.sleep(0.5)
            assert "ranks joined" in err.read_text(), err.read_text()[-2000:]
            time.sleep(1.0)
            tree = [c.pid for c in psutil.Process(p.pid).children(recursive=True)]
            assert len(tree) >= 2                                 # the two ranks (and whatever they started)
            p.send_signal(getattr(signal, how))
            rc = p.wait(timeout=90)
            t_end = time.time() + 30
            while time.time() < t_end and _pids_alive(tree):
                time.sleep(0.25)
            assert _pids_alive(tree) == [], f"survivors of {how}: {_pids_alive(tree)}"
        finally:
            for q in psutil.Process().children(recursive=True):
                q.kill()
    lines = [l for l in out.read_text().splitlines() if l.startswith("{")]
    if how == "SIGTERM":
        assert rc == 128 + 15, err.read_text()[-2000:]
    assert len(lines) == 1, (out.read_text()[-2000:], err.read_text()[-2000:])
    d = json.loads(lines[0])
    assert d["headline_pending"] is True and d["n_gpus"] == 2 and d["aborted"]["leg"] == "signal" and "provisional" in d
    assert {"metric", "value", "unit", "steps", "warmup", "ms_per_step", "config"} <= set(d)


def test_a_headline_that_hangs_costs_a_deadline_not_the_driver_s_time_out():
    """The same two ranks hung in the headline leg, nobody signalling: the leg's own deadline (here 5 s) prints the provisional
    line with `aborted` naming the leg and every rank exits 3; the launcher hands the failure on."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--steps", "4", "--warmup", "1", "--envs", "64",
                        "--leg-timeout", "5"], cwd=ROOT, capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, RC_BENCH_FAKE_LOCAL="1", RC_BENCH_HANG_LEG="headline"))
    assert r.returncode == 3, r.stderr[-2000:]                 # (the ranks' own exit code reaches the caller as it is)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.stdout[-2000:], r.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["headline_pending"] is True and d["aborted"]["leg"] == "headline" and d["aborted"]["exit_code"] == 3
    assert "deadline of 5 s" in d["aborted"]["reason"] and "exit 3" in r.stderr
