"""The multi-GPU leg on one GPU: what the trajectory gather carries, checked against the CPU oracle.

  * the half-size record (uint16 LiDAR written by the scan's store path + the 76 B/car summary) against the oracle's
    quantiser, bit for bit;
  * `rc_gather_trajectory` - RCCL bound at run time inside the C-ABI library - with a one-rank communicator, every mode;
  * two ranks sharing the GPU (gloo between them: RCCL wants one GPU per rank), each stepping its own HIP shard with
    global-id RNG streams and gathering `full`, `full-u16` and `summary` records: every rank's gathered buffer, parsed
    with the public view helpers, equals the oracle's UNSHARDED run field by field (dreamer/wrappers.py:213-219 record).
"""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORD = ("pose", "velocity", "speed", "action", "reward", "discount", "progress_total", "time")


def _oracle_rollout(track_name, total, steps, repeat, seed=4, act_seed=1):
    from helpers import make_oracle
    from oracle import racecar_oracle as ro
    from racing_dreamer_amd.track_assets import load_track
    ref = make_oracle(load_track(track_name), num_envs=total, auto_reset=True)
    ref.reset(mode=ro.RESET_RANDOM, seed=seed)
    outs = []
    for k in range(steps):
        outs.append(ref.step(ro.random_actions(act_seed, k, total), repeat=repeat))
    return outs


@pytest.mark.parametrize("transform", ["metres", "dreamer", "unit"])
def test_compact_record_is_the_quantised_scan_plus_the_summary(transform):
    import torch
    from oracle import racecar_oracle as ro
    from racing_dreamer_amd.batched_env import BatchedRaceEnv, LIDAR_TRANSFORMS
    from racing_dreamer_amd.distributed import compact_field_views, dequantise_lidar
    n = 300                                   # split > 1 path of the scan
    for n_envs in (n, 3000):                  # ... and one wave per car
        env = BatchedRaceEnv("columbia", n_envs, 1, auto_reset=True, lidar_transform=transform)
        env.enable_compact(buffers=2)
        env.reset(mode="random", seed=2)
        for k in range(3):
            env.step_random(seed=5, step=k, repeat=2)
            done = env.rotate_compact()
        env.sync()
        v = compact_field_views(done, n_envs)
        lidar = env.views["lidar"].cpu().numpy().reshape(n_envs, 1080)
        want = ro.quantise_lidar_u16(lidar, LIDAR_TRANSFORMS[transform])
        got = v["lidar_u16"].cpu().numpy()
        assert got.dtype == np.uint16 and np.array_equal(got, want)
        assert want.max() > 40000 and want.min() < 3000                      # the code range is used
        back = dequantise_lidar(v["lidar_u16"], transform).cpu().numpy()
        assert np.abs(back - lidar).max() <= (15.0 if transform == "metres" else 1.0) / 65535 * 0.5001 + 1e-6
        for name in RECORD:
            assert torch.equal(v[name], env.views[name].reshape(v[name].shape)), name
        # the slab that was filled one step earlier still holds that step's record (nothing wrote into it since)
        assert env.compact.data_ptr() != done.data_ptr()
        env.disable_compact()
        env.step_random(seed=5, step=9)
        env.sync()
        assert torch.equal(compact_field_views(done, n_envs)["reward"], v["reward"])
        env.close()


def test_compact_slab_needs_the_default_scan():
    from racing_dreamer_amd import _lib as L
    from racing_dreamer_amd.batched_env import BatchedRaceEnv
    env = BatchedRaceEnv("columbia", 64, 1)
    env.enable_compact(1)
    with pytest.raises(L.RacecarHipError, match="variant 7 only"):
        env.set_raycast_variant(5)
    env.disable_compact()
    env.set_raycast_variant(5)
    with pytest.raises(L.RacecarHipError, match="default scan only"):
        env.enable_compact(1)
    with pytest.raises(L.RacecarHipError, match="enable_compact"):
        env.gather_source("full-u16")
    env.close()


def test_rc_gather_trajectory_over_rccl_with_one_rank():
    """The C-ABI collective itself: RCCL is dlopen'ed by the library, communicator of one rank, all three payloads."""
    import torch
    from racing_dreamer_amd import _lib as L
    from racing_dreamer_amd.batched_env import BatchedRaceEnv
    env = BatchedRaceEnv("austria", 512, 1, auto_reset=True)
    with pytest.raises(L.RacecarHipError, match="rc_comm_init has not been called"):
        env.gather("full", torch.zeros(8, dtype=torch.uint8, device="cuda"))
    env.comm_init(BatchedRaceEnv.comm_unique_id(), 0, 1)
    assert env.comm_count() == 1                           # ncclCommCount of the handle's own communicator
    env.enable_compact(2)
    env.reset(mode="random", seed=1)
    dst = {m: torch.zeros(env.gather_bytes(m), dtype=torch.uint8, device="cuda") for m in ("full", "full-u16", "summary")}
    assert env.gather_bytes("full") == env.slab.numel() and env.gather_bytes("summary") == env.summary_slab.numel()
    assert env.gather_bytes("full-u16") == env.compact.numel() < 0.52 * env.gather_bytes("full")
    for k in range(3):
        env.step_random(seed=3, step=k)
        for m in dst:
            env.gather(m, dst[m])
        env.gather_wait(host_sync=True)
        for m in dst:
            assert torch.equal(dst[m], env.gather_source(m)), (k, m)
    with pytest.raises(L.RacecarHipError, match="destination too small"):
        env.gather("full", dst["summary"])
    env.close()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank(rank, world, port, total, steps, repeat, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from racing_dreamer_amd.batched_env import BatchedRaceEnv
        from racing_dreamer_amd.distributed import TrajectoryGather, shard_envs
        sh = shard_envs(total, rank, world)
        env = BatchedRaceEnv("columbia", sh.num_envs, 1, auto_reset=True, first_env=sh.first_env)   # the HIP env
        env.enable_compact(buffers=2)
        env.reset(mode="random", seed=4)
        gathers = {m: TrajectoryGather(env.gather_source(m)) for m in ("full", "full-u16", "summary")}
        got = {m: [] for m in gathers}
        for k in range(steps):
            env.step_random(seed=1, step=k, repeat=repeat)
            for m, g in gathers.items():
                g.launch(env.gather_source(m))
                got[m].append(g.wait().cpu().numpy().copy())
            env.rotate_compact()
        q.put((rank, sh.num_envs, got))
        env.close()
    except Exception as e:          # noqa: BLE001
        q.put((rank, -1, repr(e)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_hip_shards_gather_the_unsharded_oracle_run():
    import torch
    import torch.multiprocessing as mp
    from oracle import racecar_oracle as ro
    from racing_dreamer_amd.distributed import compact_field_views, slab_field_views, summary_field_views
    total, steps, repeat, world = 96, 3, 2, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, world, port, total, steps, repeat, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = _oracle_rollout("columbia", total, steps, repeat)
    for rank, n, got in results:
        assert n == total // world, got
        for k in range(steps):
            o = want[k]
            parse = {"full": lambda b: slab_field_views(b, n, False), "full-u16": lambda b: compact_field_views(b, n),
                     "summary": lambda b: summary_field_views(b, n)}
            for mode, fn in parse.items():
                g = torch.from_numpy(got[mode][k])                  # [world, bytes]: rank r's record at row r
                assert g.shape[0] == world
                views = [fn(g[r]) for r in range(world)]
                for name in RECORD:
                    cat = torch.cat([v[name] for v in views]).numpy()
                    assert np.array_equal(cat, np.asarray(o[name], np.float32).reshape(cat.shape)), (rank, k, mode, name)
                if mode == "full":
                    lidar = torch.cat([v["lidar"] for v in views]).numpy()
                    assert np.array_equal(lidar, o["lidar"]), (rank, k)
                if mode == "full-u16":
                    q16 = torch.cat([v["lidar_u16"] for v in views]).numpy()
                    assert np.array_equal(q16, ro.quantise_lidar_u16(o["lidar"])), (rank, k)


P2P_MODES = ("full-u16", "full", "summary")


def _p2p_rank(rank, world, port, total, steps, repeat, q):
    """One rank of the peer-copy gather: ONE env, one continuous rollout, the record of every step sent with
    rc_gather_trajectory_p2p - hipIpc handles exchanged once (through gloo here: the library does not care how), then per
    step one copy per peer into the peer's own buffer - `steps` steps with each payload in turn (the payload is switched
    in mid-run, as bench.py's legs do: same buffers, same export, sequence numbers run on).  The fp32 / summary payloads are
    read in place from ALTERNATING arenas: the source of gather k must survive step k + 1 (what ADVICE r2 found missing in
    the abi path of bench.py), so the result of gather k is read only after step k + 1 has been queued."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from racing_dreamer_amd.batched_env import BatchedRaceEnv
        from racing_dreamer_amd.distributed import shard_envs
        sh = shard_envs(total, rank, world)
        env = BatchedRaceEnv("columbia", sh.num_envs, 1, auto_reset=True, first_env=sh.first_env)
        blobs = [None] * world
        first = env.p2p_setup(P2P_MODES[0], rank, world)
        dist.all_gather_object(blobs, first)
        env.p2p_connect(blobs)
        second = torch.zeros(env.arena_nbytes + 64, dtype=torch.uint8, device=env.device)
        second = second[(-second.data_ptr()) % 64:][:env.arena_nbytes]
        arenas = [None, second]
        env.reset(mode="random", seed=4)
        got = {m: [] for m in P2P_MODES}
        k = 0
        for mode in P2P_MODES:
            assert env.p2p_setup(mode, rank, world) == first       # a payload switch: the same export blob
            if mode == "full-u16":
                env.enable_compact(buffers=2)
            else:
                env.disable_compact()
            for j in range(steps):
                env.step_random(seed=1, step=k, repeat=repeat)      # writes source buffer k & 1 ...
                if j > 0:                                           # ... while the gather of record k - 1 may still be
                    got[mode].append(env.gathered_p2p_host().copy())    # reading the other one: read its result only now
                env.gather_p2p()                                    # record k: asynchronous, behind step k
                k += 1
                if mode == "full-u16":
                    env.rotate_compact()
                else:
                    env.set_arena(arenas[k & 1])
            got[mode].append(env.gathered_p2p_host().copy())
        env.p2p_disconnect()                    # every rank unmaps its peers' buffers ...
        dist.barrier()
        env.p2p_teardown()                      # ... before any rank frees its own
        env.close()
        q.put((rank, sh.num_envs, got))
    except Exception as e:          # noqa: BLE001
        import traceback
        q.put((rank, -1, repr(e) + traceback.format_exc()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_hip_shards_peer_copy_gather_equals_the_unsharded_oracle_run():
    """rc_gather_trajectory_p2p between two processes that share the test GPU (hipIpc handles work on one device too):
    every rank's gathered buffer, every step, every payload == the CPU oracle's UNSHARDED run."""
    import torch
    import torch.multiprocessing as mp
    from oracle import racecar_oracle as ro
    from racing_dreamer_amd.distributed import compact_field_views, slab_field_views, summary_field_views
    total, steps, repeat, world = 96, 4, 2, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_p2p_rank, args=(r, world, port, total, steps, repeat, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=400) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = _oracle_rollout("columbia", total, steps * len(P2P_MODES), repeat)
    for rank, n, got in results:
        assert n == total // world, got
        parse = {"full": lambda b: slab_field_views(b, n, False), "full-u16": lambda b: compact_field_views(b, n),
                 "summary": lambda b: summary_field_views(b, n)}
        for mi, mode in enumerate(P2P_MODES):
            fn = parse[mode]
            assert len(got[mode]) == steps
            for j in range(steps):
                o = want[mi * steps + j]
                g = torch.from_numpy(got[mode][j])                  # [world, bytes]: rank r's record at row r
                assert g.shape[0] == world
                views = [fn(g[r]) for r in range(world)]
                for name in RECORD:
                    cat = torch.cat([v[name] for v in views]).numpy()
                    assert np.array_equal(cat, np.asarray(o[name], np.float32).reshape(cat.shape)), (rank, mode, j, name)
                if mode == "full":
                    assert np.array_equal(torch.cat([v["lidar"] for v in views]).numpy(), o["lidar"]), (rank, j)
                if mode == "full-u16":
                    q16 = torch.cat([v["lidar_u16"] for v in views]).numpy()
                    assert np.array_equal(q16, ro.quantise_lidar_u16(o["lidar"])), (rank, j)


def test_device_memory_helpers_of_the_c_abi():
    """rc_device_alloc / rc_device_free / rc_copy_from_device: what a client without the HIP runtime uses for the compact
    slab and the gather destination (examples/c_rollout.c does); here through ctypes, against the torch-owned slab."""
    import ctypes as C
    import torch
    from racing_dreamer_amd import _lib as L
    from racing_dreamer_amd.batched_env import BatchedRaceEnv
    env = BatchedRaceEnv("columbia", 128, 1, auto_reset=True)
    lib, h = env._lib, env._h
    nbytes = lib.rc_compact_bytes(C.byref(env._cfg))
    slab = C.c_void_p()
    L.check(lib.rc_device_alloc(h, nbytes, C.byref(slab)))
    assert slab.value and slab.value % 64 == 0
    L.check(lib.rc_set_compact_slab(h, slab, nbytes))
    env.reset(mode="random", seed=8)
    env.step_random(seed=1, step=0, repeat=2)
    host = np.empty(nbytes, np.uint8)
    L.check(lib.rc_copy_from_device(h, slab, host.ctypes.data, nbytes))
    env.enable_compact(buffers=1)                         # the same record into a torch-owned slab
    env.step_random(seed=1, step=1, repeat=2)
    a, b, c = env.compact_layout
    assert a == 128 * 1080 * 2 and b + c <= nbytes
    # the first slab still holds step 0's record: its summary section differs from step 1's, its layout is the same
    q0 = host[:a].view(np.uint16).reshape(128, 1080)
    assert q0.max() > 20000 and not np.array_equal(host, env.compact.cpu().numpy())
    assert rc_ok(lib.rc_device_free(h, slab)) and rc_ok(lib.rc_device_free(h, None))
    assert lib.rc_device_alloc(h, 0, C.byref(slab)) == -1 and b"zero size" in lib.rc_last_error()
    env.close()


def rc_ok(rc):
    return rc == 0


def _replay_rank(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from racing_dreamer_amd.batched_env import BatchedRaceEnv
        from racing_dreamer_amd.distributed import shard_envs
        from racing_dreamer_amd.replay import ShardedReplay, TrajectoryRing
        sh = shard_envs(64, rank, world)
        env = BatchedRaceEnv("columbia", sh.num_envs, 1, auto_reset=True, first_env=sh.first_env)
        ring = TrajectoryRing(env, capacity=8)
        ring.reset(mode="random", seed=2)
        for k in range(10):
            env.fill_random_actions(seed=3, step=k)
            ring.step(None, repeat=2)
        g = torch.Generator(device="cuda").manual_seed(5 + rank)
        batch = ShardedReplay(ring).sample(batch=8, length=3, generator=g, fields=("lidar", "reward", "time", "fresh", "done"))
        # this rank's rows, re-read from its own ring
        oldest = (ring.head + 1) % ring.capacity
        mine = slice(4 * rank, 4 * rank + 4)
        ok = True
        for j in range(4):
            t0, e = int(batch["t0"][mine][j]), int(batch["env"][mine][j])
            mid = (oldest + t0 + 1) % ring.capacity         # the middle row: never an episode boundary, never rewritten
            ok &= bool(torch.equal(batch["lidar"][mine][j, 1].cpu(), ring.fields["lidar"][mid, e, 0].cpu()))
            ok &= float(batch["time"][mine][j, 1]) == float(ring.fields["time"][mid, e, 0])
        # the same exchange as ONE native draw into ONE packed buffer and ONE collective (what bench.py's sharded headline sends)
        rep = ShardedReplay(ring)
        buf, lay = rep.draw_packed(8, 3, fields=("lidar", "reward", "time", "fresh", "done"), generator=g)
        pk = rep.exchange_packed(buf, lay)
        torch.cuda.synchronize()
        ok &= pk["lidar"].shape == (world, 4, 3, 1080) and pk["meta"].shape == (world, 4, 4) and int(pk["failed"].sum()) == 0
        local = TrajectoryRing.unpack(buf, lay)
        for n in ("lidar", "reward", "time", "fresh", "done", "meta"):
            ok &= bool(torch.equal(pk[n][rank], local[n]))
        for j in range(4):
            t0, car = int(pk["meta"][rank, j, 0]), int(pk["meta"][rank, j, 1])
            mid = (oldest + t0 + 1) % ring.capacity
            ok &= bool(torch.equal(pk["lidar"][rank, j, 1].cpu(), ring.fields["lidar"][mid, car, 0].cpu()))
        packed_sum = {n: float(pk[n].double().sum()) for n in ("lidar", "reward", "time", "meta")}
        q.put((rank, ok, {k: v.cpu().numpy() for k, v in batch.items()} | {"_packed_" + n: np.float64(v) for n, v in packed_sum.items()}))
        env.close()
    except Exception as e:          # noqa: BLE001
        q.put((rank, False, repr(e)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_replay_on_two_hip_rings():
    """Each rank keeps its shard's records in its own device-resident ring; the sampled training batch is what the
    collective carries: both ranks end up with the same 8 windows, 4 from each ring."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_replay_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (r0, ok0, a), (r1, ok1, b) = sorted(results, key=lambda t: t[0])
    assert ok0 and ok1, (a if not ok0 else b)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert a["lidar"].shape == (8, 3, 1080) and a["rank"].tolist() == [0, 0, 0, 0, 1, 1, 1, 1]
    dt = a["time"][:, 1:] - a["time"][:, :-1]
    inner = dt[a["fresh"][:, 1:] == 0]
    assert np.allclose(inner, 0.02, atol=1e-6)                      # consecutive agent steps of 2 sub-steps
