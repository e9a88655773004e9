"""CPU, world_size 2 on gloo: the multi-GPU harness logic of bench.py (clip sharding, barrier-bracketed
timing with max over ranks, whole-job aggregation).  The hot path itself has no collective: ranks are
independent replicas over clips."""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    sys.path.insert(0, REPO)
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    clips = bench.shard(7, world, rank)
    done = []

    def step():                       # rank 1 is slower: the reported time must be the max over ranks
        time.sleep(0.01 * (1 + rank))
        done.append(1)

    def allreduce_max(x):
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    dt = bench.timed_steps(step, steps=5, warmup=2, world=world, sync_fn=lambda: None, barrier_fn=dist.barrier,
                           allreduce_max_fn=allreduce_max)
    gathered = [None] * world
    dist.all_gather_object(gathered, (clips, len(done), dt))
    if rank == 0:
        torch.save(gathered, out)
    dist.destroy_process_group()


def test_two_rank_harness(tmp_path):
    out = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out)
    (c0, n0, t0), (c1, n1, t1) = res
    assert sorted(c0 + c1) == list(range(7)) and not set(c0) & set(c1)      # disjoint cover of the clips
    assert n0 == n1 == 7                                                      # 2 warm-up + exactly 5 timed steps
    assert t0 == t1                                                           # every rank reports the max
    assert t0 >= 5 * 0.02 * 0.9                                               # ... which is the slow rank's time


def test_usable_cores_positive():
    sys.path.insert(0, REPO)
    import bench
    assert 1 <= bench.usable_cores() <= (os.cpu_count() or 1)


def test_spawn_ranks_sets_rendezvous_env_and_reports_failures(tmp_path):
    """`python bench.py --gpus N` typed directly spawns its own ranks (bench.spawn_ranks): every child gets
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / one common MASTER_PORT; a failing rank's exit code is
    returned and the remaining ranks are stopped instead of waiting in a barrier forever."""
    sys.path.insert(0, REPO)
    import bench
    script = tmp_path / "child.py"
    script.write_text(
        "import os, sys, time\n"
        "r = os.environ['RANK']\n"
        "open(os.path.join(sys.argv[1], 'rank' + r), 'w').write(' '.join(os.environ[k] for k in "
        "('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')))\n"
        "if len(sys.argv) > 2 and r == '1':\n"
        "    sys.exit(7)\n"
        "if len(sys.argv) > 2:\n"
        "    time.sleep(60)\n")
    assert bench.spawn_ranks(3, [str(tmp_path)], script=str(script)) == 0
    recs = [(tmp_path / f"rank{r}").read_text().split() for r in range(3)]
    assert [r[0] for r in recs] == ["0", "1", "2"] and [r[1] for r in recs] == ["0", "1", "2"]
    assert all(r[2] == "3" and r[3] == "127.0.0.1" for r in recs) and len({r[4] for r in recs}) == 1
    t0 = time.time()
    assert bench.spawn_ranks(2, [str(tmp_path), "fail"], script=str(script)) == 7
    assert time.time() - t0 < 30                     # rank 0 (sleeping 60 s) was terminated


def test_bench_gpus_gt1_does_not_touch_gpu_in_launcher(monkeypatch):
    """The launcher branch must run before any GPU initialisation: with N > 1 and no WORLD_SIZE, main() hands over to
    spawn_ranks and exits with its code without ever calling torch.cuda.is_available()."""
    sys.path.insert(0, REPO)
    import bench
    seen = {}
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    monkeypatch.setattr(bench, "spawn_ranks", lambda n, argv, script=None: seen.update(n=n, argv=list(argv)) or 0)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: (_ for _ in ()).throw(AssertionError("GPU touched")))
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    assert seen == {"n": 4, "argv": ["--gpus", "4", "--steps", "2"]}


def test_strong_scaling_shards_and_batches():
    """--total-clips (BASELINE.json config 4: 64 Sintel clips at every N): round-robin shards cover the clips exactly once for
    even and uneven world sizes, and a rank's step is its clips in launches of at most --clips."""
    sys.path.insert(0, REPO)
    import bench
    for world in (1, 2, 3, 4, 8):
        parts = [bench.shard(64, world, r) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(64)) and max(map(len, parts)) - min(map(len, parts)) <= 1
        for p in parts:
            b = bench.batches(len(p), 8)
            assert sum(b) == len(p) and all(0 < x <= 8 for x in b) and all(x == 8 for x in b[:-1])
    assert [len(bench.shard(64, 3, r)) for r in range(3)] == [22, 21, 21]
    assert bench.batches(22, 8) == [8, 8, 6] and bench.batches(21, 8) == [8, 8, 5] and bench.batches(8, 8) == [8]
    assert bench.batches(0, 8) == []                       # more ranks than clips: that rank only joins the barriers


def test_pinned_device_env():
    """--pin visible: a rank is restricted to ONE device before the runtime starts and addresses it as device 0; an outer
    HIP_VISIBLE_DEVICES list is honoured."""
    sys.path.insert(0, REPO)
    import bench
    assert bench.pinned_device_env(3) == {"HIP_VISIBLE_DEVICES": "3", "SF_BENCH_DEVICE": "0"}
    assert bench.pinned_device_env(1, "4,5,6,7") == {"HIP_VISIBLE_DEVICES": "5", "SF_BENCH_DEVICE": "0"}


def test_timed_steps_reports_own_time():
    sys.path.insert(0, REPO)
    import bench
    own = []
    dt = bench.timed_steps(lambda: time.sleep(0.002), steps=3, warmup=1, world=1, sync_fn=lambda: None, barrier_fn=lambda: None,
                           allreduce_max_fn=lambda x: x, own=own)
    assert len(own) == 1 and 0.005 <= own[0] <= dt
