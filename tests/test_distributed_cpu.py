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
    assert bench.launch.pinned_device_env is bench.pinned_device_env
    assert bench.pinned_device_env(3) == {"HIP_VISIBLE_DEVICES": "3", "SF_BENCH_DEVICE": "0"}
    assert bench.pinned_device_env(1, "4,5,6,7") == {"HIP_VISIBLE_DEVICES": "5", "SF_BENCH_DEVICE": "0"}


def test_timed_steps_reports_own_time():
    sys.path.insert(0, REPO)
    import bench
    own = []
    dt = bench.timed_steps(lambda: time.sleep(0.002), steps=3, warmup=1, world=1, sync_fn=lambda: None, barrier_fn=lambda: None,
                           allreduce_max_fn=lambda x: x, own=own)
    assert len(own) == 3 and 0.005 <= own[0] <= dt and 0.005 <= own[1] <= own[0] and 0 <= own[2] <= own[1] + 1e-3   # (own time, host wall / CPU time inside the steps)


def test_rank_cpu_sets_numa_and_fallback():
    """Host cores of the local ranks (streamflow_amd/launch.py): disjoint, inside the allowed mask, on the NUMA node of the rank's
    GPU when the topology is known; an even split otherwise; single cores round-robin when there are fewer cores than ranks."""
    sys.path.insert(0, REPO)
    import bench
    L = bench.launch
    assert L.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    # 8 GPUs, 4 per socket, 2 x 64 cores of which the job may use every second one
    allowed = list(range(0, 128, 2))
    sets = L.rank_cpu_sets(8, allowed, [0, 0, 0, 0, 1, 1, 1, 1], {0: range(0, 64), 1: range(64, 128)})
    assert all(len(s) == 8 for s in sets) and len(set(sum(sets, []))) == 64 and set(sum(sets, [])) <= set(allowed)
    assert all(max(s) < 64 for s in sets[:4]) and all(min(s) >= 64 for s in sets[4:])
    # unknown topology (numa_node = -1) or a node without allowed cores: even split of the mask
    for nodes, cpus in (([-1] * 8, {}), ([0] * 8, {0: range(200, 208)}), (None, None)):
        sets = L.rank_cpu_sets(8, range(16), nodes, cpus)
        assert sets == [[2 * r, 2 * r + 1] for r in range(8)]
    assert L.rank_cpu_sets(4, [5, 9]) == [[5], [9], [5], [9]]
    assert L.rank_cpu_sets(1, range(4)) == [[0, 1, 2, 3]] and L.rank_cpu_sets(0, range(4)) == []


def test_eight_spawned_ranks_get_disjoint_cores_and_devices(tmp_path):
    """`python bench.py --gpus 8 --placement-only DIR`: the launcher starts 8 children; each restricts itself to ONE device
    (HIP_VISIBLE_DEVICES = its local rank, addressed as device 0) and to its own host cores before torch is imported, and exits
    without touching the GPU.  Also the external-launcher form (RANK / LOCAL_RANK / WORLD_SIZE set by torch.distributed.run)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                             "HIP_VISIBLE_DEVICES", "SF_BENCH_PIN", "SF_BENCH_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--placement-only", str(tmp_path)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    recs = [json.load(open(tmp_path / f"rank{i}.json")) for i in range(8)]
    assert [d["HIP_VISIBLE_DEVICES"] for d in recs] == [str(i) for i in range(8)] and all(d["SF_BENCH_DEVICE"] == "0" for d in recs)
    ncpu = len(os.sched_getaffinity(0))
    cores = [tuple(d["cpus"]) for d in recs]
    assert all(len(c) == max(1, ncpu // 8) for c in cores)
    if ncpu >= 8:
        assert len(set(sum(map(list, cores), []))) == 8 * (ncpu // 8)                 # pairwise disjoint
    assert all(d["torch_threads"] <= len(d["cpus"]) for d in recs)                     # thread pools sized from the rank's mask
    # external launcher: one rank of 8, outer device list honoured
    d2 = tmp_path / "ext"
    d2.mkdir()
    env2 = dict(env, RANK="5", LOCAL_RANK="5", WORLD_SIZE="8", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999", HIP_VISIBLE_DEVICES="0,1,2,3,4,6,7,5")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--placement-only", str(d2)],
                       capture_output=True, text=True, timeout=600, env=env2)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.load(open(d2 / "rank5.json"))
    assert rec["HIP_VISIBLE_DEVICES"] == "6" and rec["SF_BENCH_DEVICE"] == "0" and len(rec["cpus"]) == max(1, ncpu // 8)
    # --pin index keeps every device visible
    d3 = tmp_path / "idx"
    d3.mkdir()
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--pin", "index", "--placement-only", str(d3)],
                       capture_output=True, text=True, timeout=600, env=dict(env, RANK="2", LOCAL_RANK="2", WORLD_SIZE="8",
                                                                             MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.load(open(d3 / "rank2.json"))["HIP_VISIBLE_DEVICES"] is None
