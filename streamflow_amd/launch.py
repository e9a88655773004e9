"""Per-rank placement of the clip replicas on one node: which GPU a rank sees and which host cores it runs on.

Imported by bench.py BEFORE torch (nothing here touches the GPU runtime): a rank restricts itself to its one device with
HIP_VISIBLE_DEVICES and to its own share of the host cores with sched_setaffinity before any library starts threads -- every
rank replays a ~1,200-node HIP graph over three streams per step, and eight ranks left on one core set (or on the far NUMA
node of their GPU) would measure the host, not the kernels.  The reference has nothing to mirror here: its only multi-GPU
mechanism is nn.DataParallel inside one process (train_mf.py:146, evaluate_mf.py:1207).
"""
from __future__ import annotations

import glob
import os
import re
from typing import Dict, List, Optional, Sequence


def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    out: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def gpu_numa_nodes(sys_root: str = "/sys") -> List[int]:
    """NUMA node of every AMD GPU of the node in PCI address order (the order of HIP device indices on a default install), from
    sysfs alone: /sys/bus/pci/devices/*/{vendor,class,numa_node}.  -1 where the kernel does not say.  [] without sysfs."""
    devs = []
    for d in sorted(glob.glob(os.path.join(sys_root, "bus/pci/devices/*"))):
        try:
            with open(os.path.join(d, "vendor")) as f:
                vendor = f.read().strip()
            with open(os.path.join(d, "class")) as f:
                cls = f.read().strip()
        except OSError:
            continue
        # AMD, display controller (0x03xxxx) or processing accelerator (0x12xxxx: what the MI300 / MI355X boards report)
        if vendor != "0x1002" or not re.match(r"0x(03|12)", cls):
            continue
        try:
            with open(os.path.join(d, "numa_node")) as f:
                node = int(f.read().strip())
        except (OSError, ValueError):
            node = -1
        devs.append(node)
    return devs


def node_cpus(node: int, sys_root: str = "/sys") -> List[int]:
    try:
        with open(os.path.join(sys_root, f"devices/system/node/node{node}/cpulist")) as f:
            return parse_cpulist(f.read())
    except OSError:
        return []


def rank_cpu_sets(world: int, allowed: Sequence[int], gpu_nodes: Optional[Sequence[int]] = None,
                  cpus_of_node: Optional[Dict[int, Sequence[int]]] = None) -> List[List[int]]:
    """Disjoint core sets for the local ranks 0 .. world-1 out of `allowed` (this process' affinity mask).
    With a known topology (gpu_nodes[r] = NUMA node of rank r's GPU, cpus_of_node[n] = cores of node n) the ranks whose GPUs
    hang off one node share that node's allowed cores evenly; ranks without a usable node, and every rank when there are fewer
    cores than ranks, fall back to an even split of all allowed cores (round-robin of single cores when cores < ranks: then sets
    overlap, which is still better than every rank on every core)."""
    allowed = sorted(set(allowed))
    if world <= 0:
        return []
    if not allowed:
        return [[] for _ in range(world)]
    if len(allowed) < world:
        return [[allowed[r % len(allowed)]] for r in range(world)]
    sets: List[Optional[List[int]]] = [None] * world
    if gpu_nodes and cpus_of_node and len(gpu_nodes) >= world:
        by_node: Dict[int, List[int]] = {}
        for r in range(world):
            by_node.setdefault(int(gpu_nodes[r]), []).append(r)
        ok = all(n >= 0 and len(set(cpus_of_node.get(n, ())) & set(allowed)) >= len(rs) for n, rs in by_node.items())
        if ok:
            for n, rs in by_node.items():
                cores = sorted(set(cpus_of_node[n]) & set(allowed))
                k = len(cores) // len(rs)
                for i, r in enumerate(rs):
                    sets[r] = cores[i * k:(i + 1) * k]
    if any(s is None for s in sets):
        k = len(allowed) // world
        sets = [allowed[r * k:(r + 1) * k] for r in range(world)]
    return [list(s) for s in sets]


def pinned_device_env(local_rank: int, visible: Optional[str] = None) -> Dict[str, str]:
    """Environment that restricts a rank to ONE GPU before it initialises the runtime: the local_rank-th of the devices this
    process may see (HIP_VISIBLE_DEVICES of the parent, if set).  The rank then addresses it as device 0 (SF_BENCH_DEVICE)."""
    devs = [d for d in visible.split(",") if d != ""] if visible else None
    dev = devs[local_rank % len(devs)] if devs else str(local_rank)
    return {"HIP_VISIBLE_DEVICES": dev, "SF_BENCH_DEVICE": "0"}


def pin_rank_cpus(local_rank: int, world: int, sys_root: str = "/sys") -> List[int]:
    """Restrict THIS process to its share of the host cores (call before importing torch: OpenMP / torch size their thread pools
    from the affinity mask).  Returns the core list (the whole mask when nothing was changed: world <= 1 or no sched_setaffinity).
    SF_BENCH_AFFINITY=0 switches it off."""
    if not hasattr(os, "sched_getaffinity"):
        return []
    allowed = sorted(os.sched_getaffinity(0))
    if world <= 1 or os.environ.get("SF_BENCH_AFFINITY", "1") == "0":
        return allowed
    nodes = gpu_numa_nodes(sys_root)
    # the rank's GPU is the local_rank-th VISIBLE device: map through an outer HIP_VISIBLE_DEVICES list of plain indices
    vis = os.environ.get("SF_BENCH_OUTER_VISIBLE", "")
    if vis and nodes and all(v.isdigit() and int(v) < len(nodes) for v in vis.split(",") if v):
        nodes = [nodes[int(v)] for v in vis.split(",") if v]
    cpus = {n: node_cpus(n, sys_root) for n in set(nodes) if n >= 0}
    mine = rank_cpu_sets(world, allowed, nodes if len(nodes) >= world else None, cpus)[local_rank % world]
    if mine:
        try:
            os.sched_setaffinity(0, mine)
        except OSError:
            return allowed
    return mine or allowed
