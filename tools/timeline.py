#!/usr/bin/env python3
"""GPU timeline of the graph-replayed step from a rocprofv3 --kernel-trace CSV: busy union, idle gaps, concurrency.
usage: timeline.py <kernel_trace.csv> [first_kernel_substring=pack_f16z]   (a step starts at each launch of that kernel)"""
import csv, re, sys
from collections import defaultdict

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "pack_f16z"
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if marker in r[2]]
print(f"{len(rows)} dispatches, {len(starts)} steps (marker {marker!r})")
if len(starts) < 3:
    sys.exit(0)


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)[:60]


# the last complete step
a, b = starts[-2], starts[-1]
step = rows[a:b]
t0 = step[0][0]
t1 = max(e for _, e, _ in step)
ev = []
for s, e, n in step:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = {0: 0, 1: 0, 2: 0, 3: 0}
cur, last = 0, t0
for t, d in ev:
    busy[min(cur, 3)] += t - last
    cur += d; last = t
wall = t1 - t0
print(f"step: {len(step)} dispatches, wall {wall / 1e6:.3f} ms, sum of kernel durations {sum(e - s for s, e, _ in step) / 1e6:.3f} ms")
print("time with k kernels running: " + ", ".join(f"k={k}{'+' if k == 3 else ''}: {v / 1e6:.3f} ms ({100 * v / wall:.1f} %)" for k, v in busy.items()))
# gaps (nothing running) by the kernel that follows
gaps = defaultdict(lambda: [0, 0])
cur, last_end = 0, None
run = []
for s, e, n in sorted(step):
    run.append((s, e, n))
act_end = step[0][0]
for s, e, n in sorted(step):
    if s > act_end:
        g = gaps[short(n)]
        g[0] += s - act_end; g[1] += 1
    act_end = max(act_end, e)
tot_gap = sum(g[0] for g in gaps.values())
print(f"idle gaps: {tot_gap / 1e6:.3f} ms in {sum(g[1] for g in gaps.values())} gaps; by following kernel:")
for n, g in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"   {g[0] / 1e3:9.1f} us in {g[1]:4d} gaps (avg {g[0] / g[1] / 1e3:5.2f} us)  before {n}")
# which kernels run ALONE (k = 1), by name
alone_t = defaultdict(int)
ev2 = sorted([(s_, 1, n) for s_, e_, n in step] + [(e_, -1, n) for s_, e_, n in step])
active = {}
last = t0
for t, d_, n in ev2:
    if len(active) == 1:
        alone_t[short(next(iter(active)))] += t - last
    last = t
    if d_ == 1:
        active[n + str(t)] = 1
    else:
        for k in list(active):
            if k.startswith(n):
                del active[k]
                break
print("time alone on the GPU (k = 1) by kernel:")
for n, d_ in sorted(alone_t.items(), key=lambda kv: -kv[1])[:12]:
    print(f"   {d_ / 1e6:8.3f} ms  {n}")
# time by kernel: exclusive (alone) vs overlapped
alone = defaultdict(int); dur = defaultdict(int); cnt = defaultdict(int)
for s, e, n in step:
    dur[short(n)] += e - s; cnt[short(n)] += 1
print("kernel time by name (this step):")
for n, d in sorted(dur.items(), key=lambda kv: -kv[1])[:24]:
    print(f"   {d / 1e6:8.3f} ms  {cnt[n]:4d} x {d / cnt[n] / 1e3:8.1f} us  {n}")
