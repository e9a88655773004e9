#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: mean counter value per dispatch.
usage: pmc_summary.py <counter_collection.csv> [...]   (prints markdown)"""
import collections, csv, re, sys

def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:70]

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
counters = sorted({c for k in acc.values() for c in k})
print("| kernel | dispatches | " + " | ".join(f"{c} (mean/dispatch)" for c in counters) + " |")
print("|---|---:|" + "---:|" * len(counters))
for k, d in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
    n = max(len(v) for v in d.values())
    print(f"| `{k}` | {n} | " + " | ".join(f"{sum(d[c]) / len(d[c]):.1f}" if c in d else "-" for c in counters) + " |")
