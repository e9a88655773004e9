#!/usr/bin/env python3
"""rocprofv3 `--kernel-trace --stats --output-format csv` -> markdown table (profiles/).  usage: stats_md.py <kernel_stats.csv> [title]"""
import csv, re, sys

def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name if len(name) <= 100 else name[:97] + "..."

rows = list(csv.DictReader(open(sys.argv[1])))
if len(sys.argv) > 2:
    print(f"# {sys.argv[2]}\n")
print("| kernel | calls | total ms | avg us | min us | max us | % |")
print("|---|---:|---:|---:|---:|---:|---:|")
tot = 0.0
for r in rows:
    tot += float(r["TotalDurationNs"])
    print(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.2f} | "
          f"{float(r['MinNs']) / 1e3:.2f} | {float(r['MaxNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} |")
print(f"\ntotal kernel time: {tot / 1e6:.3f} ms over {sum(int(r['Calls']) for r in rows)} dispatches")
