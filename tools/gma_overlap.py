#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of bench.py --gma hybrid: start offsets and durations of the last GMA launches (stored-weights
stream beside the recompute kernel): do they overlap?   usage: gma_overlap.py <kernel_trace.csv>"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
g = [r for r in rows if "gma_pv" in r[2] or "gma_flash_kernel" in r[2] or "temporal_block" in r[2]]
t0 = g[-12][0]
for r in g[-12:]:
    print(f"{r[2][:44]:46s} start {(r[0] - t0) / 1e3:9.1f} us  duration {(r[1] - r[0]) / 1e3:7.1f} us")
