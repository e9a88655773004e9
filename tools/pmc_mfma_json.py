#!/usr/bin/env python3
"""Matrix-core busy fraction per kernel family from a rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE:
busy = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (sum(GRBM_GUI_ACTIVE) x 128)   (GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs / 8).
usage: pmc_mfma_json.py <counter_collection.csv> [key=value ...] > profiles/rNN_mfma_busy_<workload>.json   (read by bench.py)"""
import collections, csv, json, re, sys
sys.path.insert(0, __import__("os").path.dirname(__file__))
from traffic_json import family   # noqa: E402  (same kernel -> family map)
busy, act = collections.Counter(), collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    f = family(r["Kernel_Name"])
    if not f:
        continue
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
        busy[f] += float(r["Counter_Value"])
    elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        act[f] += float(r["Counter_Value"])
out = {"_note": "matrix-core busy fraction of SIMD-cycles per kernel family: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128)"}
for f in sorted(busy):
    if act[f] > 0:
        out[f] = round(busy[f] / (act[f] * 128.0), 4)
for kv in sys.argv[2:]:
    k, v = kv.split("=", 1)
    out[k] = int(v) if (v.isdigit() and k != "_csrc_sha") else v
print(json.dumps(out, indent=1))
