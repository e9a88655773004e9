#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.2) `*_results.db` into a per-kernel stats table (markdown), the same
figures `--stats` would print: calls, total / average / min / max duration, share.

    python tools/rocpd_stats.py gpurun_out/prof/xyz_results.db > profiles/r01_kernel_stats.md
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= 110 else name[:107] + "..."


def main(path: str) -> None:
    db = sqlite3.connect(path)
    rows = db.execute(
        "select s.display_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), "
        "max(d.end - d.start), max(s.arch_vgpr_count), max(s.sgpr_count), max(d.group_segment_size) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        "group by s.display_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | sgpr | lds B |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for name, calls, tot, avg, mn, mx, vg, sg, lds in rows:
        print(f"| `{short(name)}` | {calls} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | "
              f"{100.0 * tot / total:.1f} | {vg} | {sg} | {lds} |")
    print(f"\ntotal kernel time: {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")


if __name__ == "__main__":
    main(sys.argv[1])
