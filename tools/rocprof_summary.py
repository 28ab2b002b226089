#!/usr/bin/env python
"""Turns a rocprofv3 rocpd database (``*_results.db``) into the per-kernel text summary kept under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof/x_results.db > profiles/rNN_name_kernel_stats.txt
"""
import sqlite3
import sys


def main(path, top=45):
    c = sqlite3.connect(path)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    tot = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats summary of {path.split('/')[-1]}  (durations in microseconds)")
    print(f"# kernels: {len(rows)}  total GPU kernel time: {tot / 1e3:.3f} ms")
    print(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'%':>6}  name")
    for name, calls, total, avg, pct in rows[:top]:
        print(f"{calls:7d} {total:12.1f} {avg:10.3f} {pct:6.2f}  {name[:150]}")


if __name__ == "__main__":
    main(sys.argv[1])
