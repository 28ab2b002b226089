#!/usr/bin/env python
"""Per (kernel, grid size) summary of a rocprofv3 rocpd database: the per-name averages of tools/rocprof_summary.py mix the
32 768-agent launches of the rollout / recurrent steps with the time-batched launches of the update (51 x as many rows), this
one keeps them apart.

    python tools/rocprof_by_grid.py gpurun_out/prof/x_results.db [min_total_us] > profiles/rNN_name_by_grid.txt
"""
import sqlite3
import sys
from collections import defaultdict


def main(path, min_total_us=300.0):
    c = sqlite3.connect(path)
    views = [r[0] for r in c.execute("select name from sqlite_master where type in ('view', 'table')")]
    if "kernels" not in views:
        print("no `kernels` view; objects:", views)
        return
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    gx = "grid_x" if "grid_x" in cols else "grid_size_x"
    wx = "workgroup_x" if "workgroup_x" in cols else "workgroup_size_x"
    groups = defaultdict(list)
    for name, g, w, s, e in c.execute(f"select name, {gx}, {wx}, start, end from kernels"):
        groups[(name, g, w)].append((e - s) / 1e3)
    tot = sum(sum(v) for v in groups.values())
    print(f"# per (kernel, grid) summary of {path.split('/')[-1]}  (microseconds; grid = work-items in x, wg = workgroup size)")
    print(f"# total GPU kernel time {tot / 1e3:.3f} ms; groups below {min_total_us:.0f} us total are omitted")
    print("# p50 = median: a persistent kernel launches the same grid for its rollout-size and its time-batched calls (K1 forward: 200 + 4"
          " per four cycles), so the median is the rollout launch and the mean is not")
    print(f"{'calls':>6} {'total_us':>11} {'avg_us':>10} {'p50_us':>10} {'min_us':>10} {'max_us':>10} {'%':>6} {'grid':>10} {'wg':>5}  name")
    for (name, g, w), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        if sum(v) < min_total_us:
            continue
        print(f"{len(v):6d} {sum(v):11.1f} {sum(v) / len(v):10.2f} {sorted(v)[len(v) // 2]:10.2f} {min(v):10.2f} {max(v):10.2f} {100 * sum(v) / tot:6.2f} "
              f"{g:10d} {w:5d}  {name[:110]}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 300.0)
