#!/usr/bin/env python
"""Idle time of the GPU between consecutive kernels of a rocprofv3 rocpd database: how much of a bench cycle is NOT covered by
kernel execution, and after which kernels the device waits for the host (launch-bound stretches).

    python tools/rocprof_gaps.py gpurun_out/prof/x_results.db > profiles/rNN_name_gaps.txt
"""
import sqlite3
import sys
from collections import defaultdict


def main(path):
    c = sqlite3.connect(path)
    rows = list(c.execute("select name, start, end from kernels order by start"))
    if not rows:
        print("no kernels")
        return
    busy = sum(e - s for _, s, e in rows)
    span = rows[-1][2] - rows[0][1]
    gaps = []
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        gaps.append((max(0, s1 - e0), n0, n1))
    idle = sum(g for g, _, _ in gaps)
    print(f"# {len(rows)} kernels over {span / 1e6:.2f} ms: busy {busy / 1e6:.2f} ms, idle between kernels {idle / 1e6:.2f} ms "
          f"({100 * idle / span:.1f} %)")
    edges = [0, 1e3, 2e3, 5e3, 1e4, 2e4, 5e4, 1e5, 1e6, 1e12]
    print("# gap histogram (microseconds): count, total ms")
    for lo, hi in zip(edges, edges[1:]):
        sel = [g for g, _, _ in gaps if lo <= g < hi]
        print(f"#   [{lo / 1e3:7.0f}, {hi / 1e3:9.0f}) : {len(sel):6d}  {sum(sel) / 1e6:8.3f}")
    by = defaultdict(lambda: [0, 0.0])
    for g, n0, n1 in gaps:
        k = (n0[:70], n1[:70])
        by[k][0] += 1
        by[k][1] += g
    print("# largest idle by (kernel -> next kernel): count, total us, mean us")
    for (n0, n1), (cnt, tot) in sorted(by.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"{cnt:6d} {tot / 1e3:10.1f} {tot / cnt / 1e3:8.2f}  {n0}  ->  {n1}")


if __name__ == "__main__":
    main(sys.argv[1])
