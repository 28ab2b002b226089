#!/usr/bin/env python
"""Idle time between consecutive kernels of a rocprofv3 rocpd database: total, and grouped by the kernel that FOLLOWS
the gap (i.e. whose launch arrived late).

    python tools/gap_summary.py /tmp/prof/.../x_results.db
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if "kernel_dispatch" in t and "rocpd_" in t and not t.startswith("rocpd_info")]
    if "kernels" in tabs:
        rows = list(c.execute("select name, start, end from kernels order by start"))
    else:
        t = kd[0]
        sym = [x for x in tabs if "kernel_symbol" in x][0]
        rows = list(c.execute(f"select s.kernel_name, d.start, d.end from {t} d join {sym} s on d.kernel_id = s.id order by d.start"))
    busy = sum(e - s for _, s, e in rows)
    span = rows[-1][2] - rows[0][1]
    gaps = {}
    tot = 0
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        g = max(0, s1 - e0)
        if g > 2_000_000:      # > 2 ms: a phase boundary (host work between cycles), not a launch gap
            continue
        tot += g
        k = n1[:60]
        a = gaps.setdefault(k, [0, 0])
        a[0] += g
        a[1] += 1
    print(f"kernels {len(rows)}  busy {busy / 1e6:.2f} ms  span {span / 1e6:.2f} ms  launch gaps {tot / 1e6:.2f} ms")
    for k, (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
        print(f"{g / 1e3:10.1f} us {n:6d}x {g / n / 1e3:7.2f} us/gap  before {k}")


if __name__ == "__main__":
    main(sys.argv[1])
