#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) into the same
table `--stats` prints: per-kernel calls, total / average duration, share.
usage: rocpd_summary.py results.db [steps]   (steps divides totals -> per-step)"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([^(]+)", name)
    return (m.group(1) if m else name)[:110]


def main():
    db = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    rows = c.execute("select %s, start, end from kernels" % namecol).fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0])
        a[0] += 1
        a[1] += e - s
    tot = sum(a[1] for a in agg.values())
    print("%-112s %9s %12s %10s %7s" % ("kernel", "calls", "total_ms", "avg_us", "share"))
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-112s %9.1f %12.3f %10.2f %6.2f%%" % (k, n / steps, t / 1e6 / steps, t / n / 1e3, 100.0 * t / tot))
    print("TOTAL kernel time %.3f ms (%d dispatches) / %g steps = %.3f ms/step" % (tot / 1e6, len(rows), steps, tot / 1e6 / steps))


if __name__ == "__main__":
    main()
