#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) into the same
table `--stats` prints: per-kernel calls, total / average duration, share.
usage: rocpd_summary.py results.db [steps]            (steps divides totals -> per-step)
       rocpd_summary.py results.db --steady [K]      steady state only: the trace is cut into steps at
                                                     every `pack_filters_batched_kernel` dispatch (the
                                                     first kernel of a training step) and only the last
                                                     K complete steps (default 5) are summarised -- the
                                                     warm-up steps' one-time kernels (single filter packs,
                                                     optimizer buffer clones) stay out of the averages.
Also prints the wall-clock span of the kept steps (concurrent streams overlap: span < sum)."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([^(]+)", name)
    return (m.group(1) if m else name)[:110]


def steady_rows(rows, keep, marker="pack_filters_batched_kernel"):
    """rows sorted by start -> (rows of the last `keep` complete steps, keep)."""
    cuts = [i for i, (n, _, _) in enumerate(rows) if marker in n]
    if len(cuts) < keep + 1:
        raise SystemExit("only %d step markers (%s) in the trace, need %d" % (len(cuts), marker, keep + 1))
    lo, hi = cuts[-(keep + 1)], cuts[-1]        # the last marker opens an incomplete step: drop it
    kept = rows[lo:hi]
    span = max(e for _, _, e in kept) - min(s for _, s, _ in kept)
    print("steady state: dispatches %d..%d of %d, %d steps, wall-clock span %.3f ms/step" % (
        lo, hi, len(rows), keep, span / 1e6 / keep))
    return kept, float(keep)


def main():
    db = sys.argv[1]
    steady = len(sys.argv) > 2 and sys.argv[2] == "--steady"
    steps = 1.0 if steady else (float(sys.argv[2]) if len(sys.argv) > 2 else 1.0)
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    rows = c.execute("select %s, start, end from kernels order by start" % namecol).fetchall()
    if steady:
        rows, steps = steady_rows(rows, int(sys.argv[3]) if len(sys.argv) > 3 else 5)
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
