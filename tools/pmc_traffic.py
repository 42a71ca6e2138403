#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
separate passes: the two counters do not fit the TCC slots together).
Units/corrections as MI355X_MICROARCH.md section HBM prescribes: both counters are
in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced
(16 B/lane) stream, so the read side is doubled (all kernels here read with
16-byte loads); WRITE_SIZE is uncalibrated and used as reported.
usage: pmc_traffic.py FETCH.csv WRITE.csv out.json"""
import csv
import json
import re
import sys


def short(name):
    """Kernel family: the functor a grouped / single launch wrapper (csrc/group.h) was instantiated with
    -- 'ConvTile', 'BnBwdApplyK', ... (the names bench.py's roofline block uses) -- or the plain kernel name."""
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"ssa::k_(?:grouped|single)<(\w+)", name)
    if m:
        return "ConvTile" if m.group(1) == "ConvTileAny" else m.group(1)
    m = re.match(r"([^(]+)", name)
    n = (m.group(1) if m else name).strip()
    return re.sub(r"<.*", "", n)          # template arguments folded: one row per kernel family


def load(path, counter):
    agg = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            a = agg.setdefault(short(r["Kernel_Name"]), [0, 0.0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return agg


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(fetch, key=lambda k: -fetch[k][2]):
        n, kib, ns = fetch[k]
        wn, wkib, _ = write.get(k, [0, 0.0, 0.0])
        rd = 2.0 * kib * 1024 / n
        wr = wkib * 1024 / wn if wn else 0.0
        out[k] = {"launches": n, "avg_us": ns / n / 1e3, "read_bytes_per_launch": rd,
                  "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sha = None
    try:
        import bench
        sha = bench.source_sha()          # bench.py refuses this file once the kernel sources change
    except Exception as e:                # noqa: BLE001
        print("source_sha unavailable:", e)
    json.dump({"note": "read = 2 x FETCH_SIZE x 1024 (gfx950 correction), write = WRITE_SIZE x 1024; per launch averages",
               "source_sha": sha, "kernels": out}, open(sys.argv[3], "w"), indent=1)
    print("%-34s %8s %9s %12s %12s" % ("kernel family", "launches", "avg_us", "read MB", "write MB"))
    for k, v in list(out.items())[:22]:
        print("%-34s %8d %9.2f %12.3f %12.3f" % (k[:34], v["launches"], v["avg_us"], v["read_bytes_per_launch"] / 1e6,
                                                 v["write_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
