"""Per-kernel breakdown of ONE replayed step from a rocprofv3 --kernel-trace CSV (per-dispatch timestamps):
python tools/trace_step.py <kernel_trace.csv> [n_top]"""
import collections
import csv
import re
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    m = re.search(r"k_(single|grouped)<(.*)>\(", n)
    if m:
        return m.group(2).strip() + ("" if m.group(1) == "grouped" else " [1]")
    return n.split("(")[0][:70]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "image_resize_kernel" in r["Kernel_Name"]]
    a, b = idx[-4], idx[-2]            # two resize launches open a step
    step = rows[a:b]
    t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
    print("launches %d  span %.3f ms  sum of kernel durations %.3f ms" % (len(step), (t1 - t0) / 1e6, busy / 1e6))
    agg = collections.defaultdict(lambda: [0, 0])
    fam = collections.defaultdict(lambda: [0, 0])
    for r in step:
        k = short(r["Kernel_Name"])
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        agg[k][0] += 1
        agg[k][1] += d
        f = k.split("<")[0].split(" ")[0]
        fam[f][0] += 1
        fam[f][1] += d
    print("-- families")
    for f, (c, t) in sorted(fam.items(), key=lambda x: -x[1][1])[:top]:
        print("%8.3f ms %5d  avg %7.1f us  %s" % (t / 1e6, c, t / c / 1e3, f))
    print("-- instantiations")
    for k, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:top]:
        print("%8.3f ms %5d  avg %7.1f us  %s" % (t / 1e6, c, t / c / 1e3, k))
    if len(sys.argv) > 3:
        # every launch of the step in issue order: start (us from the step's first launch), duration, gap to the
        # previous launch's end, workgroups x threads, LDS bytes, kernel
        with open(sys.argv[3], "w") as f:
            prev_end = t0
            for r in step:
                st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
                wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0)
                grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
                f.write("%9.1f %7.1f %6.1f %6d x %4d  lds %6s  %s\n" % (
                    (st - t0) / 1e3, (en - st) / 1e3, (st - prev_end) / 1e3, grid // max(wg, 1), wg,
                    r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "?")), short(r["Kernel_Name"])))
                prev_end = en


if __name__ == "__main__":
    main()
