#!/usr/bin/env python
"""Per-kernel hardware counters from rocprofv3 --pmc passes over `bench.py` (counter_collection CSVs, one pass per
counter set: the TCC slots hold FETCH_SIZE or WRITE_SIZE, not both -- MI355X_MICROARCH.md, rocprofv3 PMC slots).

usage: pmc_counters.py out.json pass1.csv [pass2.csv ...]

Output: one row per kernel INSTANTIATION (template arguments kept: the 48-channel and 96-channel-chunk variants of a
family are told apart where they are separate kernels) and one per family, each with per-launch averages of every
counter found and the derived figures
  hbm_bytes       = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024     (gfx950: FETCH_SIZE counts 64 B per 128-B request)
  mfma_busy_frac  = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs * 256 CUs * GRBM_GUI_ACTIVE / 8)   [busy cycles summed over SIMDs;
                    GRBM_GUI_ACTIVE comes summed over the 8 XCDs; cross-check: the head GEMM's 0.41 against 0.38 from FLOPs / time]
  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  waves_per_simd  = SQ_WAVE_CYCLES * 4 / (4 * 256 * GRBM_GUI_ACTIVE)                    [SQ_WAVE_CYCLES in quad-cycles]
GRBM_GUI_ACTIVE is taken per launch from the pass that recorded it (cycles the GPU was busy = kernel duration in
shader clocks).  Values a pass did not collect are null."""
import collections
import csv
import json
import os
import re
import sys

CUS, SIMDS, XCDS, SES = 256, 4, 8, 32


def inst(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"ssa::k_(grouped|single)<(.*)>\(", name)
    if m:
        return m.group(2).strip().rstrip(" >") + (">" if "<" in m.group(2) and not m.group(2).strip().endswith(">") else "")
    return re.sub(r"\(.*", "", name).strip()[:80]


def family(i):
    f = re.sub(r"<.*", "", i)
    if f in ("ConvTileAny", "ConvTilePAny", "ConvTilePK", "ConvTileP"):
        return "ConvTile"
    return {"ConvHaloGemm3": "ConvHaloGemm", "ConvHaloReg3": "ConvHaloGemm", "ConvHaloGemm1": "ConvHaloGemm", "ConvWgradHead3": "ConvWgradHead",
            "ConvGemmWide1": "ConvGemmWide", "ConvWgradTileA": "ConvWgradTile", "BnBwdApplyXK": "BnBwdApplyK", "BnBwdFusedXK": "BnBwdFusedK", "BnBwdFusedZK": "BnBwdFusedK",
            "BnBwdApplyZK": "BnBwdApplyK", "BnBwdReduceXK": "BnBwdReduceK", "BnBwdReduceZK": "BnBwdReduceK"}.get(f, f)


def main():
    out_path, passes = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))   # inst -> counter -> [n, sum]
    dur = collections.defaultdict(lambda: [0, 0.0])
    for path in passes:
        with open(path) as f:
            for r in csv.DictReader(f):
                i = inst(r["Kernel_Name"])
                a = acc[i][r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
                if r["Counter_Name"] in ("GRBM_GUI_ACTIVE", "FETCH_SIZE"):
                    d = dur[i]
                    d[0] += 1
                    d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])

    def row(counters, n_dur, ns):
        v = {k: (s / n if n else None) for k, (n, s) in counters.items()}
        g = v.get("GRBM_GUI_ACTIVE")
        if g:
            g = g / XCDS          # rocprofv3 reports the sum over the 8 XCDs (calibrated: 17,000-21,000 counts per us)
        res = {"launches": max([n for n, _ in counters.values()] + [0]), "avg_us": (ns / n_dur / 1e3) if n_dur else None,
               "counters_per_launch": v}
        if v.get("FETCH_SIZE") is not None:
            res["read_bytes_per_launch"] = 2.0 * v["FETCH_SIZE"] * 1024
        if v.get("WRITE_SIZE") is not None:
            res["write_bytes_per_launch"] = v["WRITE_SIZE"] * 1024
        if "read_bytes_per_launch" in res and "write_bytes_per_launch" in res:
            res["hbm_bytes_per_launch"] = res["read_bytes_per_launch"] + res["write_bytes_per_launch"]
        if g:
            if v.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
                res["mfma_busy_frac"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * CUS * g)
            if v.get("SQ_WAVE_CYCLES") is not None:
                res["waves_per_simd"] = v["SQ_WAVE_CYCLES"] * 4 / (SIMDS * CUS * g)
            if v.get("SQ_BUSY_CYCLES") is not None:
                res["sq_busy_frac"] = v["SQ_BUSY_CYCLES"] / (SES * g)      # one SQ per shader engine
        if v.get("SQ_LDS_IDX_ACTIVE"):
            res["lds_conflict_frac"] = (v.get("SQ_LDS_BANK_CONFLICT") or 0.0) / v["SQ_LDS_IDX_ACTIVE"]
        return res
    insts = {i: row(c, dur[i][0], dur[i][1]) for i, c in acc.items()}
    fam_acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    fam_dur = collections.defaultdict(lambda: [0, 0.0])
    for i, c in acc.items():
        for k, (n, s) in c.items():
            fam_acc[family(i)][k][0] += n
            fam_acc[family(i)][k][1] += s
        fam_dur[family(i)][0] += dur[i][0]
        fam_dur[family(i)][1] += dur[i][1]
    fams = {f: row(c, fam_dur[f][0], fam_dur[f][1]) for f, c in fam_acc.items()}
    sha = None
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        sha = bench.source_sha()
    except Exception as e:                # noqa: BLE001
        print("source_sha unavailable:", e)
    json.dump({"note": __doc__, "source_sha": sha, "kernels": fams, "instantiations": insts}, open(out_path, "w"), indent=1)
    order = sorted(insts, key=lambda i: -(insts[i]["avg_us"] or 0) * insts[i]["launches"])
    print("%-58s %6s %8s %9s %9s %7s %7s %7s" % ("instantiation", "n", "avg us", "read MB", "write MB", "mfma", "ldsconf", "w/simd"))
    for i in order[:40]:
        r = insts[i]
        f = lambda k, s=1.0: ("%9.3f" % (r[k] * s)) if r.get(k) is not None else "        -"   # noqa: E731
        print("%-58s %6d %8.1f %s %s %s %s %s" % (i[:58], r["launches"], r["avg_us"] or 0, f("read_bytes_per_launch", 1e-6),
                                               f("write_bytes_per_launch", 1e-6), f("mfma_busy_frac")[2:], f("lds_conflict_frac")[2:],
                                               f("waves_per_simd")[2:]))


if __name__ == "__main__":
    main()
