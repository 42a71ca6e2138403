#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys
src = sys.argv[1]
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function",
                      "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"], capture_output=True, text=True).stderr
rows, cur = [], None
for l in out.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
    for k, tag in (("VGPRs", "v"), ("AGPRs", "a"), ("ScratchSize \[bytes/lane\]", "scratch"), ("Occupancy \[waves/SIMD\]", "occ"),
                   ("SGPRs Spill", "sspill"), ("VGPRs Spill", "vspill"), ("SGPRs", "s"), ("LDS Size \[bytes/block\]", "lds")):
        mm = re.search(r"remark:\s+" + k + r": (\d+)", l)
        if mm and cur is not None:
            cur[tag] = int(mm.group(1))
    if "error" in l:
        print(l)
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = name.replace("(anonymous namespace)::", "").replace("void ssa::", "")
    print("%-70s v=%3d a=%3d s=%3d occ=%d scratch=%d sspill=%d vspill=%d" % (
        name[:70], r.get("v", -1), r.get("a", -1), r.get("s", -1), r.get("occ", -1), r.get("scratch", -1), r.get("sspill", -1), r.get("vspill", -1)))
