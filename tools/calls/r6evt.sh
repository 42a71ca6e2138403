#!/bin/bash
# Round 6: which launches surround the copyBuffer / pack_filter launches of an evaluation forward (configs[1], fp16).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r6evt}
mkdir -p gpurun_out/${T}_prof
SSA_ACT_DTYPE=fp16 timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${T}_prof -o $T -- python tools/eval_bench.py 6 c1 > gpurun_out/${T}_rocprof.log 2>&1
f=$(ls gpurun_out/${T}_prof/*/*kernel_trace.csv gpurun_out/${T}_prof/*kernel_trace.csv 2>/dev/null | head -1)
python - "$f" <<'PY' > gpurun_out/${T}_seq.txt
import csv, sys, re
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
def short(n):
    n=n.replace("(anonymous namespace)::","")
    m=re.search(r"k_(single|grouped)<(.*)>\(", n)
    return (m.group(2) if m else n.split("(")[0])[:48]
idx=[i for i,r in enumerate(rows) if "image_resize_kernel" in r["Kernel_Name"]]
a,b=idx[-2],idx[-1]
step=rows[a:b]
print("launches in the last forward:", len(step))
for i,r in enumerate(step):
    n=short(r["Kernel_Name"])
    if "copyBuffer" in n or "pack_filter" in n or "at::native" in n or "fillBuffer" in n:
        prev=short(step[i-1]["Kernel_Name"]) if i else ""
        nxt=short(step[i+1]["Kernel_Name"]) if i+1<len(step) else ""
        print("%4d %-40s grid %s wg %s | after %-34s before %s" % (i, n, r.get("Grid_Size_X", r.get("Grid_Size","?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size","?")), prev, nxt))
PY
rm -rf gpurun_out/${T}_prof
head -90 gpurun_out/${T}_seq.txt
