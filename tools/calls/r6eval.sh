#!/bin/bash
# Round 6: where the evaluation forward's time goes -- rocprofv3 kernel statistics of configs[1] (1024 x 2048, single
# scale, fp16 storage) through graph_eval.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r6eval}
mkdir -p gpurun_out/${T}_prof
SSA_ACT_DTYPE=fp16 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof -o $T -- python tools/eval_bench.py 20 c1 > gpurun_out/${T}_rocprof.log 2>&1
cp $(ls gpurun_out/${T}_prof/*/*kernel_stats.csv gpurun_out/${T}_prof/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/${T}_kernel_stats.csv
rm -rf gpurun_out/${T}_prof
grep '^{' gpurun_out/${T}_rocprof.log | tail -1 | cut -c1-300
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${T}_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    n=r["Name"].replace("(anonymous namespace)::","")
    import re
    m=re.search(r"k_(single|grouped)<(.*)>\(", n)
    n=m.group(2)[:60] if m else n.split("(")[0][:60]
    print("%-62s %6s calls %8.1f us avg %5.1f %%" % (n, r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
