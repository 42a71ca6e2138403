#!/bin/bash
# Round 6 call L: the step on the current sources taken apart -- bench line (roofline + CPU baseline), rocprofv3 kernel
# stats + one replayed step (per family, every launch in order), PMC passes.   bash tools/calls/r6l.sh [tag]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r6l}
mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/${T}_bench.log 2>&1
grep -h '^{' gpurun_out/${T}_bench.log > gpurun_out/${T}_bench_line.json
mkdir -p gpurun_out/${T}_prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof -o $T -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --eager-steps 0 > gpurun_out/${T}_rocprof.log 2>&1
f=$(ls gpurun_out/${T}_prof/*/*kernel_trace.csv gpurun_out/${T}_prof/*kernel_trace.csv 2>/dev/null | head -1)
python tools/trace_step.py "$f" 110 gpurun_out/${T}_launches.txt > gpurun_out/${T}_trace_step.txt 2>&1
cp $(ls gpurun_out/${T}_prof/*/*kernel_stats.csv gpurun_out/${T}_prof/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/${T}_kernel_stats.csv
rm -rf gpurun_out/${T}_prof
python -c 'import sys,json; d=json.load(open(sys.argv[1])); print(round(d["ms_per_step"],2), "ms", round(d["value"],2), "img/s; bf16", d.get("ms_per_step_bf16"), "roofline frac", d["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"])' gpurun_out/${T}_bench_line.json
head -60 gpurun_out/${T}_trace_step.txt
bash tools/calls/pmc.sh ${T} > gpurun_out/${T}_pmc_call.log 2>&1
tail -5 gpurun_out/${T}_pmc_call.log
