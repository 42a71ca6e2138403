#!/bin/bash
# Re-measure after a late kernel change: loss/elementwise kernel tests, PMC passes (-> profiles/r03_pmc.*), the bench line,
# rocprofv3 stats + one-step trace.  bash tools/calls/final.sh [tag]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r3fin}
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_group_gpu.py tests/test_fuse_bwd_gpu.py -q -m gpu > gpurun_out/${T}_tests.log 2>&1
tail -1 gpurun_out/${T}_tests.log
bash tools/calls/pmc.sh ${T}_pmc > gpurun_out/${T}_pmc_call.log 2>&1
cp gpurun_out/${T}_pmc_pmc.json profiles/r03_pmc.json
cp gpurun_out/${T}_pmc_pmc.txt profiles/r03_pmc.txt
timeout 500 python bench.py > gpurun_out/${T}_bench.log 2>&1
grep -h '^{' gpurun_out/${T}_bench.log > gpurun_out/${T}_bench_line.json
mkdir -p gpurun_out/${T}_prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof -o $T -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --eager-steps 0 > gpurun_out/${T}_rocprof.log 2>&1
f=$(ls gpurun_out/${T}_prof/*/*kernel_trace.csv gpurun_out/${T}_prof/*kernel_trace.csv 2>/dev/null | head -1)
python tools/trace_step.py "$f" 110 gpurun_out/${T}_launches.txt > gpurun_out/${T}_trace_step.txt 2>&1
cp $(ls gpurun_out/${T}_prof/*/*kernel_stats.csv gpurun_out/${T}_prof/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/${T}_kernel_stats.csv
rm -rf gpurun_out/${T}_prof
python -c 'import sys,json; d=json.load(open(sys.argv[1])); print(round(d["ms_per_step"],2), "ms", round(d["value"],2), "img/s; roofline frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "mfma_busy", d["roofline"].get("mfma_busy_frac"))' gpurun_out/${T}_bench_line.json
head -12 gpurun_out/${T}_trace_step.txt
