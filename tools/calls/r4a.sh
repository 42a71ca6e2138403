#!/bin/bash
# Round 4, first call: kernel tests on the rewritten trunk conv / BatchNorm kernels, their micro-benchmarks
# (tools/tilebench.py, tools/bnbench.py at 2 / 4 / 8 rows per thread), one bench line and a one-step trace.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r4a}
mkdir -p gpurun_out/${T}_prof
log=gpurun_out/$T.log
: > "$log"
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_group_gpu.py tests/test_fuse_bwd_gpu.py -q -m gpu -x > gpurun_out/${T}_tests.log 2>&1
echo "tests rc=$?: $(tail -1 gpurun_out/${T}_tests.log)" >> "$log"
timeout 200 python tools/tilebench.py 20 > gpurun_out/${T}_tilebench.txt 2>&1
echo "tilebench rc=$?" >> "$log"; tail -22 gpurun_out/${T}_tilebench.txt >> "$log"
for r in 2 4 8; do
  timeout 100 env SSA_BN_ROWS_APPLY=$r SSA_BN_ROWS_BWD=$r SSA_BN_ROWS_REDUCE=$r python tools/bnbench.py 30 >> gpurun_out/${T}_bnbench.txt 2>&1
done
cat gpurun_out/${T}_bnbench.txt >> "$log"
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline"
line() { grep -h '^{' "$1" | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],2), "ms", d["config"]["library_launches_per_step"], "launches, loss", round(d["config"]["loss"],4))' 2>&1 | tail -1; }
timeout 200 $B > gpurun_out/${T}_bench_default.log 2>&1; echo "default: $(line gpurun_out/${T}_bench_default.log)" >> "$log"
for spec in "$@"; do
  [ "$spec" = "$T" ] && continue
  name=${spec%%=*}; envs=${spec#*=}
  timeout 150 env ${envs//,/ } $B > gpurun_out/${T}_bench_$name.log 2>&1; echo "$name [$envs]: $(line gpurun_out/${T}_bench_$name.log)" >> "$log"
done
timeout 240 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${T}_prof -o $T -- $B > gpurun_out/${T}_rocprof.log 2>&1
f=$(ls gpurun_out/${T}_prof/*/*kernel_trace.csv gpurun_out/${T}_prof/*kernel_trace.csv 2>/dev/null | head -1)
python tools/trace_step.py "$f" 70 gpurun_out/${T}_launches.txt > gpurun_out/${T}_trace_step.txt 2>&1
rm -rf gpurun_out/${T}_prof
head -40 gpurun_out/${T}_trace_step.txt >> "$log"
tail -5 gpurun_out/${T}_bench_default.log >> "$log"
cat "$log"
