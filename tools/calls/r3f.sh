#!/bin/bash
# Round 3, call F: software-pipelined weight-gradient tile kernel (unconditional prefetch loads, MFMA register ring).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=r3f
mkdir -p gpurun_out/${T}_prof
log=gpurun_out/$T.log
: > "$log"
run() { local name=$1 t=$2; shift 2; timeout "$t" "$@" > "gpurun_out/${T}_$name.log" 2>&1; echo "$name rc=$?" >> "$log"; }
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline"
run tests 300 python -m pytest tests/test_group_gpu.py tests/test_fuse_bwd_gpu.py tests/test_kernels_gpu.py -q -m gpu -x
tail -2 gpurun_out/${T}_tests.log >> "$log"
run bench 150 $B
run bench2 150 $B
run rocprof 240 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${T}_prof -o $T -- $B
f=$(ls gpurun_out/${T}_prof/*/*kernel_trace.csv gpurun_out/${T}_prof/*kernel_trace.csv 2>/dev/null | head -1)
python tools/trace_step.py "$f" 70 > gpurun_out/${T}_trace_step.txt 2>&1
rm -rf gpurun_out/${T}_prof
for n in bench bench2; do
  echo "$n: $(grep -h '^{' gpurun_out/${T}_$n.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],2), "ms", d["config"]["library_launches_per_step"], "launches, loss", d["config"]["loss"])' 2>&1 | tail -1)" >> "$log"
done
head -30 gpurun_out/${T}_trace_step.txt >> "$log"
cat "$log"
