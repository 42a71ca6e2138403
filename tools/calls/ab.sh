#!/bin/bash
# Generic measurement call: bash tools/calls/ab.sh <tag> [name=ENV1=v,ENV2=v ...]
# Runs the op-level / grouped / fused-backward GPU tests, `bench.py` for the default configuration and for every named
# environment variant, and a rocprofv3 kernel trace of the default configuration summarised by tools/trace_step.py.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=$1; shift
mkdir -p gpurun_out/${T}_prof
log=gpurun_out/$T.log
: > "$log"
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline"
timeout 400 python -m pytest tests/test_group_gpu.py tests/test_fuse_bwd_gpu.py tests/test_kernels_gpu.py -q -m gpu -x > gpurun_out/${T}_tests.log 2>&1
echo "tests rc=$?: $(tail -1 gpurun_out/${T}_tests.log)" >> "$log"
line() { grep -h '^{' "$1" | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],2), "ms", d["config"]["library_launches_per_step"], "launches, loss", round(d["config"]["loss"],4))' 2>&1 | tail -1; }
timeout 150 $B > gpurun_out/${T}_bench_default.log 2>&1; echo "default: $(line gpurun_out/${T}_bench_default.log)" >> "$log"
for spec in "$@"; do
  name=${spec%%=*}; envs=${spec#*=}
  timeout 150 env ${envs//,/ } $B > gpurun_out/${T}_bench_$name.log 2>&1; echo "$name [$envs]: $(line gpurun_out/${T}_bench_$name.log)" >> "$log"
done
timeout 150 $B > gpurun_out/${T}_bench_default2.log 2>&1; echo "default again: $(line gpurun_out/${T}_bench_default2.log)" >> "$log"
timeout 240 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${T}_prof -o $T -- $B > gpurun_out/${T}_rocprof.log 2>&1
f=$(ls gpurun_out/${T}_prof/*/*kernel_trace.csv gpurun_out/${T}_prof/*kernel_trace.csv 2>/dev/null | head -1)
python tools/trace_step.py "$f" 70 gpurun_out/${T}_launches.txt > gpurun_out/${T}_trace_step.txt 2>&1
head -1 "$f" > gpurun_out/${T}_trace_header.txt
rm -rf gpurun_out/${T}_prof
head -32 gpurun_out/${T}_trace_step.txt >> "$log"
cat "$log"
