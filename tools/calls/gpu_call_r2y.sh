#!/bin/bash
# Round 2, call Y: bench line + rocprofv3 trace of the final default (weight-gradient flush at the end of backward).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out/r2y_prof
export TMPDIR=/tmp
log=gpurun_out/r2y.log
: > "$log"
run() { local name=$1 t=$2; shift 2; echo "== $name" >> "$log"; timeout "$t" "$@" > "gpurun_out/r2y_$name.log" 2>&1; echo "$name rc=$?" >> "$log"; }
run bench 500 python bench.py
run rocprof 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2y_prof -o r2y -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
run group_tests 300 python -m pytest tests/test_group_gpu.py tests/test_fuse_bwd_gpu.py tests/test_e2e_gpu.py -q -m gpu
tail -3 gpurun_out/r2y_group_tests.log >> "$log"
cat "$log"
grep -h '^{' gpurun_out/r2y_bench.log | cut -c1-260
