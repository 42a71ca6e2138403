#!/bin/bash
# Round 2, call Q: BatchNorm apply kernels issue their data loads before the coefficient prologue; strip sweep; trace.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out/r2q_prof
export TMPDIR=/tmp
log=gpurun_out/r2q.log
: > "$log"
run() { local name=$1 t=$2; shift 2; echo "== $name" >> "$log"; timeout "$t" "$@" > "gpurun_out/r2q_$name.log" 2>&1; echo "$name rc=$?" >> "$log"; }
b() { local name=$1; shift; run "bench_$name" 120 env "$@" python bench.py --no-cpu-baseline --no-roofline; grep -h '^{' "gpurun_out/r2q_bench_$name.log" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2), d['config']['library_launches_per_step'], d['config']['loss'])" >> "$log" 2>&1; }
run tests 600 python -m pytest tests/test_kernels_gpu.py tests/test_group_gpu.py tests/test_fuse_bwd_gpu.py -q -m gpu
tail -4 gpurun_out/r2q_tests.log >> "$log"
b default SSA_X=0
b strip16 SSA_WGRAD_STRIP=16
b strip4 SSA_WGRAD_STRIP=4
b default2 SSA_X=0
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2q_prof -o r2q -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r2q_prof_bench.log 2>&1
echo "prof rc=$?" >> "$log"
cat "$log"
