#!/bin/bash
# Round 2, call T: head weight gradient on 2-row tiles (2-3 workgroups per CU); BatchNorm reduce rows.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out
log=gpurun_out/r2t.log
: > "$log"
run() { local name=$1 t=$2; shift 2; echo "== $name" >> "$log"; timeout "$t" "$@" > "gpurun_out/r2t_$name.log" 2>&1; echo "$name rc=$?" >> "$log"; }
b() { local name=$1; shift; run "bench_$name" 120 env "$@" python bench.py --no-cpu-baseline --no-roofline; grep -h '^{' "gpurun_out/r2t_bench_$name.log" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2), d['config']['library_launches_per_step'], d['config']['loss'])" >> "$log" 2>&1; }
run tests 600 python -m pytest tests/test_kernels_gpu.py tests/test_group_gpu.py -q -m gpu
tail -3 gpurun_out/r2t_tests.log >> "$log"
b default SSA_X=0
b th4 SSA_WGRAD_HEAD_TH=4
b wgs3 SSA_WGRAD_HEAD_WGS=3
b wgs1 SSA_WGRAD_HEAD_WGS=1
b bnr4 SSA_BN_ROWS_REDUCE=4
b default2 SSA_X=0
grep -v "^==\|rc=0" "$log"
