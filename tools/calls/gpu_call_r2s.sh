#!/bin/bash
# Round 2, call S: BatchNorm rows-per-thread sweep after the prologue fix.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out
log=gpurun_out/r2s.log
: > "$log"
run() { local name=$1 t=$2; shift 2; echo "== $name" >> "$log"; timeout "$t" "$@" > "gpurun_out/r2s_$name.log" 2>&1; echo "$name rc=$?" >> "$log"; }
b() { local name=$1; shift; run "bench_$name" 120 env "$@" python bench.py --no-cpu-baseline --no-roofline; grep -h '^{' "gpurun_out/r2s_bench_$name.log" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2), d['config']['library_launches_per_step'], d['config']['loss'])" >> "$log" 2>&1; }
b default SSA_X=0
b a4r8 SSA_BN_ROWS_APPLY=4 SSA_BN_ROWS_REDUCE=8
b a4r16 SSA_BN_ROWS_APPLY=4 SSA_BN_ROWS_REDUCE=16
b a2r8 SSA_BN_ROWS_APPLY=2 SSA_BN_ROWS_REDUCE=8
b a2r4 SSA_BN_ROWS_APPLY=2 SSA_BN_ROWS_REDUCE=4
b a8r8 SSA_BN_ROWS_APPLY=8 SSA_BN_ROWS_REDUCE=8
b default2 SSA_X=0
grep -v "^==\|rc=0" "$log"
