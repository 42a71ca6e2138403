#!/bin/bash
# Round 2, call X: weight-gradient flush threshold sweep; clean re-run of the 1024^2 parity tests and the attnscale tests.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out
log=gpurun_out/r2x.log
: > "$log"
run() { local name=$1 t=$2; shift 2; echo "== $name" >> "$log"; timeout "$t" "$@" > "gpurun_out/r2x_$name.log" 2>&1; echo "$name rc=$?" >> "$log"; }
b() { local name=$1; shift; run "bench_$name" 120 env "$@" python bench.py --no-cpu-baseline --no-roofline; grep -h '^{' "gpurun_out/r2x_bench_$name.log" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2), d['config']['library_launches_per_step'], d['config']['loss'])" >> "$log" 2>&1; }
b default SSA_X=0
b flush48 SSA_WGRAD_FLUSH_AT=48
b flush192 SSA_WGRAD_FLUSH_AT=192
b flush1000 SSA_WGRAD_FLUSH_AT=1000
b default2 SSA_X=0
run parity 600 python -m pytest tests/test_parity_1024_gpu.py tests/test_attnscale_gpu.py -q -m gpu
tail -3 gpurun_out/r2x_parity.log >> "$log"
grep -v "^==\|rc=0" "$log"
