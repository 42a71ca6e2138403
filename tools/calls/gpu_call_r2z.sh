#!/bin/bash
# Round 2, call Z: knob sweep under the single weight-gradient flush.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out
log=gpurun_out/r2z.log
: > "$log"
run() { local name=$1 t=$2; shift 2; echo "== $name" >> "$log"; timeout "$t" "$@" > "gpurun_out/r2z_$name.log" 2>&1; echo "$name rc=$?" >> "$log"; }
b() { local name=$1; shift; run "bench_$name" 120 env "$@" python bench.py --no-cpu-baseline --no-roofline; grep -h '^{' "gpurun_out/r2z_bench_$name.log" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2), d['config']['library_launches_per_step'], d['config']['loss'])" >> "$log" 2>&1; }
b default SSA_X=0
b strip16 SSA_WGRAD_STRIP=16
b strip4 SSA_WGRAD_STRIP=4
b wgs320 SSA_WGRAD_WGS=320
b wgs1280 SSA_WGRAD_WGS=1280
b default2 SSA_X=0
grep -v "^==\|rc=0" "$log"
