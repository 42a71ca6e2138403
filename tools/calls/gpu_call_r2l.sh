#!/bin/bash
# Round 2, call L: tile-balanced filter repack, stride-2 data gradient by output parity.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out
log=gpurun_out/r2l.log
: > "$log"
run() { local name=$1 t=$2; shift 2; echo "== $name" >> "$log"; timeout "$t" "$@" > "gpurun_out/r2l_$name.log" 2>&1; echo "$name rc=$?" >> "$log"; }
b() { local name=$1; shift; run "bench_$name" 120 env "$@" python bench.py --no-cpu-baseline --no-roofline; grep -h '^{' "gpurun_out/r2l_bench_$name.log" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2), d['config']['library_launches_per_step'], d['config']['loss'])" >> "$log" 2>&1; }
run tests 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x
tail -5 gpurun_out/r2l_tests.log >> "$log"
b default SSA_X=0
b old SSA_PACK_TILED=0 SSA_DGRAD_S2=0
b pack_only SSA_DGRAD_S2=0
b default2 SSA_X=0
cat "$log"
