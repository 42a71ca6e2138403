#!/bin/bash
# Round 2, call U: ConvTileAny variant with three n-blocks per workgroup (halo read once per 96 output channels).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out
log=gpurun_out/r2u.log
: > "$log"
run() { local name=$1 t=$2; shift 2; echo "== $name" >> "$log"; timeout "$t" "$@" > "gpurun_out/r2u_$name.log" 2>&1; echo "$name rc=$?" >> "$log"; }
b() { local name=$1; shift; run "bench_$name" 120 env "$@" python bench.py --no-cpu-baseline --no-roofline; grep -h '^{' "gpurun_out/r2u_bench_$name.log" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2), d['config']['library_launches_per_step'], d['config']['loss'])" >> "$log" 2>&1; }
run tests 600 env SSA_TILE_NB3=1 python -m pytest tests/test_group_gpu.py tests/test_fuse_bwd_gpu.py -q -m gpu
tail -3 gpurun_out/r2u_tests.log >> "$log"
b default SSA_X=0
b nb3 SSA_TILE_NB3=1
b default2 SSA_X=0
b nb3b SSA_TILE_NB3=1
grep -v "^==\|rc=0" "$log"
