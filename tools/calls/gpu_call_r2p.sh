#!/bin/bash
# Round 2, call P: weight-gradient tile kernel: 192/384 channels as sub-problems of the 96-channel instantiation,
# 48- and 96-channel instantiations in one launch.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out
log=gpurun_out/r2p.log
: > "$log"
run() { local name=$1 t=$2; shift 2; echo "== $name" >> "$log"; timeout "$t" "$@" > "gpurun_out/r2p_$name.log" 2>&1; echo "$name rc=$?" >> "$log"; }
b() { local name=$1; shift; run "bench_$name" 120 env "$@" python bench.py --no-cpu-baseline --no-roofline; grep -h '^{' "gpurun_out/r2p_bench_$name.log" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2), d['config']['library_launches_per_step'], d['config']['loss'])" >> "$log" 2>&1; }
run tests 600 python -m pytest tests/test_kernels_gpu.py tests/test_group_gpu.py tests/test_fuse_bwd_gpu.py -q -m gpu
tail -4 gpurun_out/r2p_tests.log >> "$log"
b default SSA_X=0
b noc96 SSA_WGRAD_C96=0
b noany SSA_WGRAD_ANY=0
b old SSA_WGRAD_C96=0 SSA_WGRAD_ANY=0
b strip8 SSA_WGRAD_STRIP=8
b default2 SSA_X=0
cat "$log"
