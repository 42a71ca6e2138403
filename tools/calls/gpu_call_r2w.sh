#!/bin/bash
# Round 2, call W: conv_halo_gemm with software-pipelined fragment reads (SSA_HALO_PRE=1).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out
log=gpurun_out/r2w.log
: > "$log"
run() { local name=$1 t=$2; shift 2; echo "== $name" >> "$log"; timeout "$t" "$@" > "gpurun_out/r2w_$name.log" 2>&1; echo "$name rc=$?" >> "$log"; }
b() { local name=$1; shift; run "bench_$name" 120 env "$@" python bench.py --no-cpu-baseline --no-roofline; grep -h '^{' "gpurun_out/r2w_bench_$name.log" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2), d['config']['library_launches_per_step'], d['config']['loss'])" >> "$log" 2>&1; }
run tests 300 env SSA_HALO_PRE=1 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "conv_fwd_bwd"
tail -3 gpurun_out/r2w_tests.log >> "$log"
b default SSA_X=0
b pre SSA_HALO_PRE=1
b default2 SSA_X=0
b pre2 SSA_HALO_PRE=1
grep -v "^==\|rc=0" "$log"
