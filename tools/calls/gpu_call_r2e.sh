#!/bin/bash
# Round 2, call E: BN block reduce without LDS atomics, per-pixel bilinear for the logits, igemm tile choice
# reverted; the teacher-forced parity test with the calibrated parameter-gradient tolerance.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out
log=gpurun_out/r2e.log
: > "$log"
run() { local name=$1 t=$2; shift 2; echo "== $name" >> "$log"; timeout "$t" "$@" > "gpurun_out/r2e_$name.log" 2>&1; echo "$name rc=$?" >> "$log"; }
b() { local name=$1; shift; run "bench_$name" 120 env "$@" python bench.py --no-cpu-baseline --no-roofline; grep -h '^{' "gpurun_out/r2e_bench_$name.log" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2), d['config']['library_launches_per_step'], d['config']['loss'])" >> "$log" 2>&1; }
run tests 500 python -m pytest tests/test_kernels_gpu.py tests/test_group_gpu.py -q -m gpu
b default SSA_X=0
b bnrows_old SSA_BN_ROWS_APPLY=2 SSA_BN_ROWS_REDUCE=4
b nounify SSA_GROUP_UNIFY=0
b default2 SSA_X=0
run teacher 400 python -m pytest tests/test_parity_1024_gpu.py -q -s -m gpu -k teacher
run bench_full 200 env SSA_DUMP_KERNELS=1 python bench.py --no-cpu-baseline
cat "$log"
