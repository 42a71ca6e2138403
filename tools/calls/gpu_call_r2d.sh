#!/bin/bash
# Round 2, call D: re-run of what call C flagged (tolerance calibration), the new weight-gradient reduce,
# and A/B timings of the launch-geometry knobs.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out
log=gpurun_out/r2d.log
: > "$log"
run() { local name=$1 t=$2; shift 2; echo "== $name" >> "$log"; timeout "$t" "$@" > "gpurun_out/r2d_$name.log" 2>&1; echo "$name rc=$?" >> "$log"; }
b() { local name=$1; shift; run "bench_$name" 120 env "$@" python bench.py --no-cpu-baseline --no-roofline; grep -h '^{' "gpurun_out/r2d_bench_$name.log" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2), d['config']['library_launches_per_step'], d['config']['loss'])" >> "$log" 2>&1; }
run tests 700 python -m pytest tests/test_group_gpu.py tests/test_kernels_gpu.py tests/test_fuse_bwd_gpu.py tests/test_deepv3_gpu.py tests/test_ddp_graph_gpu.py -q -m gpu
b default SSA_X=0
b nounify SSA_GROUP_UNIFY=0
b bnrows4 SSA_BN_ROWS_APPLY=4 SSA_BN_ROWS_REDUCE=8
b bnrows8 SSA_BN_ROWS_APPLY=8 SSA_BN_ROWS_REDUCE=16
b bnrows16 SSA_BN_ROWS_APPLY=16 SSA_BN_ROWS_REDUCE=32
b strip8 SSA_WGRAD_STRIP=8
b strip32 SSA_WGRAD_STRIP=32
b nowgradtile SSA_WGRAD_TILE=0
run teacher 600 python -m pytest tests/test_parity_1024_gpu.py -q -s -m gpu -k teacher
run bench_full 200 env SSA_DUMP_KERNELS=1 python bench.py --no-cpu-baseline
cat "$log"
