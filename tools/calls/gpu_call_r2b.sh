#!/bin/bash
# Round 2, call B: first contact of the lockstep/grouped rewrite with hardware.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out
log=gpurun_out/r2b.log
: > "$log"
run() { local name=$1 t=$2; shift 2; echo "== $name" >> "$log"; timeout "$t" "$@" > "gpurun_out/r2b_$name.log" 2>&1; echo "$name rc=$?" >> "$log"; }
run group_tests 300 python -m pytest tests/test_group_gpu.py tests/test_fuse_bwd_gpu.py -q -x -s -m gpu
run kernel_tests 400 python -m pytest tests/test_kernels_gpu.py -q -m gpu
run bench 200 env SSA_DUMP_KERNELS=1 python bench.py --no-cpu-baseline
run e2e 400 python -m pytest tests/test_e2e_gpu.py -q -s -m gpu
run bench_dist1 200 env SSA_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-roofline
cat "$log"
