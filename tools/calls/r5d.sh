#!/bin/bash
# Round 5, call D: the sign-byte ReLU mask of the BatchNorm passes (csrc/bn.hip) on the device: BN / block / fused-backward
# tests, the step with and without it on one box, the BN micro-benchmark.        bash tools/calls/r5d.sh [tag]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r5d}
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_group_gpu.py tests/test_fuse_bwd_gpu.py -q -m gpu -x > gpurun_out/${T}_tests.log 2>&1
echo "kernel / group / fused-backward tests rc=$?: $(tail -1 gpurun_out/${T}_tests.log)"
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --eager-steps 0"
line() { grep -h '^{' "$1" | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],2), "ms", d["config"]["library_launches_per_step"], "launches, loss", round(d["config"]["loss"],4))' 2>&1 | tail -1; }
run() { name=$1; shift; timeout 200 env "$@" $B > gpurun_out/${T}_bench_$name.log 2>&1; echo "$name [$*]: $(line gpurun_out/${T}_bench_$name.log)"; }
run default A=1
run nomask SSA_BN_SIGN_MASK=0
run default2 A=1
run nomask2 SSA_BN_SIGN_MASK=0
timeout 200 python tools/bnbench.py > gpurun_out/${T}_bnbench.txt 2>&1; tail -12 gpurun_out/${T}_bnbench.txt
timeout 200 env SSA_BN_SIGN_MASK=0 python tools/bnbench.py > gpurun_out/${T}_bnbench_nomask.txt 2>&1; tail -12 gpurun_out/${T}_bnbench_nomask.txt
