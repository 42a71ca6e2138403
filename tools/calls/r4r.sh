#!/bin/bash
# Second device run of conv_tile_q.hip and first of the branch-free halo staging / pinned halo loads in conv_tile_p.hip:
# op-level tests, micro-benchmark (both kernels), the step (default and SSA_TILE_Q=1).   bash tools/calls/r4r.sh [tag]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r4r}
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_conv_tile_q_gpu.py tests/test_kernels_gpu.py tests/test_group_gpu.py tests/test_fuse_bwd_gpu.py -q -m gpu > gpurun_out/${T}_tests.log 2>&1
echo "tests: $(tail -1 gpurun_out/${T}_tests.log)"
grep -E "^FAILED|^ERROR" gpurun_out/${T}_tests.log | head -10
timeout 200 python tools/tilebench.py 20 --q 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_tilebench.txt
cat gpurun_out/${T}_tilebench.txt
timeout 200 env SSA_TILE_Q_PB=2 python tools/tilebench.py 20 --q 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_tilebench_pb2.txt
head -14 gpurun_out/${T}_tilebench_pb2.txt
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --eager-steps 0"
timeout 200 $B > gpurun_out/${T}_bench_default.log 2>&1
timeout 200 env SSA_TILE_Q=1 $B > gpurun_out/${T}_bench_q.log 2>&1
timeout 200 env SSA_TILE_Q=1 SSA_TILE_Q_PB=2 $B > gpurun_out/${T}_bench_q_pb2.log 2>&1
for f in default q q_pb2; do grep -h '^{' gpurun_out/${T}_bench_$f.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["ms_per_step"],2), "ms", d["config"]["loss"])' $f || tail -3 gpurun_out/${T}_bench_$f.log; done
timeout 300 env SSA_TILE_Q=1 python -m pytest tests/test_group_gpu.py tests/test_fuse_bwd_gpu.py -q -m gpu > gpurun_out/${T}_tests_q_env.log 2>&1
echo "group / fuse tests under SSA_TILE_Q=1: $(tail -1 gpurun_out/${T}_tests_q_env.log)"
grep -E "^FAILED|^ERROR" gpurun_out/${T}_tests_q_env.log | head
