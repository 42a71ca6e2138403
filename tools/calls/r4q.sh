#!/bin/bash
# First device run of the 48-channel-block trunk conv (csrc/conv_tile_q.hip, opt-in): lane-map probes + op-level tests,
# micro-benchmark against conv_tile_p.hip, the step with SSA_TILE_Q=1 against the default.   bash tools/calls/r4q.sh [tag]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r4q}
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_conv_tile_q_gpu.py -q -m gpu -x > gpurun_out/${T}_tests.log 2>&1
echo "q tests: $(tail -1 gpurun_out/${T}_tests.log)"
grep -E "FAILED|Error|assert" gpurun_out/${T}_tests.log | head -10
timeout 200 python tools/tilebench.py 20 --q 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_tilebench.txt
cat gpurun_out/${T}_tilebench.txt
for pb in 4 2; do
  timeout 200 env SSA_TILE_Q_PB=$pb python tools/tilebench.py 20 --q 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_tilebench_pb$pb.txt
  head -12 gpurun_out/${T}_tilebench_pb$pb.txt
done
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --eager-steps 0"
timeout 200 $B > gpurun_out/${T}_bench_default.log 2>&1
timeout 200 env SSA_TILE_Q=1 $B > gpurun_out/${T}_bench_q.log 2>&1
timeout 200 env SSA_TILE_Q=1 SSA_TILE_Q_WGS=768 $B > gpurun_out/${T}_bench_q768.log 2>&1
for f in default q q768; do grep -h '^{' gpurun_out/${T}_bench_$f.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["ms_per_step"],2), "ms", d["config"]["loss"])' $f || tail -3 gpurun_out/${T}_bench_$f.log; done
timeout 300 env SSA_TILE_Q=1 python -m pytest tests/test_group_gpu.py tests/test_fuse_bwd_gpu.py -q -m gpu > gpurun_out/${T}_tests_q_env.log 2>&1
echo "group / fuse tests under SSA_TILE_Q=1: $(tail -1 gpurun_out/${T}_tests_q_env.log)"
