#!/bin/bash
# LDS-address form of the filter DMAs + interior fast path of the halo fetch in conv_tile_p.hip / conv_tile_q.hip:
# op-level tests, micro-benchmark of both kernels, the step (default twice, SSA_TILE_Q=1 once).   bash tools/calls/r4u.sh [tag]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r4u}
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_conv_tile_q_gpu.py tests/test_kernels_gpu.py tests/test_group_gpu.py tests/test_fuse_bwd_gpu.py -q -m gpu > gpurun_out/${T}_tests.log 2>&1
echo "tests: $(tail -1 gpurun_out/${T}_tests.log)"
grep -E "^FAILED|^ERROR" gpurun_out/${T}_tests.log | head -10
timeout 200 python tools/tilebench.py 20 --q 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_tilebench.txt
cat gpurun_out/${T}_tilebench.txt
B="bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --eager-steps 0"
timeout 200 python $B > gpurun_out/${T}_bench_default.log 2>&1
timeout 200 env SSA_TILE_Q=1 python $B > gpurun_out/${T}_bench_q.log 2>&1
timeout 200 python $B > gpurun_out/${T}_bench_default2.log 2>&1
for f in default q default2; do grep -h '^{' gpurun_out/${T}_bench_$f.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["ms_per_step"],2), "ms", d["config"]["loss"])' $f || tail -3 gpurun_out/${T}_bench_$f.log; done
