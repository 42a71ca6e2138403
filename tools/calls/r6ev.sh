#!/bin/bash
# Round 6 call EV: evaluation-mode BatchNorm coefficients from a registry refreshed by one batched launch per forward
# (were one launch per layer and scale pass): tests, the evaluation rows, kernel statistics of configs[1].
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r6ev}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_group_gpu.py tests/test_e2e_gpu.py tests/test_graphed_step_gpu.py tests/test_siblings_gpu.py tests/test_attnscale_gpu.py tests/test_deepv3_gpu.py -q -x -m gpu > gpurun_out/${T}_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/${T}_tests.log
timeout 600 python -m pytest tests/test_parity_eval_gpu.py tests/test_fp16_storage_gpu.py -q -x -m gpu > gpurun_out/${T}_parity_eval.log 2>&1
echo "eval parity rc=$?"; tail -3 gpurun_out/${T}_parity_eval.log
timeout 300 env SSA_ACT_DTYPE=fp16 python tools/eval_bench.py 10 2>&1 | grep '^{' > gpurun_out/${T}_eval_bench.json
timeout 300 env SSA_ACT_DTYPE=fp16 python tools/eval_bench.py 3 mapillary-ref 2>&1 | grep '^{' >> gpurun_out/${T}_eval_bench.json
python -c 'import sys,json
for l in open(sys.argv[1]):
    d=json.loads(l); print(d["config"], d["storage"], round(d["ms_per_image"],2), "ms captured,", round(d["eager_ms_per_image"],2), "eager")' gpurun_out/${T}_eval_bench.json
bash tools/calls/r6eval.sh ${T}p | tail -14
