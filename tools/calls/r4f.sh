#!/bin/bash
# Round 4: the new parity tests (device-side teacher, full-size eval shapes) and the fp16-storage build.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r4f}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_eval_gpu.py -q -m gpu -x -s --durations=10 > gpurun_out/${T}_eval_parity.log 2>&1
echo "eval parity rc=$?"; grep -v "^  op \|^$" gpurun_out/${T}_eval_parity.log | tail -40
timeout 900 python -m pytest tests/test_fp16_storage_gpu.py -q -m gpu --durations=10 > gpurun_out/${T}_fp16.log 2>&1
echo "fp16 rc=$?"; tail -25 gpurun_out/${T}_fp16.log
grep "fp16 storage\|rel err" gpurun_out/fp16_e2e_eval.log | head -20
timeout 300 python -m pytest tests/test_siblings_gpu.py -q -m gpu -k "OCRNetASPP" > gpurun_out/${T}_sib.log 2>&1
echo "siblings rc=$?"; tail -3 gpurun_out/${T}_sib.log
