#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r4g}
mkdir -p gpurun_out
timeout 300 env SSA_ACT_DTYPE=fp16 python tools/debug_nscale.py > gpurun_out/${T}_dbg_f16.log 2>&1; grep -v amdgpu.ids gpurun_out/${T}_dbg_f16.log | tail -60
timeout 300 python tools/debug_nscale.py > gpurun_out/${T}_dbg_bf16.log 2>&1; grep -v amdgpu.ids gpurun_out/${T}_dbg_bf16.log | tail -25
timeout 900 python -m pytest tests/test_parity_eval_gpu.py -q -m gpu -x -s --durations=10 > gpurun_out/${T}_eval_parity.log 2>&1
echo "eval parity rc=$?"; grep -v "^  op \|^$" gpurun_out/${T}_eval_parity.log | tail -40
