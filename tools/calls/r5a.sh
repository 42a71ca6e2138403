#!/bin/bash
# Round 5, call A: the re-conditioned evaluation parity tests (calibrated BatchNorm buffers, 1/f images, operand-range
# assertions) on both storage builds, with durations.      bash tools/calls/r5a.sh
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_eval_gpu.py "tests/test_e2e_gpu.py::test_eval_nscale" \
  "tests/test_fp16_storage_gpu.py::test_eval_teacher_forced_on_the_fp16_build" \
  -q -m gpu -s --durations=15 > gpurun_out/r5a_tests.log 2>&1
echo "tests rc=$?"
grep -E "operand ranges|comparisons|passed|failed|FAIL|nscale |forward .* s$|Error|error|vs fp64" gpurun_out/r5a_tests.log | head -60
tail -25 gpurun_out/r5a_tests.log
