#!/bin/bash
# Round 5, call I: the CPU sides of the parity tests on 16 intra-op threads (tests/conftest.py) instead of torch's default
# of one per visible CPU: durations of the CPU-heavy tests.     bash tools/calls/r5i.sh
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r5i}
mkdir -p gpurun_out
python -c "import torch, os; print('torch default intra-op threads on this box:', torch.get_num_threads(), 'visible CPUs', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))"
timeout 600 python -m pytest tests/test_e2e_gpu.py::test_train_step "tests/test_siblings_gpu.py::test_sibling_train_step" \
  tests/test_parity_1024_gpu.py "tests/test_siblings_gpu.py::test_sibling_eval_op_by_op[mscale.HRNet]" -q -m gpu --durations=10 > gpurun_out/${T}_tests.log 2>&1
echo "rc=$?: $(tail -1 gpurun_out/${T}_tests.log)"
grep -E "s call|s setup" gpurun_out/${T}_tests.log | head -12
