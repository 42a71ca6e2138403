#!/bin/bash
# Head-kernel change: conv kernel tests on the device, then the micro-benchmark, then a bench line.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
T=${1:-head2}
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "conv" > gpurun_out/${T}_tests.log 2>&1
tail -3 gpurun_out/${T}_tests.log
timeout 300 python tools/headbench.py 10 > gpurun_out/${T}_bench.log 2>&1
grep -v "Warning\|super()\|amdgpu.ids" gpurun_out/${T}_bench.log | tail -12
timeout 200 python bench.py --no-cpu-baseline --no-roofline --eager-steps 0 > gpurun_out/${T}_step.log 2>&1
grep -h '^{' gpurun_out/${T}_step.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print("step", round(d["ms_per_step"],2), "ms")'
