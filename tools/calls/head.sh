#!/bin/bash
# Head-kernel micro-benchmarks: product library and the experiment builds named on the command line.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
T=${1:-head}; shift
timeout 300 python tools/headbench.py 10 > gpurun_out/${T}_base.log 2>&1
for n in "$@"; do timeout 300 python tools/headbench.py 10 --lib $n > gpurun_out/${T}_$n.log 2>&1; done
for f in gpurun_out/${T}_*.log; do echo "== $f"; grep -v "Warning\|super()\|amdgpu.ids" $f | tail -12; done
