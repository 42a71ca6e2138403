#!/bin/bash
# Round 6 call U: LDS-tiled separable backward of the few-channel fp32 upsampling resizes: the bilinear tests, the
# micro-benchmark with the tile kernel on / off, the step on / off.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r6u}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "bilinear or scale_fusion" > gpurun_out/${T}_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/${T}_tests.log
for f in 1 0; do SSA_BILINEAR_BWD_TILE=$f timeout 200 python tools/bilinbench.py 20 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/${T}_bilinbench.txt; done
for f in 1 0 1 0; do
  SSA_BILINEAR_BWD_TILE=$f timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --eager-steps 0 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
    print("tile=$f: %.3f ms  %.2f img/s" % (d["ms_per_step"], d["value"]))
except Exception as e:
    print("tile=$f failed:", e); print(open("gpurun_out/${T}_bench.err").read()[-1500:])
PY
done
timeout 300 python -m pytest tests/test_e2e_gpu.py tests/test_parity_1024_gpu.py -q -x -m gpu > gpurun_out/${T}_parity.log 2>&1
echo "parity rc=$?"; tail -3 gpurun_out/${T}_parity.log
