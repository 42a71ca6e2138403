#!/bin/bash
# Round 6 call M: first device run of the register-fed 3x3 head kernel (csrc/conv_halo_reg.hip): its tests, the head
# micro-benchmark with it on / off, the step with it on / off and with the wide-layer BatchNorm chunking on / off.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r6m}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "halo_reg or conv_fwd_bwd or bn_" > gpurun_out/${T}_kernel_tests.log 2>&1
echo "kernel tests rc=$?"; tail -4 gpurun_out/${T}_kernel_tests.log
echo "== headbench 3x3, register-fed"; timeout 300 python tools/headbench.py 20 --only-3x3 --fwd-only 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${T}_headbench_reg.txt
echo "== headbench 3x3, LDS ring"; SSA_HALO3_REG=0 timeout 300 python tools/headbench.py 20 --only-3x3 --fwd-only 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${T}_headbench_ring.txt
for cfg in "1 6" "0 6" "1 0" "1 6"; do
  set -- $cfg
  SSA_HALO3_REG=$1 SSA_BN_WIDE_CHUNKS=$2 timeout 400 python bench.py --dtype bf16 --steps 30 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/${T}_bench_reg$1_chunks$2.json 2> gpurun_out/${T}_bench.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${T}_bench_reg$1_chunks$2.json").read().strip().splitlines()[-1])
    f = d["roofline"]["families"]
    print("reg=$1 chunks=$2: %.2f ms | halo3 %.3f  bn apply %.3f  bwd apply %.3f  bwd reduce %.3f" % (d["ms_per_step"], (d["roofline"]["mfma_3x3"]["kernels"].get("ConvHaloGemm3") or d["roofline"]["mfma_3x3"]["kernels"].get("ConvHaloReg3") or {}).get("ms_per_step", -1), f["BnApplyTrainK"]["ms_per_step"], f["BnBwdApplyK"]["ms_per_step"], f["BnBwdReduceK"]["ms_per_step"]))
except Exception as e:
    print("reg=$1 chunks=$2 failed:", e); print(open("gpurun_out/${T}_bench.err").read()[-1500:])
PY
done
