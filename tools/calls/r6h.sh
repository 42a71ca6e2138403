#!/bin/bash
# Round 6 call H: the wide GEMM on the memory-bound 1x1 convs of layer1 / the fuse layers (SSA_GEMM_WIDE_SMALL 0 / 1 / 2).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r6h}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_group_gpu.py tests/test_fuse_bwd_gpu.py -q -x -m gpu > gpurun_out/${T}_kernel_tests.log 2>&1
echo "kernel tests rc=$?"; tail -5 gpurun_out/${T}_kernel_tests.log
for v in 2 1 0 2 0; do
  SSA_GEMM_WIDE_SMALL=$v timeout 400 python bench.py --dtype bf16 --steps 30 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/${T}_bench_small$v.json 2> gpurun_out/${T}_bench.err
  python - <<PY
import json
j = json.load(open("gpurun_out/${T}_bench_small$v.json"))
f = j["roofline"]["families"]
print("small=$v: ms", round(j["ms_per_step"], 3), "loss", round(j["config"]["loss"], 3), " ".join("%s %.3f(%d)" % (k, v["ms_per_step"], v["launches_per_step"]) for k, v in f.items() if k in ("ConvIgemm", "ConvGemmWide", "ConvHaloGemm", "ConvWgradTr")))
PY
done
timeout 600 python -m pytest tests/test_parity_1024_gpu.py tests/test_e2e_gpu.py -q -x -m gpu > gpurun_out/${T}_parity.log 2>&1; echo "parity rc=$?"; tail -3 gpurun_out/${T}_parity.log
